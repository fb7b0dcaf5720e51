"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py (steady-state step = between the last two
sample_update_kernel launches).  Usage: python benchmarks/trace_step.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "sample_update_kernel" in r["Kernel_Name"]]
    step = rows[idx[-2]:idx[-1]]
    wall = (int(rows[idx[-1]]["Start_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6
    print(f"step wall {wall:.2f} ms, busy {busy:.2f} ms, launches {len(step)}")
    agg = collections.defaultdict(lambda: [0, 0])
    byk = collections.defaultdict(lambda: [0, 0])
    for r in step:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        a = agg[(k, f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")]
        a[0] += d
        a[1] += 1
        b = byk[k]
        b[0] += d
        b[1] += 1
    # idle time charged to the kernel that FOLLOWS the gap (the dependent launch that could not start earlier)
    gaps = collections.defaultdict(lambda: [0, 0])
    for a_, b_ in zip(step[:-1], step[1:]):
        gdur = int(b_["Start_Timestamp"]) - int(a_["End_Timestamp"])
        if gdur > 0:
            gk = gaps[short(b_["Kernel_Name"])]
            gk[0] += gdur
            gk[1] += 1
    print(f"idle between kernels {sum(v[0] for v in gaps.values()) / 1e6:.2f} ms")
    print("-- gap before kernel (top 15)")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:15]:
        print(f"{k:<52s}{v[1]:>5d}{v[0] / 1e3:>10.1f} us{v[0] / v[1] / 1e3:>9.2f}")
    print("-- by kernel")
    for k, v in sorted(byk.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"{k:<52s}{v[1]:>5d}{v[0] / 1e3:>10.1f} us{v[0] / v[1] / 1e3:>9.1f}")
    print("-- by kernel and grid")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{k[0]:<52s}{k[1]:>18s}{v[1]:>5d}{v[0] / 1e3:>10.1f} us{v[0] / v[1] / 1e3:>9.1f}")


if __name__ == "__main__":
    main()
