"""Reduce the rocprofv3 CSV outputs of tests/run_profile.sh to one text summary (kernel-time table + per-kernel PMC means)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "at::native" in name:
        return "torch:" + name.split("at::native::")[-1][:50]
    return name.split("(")[0][:70]


def kernel_stats(root):
    rows = []
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    return rows


def pmc(root, sub):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return agg


def main():
    root = sys.argv[1]
    ks = kernel_stats(root)
    print("== kernel time (rocprofv3 --kernel-trace --stats) ==")
    tot = 0.0
    for r in ks:
        tot += float(r.get("TotalDurationNs", 0))
    print(f"{'kernel':<72s}{'calls':>8s}{'total ms':>10s}{'avg us':>10s}{'%':>7s}")
    for r in sorted(ks, key=lambda r: -float(r.get("TotalDurationNs", 0)))[:40]:
        t = float(r["TotalDurationNs"])
        print(f"{short(r['Name']):<72s}{int(r['Calls']):>8d}{t / 1e6:>10.2f}{float(r['AverageNs']) / 1e3:>10.1f}{100 * t / tot:>7.2f}")
    for sub, title in (("pmc_fetch", "FETCH_SIZE (KiB units as reported; gfx950 wide reads are under-counted 2x)"), ("pmc_write", "WRITE_SIZE"),
                       ("pmc_mfma", "MFMA busy")):
        agg = pmc(root, sub)
        if not agg:
            print(f"== {title}: no data ==")
            continue
        print(f"== {title}: mean per dispatch ==")
        names = sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values()))[:25]
        for k in names:
            cells = "  ".join(f"{c}={v[0] / max(v[1], 1):.4g} (n={v[1]})" for c, v in agg[k].items())
            print(f"{k:<72s}{cells}")


if __name__ == "__main__":
    main()
