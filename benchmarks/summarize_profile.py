"""Reduce the rocprofv3 CSV outputs of benchmarks/run_profile.sh to one text summary (kernel-time table + per-kernel PMC means)."""
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


def traffic_json(root, out_path):
    """Per-kernel HBM bytes per launch: FETCH_SIZE and WRITE_SIZE are reported in KiB; gfx950 counts a wide (16 B/lane)
    coalesced read at half its bytes (MI355X_MICROARCH.md, HBM section) -> reads x2."""
    import json
    fetch, write = pmc(root, "pmc_fetch"), pmc(root, "pmc_write")
    ks = {short(r["Name"]): r for r in kernel_stats(root)}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py, benchmarks/run_profile.sh",
           "correction": "hbm_bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024", "kernels": {}}
    # all template variants of a kernel (e.g. wconv_kernel<GN, NB>) are merged, weighted by their launches: the figure is the
    # mean over the same population of launches as bench.py's `algorithmic_bytes_per_launch`
    merged = {}
    for k in fetch:
        f = fetch[k].get("FETCH_SIZE")
        w = write.get(k, {}).get("WRITE_SIZE")
        if not f or not w:
            continue
        key = k.split("<")[0]
        m = merged.setdefault(key, {"variants": [], "n": 0, "fetch": 0.0, "write": 0.0, "ns": 0.0, "calls": 0})
        m["variants"].append(k)
        m["n"] += f[1]
        m["fetch"] += f[0] / max(f[1], 1) * f[1] * 1024 * 2
        m["write"] += w[0] / max(w[1], 1) * f[1] * 1024
        if k in ks:
            m["ns"] += float(ks[k]["TotalDurationNs"])
            m["calls"] += int(ks[k]["Calls"])
    for key, m in merged.items():
        rec = {"variants": sorted(m["variants"]), "launches_sampled": m["n"], "fetch_bytes_per_launch": m["fetch"] / m["n"],
               "write_bytes_per_launch": m["write"] / m["n"], "hbm_bytes_per_launch": (m["fetch"] + m["write"]) / m["n"]}
        if m["calls"]:
            rec["avg_us_kernel_trace"] = m["ns"] / m["calls"] / 1e3
        out["kernels"][key] = rec
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2])
