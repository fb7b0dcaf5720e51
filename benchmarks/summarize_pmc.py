"""Mean PMC counter values per (kernel, grid size) from rocprofv3 counter_collection CSVs under <root>/*/.
Usage: python benchmarks/summarize_pmc.py <root> [kernel-name substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if sub not in r["Kernel_Name"]:
                    continue
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
                a = agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for (name, grid), cs in sorted(agg.items()):
        print(f"{name}  grid={grid}")
        for c, (s, n) in sorted(cs.items()):
            print(f"    {c:<36s}{s / max(n, 1):>16.5g}   (n={n})")


if __name__ == "__main__":
    main()
