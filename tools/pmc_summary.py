"""Summarise rocprofv3 --pmc passes (one counter per pass, --output-format csv) into the per-kernel table committed
under profiles/: python tools/pmc_summary.py OUT.csv DIR_FETCH DIR_WRITE"""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    rows = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Counter_Name"], r["Kernel_Name"][:110], int(r["Grid_Size"]))
            rows[k][0] += 1
            rows[k][1] += float(r["Counter_Value"])
    return rows


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    allrows = {}
    for d in dirs:
        allrows.update(load(d))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["counter (KB per dispatch; on gfx950 FETCH_SIZE reads 1/2 of wide streaming reads: double it)", "kernel",
                    "grid_size", "dispatches", "avg_KB"])
        for (c, k, g), (n, tot) in sorted(allrows.items(), key=lambda kv: -kv[1][1]):
            if tot / max(n, 1) < 64:  # < 64 KB per dispatch: noise
                continue
            w.writerow([c, k, g, n, round(tot / n, 1)])


if __name__ == "__main__":
    main()
