"""Normalise rocprofv3 --pmc passes of bench.py into the per-kernel MFMA / LDS table committed under profiles/:

    python tools/pmc_mfma_summary.py OUT.csv DIR_MFMA_PASS [DIR_LDS_PASS]

DIR_MFMA_PASS: SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
DIR_LDS_PASS : SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES
Columns: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the busy CUs' SIMD cycles with
the matrix pipe executing (clock independent; 32 cycles per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32);
valu_coexec = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES (matrix cycles with a vector instruction beside);
lds_active = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES, lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
avg_us is the dispatch duration UNDER the profiler (slower than the un-profiled kernel_stats)."""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:110], int(r["Grid_Size"]))
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:
                disp[k].add(r["Dispatch_Id"])
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, disp, dur


def main():
    out = sys.argv[1]
    a1, d1, t1 = load(sys.argv[2])
    a2 = load(sys.argv[3])[0] if len(sys.argv) > 3 else {}
    rows = []
    for k, c in a1.items():
        n = len(d1[k])
        busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0 or t1[k] / n < 2.0:
            continue
        l = a2.get(k, {})
        lb = l.get("SQ_BUSY_CU_CYCLES", 0.0)
        rows.append([k[0], k[1], n, round(t1[k] / n, 1), round(mf / (4.0 * busy), 4),
                     round(c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / mf, 4) if mf > 0 else "",
                     round(l.get("SQ_LDS_IDX_ACTIVE", 0.0) / lb, 4) if lb > 0 else "",
                     round(l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"], 4) if l.get("SQ_LDS_IDX_ACTIVE", 0) > 0 else ""])
    rows.sort(key=lambda r: -r[2] * r[3])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_size", "dispatches", "avg_us", "mfma_busy", "valu_coexec", "lds_active", "lds_conflict"])
        w.writerows(rows)


if __name__ == "__main__":
    main()
