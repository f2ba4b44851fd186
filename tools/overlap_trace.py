"""Where does the pipelined step lose time against encode-alone?  Reads a rocprofv3 --kernel-trace CSV of bench.py
and reports, over the timed steady state: busy time of the encoder kernels, the gaps between consecutive encoder
kernels (in-order stream), and how much decode-kernel time ran inside / outside those gaps.
    python tools/overlap_trace.py <kernel_trace.csv>"""
import csv
import sys

ENC = ("cn_mlp_fused", "cn_dwconv", "cn_gemm2_kernelILi128", "cn_gemm2_kernel<128", "cn_ln_patchify", "cn_logmel", "cn_stem",
       "cn_frame_mean", "cn_clip_pool")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = rows[-1][1]
win = [x for x in rows if x[0] > t_end - 60_000_000]  # last 60 ms: steady state of the timed region
enc = [x for x in win if any(k in x[2] for k in ENC)]
dec = [x for x in win if not any(k in x[2] for k in ENC)]
span = enc[-1][1] - enc[0][0]
busy = sum(e - s for s, e, _ in enc)
gaps = [enc[i + 1][0] - enc[i][1] for i in range(len(enc) - 1)]
big = sorted(gaps)[-10:]
print(f"window {span / 1e6:.2f} ms: encoder kernels {len(enc)}, busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), "
      f"gaps total {sum(g for g in gaps if g > 0) / 1e6:.2f} ms, median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, "
      f"10 largest {[round(g / 1e3) for g in big]} us")
dbusy = sum(e - s for s, e, _ in dec)
print(f"decode-side kernels {len(dec)}, summed duration {dbusy / 1e6:.2f} ms")
# encoder kernels that overlap a decode kernel vs not: mean duration ratio per kernel name
import collections
dur = collections.defaultdict(lambda: [[], []])
j = 0
for s, e, n in enc:
    ov = any(ds < e and de > s for ds, de, _ in dec)
    dur[n[:40]][1 if ov else 0].append(e - s)
for n, (a, b) in sorted(dur.items(), key=lambda kv: -sum(kv[1][0] + kv[1][1]))[:8]:
    ma = sum(a) / len(a) / 1e3 if a else 0
    mb = sum(b) / len(b) / 1e3 if b else 0
    print(f"{n:42s} alone {len(a):4d} x {ma:7.1f} us   beside decode {len(b):4d} x {mb:7.1f} us")
