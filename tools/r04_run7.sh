cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_h}
python -m pytest tests/test_gpu_bf16_parity.py tests/test_gpu_parity.py tests/test_gpu_exact.py -x -q -m gpu -k "encode or block or downsample or invariant" 2>&1 | tail -3
bash tools/lab/enc_only_trace.sh bf16 ${T}_enc_bf16 2>&1 | grep -E "encode alone|dwconv|mlp_r"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_fetch -o run -- python3 $GRAFT_REPO_ROOT/tools/lab/enc_only.py 3 24 bf16 > $GRAFT_REPO_ROOT/gpurun_out/${T}_fetch.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, collections
T=os.environ["T"]
acc=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(f"gpurun_out/{T}_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE":
            a=acc[r["Kernel_Name"][:60]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(acc.items(), key=lambda kv:-kv[1][1])[:12]:
    print(f"{k:60s} n {n:4d} FETCH_SIZE avg {v/n/1024:9.1f} MB raw")
PY
rm -rf gpurun_out/${T}_fetch
