cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_g}
python -m pytest tests/test_gpu_bf16_parity.py tests/test_gpu_parity.py -x -q -m gpu -k "encode or block or downsample or topk" 2>&1 | tail -4
bash tools/lab/enc_only_trace.sh bf16 ${T}_enc_bf16 2>&1 | grep -E "encode alone|dwconv"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_fetch_calib -o run -- $GRAFT_REPO_ROOT/tools/lab/fetch_calib > $GRAFT_REPO_ROOT/gpurun_out/${T}_fetch_calib.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
T=os.environ["T"]
for f in glob.glob(f"gpurun_out/{T}_fetch_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_"):
            print(r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"])
PY
rm -rf gpurun_out/${T}_fetch_calib
