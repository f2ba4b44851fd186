cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export T=${1:-r04_m}
python tools/lab/enc_with_traffic.py 20 2>&1 | grep "copy of"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${T}_decfetch -o run -- python3 $GRAFT_REPO_ROOT/tools/lab/dec_only.py 5 bf16 nograph > $GRAFT_REPO_ROOT/gpurun_out/${T}_decfetch.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, collections
T=os.environ["T"]
acc=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(f"gpurun_out/{T}_decfetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE":
            a=acc[r["Kernel_Name"][:50]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(acc.items(), key=lambda kv:-kv[1][1])[:8]:
    print(f"{k:50s} n {n:5d} FETCH_SIZE avg {v/n:9.1f} KB raw  total {v/1024:8.1f} MB raw")
PY
rm -rf gpurun_out/${T}_decfetch
