#!/bin/bash
# One measurement round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh TAG      -> gpurun_out/TAG_{bench.log,kernel_stats.csv,pmc_hbm_traffic.csv}
# Kernel trace and the two PMC passes are separate rocprofv3 runs (never combined), each under `timeout`.
TAG=${1:-r01_x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $ROOT/bench.py > $OUT/${TAG}_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- python3 $ROOT/bench.py --steps 10 > $OUT/${TAG}_trace.log 2>&1
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-clips 0 > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-clips 0 > $OUT/${TAG}_pmc_write.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_hbm_traffic.csv $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
ls -la $OUT | grep ${TAG}
tail -1 $OUT/${TAG}_bench.log | cut -c1-300
