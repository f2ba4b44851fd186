#!/bin/bash
# One measurement round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh TAG  -> gpurun_out/TAG_{bench.log,kernel_stats.csv,pmc_hbm_traffic.csv,pmc_mfma.csv}
# Kernel trace and the PMC passes are separate rocprofv3 runs (never combined with a trace), each under `timeout`;
# the program itself follows `--` (no env / shell wrapper: the profiler initialises the GPU before it starts).
TAG=${1:-r02_x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 python3 $ROOT/bench.py > $OUT/${TAG}_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- python3 $ROOT/bench.py --steps 12 --cpu-clips 0 --parity-clips 0 > $OUT/${TAG}_trace.log 2>&1
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
P="python3 $ROOT/bench.py --steps 4 --warmup 1 --cpu-clips 0 --parity-clips 0"   # (4 steps: the default decode group of 4 batches)
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o run -- $P > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o run -- $P > $OUT/${TAG}_pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/${TAG}_pmc_mfma_raw -o run -- $P > $OUT/${TAG}_pmc_mfma.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/${TAG}_pmc_lds_raw -o run -- $P > $OUT/${TAG}_pmc_lds.log 2>&1
# which build the PMC tables describe: bench.py only quotes them for the library they were measured on
# (the SOURCE hash build.py records beside the library -- csrc/ + header + flags -- not the binary's: a rebuild changes the latter)
python3 -c "import hashlib,json,sys; d=json.load(open('$ROOT/conette-audio-captioning_amd/libconette_hip.build.json')); print(json.dumps({'source_sha256': d['source_sha256'], 'library_sha256': hashlib.sha256(open('$ROOT/conette-audio-captioning_amd/libconette_hip.so','rb').read()).hexdigest()}))" > $OUT/${TAG}_pmc_meta.json
python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_hbm_traffic.csv $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
python3 $ROOT/tools/pmc_mfma_summary.py $OUT/${TAG}_pmc_mfma.csv $OUT/${TAG}_pmc_mfma_raw $OUT/${TAG}_pmc_lds_raw
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma_raw $OUT/${TAG}_pmc_lds_raw
ls -la $OUT | grep ${TAG}
tail -1 $OUT/${TAG}_bench.log | cut -c1-300
