#!/bin/bash
# Profiles of the benchmark's timed region for profiles/ (run on the GPU box from the repo root: bash tools/profile_round.sh r02).
# One rocprofv3 run per kind: kernel trace + stats, then the counters in their own runs (SQ set, FETCH_SIZE, WRITE_SIZE), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; the program directly after `--`.
set -o pipefail
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --timed-only --steps 20 --warmup 5 > $OUT/bench_stats.json 2> $OUT/stats.log
echo "stats done: $(tail -c 300 $OUT/bench_stats.json | head -c 200)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_sq.json 2> $OUT/sq.log
echo "sq done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_fetch.json 2> $OUT/fetch.log
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_write.json 2> $OUT/write.log
echo "write done"
cd $ROOT
ls $OUT/*/*/ | head -30
