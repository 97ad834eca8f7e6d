#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_mode.sh <tag> <bench.py flags...>  -- rocprofv3 kernel statistics of one bench.py mode
set -e
TAG="$1"; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --timed-only --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/err.log
cd $ROOT
python3 tools/timeline.py $OUT/stats FusedOptimizer > $OUT/timeline.txt 2>&1 || true
python3 tools/kstats.py $OUT/stats 25 60 > $OUT/kstats.txt 2>&1 || true
