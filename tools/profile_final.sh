#!/bin/bash
# usage (GPU box, repo root): bash tools/profile_final.sh -- the end-of-round profile set: tools/profile_round.sh ${TAG:-r05b} (kernel
# statistics + five counter passes of the hot step), counter summary + traffic.json, kernel statistics of --mode full and --mode vfe
set -e
ROOT=$(pwd)
bash tools/profile_round.sh ${TAG:-r05b} > gpurun_out/${TAG:-r05b}_profile_round.log 2>&1 || { tail -30 gpurun_out/${TAG:-r05b}_profile_round.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/prof_${TAG:-r05b} gpurun_out/${TAG:-r05b}_pmc_summary.json gpurun_out/${TAG:-r05b}_traffic.json > gpurun_out/${TAG:-r05b}_pmc_summary.log 2>&1 || { tail gpurun_out/${TAG:-r05b}_pmc_summary.log; exit 1; }
python3 tools/pmc_table.py gpurun_out/prof_${TAG:-r05b} > gpurun_out/${TAG:-r05b}_pmc_table.txt 2>&1 || true
python3 tools/kstats.py gpurun_out/prof_${TAG:-r05b}/stats 25 60 > gpurun_out/${TAG:-r05b}_hot_kstats.txt 2>&1 || true
python3 tools/timeline.py gpurun_out/prof_${TAG:-r05b}/stats FusedOptimizer > gpurun_out/${TAG:-r05b}_hot_timeline.txt 2>&1 || true
bash tools/prof_mode.sh ${TAG:-r05b}_full --mode full
bash tools/prof_mode.sh ${TAG:-r05b}_vfe --mode vfe
tail -3 gpurun_out/${TAG:-r05b}_pmc_summary.log
