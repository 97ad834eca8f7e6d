#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_run.sh <tag> <script.py> [args]  -- three counter passes of a small script,
# CSVs under gpurun_out/pmc_<tag>/ (developer tool; the program directly after `--`)
set -euo pipefail
TAG="$1"; shift
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
SCRIPT="$ROOT/$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/tcc" -- python3 "$SCRIPT" "$@" > "$OUT/tcc.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$SCRIPT" "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$SCRIPT" "$@" > "$OUT/sq.log" 2>&1
python3 "$ROOT/tools/pmc_table.py" "$OUT"
