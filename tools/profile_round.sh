#!/bin/bash
# Profiles of the benchmark's timed region for profiles/ (run on the GPU box from the repo root:
#   bash tools/profile_round.sh r03 [extra bench.py flags, e.g. --mode full]).
# One rocprofv3 run per kind: kernel trace + stats, then the counters in their own runs (SQ set, FETCH_SIZE, WRITE_SIZE), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; the program directly after `--` (no env / bash -c hop: the profiler's
# preloaded library has initialised the GPU by then).
set -euo pipefail
TAG="${1:-r03}"
shift || true
EXTRA=("$@")
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

run_pass() {   # name, steps, warmup, rocprofv3 options...
    local name="$1" steps="$2" warm="$3"
    shift 3
    local dir="$OUT/$name"
    rm -rf "$dir"
    if ! rocprofv3 "$@" --output-format csv -d "$dir" -- python3 "$ROOT/bench.py" --timed-only --steps "$steps" --warmup "$warm" \
            "${EXTRA[@]}" > "$OUT/bench_$name.json" 2> "$OUT/$name.log"; then
        echo "pass $name FAILED (rocprofv3 / bench.py exit code); last lines of $OUT/$name.log:" >&2
        tail -n 20 "$OUT/$name.log" >&2
        exit 1
    fi
    if [ -z "$(find "$dir" -name '*.csv' -print -quit 2>/dev/null)" ]; then
        echo "pass $name wrote no CSV under $dir" >&2
        exit 1
    fi
    echo "$name done: $(tail -c 300 "$OUT/bench_$name.json" | head -c 200 || true)"
}

run_pass stats 20 5 --kernel-trace --stats
run_pass pmc_sq 4 2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace
run_pass pmc_sq2 4 2 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace
run_pass pmc_sq3 4 2 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace
run_pass pmc_fetch 4 2 --pmc FETCH_SIZE --kernel-trace
run_pass pmc_write 4 2 --pmc WRITE_SIZE --kernel-trace
cd "$ROOT"
find "$OUT" -name '*.csv' | sort
