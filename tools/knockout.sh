#!/bin/bash
# What does each kernel class cost on the CRITICAL PATH of a step?  Runs bench.py --timed-only with one class of kernels not
# launched at a time (MVX_KNOCKOUT, modules/frames.py: results are garbage, only the step time is read) and prints
# ms_per_step next to the complete step's.  On the GPU box:  bash tools/knockout.sh [bench flags] > gpurun_out/knockout.txt
set -uo pipefail
EXTRA=("$@")
run() {
    local ko="$1"
    local ms
    ms=$(MVX_KNOCKOUT="$ko" python3 bench.py --timed-only --steps 10 --warmup 3 "${EXTRA[@]}" 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("%.3f" % d["ms_per_step"])')
    printf '%-40s %s ms/step\n' "${ko:-<complete step>}" "$ms"
}
run ""
for k in wgrad_bg lin_wgrad "wgrad_bg,lin_wgrad" gather_fwd gather_dgrad "gather_fwd,gather_dgrad" lin_fwd lin_dgrad bn_bwd_rows bn_bwd_grid bn_bwd_tiles bn_apply_rows bn_apply_cml sample sparse_out tap_sums \
         "bn_bwd_rows,bn_bwd_grid,bn_bwd_tiles,bn_apply_rows,bn_apply_cml" "lin_fwd,lin_dgrad,lin_wgrad"; do
    run "$k"
done
