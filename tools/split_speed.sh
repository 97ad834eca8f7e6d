#!/bin/bash
# Step rate of bench.py --convmath bf16x3 per choice of bf16x3 row GEMMs (MVX_ROW_SPLIT, modules/_hip.py row_split).
set -euo pipefail
out=gpurun_out/split_speed.txt
: > "$out"
for rows in "none" "dgrad" "fusion_768x768,dgrad" "fusion_768x768,fusion_128x768,dgrad" "fusion_768x768,conv1,rpn,dgrad" "fusion,vfe,conv1,rpn,dgrad"; do
  for mode in hot full; do
    v=$(MVX_ROW_SPLIT="$rows" timeout -k 10 200 python bench.py --mode $mode --convmath bf16x3 --timed-only --steps 20 --warmup 5 \
        | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['value'])")
    echo "$mode rows[$rows] $v frames/s" | tee -a "$out"
  done
done
