set -e
bash tools/prof_mode.sh r05s_vfe --mode vfe
python3 tools/timeline.py gpurun_out/prof_r05s_vfe/stats FusedOptimizer 1 > gpurun_out/prof_r05s_vfe/timeline_q1.txt 2>&1 || true
