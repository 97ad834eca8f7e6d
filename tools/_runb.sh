set -e
timeout -k 10 900 python -m pytest tests/test_frames_gpu.py tests/test_configs_gpu.py -x -q > gpurun_out/r05m_t1.log 2>&1 || { tail -40 gpurun_out/r05m_t1.log; exit 1; }
tail -2 gpurun_out/r05m_t1.log
for k in 0 1 0 1; do
  MVX_TAPS_ON_SIDE=$k python bench.py --timed-only --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hot side $k', d['value'], d['ms_per_step'])"
done
