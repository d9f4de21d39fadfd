#!/bin/bash
# tracking tests, then the bench's per-kernel numbers
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r3_p_trk.log 2>&1 || { tail -30 gpurun_out/r3_p_trk.log; exit 1; }
tail -2 gpurun_out/r3_p_trk.log
timeout -k 10 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r3_p_bench.json 2> gpurun_out/r3_p_bench.err || { tail -20 gpurun_out/r3_p_bench.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r3_p_bench.json').read().strip().splitlines()[-1])
print(r['x_realtime'], r['ms_per_step'], r.get('kernels_ms_per_launch'))
print(r['roofline'])
print({k: (v.get('x_realtime') if isinstance(v, dict) else v) for k, v in r.get('closed_loop', {}).items()})
PY
