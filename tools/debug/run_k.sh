#!/bin/bash
# planner v4: tracking tests (plain and with the chain's checks on), the chain alone, then the bench
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_tracking.py -x -q > gpurun_out/r3_k_trk.log 2>&1 || { tail -30 gpurun_out/r3_k_trk.log; exit 1; }
tail -3 gpurun_out/r3_k_trk.log
GNSSCORR_PLAN_VERIFY=1 timeout -k 10 300 python -m pytest tests/test_gpu_tracking.py -x -q > gpurun_out/r3_k_trkv.log 2>&1 || { tail -30 gpurun_out/r3_k_trkv.log; exit 1; }
tail -3 gpurun_out/r3_k_trkv.log
GNSSCORR_LIB=$PWD/tools/variants/lib_planprof.so NCH=4 timeout -k 10 100 python tools/debug/plan_alone.py 2>&1 | tail -5 || exit 1
BENCH_PLAN_STATS=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r3_k_bench.json 2> gpurun_out/r3_k_bench.err || { tail -20 gpurun_out/r3_k_bench.err; exit 1; }
grep "planner" gpurun_out/r3_k_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r3_k_bench.json').read().strip().splitlines()[-1])
print(r['x_realtime'], r['ms_per_step'], r.get('kernels_ms_per_launch'))
PY
