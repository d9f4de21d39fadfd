set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 bash tools/trk_trace.sh > gpurun_out/r3_pstrace2.log 2>&1; cat gpurun_out/r3_pstrace2.log
timeout -k 10 600 python -m pytest tests/test_gpu_tracking.py tests/test_gpu_loop.py -x -q -m gpu > gpurun_out/r3_trk_f.log 2>&1; rc=$?; tail -3 gpurun_out/r3_trk_f.log
[ $rc -eq 0 ] || exit 1
BENCH_NO_HOSTFED=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-acq > gpurun_out/r3_b4.json 2> gpurun_out/r3_b4.err; python -c "
import json; d=json.load(open('gpurun_out/r3_b4.json')); print(d['x_realtime'], d['kernels_ms_per_launch'], d['roofline']['frac'], d.get('integrity'), d.get('gpu_clocks')); print({k:(v['x_realtime'] if isinstance(v,dict) else v) for k,v in d['closed_loop'].items() if k!='note'}); print(d['cpu_baseline'])"
