set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_acq.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r3_acq_g.log 2>&1; rc=$?; tail -3 gpurun_out/r3_acq_g.log
[ $rc -eq 0 ] || exit 1
BENCH_NO_HOSTFED=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu --loop-periods 0 > gpurun_out/r3_b5.json 2> gpurun_out/r3_b5.err; python -c "
import json; d=json.load(open('gpurun_out/r3_b5.json')); print(d['x_realtime'], d['kernels_ms_per_launch'], d['roofline']['frac']); a=d['acquisition']; print(a['ms_per_32sv_search'], a['kernels_ms_per_search'], a['roofline']['frac'])"
