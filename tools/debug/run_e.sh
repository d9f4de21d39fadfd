set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dist_gloo.py tests/test_gpu_loop.py -x -q -m gpu > gpurun_out/r3_gpu_e.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3_gpu_e.log; tail -8 gpurun_out/r3_gpu_e.log
[ $rc -eq 0 ] || exit 1
GNSSCORR_LIB=tools/variants/lib_tailprof.so timeout -k 10 200 python tools/debug/tail_prof.py > gpurun_out/r3_tailprof2.log 2>&1; cat gpurun_out/r3_tailprof2.log
BENCH_NO_HOSTFED=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-acq --no-cpu > gpurun_out/r3_b2.json 2> gpurun_out/r3_b2.err; python -c "
import json; d=json.load(open('gpurun_out/r3_b2.json')); print(json.dumps(d['closed_loop'], indent=1))" | grep -v note
