set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_loop.py -x -q -m gpu > gpurun_out/r3_loop_b.log 2>&1; echo "default rc=$?" >> gpurun_out/r3_loop_b.log
GNSSCORR_LIB=tools/variants/lib_v2.so timeout -k 10 120 python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "tie_in_the_top or bench_configuration" > gpurun_out/r3_loop_v2.log 2>&1; echo "v2 rc=$?" >> gpurun_out/r3_loop_v2.log
tail -4 gpurun_out/r3_loop_b.log; tail -4 gpurun_out/r3_loop_v2.log
