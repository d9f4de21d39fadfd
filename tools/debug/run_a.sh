set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/debug/loop_drift.py 350 > gpurun_out/r3_drift.log 2>&1; echo "drift rc=$?" >> gpurun_out/r3_drift.log
GNSSCORR_LIB=tools/variants/lib_inl.so timeout -k 10 120 python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "tie_in_the_top or bench_configuration" > gpurun_out/r3_loop_inl.log 2>&1; echo "inl rc=$?" >> gpurun_out/r3_loop_inl.log
tail -5 gpurun_out/r3_drift.log; tail -8 gpurun_out/r3_loop_inl.log
