#!/bin/bash
# end-of-round record: GPU suite, default bench, rocprofv3 stats + FETCH/WRITE passes (tools/prof.sh), SQ counters of the correlator
cd /root/repo
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/q_tests.log 2>&1 || { tail -30 gpurun_out/q_tests.log; exit 1; }
tail -2 gpurun_out/q_tests.log
timeout -k 10 400 python bench.py > gpurun_out/q_bench_default.json 2> gpurun_out/q_bench_default.err || { tail -20 gpurun_out/q_bench_default.err; exit 1; }
tail -c 600 gpurun_out/q_bench_default.json; echo
timeout -k 10 500 bash tools/prof.sh q > gpurun_out/q_prof.log 2>&1 || { tail -20 gpurun_out/q_prof.log; exit 1; }
timeout -k 10 400 bash tools/pmc_trk.sh > gpurun_out/q_pmc_sq_trk_corr.txt 2>&1 || { tail -20 gpurun_out/q_pmc_sq_trk_corr.txt; exit 1; }
tail -20 gpurun_out/q_pmc_sq_trk_corr.txt
