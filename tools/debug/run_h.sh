set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_all2.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3_gpu_all2.log; tail -4 gpurun_out/r3_gpu_all2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 bash tools/prof.sh h > gpurun_out/r3_prof_h.log 2>&1; tail -3 gpurun_out/r3_prof_h.log
timeout -k 10 500 bash tools/pmc_trk.sh > gpurun_out/pmc_trk_h.txt 2>&1; tail -25 gpurun_out/pmc_trk_h.txt
