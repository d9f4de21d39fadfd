set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_all.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3_gpu_all.log; tail -6 gpurun_out/r3_gpu_all.log
[ $rc -eq 0 ] || exit 1
BENCH_NO_HOSTFED=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-acq --no-cpu > gpurun_out/r3_b3.json 2> gpurun_out/r3_b3.err; python -c "
import json; d=json.load(open('gpurun_out/r3_b3.json')); print(json.dumps(d['closed_loop'], indent=1))" | grep -v note
