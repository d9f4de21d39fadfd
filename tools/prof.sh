#!/bin/bash
# rocprofv3 passes for profiles/: kernel stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r1}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu > gpurun_out/prof_${R}_bench.json 2> gpurun_out/prof_${R}_bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_${R}_fetch -- python3 bench.py --steps 1 --warmup 1 --acq-steps 2 --no-cpu > /dev/null 2> gpurun_out/prof_${R}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_${R}_write -- python3 bench.py --steps 1 --warmup 1 --acq-steps 2 --no-cpu > /dev/null 2> gpurun_out/prof_${R}_write.err
find gpurun_out -name "*.csv" | head -20
