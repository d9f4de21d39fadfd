cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then export GNSSCORR_LIB=$PWD/tools/variants/lib_old.so; else unset GNSSCORR_LIB; fi
  BENCH_NO_HOSTFED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-acq --loop-periods 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', round(d['x_realtime'],1), {k: round(x,4) for k,x in d['kernels_ms_per_launch'].items()})"
done; done
