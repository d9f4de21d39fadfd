for a in 0 1 2 3; do
  GNSSCORR_PLAN_DBG=$a timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu --no-acq --loop-periods 0 2>/dev/null | tail -1 > gpurun_out/pd_$a.json
  python -c "import json; d=json.load(open('gpurun_out/pd_$a.json')); print('plan dbg', $a, d['kernels_ms_per_launch'])"
done
