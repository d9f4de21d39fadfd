for a in 1 2; do
  GNSSCORR_TRK_NIT=$a timeout -k 10 120 python bench.py --steps 2 --warmup 1 --inner 8 --no-cpu --no-acq --loop-periods 0 2>/dev/null | tail -1 > gpurun_out/nit_$a.json
  python -c "import json; d=json.load(open('gpurun_out/nit_$a.json')); print('nit', $a, d['kernels_ms_per_launch'])"
done
