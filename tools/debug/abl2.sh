for a in 0 4 8 12; do
  BENCH_NO_HOSTFED=1 GNSSCORR_TRK_ABLATE=$a timeout -k 10 120 python bench.py --steps 2 --warmup 1 --inner 8 --no-cpu --no-acq --loop-periods 0 2>/dev/null | tail -1 > gpurun_out/abl_$a.json
  python -c "import json; d=json.load(open('gpurun_out/abl_$a.json')); print('ablate', $a, d['kernels_ms_per_launch']['trk_corr'])"
done
