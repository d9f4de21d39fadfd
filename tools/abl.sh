python -m pytest tests/test_gpu_tracking.py -m gpu -q -x 2>&1 | tail -2
for nit in 2 4; do echo "nit=$nit"; GNSSCORR_TRK_NIT=$nit python bench.py --steps 10 --warmup 2 --no-cpu --no-acq 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['x_realtime'], d['roofline']['frac'], d['kernels_ms_per_step'])"; done
