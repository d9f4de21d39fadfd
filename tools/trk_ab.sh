#!/bin/bash
python -m pytest tests/test_gpu_tracking.py tests/test_gpu_symbols.py -x -q -m gpu 2>&1 | tail -8 || exit 1
for nit in 1 2; do echo "prefix NIT=$nit"; GNSSCORR_TRK_NIT=$nit python bench.py --steps 30 --warmup 3 --no-cpu --no-acq 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms_per_step'], d['x_realtime'])"; done
echo replica; GNSSCORR_TRK_ALGO=replica python bench.py --steps 30 --warmup 3 --no-cpu --no-acq 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms_per_step'], d['x_realtime'])"
