#!/bin/bash
# terrain1m at 16 spp with different numbers of samples in flight
for b in 1 2 4 0; do
  timeout -k 10 200 python bench.py --config terrain1m --spp 16 --batch-spp $b --steps 2 --warmup 1 --no-cpu --no-count --no-others | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b', round(d['value']), 'Mrays/s', round(d['ms_per_step'],1), 'ms', d['kernel_ms_per_step'])"
done
