mkdir -p gpurun_out
for t in 131072 262144 524288 1048576 2097152; do
  for spp in 4 16; do
    PT_TAIL=$t timeout -k 10 200 python bench.py --config blob7m --spp $spp --steps 3 --warmup 1 --no-cpu --no-count --no-others > gpurun_out/sw.json 2>gpurun_out/sw.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('tail', $t, 'spp', $spp, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/sweep_tail.txt
  done
done
