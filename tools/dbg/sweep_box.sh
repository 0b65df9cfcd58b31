var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  for spp in 64 16; do
    env $var=$v timeout -k 10 250 python bench.py --config box1k --spp $spp --steps 5 --warmup 2 --no-cpu --no-count --no-others > gpurun_out/sw.json 2>gpurun_out/sw.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('$var=$v', 'box1k', $spp, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})" | tee -a gpurun_out/sweep_box.txt
  done
done
