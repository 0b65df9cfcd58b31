# rates of the three large configurations at their benchmark spp (and the 7 M scene at 4 spp)
mkdir -p gpurun_out
for cs in atrium260k:32 blob7m:16 blob7m:4 terrain1m:2; do
  cfg=${cs%%:*}; spp=${cs##*:}
  timeout -k 10 250 python bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu --no-count --no-others > gpurun_out/b3.json 2>gpurun_out/b3.err || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/b3.json').read().strip().splitlines()[-1]); print('$cfg', $spp, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})" | tee -a gpurun_out/bench3.txt
done
