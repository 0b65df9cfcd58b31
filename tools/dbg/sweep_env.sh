# usage: sweep_env.sh VAR v1 v2 ... : rates of the large configurations under each value of an environment knob
var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  for cs in atrium260k:32 blob7m:16 blob7m:4; do
    cfg=${cs%%:*}; spp=${cs##*:}
    env $var=$v timeout -k 10 250 python bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu --no-count --no-others > gpurun_out/sw.json 2>gpurun_out/sw.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('$var=$v', '$cfg', $spp, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})" | tee -a gpurun_out/sweep_env.txt
  done
done
