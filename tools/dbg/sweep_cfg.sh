# usage: sweep_cfg.sh VAR v1 v2 ... : rates and per-ray work of the four configurations under each value of an environment knob
var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  for cs in box1k:64 atrium260k:32 blob7m:16 terrain1m:2; do
    cfg=${cs%%:*}; spp=${cs##*:}
    env $var=$v timeout -k 10 250 python bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu --no-others > gpurun_out/bc.json 2>gpurun_out/bc.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/bc.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$var=$v', '$cfg', $spp, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'boxes/ray', round(r.get('node_boxes_per_ray',0),1), 'tris/ray', round(r.get('tri_tests_per_ray',0),2), 'shadow', round(r.get('shadow_node_boxes_per_ray',0),1), round(r.get('shadow_tri_tests_per_ray',0),2))" | tee -a gpurun_out/sweep_cfg.txt
  done
done
