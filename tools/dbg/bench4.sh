# rates of the four configurations at their benchmark spp (and low-spp frames of the room and the 7 M scene)
mkdir -p gpurun_out
for cs in box1k:64 box1k:16 atrium260k:32 blob7m:16 blob7m:4 terrain1m:2; do
  cfg=${cs%%:*}; spp=${cs##*:}
  timeout -k 10 250 python bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu --no-others > gpurun_out/b4.json 2>gpurun_out/b4.err || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/b4.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$cfg', $spp, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'boxes/ray', round(r.get('node_boxes_per_ray',0),1), 'shadow boxes/ray', round(r.get('shadow_node_boxes_per_ray',0),1))" | tee -a gpurun_out/bench4.txt
done
