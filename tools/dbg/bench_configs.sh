# rates and per-ray work of the four configurations at their benchmark spp (for A/B runs of a kernel change)
mkdir -p gpurun_out
for cs in box1k:64 atrium260k:32 blob7m:16 terrain1m:2; do
  cfg=${cs%%:*}; spp=${cs##*:}
  timeout -k 10 250 python bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu --no-others > gpurun_out/bc.json 2>gpurun_out/bc.err || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/bc.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$cfg', $spp, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, 'boxes/ray', round(r.get('node_boxes_per_ray',0),1), 'shadow boxes/ray', round(r.get('shadow_node_boxes_per_ray',0),1), 'shadow tris/ray', round(r.get('shadow_tri_tests_per_ray',0),2))" | tee -a gpurun_out/bench_configs.txt
done
