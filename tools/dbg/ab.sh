#!/bin/bash
# One line per large configuration (and the room): G rays/s, frame time, kernel times.  usage: tools/dbg/ab.sh [label] [configs...]
L=${1:-run}; shift
CFGS=${@:-"atrium260k blob7m terrain1m box1k"}
for c in $CFGS; do
  spp=""; [ "$c" = "atrium260k" ] && spp="--spp 32"; [ "$c" = "terrain1m" ] && spp="--spp 4"
  timeout -k 10 300 python bench.py --config $c $spp --steps 2 --warmup 1 --no-cpu --no-others | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$L', '$c', round(d['value']), 'Mrays/s', round(d['ms_per_step'],1), 'ms', {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'boxes/ray', round(r.get('node_boxes_per_ray') or 0,1), 'tris', round(r.get('tri_tests_per_ray') or 0,2), 'shadow boxes', round(r.get('shadow_node_boxes_per_ray') or 0,1))"
done
