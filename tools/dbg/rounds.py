"""Lane use of the eight-child walk by phase (library built with EXTRA=-DPT_DEBUG_ROUNDS).
usage: python tools/dbg/rounds.py config [spp]"""
import ctypes as C, sys
sys.path.insert(0, '.')
name = sys.argv[1]; spp = sys.argv[2] if len(sys.argv) > 2 else '2'
sys.argv = ['bench.py', '--config', name, '--spp', spp, '--steps', '1', '--warmup', '0', '--no-cpu', '--no-others']
import bench
from pathtracer_amd import api
bench.main()
out = (C.c_ulonglong * 24)()
assert api.lib().pt_debug_rounds(out) == 0
for k, tag in ((0, 'closest'), (8, 'visibility')):
    r, rn, ln, lb, rl, ll, ev, evl = [out[k + i] for i in range(8)]
    if r == 0: continue
    print('%s %s: rounds %d; node part ran in %.2f of them with %.1f lanes (of %.1f lanes that hold a ray); leaf part ran in %.2f with %.1f lanes'
          % (name, tag, r, rn / r, ln / max(1, rn), lb / r, rl / r, ll / max(1, rl)))
    print('   refill events %d (one per %.1f rounds), %.1f rays handed out per event' % (ev, r / max(1, ev), evl / max(1, ev)))
    a, b = (out[18], out[19]) if k else (out[16], out[17])
    print('   wave time after the work queue ran dry: %.3f of all wave time' % (a / max(1, b)))
