"""Start-up cost of a large scene with and without the binary scene cache:
OBJ text -> parse + both tree levels built  versus  cache file -> arrays.  usage: cache_timing.py [n]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import host, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
meshes, mats = scenes.dense_blob(n)
d = tempfile.mkdtemp()
obj = os.path.join(d, "blob.obj")
t0 = time.perf_counter(); scenes.write_obj(obj, meshes, mats); t_write = time.perf_counter() - t0
t0 = time.perf_counter(); s = host.Scene.from_obj(obj); s.export; t_parse = time.perf_counter() - t0
cache = os.path.join(d, "blob.ptscene")
t0 = time.perf_counter(); s.save_cache(cache); t_save = time.perf_counter() - t0
t0 = time.perf_counter(); c = host.Scene.from_cache(cache); t_load = time.perf_counter() - t0
print("triangles %d  obj %.1f MB  cache %.1f MB" % (scenes.count_tris(meshes), os.path.getsize(obj) / 1e6, os.path.getsize(cache) / 1e6))
print("parse + build %.2f s   save %.2f s   load %.3f s   (%.0fx)" % (t_parse, t_save, t_load, t_parse / max(t_load, 1e-9)))
