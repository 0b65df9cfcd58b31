"""Seed files for hostcheck: two PNGs, a PPM, a textured OBJ with its MTL and map, a scene cache."""
import os, struct, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from pathtracer_amd import host, scenes

d = sys.argv[1]
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(3)


def png(path, rows, color_type, palette=None):
    h = len(rows)
    w = len(rows[0]) // {0: 1, 2: 3, 3: 1}[color_type]
    raw = b"".join(b"\x00" + bytes(r) for r in rows)

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
    if palette is not None:
        body += chunk(b"PLTE", palette)
    body += chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + body)


img = (rng.integers(0, 4, (24, 31, 3)) * 70).astype(np.uint8)
png(os.path.join(d, "seed.png"), [r.tobytes() for r in img], 2)
idx = rng.integers(0, 16, (20, 20)).astype(np.uint8)
png(os.path.join(d, "seed_pal.png"), [r.tobytes() for r in idx], 3, rng.integers(0, 256, 48).astype(np.uint8).tobytes())
open(os.path.join(d, "seed.ppm"), "wb").write(b"P6\n31 24\n255\n" + img.tobytes())
meshes, mats = scenes.box_room(3)
scenes.write_textured_obj(os.path.join(d, "seed.obj"), meshes, mats, {0: img})
host.Scene.from_obj(os.path.join(d, "seed.obj")).save_cache(os.path.join(d, "seed.ptscene"))
