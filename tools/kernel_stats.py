"""Prints VGPR / SGPR / scratch / static LDS per kernel from a hipcc -S assembly file (names demangled), with
the waves per SIMD the register count allows (512 registers per SIMD lane, allocated in steps of 8, at most 8
waves).  usage: kernel_stats.py pt_gpu.s"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows = []
for blk in re.findall(r"- \.agpr_count.*?\.wavefront_size: +\d+", txt, re.S):
    g = lambda k: re.search(r"\.%s: +(\S+)" % k, blk).group(1)
    rows.append((g("name"), int(g("vgpr_count")), int(g("sgpr_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (_, v, s, sc, lds), n in zip(rows, names):
    alloc = max(8, (v + 7) // 8 * 8)
    print("%-64s vgpr %3d sgpr %3d scratch %4d static-lds %5d  waves/SIMD by registers %d" % (n.split("(")[0][:64], v, s, sc, lds, min(8, 512 // alloc)))
