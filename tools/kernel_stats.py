"""Prints VGPR / SGPR / scratch / LDS per kernel from a hipcc -S assembly file."""
import re, sys
txt = open(sys.argv[1]).read()
for blk in re.findall(r"- \.agpr_count.*?\.wavefront_size: +\d+", txt, re.S):
    g = lambda k: re.search(r"\.%s: +(\S+)" % k, blk).group(1)
    print("%-90s vgpr %3s sgpr %3s scratch %4s lds %6s" % (g("name")[:90], g("vgpr_count"), g("sgpr_count"),
          g("private_segment_fixed_size"), g("group_segment_fixed_size")))
