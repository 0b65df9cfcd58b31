"""Per-kernel registers, scratch, static LDS and waves per SIMD from the compiler's resource remarks:
   (cd pathtracer_amd/csrc && make asm 2> /tmp/asm.log); python tools/kernel_resources.py /tmp/asm.log"""
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for r, d in zip(rows, dem):
    d = d.split("(")[0]
    print("%-64s vgpr %3s sgpr %3s scratch %4s static-lds %5s  waves/SIMD by registers %s" % (
        d, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
