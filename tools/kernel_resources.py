"""Print VGPR / SGPR / spill / LDS / occupancy per kernel of one .hip source (gfx950)."""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^ ]+ +(?:Function )?Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
    if m and "remark" in line:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: .*?    ([A-Za-z ]+(?:\[[^\]]+\])?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    print("%-28s vgpr %3d agpr %3d sgpr %3d spill %4d scratch %5d occ %d lds %6d" % (
        name[:28], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
        r.get("VGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1),
        r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
