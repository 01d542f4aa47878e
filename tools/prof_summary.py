"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid, LDS) so that launches of one
template instantiation with different roles (residual block vs fusion vs prediction) are
told apart.  Usage: python tools/prof_summary.py <kernel_trace.csv> [min_calls]"""
import csv
import re
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if name.startswith("at::native") or name.startswith("__amd"):
            name = name[:60]
        key = (name, "%sx%sx%s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
                                   int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]), r["Grid_Size_Z"]),
               r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"])
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in rows.values())
print("%-34s %-12s %7s %5s %5s %7s %6s %12s %10s %6s" % ("kernel", "workgroups", "LDS", "vgpr", "agpr", "scratch",
                                                    "calls", "avg_us", "total_ms", "%"))
for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / total < 0.0005:
        continue
    print("%-34s %-12s %7s %5s %5s %7s %6d %12.1f %10.2f %6.2f" % (key[0][:34], key[1], key[2], key[3], key[4], key[5],
                                                           len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6,
                                                           100.0 * sum(v) / total))
