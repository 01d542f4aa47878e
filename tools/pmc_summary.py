"""Join rocprofv3 counter_collection.csv with kernel_trace.csv: per (kernel, grid) averages of
each counter, kernel duration, and derived clock / MFMA utilisation."""
import csv
import re
import sys
from collections import defaultdict

cc, kt = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(cc)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if name.startswith("at::") or name.startswith("__amd"):
        continue
    key = (name[:30], "%d" % (int(r["Grid_Size"]) // int(r["Workgroup_Size"])))
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
for key in acc:
    n = len(cnt[key])
    d = sum(dur[i] for i in cnt[key] if i in dur) / max(1, n)
    vals = {k: v / n for k, v in acc[key].items()}
    line = "%-30s %-8s n=%3d dur_us=%9.1f " % (key[0], key[1], n, d / 1e3)
    if "GRBM_GUI_ACTIVE" in vals and d:
        line += "clk_GHz=%.2f " % (vals["GRBM_GUI_ACTIVE"] / 8 / d)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
        line += "mfma_util=%.1f%% " % (100 * vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8 * 1024))
    line += " ".join("%s=%.3g" % (k, v) for k, v in sorted(vals.items()))
    print(line)
