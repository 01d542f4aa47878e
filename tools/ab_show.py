"""Print value / stage times of the bench lines under gpurun_out/<dir> (A/B runs of library variants)."""
import glob, json, sys
for f in sorted(glob.glob("gpurun_out/%s/*.json" % sys.argv[1])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, e)
        continue
    st = d.get("stage_ms", {})
    print("%-40s %8.0f %7.3f  %s" % (f.split("/")[-1], d["value"], d["ms_per_step"],
                                      {k[:6]: round(v, 4) for k, v in st.items() if isinstance(v, float) and v > 0.05}))
