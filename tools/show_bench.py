"""Print the headline numbers of a bench.py JSON line."""
import json, sys
d = json.load(open(sys.argv[1]))
print("%.0f contours/s  %.3f ms/step  roofline %.1f (%.3f)" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"]))
try:
    print({k: round(v, 4) for k, v in d["stage_ms"].items() if not isinstance(v, str)})
except BrokenPipeError:
    pass
