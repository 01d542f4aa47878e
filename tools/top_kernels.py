"""Print per-kernel totals of a rocprofv3 results database (short names): python tools/top_kernels.py <db> [n]."""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for name, calls, total, avg, pct in con.execute("select * from top_kernels limit %d" % n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    print("%-40s calls %5d  avg %9.1f us  total %9.1f us  %5.1f %%" % ((m.group(1) if m else name[:40]), calls, avg, total / 1, pct))
