"""Hardware counters of a rocprofv3 --pmc rocpd database averaged per (kernel, grid, workgroup, LDS bytes):
  python3 tools/rocpd_pmc_by_shape.py results.db [name filter, default k_ntt]   (one --pmc pass of up to 8 SQ counters)"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "k_ntt"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name, grid, wg, lds, ctr, val, dur in db.execute(
        "select kernel_name, grid_size_x, workgroup_size_x, lds_block_size, counter_name, value, duration from counters_collection"):
    if flt not in name:
        continue
    k = (re.sub(r"\(.*", "", name).replace("void ps::", "").replace("ps::", ""), grid, wg, lds)
    a = acc[k][ctr]
    a[0] += 1
    a[1] += val
    d = acc[k]["duration_us"]
    d[0] += 1
    d[1] += dur / 1e3
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["duration_us"][1]):
    n = max(v[0] for kk, v in c.items() if kk != "duration_us")
    print("%-20s grid %8d wg %4d lds %6d  x%-4d %s" % (k[0], k[1], k[2], k[3], n, "  ".join(
        "%s %.4g" % (x.replace("SQ_", ""), c[x][1] / c[x][0]) for x in sorted(c))))
