"""Durations of the NTT pass kernels by launch shape, from a rocprofv3 rocpd database of a quotient run:
python3 tools/ntt_pass_times.py results.db   ->  kernel, grid, workgroup, LDS bytes: calls, average us"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
want = [c for c in ("name", "start", "end", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "lds_size", "lds_block_size", "group_segment_size") if c in cols]
acc = collections.defaultdict(list)
for row in db.execute("select %s from kernels" % ", ".join(want)):
    d = dict(zip(want, row))
    if "k_ntt_pass" not in d["name"]:
        continue
    key = (d["name"].split("(")[0].replace("void ps::", ""),) + tuple(d[c] for c in want[3:])
    acc[key].append((d["end"] - d["start"]) / 1e3)
print("columns:", want[3:])
for key, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s %-36s calls %4d  avg %8.1f us  total %9.1f us" % (key[0], str(key[1:]), len(v), sum(v) / len(v), sum(v)))
