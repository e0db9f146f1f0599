"""Where the quotient phase of one Groth16 proof goes: every kernel between a proof's k_check_gates and the first sort
kernel of its sums (all on the quotient's stream), grouped by kernel and launch shape, with the idle gaps between them.
  python3 tools/quotient_breakdown.py results.db [nth proof, default -1]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nth = int(sys.argv[2]) if len(sys.argv) > 2 else -1
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gcol = "grid_x" if "grid_x" in cols else "grid_size_x"
wcol = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
rows = list(db.execute(f"select name, start, end, stream_id, {gcol}, {wcol} from kernels order by start"))
short = lambda n: re.sub(r"\(.*", "", n).replace("void ps::", "").replace("ps::", "")
gates = [i for i, r in enumerate(rows) if "k_check_gates" in r[0]]
i0 = gates[nth]
qs = rows[i0][3]
i1 = next(i for i in range(i0, len(rows)) if "k_sort_count" in rows[i][0])
win = [r for r in rows[i0:i1] if r[3] == qs]
t0, t1 = win[0][1], win[-1][2]
busy = sum(r[2] - r[1] for r in win)
acc = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, st, g, w in win:
    a = acc[(short(n), g, w)]
    a[0] += 1
    a[1] += (e - s) / 1e3
gaps = [win[i + 1][1] - win[i][2] for i in range(len(win) - 1)]
print("quotient window %.3f ms: %d kernels, busy %.3f ms, gaps %.3f ms (median gap %.1f us)" % (
    (t1 - t0) / 1e6, len(win), busy / 1e6, sum(g for g in gaps if g > 0) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3))
for (n, g, w), (cnt, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-22s grid %8d wg %4d  x%3d  avg %7.1f us  total %8.1f us" % (n, g, w, cnt, us / cnt, us))
