"""Compact kernel timeline from a rocprofv3 rocpd database: every kernel longer than --min-ms inside
[t0 - before, t0 + after], where t0 is the start of the --nth launch of the kernel matching --anchor.
Consecutive launches of the same kernel on the same stream are merged.
Usage: python tools/rocpd_timeline.py results.db --anchor k_check_gates --nth -1 --before 1 --after 60"""
import argparse
import re
import sqlite3


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ps::", "").replace("ps::", "")[:34]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--anchor", default="k_check_gates")
    ap.add_argument("--nth", type=int, default=-1)
    ap.add_argument("--before", type=float, default=1.0, help="ms")
    ap.add_argument("--after", type=float, default=60.0, help="ms")
    ap.add_argument("--min-ms", type=float, default=0.25)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
    anchors = [r for r in rows if a.anchor in r[0]]
    t0 = anchors[a.nth][1]
    merged = []
    for name, s, e, st in rows:
        if s < t0 - a.before * 1e6 or s > t0 + a.after * 1e6:
            continue
        n = short(name)
        if merged and merged[-1][0] == n and merged[-1][3] == st:
            merged[-1][2] = e
            merged[-1][4] += e - s
            merged[-1][5] += 1
        else:
            merged.append([n, s, e, st, e - s, 1])
    for n, s, e, st, busy, cnt in merged:
        if busy / 1e6 >= a.min_ms:
            print(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f} ms  busy {busy / 1e6:7.3f}  x{cnt:<3d} s{st} {n}")


if __name__ == "__main__":
    main()
