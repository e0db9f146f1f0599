"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats` writes `<name>_results.db` on ROCm 7.2), as CSV in the column
layout of rocprofv3's kernel_stats.csv.  Usage: python tools/rocpd_stats.py results.db > stats.csv"""
import collections
import csv
import sqlite3
import statistics
import sys


def main(path):
    db = sqlite3.connect(path)
    durs = collections.defaultdict(list)
    for name, start, end in db.execute("select name, start, end from kernels"):
        durs[name].append(end - start)
    total = sum(sum(v) for v in durs.values()) or 1
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / total, 4), min(v), max(v),
                    statistics.pstdev(v) if len(v) > 1 else 0.0])


if __name__ == "__main__":
    main(sys.argv[1])
