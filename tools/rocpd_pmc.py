"""Per-kernel averages of hardware counters from one or more rocprofv3 --pmc rocpd databases
(one counter per pass, as the HBM section of the MI355X guide prescribes), as JSON:
  python tools/rocpd_pmc.py fetch.db write.db > pmc.json"""
import collections
import json
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*", "", name)


def main(paths):
    out = collections.defaultdict(dict)
    for path in paths:
        db = sqlite3.connect(path)
        acc = collections.defaultdict(lambda: [0, 0.0])
        for kernel, counter, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
            a = acc[(counter, short(kernel))]
            a[0] += 1
            a[1] += float(value)
        for (counter, kernel), (cnt, total) in acc.items():
            out[counter][kernel] = {"launches": cnt, "avg": total / cnt}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
