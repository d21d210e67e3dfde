#!/usr/bin/env python3
"""Summarise rocprofv3 output databases (rocpd sqlite) written by tools/gpu_prof.sh into a small text report.

usage: prof_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_]+|__amd_rocclr_[A-Za-z]+)", name)
    return m.group(1) if m else name[:40]


def main(folder):
    kt = glob.glob(os.path.join(folder, "kt", "*.db"))
    if kt:
        c = sqlite3.connect(kt[0])
        print("== rocprofv3 --kernel-trace --stats (top kernels) ==")
        print("%-28s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
        for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%-28s %8d %14.1f %12.2f %7.2f" % (short(name), calls, total, avg, pct))
        print()
    for db in sorted(glob.glob(os.path.join(folder, "pmc*", "*.db"))):
        c = sqlite3.connect(db)
        agg = defaultdict(lambda: defaultdict(float))
        dur = defaultdict(float)
        seen = set()
        for kname, cname, val, did, d in c.execute("select kernel_name,counter_name,value,dispatch_id,duration from counters_collection"):
            k = short(kname)
            agg[k][cname] += val
            if (did, k) not in seen:
                seen.add((did, k))
                dur[k] += d
        names = sorted({n for k in agg for n in agg[k]})
        print("== %s: hardware counters summed over all dispatches of a kernel ==" % os.path.basename(os.path.dirname(db)))
        print("%-24s %12s " % ("kernel", "dur_us") + " ".join("%22s" % n for n in names))
        for k in sorted(agg, key=lambda k: -dur[k]):
            if k.startswith("__amd"):
                continue
            print("%-24s %12.1f " % (k, dur[k] / 1e3) + " ".join("%22.4g" % agg[k].get(n, 0) for n in names))
        print()


if __name__ == "__main__":
    main(sys.argv[1])
