#!/bin/bash
# Usage (on the GPU box): bash tools/gpu_launch_trace.sh <tag> <workload> <spp>   -- every kernel launch of one render in order, with its duration
set -u
TAG=$1; WL=$2; SPP=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py --workload $WL --spp $SPP --traversal fast --repeat 2 > $OUT/kt.log 2>&1 || exit 1
python3 - $OUT > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_launches.txt <<'PY'
import glob, re, sqlite3, sys
db = glob.glob(sys.argv[1] + "/kt/*.db")[0]
c = sqlite3.connect(db)
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = [n for n in names if n == "kernels"] or [n for n in names if "kernel" in n.lower()]
print("# tables:", names)
cols = [r[1] for r in c.execute("pragma table_info(%s)" % view[0])]
print("# columns of", view[0], cols)
rows = list(c.execute("select name, start, end from %s order by start" % view[0]))
t0 = rows[0][1]
half = len(rows) // 2
for name, s, e in rows[half:]:      # the second repeat
    m = re.search(r"(k_[a-z_]+)", name)
    print("%-22s start %10.3f ms  dur %9.3f ms" % (m.group(1) if m else name[:22], (s - t0) / 1e6, (e - s) / 1e6))
PY
rm -rf $OUT
