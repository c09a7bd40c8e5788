#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of one bench.py leg -> gpurun_out/<tag>_kernel_stats.csv (names shortened)
#   bash tools/prof_leg.sh round3_foreign --headline-only --foreign
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_kt -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${TAG}_kt.err || { tail -5 $R/gpurun_out/prof_${TAG}_kt.err; exit 1; }
python3 - <<PY
import csv, glob
R, TAG = "$R", "$TAG"
for f in glob.glob(f"{R}/gpurun_out/prof_{TAG}_kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    with open(f"{R}/gpurun_out/{TAG}_kernel_stats.csv", "w", newline="") as o:
        w = csv.writer(o); w.writerow(rows[0])
        for r in rows[1:]: w.writerow([r[0].replace("lz4f::", "").replace("void ", "").split("(")[0]] + r[1:])
PY
head -n 16 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
