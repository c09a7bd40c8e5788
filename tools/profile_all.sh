#!/bin/bash
# GPU box: every rocprofv3 summary a round's numbers are checked against, from ONE tree:
#   GIT_HEAD=$(git rev-parse HEAD) gpurun -- "GIT_HEAD=$GIT_HEAD bash tools/profile_all.sh round4"     (then copy gpurun_out/round4_* into profiles/)
# headline: kernel stats + FETCH_SIZE / WRITE_SIZE (profile_round.sh) + instruction counts (pmc_sq.sh) + pass E1's wait / LDS counters;
# side legs foreign, cfg2, bck, linked, dense (text in 4 MiB blocks): kernel stats (prof_leg.sh); cfg2 and foreign also their counters (prof_pmc.sh).  Every JSON carries GIT_HEAD.
TAG=${1:-round4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GIT_HEAD=${GIT_HEAD:-unknown}
set -x
bash $R/tools/profile_round.sh $TAG > $R/gpurun_out/${TAG}_profile_round.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_profile_round.log; exit 1; }
bash $R/tools/pmc_sq.sh $TAG > $R/gpurun_out/${TAG}_sq.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_sq.log; exit 1; }
for leg in foreign cfg2 bck linked dense; do
  bash $R/tools/prof_leg.sh ${TAG}_$leg --legs $leg > $R/gpurun_out/${TAG}_${leg}_leg.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_${leg}_leg.log; exit 1; }
done
bash $R/tools/prof_pmc.sh ${TAG}_cfg2 --legs cfg2 > $R/gpurun_out/${TAG}_cfg2_pmc.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_cfg2_pmc.log; exit 1; }
bash $R/tools/prof_pmc.sh ${TAG}_foreign --legs foreign > $R/gpurun_out/${TAG}_foreign_pmc.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_foreign_pmc.log; exit 1; }
# the bench JSON lines of the runs under the profiler get the tree's name too
python3 - <<PY
import glob, json, os
for f in glob.glob("$R/gpurun_out/${TAG}*_bench_under_rocprof.json"):
    try:
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        d = json.loads(lines[-1]); d["git_head"] = os.environ.get("GIT_HEAD", "unknown")
        open(f, "w").write(json.dumps(d) + "\n")
    except Exception as e: print("skip", f, e)
PY
ls $R/gpurun_out | grep "^${TAG}" | head -40
