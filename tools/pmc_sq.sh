#!/bin/bash
# GPU box: scalar / vector instruction counts per kernel of the headline step (rocprofv3 --pmc, a pass of its own) -> gpurun_out/<tag>_sq.json
TAG=${1:-round2b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--headline-only --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/prof_${TAG}_sq.err || exit 1
python3 - <<PY
import csv, glob, json, collections
R, TAG = "$R", "$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(f"{R}/gpurun_out/prof_{TAG}_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("lz4f::", "").replace("void ", "").split("(")[0]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
json.dump({"git_head": __import__("os").environ.get("GIT_HEAD", "unknown"), "note": "rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR around bench.py --headline-only; per-launch means (wave instructions)", "kernels": out}, open(f"{R}/gpurun_out/{TAG}_sq.json", "w"), indent=1)
for k in ("k_find_matches", "k_emit_gather<4>", "k_copy_indexed<FzCfg<8> >", "k_parse_indexed"):
    if k in out: print(k, {c: round(v / 1e6, 1) for c, v in out[k].items()}, "M")
PY
