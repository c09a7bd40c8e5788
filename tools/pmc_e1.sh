#!/bin/bash
# GPU box: what pass E1's waves wait for - LDS pipeline occupancy and bank conflicts, issue stalls, per launch of every kernel of the
# headline step (rocprofv3 --pmc in passes of their own) -> gpurun_out/<tag>_e1pmc.json
TAG=${1:-round3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--headline-only --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 -L > $R/gpurun_out/${TAG}_counters_list.txt 2>&1
i=0
for SET in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_e1pmc$i -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/prof_${TAG}_e1pmc$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, json, collections
R, TAG = "$R", "$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(f"{R}/gpurun_out/prof_{TAG}_e1pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("lz4f::", "").replace("void ", "").split("(")[0]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: {c: acc[k][c] / cnt[k][c] for c in sorted(acc[k])} for k in acc}
json.dump({"note": "rocprofv3 --pmc (three passes) around bench.py --headline-only; per-launch means", "kernels": out}, open(f"{R}/gpurun_out/{TAG}_e1pmc.json", "w"), indent=1)
for k in ("k_find_matches", "k_emit_gather<4>", "k_copy_indexed<FzCfg<8> >"):
    if k in out: print(k, {c: round(v / 1e6, 2) for c, v in out[k].items()}, "M")
PY
