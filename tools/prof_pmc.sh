#!/bin/bash
# GPU box: SQ / LDS / TCC counters of one bench.py leg, per kernel (rocprofv3 --pmc passes of their own) -> gpurun_out/<tag>_pmc.json
#   bash tools/prof_pmc.sh round3_cfg2 --legs cfg2
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_pmc$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/prof_${TAG}_pmc$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, json, collections
R, TAG = "$R", "$TAG"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(f"{R}/gpurun_out/prof_{TAG}_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("lz4f::", "").replace("void ", "").split("(")[0]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: dict({c: acc[k][c] / cnt[k][c] for c in sorted(acc[k])}, launches_seen=max(cnt[k].values())) for k in acc}
for k, v in out.items():
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        v["hbm_bytes_corrected"] = int(2 * v.get("FETCH_SIZE", 0.0) * 1024 + v.get("WRITE_SIZE", 0.0) * 1024)
json.dump({"git_head": __import__("os").environ.get("GIT_HEAD", "unknown"), "note": "rocprofv3 --pmc (four passes: SQ waits, SQ instruction counts + LDS, FETCH_SIZE, WRITE_SIZE; --kernel-trace only) around bench.py %s; per-launch means over ALL launches of a kernel in the run "
                   "(the headline's launches included).  FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE reads half of wide coalesced reads on gfx950)" % "$*",
           "kernels": out}, open(f"{R}/gpurun_out/{TAG}_pmc.json", "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", 0))[:8]:
    print(k, {c: round(v / 1e6, 2) for c, v in out[k].items() if c != "launches_seen"}, "M")
PY
