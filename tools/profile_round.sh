#!/bin/bash
# GPU box: the rocprofv3 summaries this round's numbers come from -> gpurun_out/prof_<tag>/ and profiles/<tag>_*.
#   bash tools/profile_round.sh round2        (then copy gpurun_out/<tag>_* into profiles/ and commit)
# Three runs of `bench.py --headline-only`: --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE (counters in passes of
# their own, no other tracing: MI355X_MICROARCH.md, rocprofv3 PMC slots).
TAG=${1:-round2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--headline-only --steps 4 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_kt -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${TAG}_kt.err || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/prof_${TAG}_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py $ARGS > /dev/null 2> $R/gpurun_out/prof_${TAG}_write.err || exit 1
python3 - <<PY
import csv, glob, json, collections, os
R, TAG = "$R", "$TAG"
def short(n):
    n = n.replace("lz4f::", "").replace("void ", "")
    return n.split("(")[0]
# kernel stats (names shortened)
for f in glob.glob(f"{R}/gpurun_out/prof_{TAG}_kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    with open(f"{R}/gpurun_out/{TAG}_kernel_stats.csv", "w", newline="") as o:
        w = csv.writer(o); w.writerow(rows[0])
        for r in rows[1:]: w.writerow([short(r[0])] + r[1:])
def counters(d):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}
fe, wr = counters(f"prof_{TAG}_fetch"), counters(f"prof_{TAG}_write")
kernels = {}
for k in sorted(set(fe) | set(wr)):
    f_raw = fe.get(k, 0.0) * 1024; w_b = wr.get(k, 0.0) * 1024
    kernels[k] = {"fetch_bytes_raw": int(f_raw), "fetch_bytes_x2": int(2 * f_raw), "write_bytes": int(w_b), "hbm_bytes_corrected": int(2 * f_raw + w_b)}
groups = {"find_matches": [k for k in kernels if k.startswith("k_find_matches")], "emit": [k for k in kernels if k.startswith("k_emit_gather")],
          "decode": [k for k in kernels if k.split("<")[0] in ("k_check_index", "k_parse_indexed", "k_resolve_direct", "k_copy_indexed", "k_copy_selffed", "k_dense_gate", "k_finish_decode")]}
json.dump({"git_head": os.environ.get("GIT_HEAD", "unknown"), "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around `python3 bench.py --headline-only --steps 4 --warmup 2 --no-cpu-baseline`; "
                   "per-launch means.  Units: FETCH_SIZE/WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reads exactly 1/2 of the bytes of wide (16 B/lane) "
                   "coalesced streaming reads -> read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16 B/lane stores.  k_find_matches reads its input by LDS-DMA, 16 B per lane (the calibrated shape).",
           "workload": "4 GiB synth50, 4 MiB independent blocks, 1 GPU, decode from the stream alone (in-band trailer)", "kernels": kernels, "groups": groups},
          open(f"{R}/gpurun_out/{TAG}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: kernels[k] for g in groups.values() for k in g}, indent=1))
PY
tail -n 30 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
