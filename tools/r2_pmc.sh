#!/bin/bash
# dev: SQ / LDS counters of the compress kernels (each --pmc set is its own run)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_e1a -- python3 $R/tools/quick_bench.py 1024 > $R/gpurun_out/pmc_e1a.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_e1b -- python3 $R/tools/quick_bench.py 1024 > $R/gpurun_out/pmc_e1b.log 2>&1
ls -R $R/gpurun_out/pmc_e1a | head; 
python3 - <<PY
import csv, glob, collections
for d in ("pmc_e1a", "pmc_e1b"):
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "find_matches" in k or "emit" in k:
                print(d, k, {c: "%.3g" % x for c, x in v.items()})
PY
