#!/usr/bin/env python3
"""Development (GPU box): pass E1's time by input size and tiles per workgroup (LZ4F_MI355X_E1_RUN), synth50 in 4 MiB independent blocks; each setting in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = int(sys.argv[1]) << 20
src = synth50_device(n, 1234, "cuda"); eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
best = None
for it in range(5):
    eng.compress_async(src, frame, p); r = eng.result(); t = eng.get_timing()
    if it and (best is None or t["find_matches"] < best): best = t["find_matches"]
print("RES %%.3f %%.4f" %% (best, n / r.size))
''' % ROOT
for mib in (64, 256, 1024, 2048):
    row = []
    for run in ("", "4", "16", "32", "64", "128"):
        env = dict(os.environ)
        if run: env["LZ4F_MI355X_E1_RUN"] = run
        r = subprocess.run([sys.executable, "-c", CHILD, str(mib)], env=env, capture_output=True, text=True, timeout=200)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES ")]
        row.append("%s: %s" % (run or "default", line[0][4:] if line else "FAILED"))
    print("%5d MiB  " % mib + "   ".join(row), flush=True)
