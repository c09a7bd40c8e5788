#!/usr/bin/env python3
"""Development (GPU box): one compress of synth50 / text with an -DE1_DEBUG build (LZ4F_MI355X_LIB=...): pass E1's phase stamps on stderr."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
what = sys.argv[1] if len(sys.argv) > 1 else "s50"
n = 1 << 30
src = synth50_device(n, 1234, "cuda") if what == "s50" else torch.from_numpy(datagen.synth_text(64 << 20, 99)).cuda().repeat(16)
eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
for it in range(2):
    eng.compress_async(src, frame, p); r = eng.result(); t = eng.get_timing()
print(what, "e1 ms", round(t["find_matches"], 3), "ratio", round(n / r.size, 4))
