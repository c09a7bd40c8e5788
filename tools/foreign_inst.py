#!/usr/bin/env python3
"""Development (GPU box): the foreign 4 GiB frame, a NEW instance per round (the encoder's waves race: every compress gives other bytes):
decode time per instance (best of 3) - how much does the stretch finder's time depend on the instance?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 4 << 30
src = synth50_device(n, 1234, "cuda"); eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
back = torch.zeros(n, dtype=torch.uint8, device="cuda")
for inst in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    eng.compress_async(src, frame, p); rc = eng.result()
    best = None
    for it in range(3):
        eng.decompress_frame_async(frame, int(rc.size), back); r = eng.result(); t = eng.get_timing()
        v = (t["decompress_total"], t["walk"], t["decode"], t["decode_parse"], t["decode_copy"])
        if best is None or v < best: best = v
    print("instance %2d size %d: total %.3f walk %.3f decode %.3f (parse kernels %.3f copy %.3f -> in front of them %.3f) ok=%s" % (inst, rc.size, *best, best[2] - best[3] - best[4], bool(torch.equal(back, src))), flush=True)
