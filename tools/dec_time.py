#!/usr/bin/env python3
"""Development (GPU box): the headline decode (in-band index) of 4 GiB synth50: parse / copy ms per library variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 4 << 30
src = synth50_device(n, 1234, "cuda"); eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda")
eng.compress_async(src, frame, p, inband=True); rc = eng.result()
back = torch.zeros(n, dtype=torch.uint8, device="cuda")
best = None
for it in range(6):
    eng.decompress_frame_async(frame, int(rc.size), back); r = eng.result(); t = eng.get_timing()
    if best is None or t["decompress_total"] < best[0]: best = (t["decompress_total"], t["decode_parse"], t["decode_copy"])
print("%s: decode total %.3f ms (parse kernels %.3f, copy %.3f) ok=%s" % (os.environ.get("LZ4F_MI355X_LIB", "base").split("lib_")[-1], *best, bool(torch.equal(back, src))))
