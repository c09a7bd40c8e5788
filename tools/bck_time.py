#!/usr/bin/env python3
"""Development (GPU box): the headline step with block checksums: whole-call times, and the pieces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 4 << 30
src = synth50_device(n, 1234, "cuda"); eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=7, blockMode=1, blockChecksum=1)
frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda"); back = torch.zeros(n, dtype=torch.uint8, device="cuda")
for inband in (True, False):
    best = None
    for it in range(4):
        eng.compress_async(src, frame, p, inband=inband); rc = eng.result(); tc = eng.get_timing()
        eng.decompress_frame_async(frame, int(rc.size), back); r = eng.result(); td = eng.get_timing()
        tot = tc["compress_total"] + td["decompress_total"]
        if best is None or tot < best[0]: best = (tot, tc["compress_total"], td["decompress_total"], tc["xxh32_write"], td["xxh32_verify"], td["walk"], td["decode"])
    print("block checksums on, %s: step %.3f ms = compress %.3f + decompress %.3f (xxh32 write %.3f, verify %.3f beside walk %.3f + decode %.3f) ok=%s" % ("in-band index" if inband else "bare frame", *best, bool(torch.equal(back, src))))
