#!/usr/bin/env python3
"""Development (GPU box): two halves of the bench input as two frames, decoded one after the other on one stream and side by side on two
engines with a stream each: is there anything to gain from overlapping one half's parse kernels with the other half's copy kernel?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 2 << 30
src = synth50_device(2 * n, 1234, "cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
eA, eB = Engine(0, sA), Engine(0, sB)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
fr, sz, back = [], [], []
for i, e in enumerate((eA, eB)):
    f = torch.empty(e.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(e.stream): e.compress_async(src[i * n:(i + 1) * n], f, p, inband=True)
    sz.append(int(e.result().size)); fr.append(f); back.append(torch.empty(n, dtype=torch.uint8, device="cuda"))
def run(mode):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "one after the other":
        for i in range(2):
            with torch.cuda.stream(sA): eA.decompress_frame_async(fr[i], sz[i], back[i])
            eA.stream.synchronize()
    else:
        with torch.cuda.stream(sA): eA.decompress_frame_async(fr[0], sz[0], back[0])
        with torch.cuda.stream(sB): eB.decompress_frame_async(fr[1], sz[1], back[1])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for mode in ("one after the other", "side by side"):
    ts = [run(mode) for _ in range(8)]
    print("%-20s best %.3f ms median %.3f" % (mode, min(ts), sorted(ts)[4]))
print("ok", bool(torch.equal(back[0], src[:n]) and torch.equal(back[1], src[n:])))
