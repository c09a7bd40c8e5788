#!/usr/bin/env python3
"""Development (GPU box): replay decode_spx.cuh's choice of lane starts on the host for one bench frame instance: per block, which 32 KiB segments
get no start (pairs 0xF? 0xFF in the first KiB that shows one, none of them a token) and how long the runs of such segments are."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 1 << 30
src = synth50_device(4 << 30, 1234, "cuda")[:n].contiguous(); eng = Engine(0)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
eng.compress_async(src, frame, p); rc = eng.result()
f = frame[:rc.size].cpu().numpy()
SEG, SCAN = 32768, 8192
pos = 7; b = 0; worst = (0, 0, 0); nostart_total = 0
while True:
    w = int.from_bytes(f[pos:pos + 4].tobytes(), "little"); pos += 4
    if w == 0: break
    csz = w & 0x7FFFFFFF
    if w >> 31: pos += csz; b += 1; continue
    blk = f[pos:pos + csz]
    # true token positions
    toks = {}; q = 0; prev = None
    while q < csz:
        t = int(blk[q]); toks[q] = None
        if prev is not None: toks[prev] = q
        prev = q; q += 1
        lit = t >> 4
        if lit == 15:
            while True:
                e = int(blk[q]); q += 1; lit += e
                if e != 255: break
        q += lit
        if q >= csz: break
        q += 2
        if (t & 15) == 15:
            while True:
                e = int(blk[q]); q += 1
                if e != 255: break
    pair = ((blk[:-1] & 0xF0) == 0xF0) & (blk[1:] == 0xFF)
    nl = min((csz + SEG - 1) // SEG, 128)
    run = 0
    for u in range(1, nl):
        a0 = u * SEG; cands = []
        a = a0
        while a < a0 + SCAN and a + 1048 < csz and not cands:
            idx = np.nonzero(pair[a:a + 1024])[0]
            cands = [a + int(i) for i in idx[:6]]
            a += 1024
        def likely(c):                           # spx_likely_token: a true token whose successor announces a long literal run (or is the end)
            if c not in toks: return False
            nx = toks[c]
            return nx is None or nx + 1 >= csz or ((int(blk[nx]) & 0xF0) == 0xF0 and int(blk[nx + 1]) == 0xFF)
        ok = any(likely(c) for c in cands)
        if cands and not ok:
            run += 1; nostart_total += 1
            if run > worst[0]: worst = (run, b, u)
        else: run = 0
    pos += csz; b += 1
print("blocks", b, "segments with pairs but no true token among the first six of their first KiB with one:", nostart_total, "longest run of them:", worst)
