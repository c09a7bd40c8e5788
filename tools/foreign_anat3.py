#!/usr/bin/env python3
"""Development (GPU box): the sequences of one block of the bench frame around a payload position (tools/foreign_anat3.py BLOCK POS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
B, POS = int(sys.argv[1]), int(sys.argv[2])
n = 4 << 30
src = synth50_device(n, 1234, "cuda"); eng = Engine(0)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
for inst in range(6):
    eng.compress_async(src, frame, p); rc = eng.result()
    pos = 7
    for b in range(B):
        w = int.from_bytes(frame[pos:pos + 4].cpu().numpy().tobytes(), "little"); pos += 4 + (w & 0x7FFFFFFF)
    w = int.from_bytes(frame[pos:pos + 4].cpu().numpy().tobytes(), "little"); csz = w & 0x7FFFFFFF
    blk = frame[pos + 4:pos + 4 + csz].cpu().numpy()
    q = 0; op = 0; seqs = []
    while q < csz:
        t = int(blk[q]); at = q; q += 1
        lit = t >> 4
        if lit == 15:
            while True:
                e = int(blk[q]); q += 1; lit += e
                if e != 255: break
        q += lit
        if q >= csz: break
        off = int(blk[q]) | (int(blk[q + 1]) << 8); q += 2
        ml = t & 15
        if ml == 15:
            while True:
                e = int(blk[q]); q += 1; ml += e
                if e != 255: break
        seqs.append((at, op, lit, ml + 4, off)); op += lit + ml + 4
    toks = set(x[0] for x in seqs)
    def walk(at, stop):
        q = at; hops = 0; trail = []
        while q < stop and q + 8 < csz and hops < 5000:
            t = int(blk[q]); q0 = q; q += 1
            lit = t >> 4
            if lit == 15:
                while q < csz:
                    e = int(blk[q]); q += 1; lit += e
                    if e != 255: break
            q += lit + 2
            if (t & 15) == 15:
                while q < csz:
                    e = int(blk[q]); q += 1
                    if e != 255: break
            hops += 1
            if len(trail) < 6: trail.append((q0, q - q0))
        return hops, q, trail
    for u in range(1, min((csz - 8192 - 2049) // 32768 + 1, 128)):
        a0 = u * 32768; cands = []; a = a0
        while a < a0 + 8192 and a + 1048 < csz and not cands:
            seg = blk[a:a + 1025]
            idx = np.nonzero(((seg[:-1] & 0xF0) == 0xF0) & (seg[1:] == 0xFF))[0]
            cands = [a + int(i) for i in idx[:6]]; a += 1024
        nxt = min([t for t in toks if t >= (u + 1) * 32768] or [csz])
        for c in cands:
            h, land, trail = walk(c, nxt)
            if h > 200: print("   lane %d candidate %d (true token: %s): %d hops to %d (next lane's region starts %d); first hops %s" % (u, c, c in toks, h, land, nxt, trail))
    near = [s for s in seqs if POS - 2000 <= s[0] <= POS + 36000]
    short = [s for s in near if s[2] < 100]
    print("instance %d block %d: csize %d, %d sequences, %d between payload %d and %d of which %d have < 100 literals" % (inst, B, csz, len(seqs), len(near), POS - 2000, POS + 36000, len(short)))
    if len(short) > 50:
        for s in near[:3] + short[:12]: print("   at %d out %d (tile %d + %d): lit %d mlen %d off %d" % (s[0], s[1], s[1] >> 16, s[1] & 65535, s[2], s[3], s[4]))
        x = src[B * (4 << 20) + short[0][1]: B * (4 << 20) + short[0][1] + 64].cpu().numpy()
        print("   input there:", x[:48].tobytes().hex())
        break
