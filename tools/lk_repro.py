#!/usr/bin/env python3
"""Dev tool (GPU box): linked frames that start with a long incompressible run - which lengths does the window kernel get wrong?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
eng = Engine(0)
rng = np.random.default_rng(1)
tail = np.frombuffer(datagen.structured(2 << 20, 77), dtype=np.uint8)
for bsid in (5,):
    for L in [2 * 262144 + x for x in (64603, 1000, 20000, 40000, 60000, 62000, 63000, 64000, 65000, 65536, 66000, 68000, 70000, 90000, 130000, 131072, 140000, 200000)] + [262144 + 64603, 3 * 262144 + 64603, 23 * 262144 + 64603]:
        data = np.concatenate([rng.integers(0, 256, L, dtype=np.uint8), tail]).copy()
        p = conduit.make_preferences(blockSizeID=bsid, blockMode=0)
        bs = 1 << (8 + 2 * bsid); src = torch.from_numpy(data).cuda(); nb = (src.numel() + bs - 1) // bs
        frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda"); table = eng.new_table(nb)
        eng.compress_async(src, frame, p, table); r = eng.result()
        back = torch.zeros_like(src)
        eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo); r2 = eng.result()
        ok = r2.size == src.numel() and bool(torch.equal(back, src))
        if not ok:
            d = (back != src).nonzero().flatten()
            print("bsid", bsid, "L", L, "WRONG: first", int(d[0]), "last", int(d[-1]), "count", int(d.numel()), flush=True)
        else: print("bsid", bsid, "L", L, "ok", flush=True)
