#!/usr/bin/env python3
"""Dev tool (GPU box): soak of the indexed decode path - random structured / synth50 / text inputs, random sizes and block
sizes, every frame decoded with and without the index and compared with the source.  argv: cases [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = Engine(0)
used = bad = 0
t0 = time.time()
for i in range(cases):
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(1, 40 << 20))
    if kind == 0: data = np.frombuffer(datagen.structured(n, int(rng.integers(1 << 30))), dtype=np.uint8).copy()
    elif kind == 1: data = datagen.synth50(max(1024, n & ~1023), int(rng.integers(1 << 30)))[:n]
    elif kind == 2: data = datagen.synth_text(n, int(rng.integers(1 << 30)))
    else:
        parts = []
        while sum(len(p) for p in parts) < n:
            m = int(rng.integers(1, 3 << 20))
            parts.append(rng.integers(0, 256, m, dtype=np.uint8) if rng.integers(0, 2) else np.frombuffer(datagen.structured(m, int(rng.integers(1 << 30))), dtype=np.uint8))
        data = np.concatenate(parts)[:n].copy()
    bsid = int(rng.integers(5, 8))
    p = conduit.make_preferences(blockSizeID=bsid, blockMode=1, blockChecksum=int(rng.integers(0, 2)))
    bs = 1 << (8 + 2 * bsid)
    src = torch.from_numpy(data).cuda(); nb = (src.numel() + bs - 1) // bs
    frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
    table = eng.new_table(nb)
    index = torch.zeros(eng.index_size(src.numel(), p) * int(rng.choice([1, 1, 4, 16])), dtype=torch.uint8, device="cuda")
    eng.compress_async(src, frame, p, table, index); r = eng.result()
    used += int(index[:4].cpu().numpy().view(np.uint32)[0] == 0x3258494C)
    for ix in (index, None, index):
        back = torch.empty_like(src)
        eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo, ix); r2 = eng.result()
        if r2.size != src.numel() or not torch.equal(back, src):
            bad += 1; print("MISMATCH case", i, "kind", kind, "n", n, "bsid", bsid, "indexed", ix is not None, flush=True)
    if i % 10 == 9: print("case", i + 1, "usable indexes", used, "bad", bad, "%.0f s" % (time.time() - t0), flush=True)
print("done: cases", cases, "usable indexes", used, "mismatches", bad)
sys.exit(1 if bad else 0)
