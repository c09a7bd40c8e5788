#!/usr/bin/env python3
"""Development (GPU box): device-resident decode of FOREIGN frames (the oracle's = liblz4's bytes, no trailer) by input size, data class and framing - a look for cliffs
(a decoder chosen wrongly for small inputs shows as a step in the GiB/s column)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import datagen
from lz4_frame_conduit_amd.device import Engine
eng = Engine(0); eng.set_timing(True)
big = {"synth50": datagen.synth50(256 << 20, 7), "text": datagen.synth_text(256 << 20, 8)}
for name, data in big.items():
    for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=1), dict(bsid=4, indep=0)):
        row = []
        for mib in (1, 4, 12, 32, 96, 256):
            d = data[:mib << 20]
            fr = oracle.conduit_compress(d.tobytes(), oracle.mkprefs(**kw))
            dev = torch.from_numpy(np.frombuffer(fr + bytes(64), dtype=np.uint8).copy()).cuda()
            back = torch.zeros(len(d) + 64, dtype=torch.uint8, device="cuda")
            best = None
            for it in range(3):
                eng.decompress_frame_async(dev, len(fr), back); r = eng.result(); t = eng.get_timing()["decompress_total"]
                best = t if best is None or t < best else best
            ok = r.size == len(d) and bool(torch.equal(back[:len(d)], torch.from_numpy(d).cuda()))
            row.append("%d MiB %.2f ms %.1f GiB/s%s" % (mib, best, len(d) / 2**30 / (best / 1e3), "" if ok else " WRONG"))
        print("%-8s bsid %d %s: " % (name, kw["bsid"], "independent" if kw["indep"] else "linked") + " | ".join(row), flush=True)
