#!/usr/bin/env python3
"""Development (GPU box): text in 4 MiB independent blocks, framed by the oracle (= liblz4's bytes), bare frame, device-resident: what `lz4 -c` writes of
text by default (test/Main.hs:33-36).  Such a block is one match chain from its first byte to its last (61 % of the matches reach less than 4 KiB back), so a
block is a wave's work and the speed is the number of blocks in flight: 13 GiB/s for 1 GiB (256 blocks), 52 for 4 GiB, 90 for 8 GiB (NOTES_r4.md).
    tools/text_big_blocks.py [eighths of a GiB, default 8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import datagen
from lz4_frame_conduit_amd.device import Engine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tile = datagen.synth_text(128 << 20, 99)
fr = oracle.conduit_compress(tile.tobytes(), oracle.mkprefs(bsid=7, indep=1))
body = fr[7:-4]
frame = fr[:7] + body * reps + fr[-4:]
n = len(tile) * reps
dev = torch.from_numpy(np.frombuffer(frame + bytes(64), dtype=np.uint8).copy()).cuda()
back = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
eng = Engine(0); eng.set_timing(True)
best = None
for it in range(4):
    eng.decompress_frame_async(dev, len(frame), back); r = eng.result(); t = eng.get_timing()
    if it and (best is None or t["decompress_total"] < best): best = t["decompress_total"]
src = torch.from_numpy(tile).cuda()
ok = r.size == n and all(bool(torch.equal(back[i * len(tile):(i + 1) * len(tile)], src)) for i in range(reps))
print("%d MiB of text in 4 MiB independent blocks: decompress %.2f ms = %.1f GiB/s, ok=%s, path %s" % (n >> 20, best, n / 2**30 / (best / 1e3), ok, hex(int(r.flags) >> 12)))
