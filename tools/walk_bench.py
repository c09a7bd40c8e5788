#!/usr/bin/env python3
"""Dev tool (GPU box): a device-resident frame decoded as a foreign one (dev_decompressFrame: header peek, size-word walk, decode), per-kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096) << 20
bsid = int(sys.argv[2]) if len(sys.argv) > 2 else 7
kind = sys.argv[3] if len(sys.argv) > 3 else "synth50"
if kind == "text": src = torch.from_numpy(datagen.synth_text(64 << 20, 99)).cuda().repeat(n // (64 << 20))
else: src = synth50_device(n, 1234)
eng = Engine(0); p = conduit.make_preferences(blockSizeID=bsid, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda"); back = torch.empty_like(src)
eng.set_timing(True)
inband = os.environ.get("INBAND") == "1"
if inband: frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda")
eng.compress_async(src, frame, p, inband=inband); r = eng.result()
for it in range(3):
    back.zero_()
    eng.decompress_frame_async(frame, int(r.size), back); r2 = eng.result(); t = eng.get_timing()
print("ok", bool(r2.size == n and torch.equal(back, src)), {k: round(v, 3) for k, v in t.items() if v})
