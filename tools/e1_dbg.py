#!/usr/bin/env python3
"""Dev tool (GPU box): one small compress through the device API (argv: MiB, bsid, linked, kind)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0625
bsid = int(sys.argv[2]) if len(sys.argv) > 2 else 4
linked = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kind = sys.argv[4] if len(sys.argv) > 4 else "random"
n = int(mib * (1 << 20))
if kind == "random": data = np.random.default_rng(7).integers(0, 256, n, dtype=np.uint8)
elif kind == "text": data = datagen.synth_text(n, 99)
else: data = datagen.synth50(n, 1234)
src = torch.from_numpy(data).cuda(); eng = Engine(0)
p = conduit.make_preferences(blockSizeID=bsid, blockMode=0 if linked else 1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
print("launch", flush=True)
eng.compress_async(src, frame, p); r = eng.result()
print("size", r.size, "status", r.status, flush=True)
out, used = oracle.decompress_frame(frame[:r.size].cpu().numpy().tobytes(), cap=n + 64)
print("oracle roundtrip", out == data.tobytes(), "ratio %.4f" % (n / r.size))
