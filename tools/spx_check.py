#!/usr/bin/env python3
"""Development (GPU box): foreign frames of big independent blocks through the stretch-parallel self-index (decode_spx.cuh): same bytes
as the input, which path ran, how long."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
eng = Engine(0); eng.set_timing(True)
rng = np.random.default_rng(3)
def mk(name, n):
    if name == "synth50": return np.concatenate([datagen.synth50(n & ~1023, 5), rng.integers(0, 256, n & 1023, dtype=np.uint8)])
    if name == "text": return datagen.synth_text(n, 7)
    if name == "structured": return np.frombuffer(datagen.structured(n, 9), dtype=np.uint8)
    if name == "noise": return rng.integers(0, 256, n, dtype=np.uint8)
    if name == "zeros": return np.zeros(n, dtype=np.uint8)
    if name == "mix":
        return np.concatenate([datagen.synth50((n // 4) & ~1023, 1), datagen.synth_text(n // 4, 2), rng.integers(0, 256, n // 4, dtype=np.uint8), np.zeros(n - 2 * (n // 4) - ((n // 4) & ~1023), dtype=np.uint8)])
bad = 0
for name in ("synth50", "text", "structured", "noise", "zeros", "mix"):
    for n in (9 << 20, (16 << 20) + 12345):
        for bsid in (5, 6, 7):
            data = mk(name, n)
            ref = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(bsid=bsid, indep=1))      # == liblz4's bytes
            dev = torch.from_numpy(np.frombuffer(ref + bytes(64), dtype=np.uint8).copy()).cuda()
            back = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
            eng.decompress_frame_async(dev, len(ref), back)
            r = eng.result(); t = eng.get_timing()
            ok = r.size == len(data) and back[:len(data)].cpu().numpy().tobytes() == data.tobytes()
            bad += not ok
            print("%-10s n=%9d bsid=%d path=%03x %s decode %.3f ms" % (name, n, bsid, r.flags >> 12, "ok" if ok else "MISMATCH", t["decode"]), flush=True)
# the headline shape: 1 GiB synth50, bare frame from this library's encoder
n = 1 << 30
src = synth50_device(n, 1234, "cuda")
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
eng.compress_async(src, frame, p); rc = eng.result()
back = torch.zeros(n, dtype=torch.uint8, device="cuda")
for it in range(3):
    back.zero_()
    eng.decompress_frame_async(frame, int(rc.size), back); r = eng.result(); t = eng.get_timing()
    print("1 GiB synth50 bare frame: path=%03x ok=%s walk %.3f decode %.3f (parse part %.3f, copy %.3f) finish %.3f ms" % (r.flags >> 12, bool(torch.equal(back, src)), t["walk"], t["decode"], t["decode_parse"], t["decode_copy"], t["finish"]), flush=True)
os.environ["X"] = "1"
print("mismatches:", bad)
sys.exit(1 if bad else 0)
