#!/usr/bin/env python3
"""Dev tool (GPU box): the gather emit kernel must write the same frame and the same index as the record-at-a-time one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(3)
eng = Engine(0); eng.set_timing(True)
bad = 0
def one(src, bsid, indep, label):
    global bad
    p = conduit.make_preferences(blockSizeID=bsid, blockMode=indep, blockChecksum=int(rng.integers(0, 2)))
    bs = 1 << (8 + 2 * bsid); nb = (src.numel() + bs - 1) // bs
    out = []
    for serial in (True, False):
        for k in ("LZ4F_MI355X_EMIT_SERIAL", "LZ4F_MI355X_LAYOUT_SERIAL"):
            if serial: os.environ[k] = "1"
            else: os.environ.pop(k, None)
        frame = torch.zeros(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
        table = eng.new_table(nb); index = torch.zeros(eng.index_size(src.numel(), p) * 16, dtype=torch.uint8, device="cuda")
        eng.compress_async(src, frame, p, table, index); r = eng.result()
        out.append((frame, index, r.size, eng.get_timing()["emit"], eng.get_timing()["layout"], table.clone()))
    same = out[0][2] == out[1][2] and torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][5][:nb * 24], out[1][5][:nb * 24])
    if not same:
        bad += 1
        d = (out[0][0] != out[1][0]).nonzero()
        print("DIFF", label, "sizes", out[0][2], out[1][2], "first frame diff", int(d[0]) if len(d) else None, "index equal", bool(torch.equal(out[0][1], out[1][1])), flush=True)
    return (out[0][3], out[0][4]), (out[1][3], out[1][4])
for i in range(cases):
    kind = int(rng.integers(0, 4)); n = int(rng.integers(1, 24 << 20))
    if kind == 0: data = np.frombuffer(datagen.structured(n, int(rng.integers(1 << 30))), dtype=np.uint8).copy()
    elif kind == 1: data = datagen.synth50(max(1024, n & ~1023), int(rng.integers(1 << 30)))[:n]
    elif kind == 2: data = datagen.synth_text(n, int(rng.integers(1 << 30)))
    else: data = np.concatenate([rng.integers(0, 256, n // 2, dtype=np.uint8), np.zeros(n - n // 2, dtype=np.uint8)])
    one(torch.from_numpy(data).cuda(), int(rng.integers(4, 8)), int(rng.integers(0, 2)), "case %d kind %d n %d" % (i, kind, n))
print("cases", cases, "differences", bad)
for name, src in (("synth50 4 GiB", synth50_device(4 << 30, 1234)),):
    a, b = one(src, 7, 1, name)
    print(name, "emit/layout ms serial %.3f %.3f  new %.3f %.3f" % (a + b))
t = torch.from_numpy(datagen.synth_text(64 << 20, 99)).cuda().repeat(16)
a, b = one(t, 7, 1, "text 1 GiB"); print("text 1 GiB emit/layout ms serial %.3f %.3f  new %.3f %.3f" % (a + b))
a, b = one(t[: (64 << 20) + 12345], 4, 1, "text 64 KiB blocks"); print("text 64 MiB, 64 KiB blocks: emit/layout ms serial %.3f %.3f  new %.3f %.3f" % (a + b))
sys.exit(1 if bad else 0)
