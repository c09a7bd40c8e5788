#!/usr/bin/env python3
"""Development (GPU box): where this encoder's frames are bigger than liblz4's (the oracle's) on one input: sequences, literal bytes, match bytes, bytes of
sequence overhead, and how the literal bytes fall within the 1 KiB helpings pass E1 hands out (position of the literal in its helping).
    tools/ratio_anatomy.py [structured|text|synth50|rows256|real] [MiB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
what = sys.argv[1] if len(sys.argv) > 1 else "structured"
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 8) << 20
if what == "structured": data = np.frombuffer(datagen.structured(n, 8), dtype=np.uint8).copy()
elif what == "text": data = datagen.synth_text(n, 4)
elif what == "rows256":
    rng = np.random.default_rng(3); a = rng.integers(0, 256, n, dtype=np.uint8).reshape(-1, 256); odd = np.arange(1, a.shape[0], 2)
    back = rng.integers(1, 100, odd.size) * 2 + 1; srcr = np.maximum(odd - back, 0); srcr -= srcr % 2; a[odd] = a[srcr]; data = a.reshape(-1)
elif what == "real":                                                     # the image's own Python sources and headers, as tools/real_text.py takes them
    buf = bytearray()
    for root in ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include"):
        for dp, dn, fn in os.walk(root):
            for f in sorted(fn):
                if f.endswith((".py", ".h", ".hpp", ".txt", ".md", ".rst", ".json")):
                    try: buf += open(os.path.join(dp, f), "rb").read()
                    except OSError: pass
            if len(buf) >= n: break
        if len(buf) >= n: break
    data = np.frombuffer(bytes(buf[:n]), dtype=np.uint8).copy()
else: data = datagen.synth50(n, 3)
def seqs(frame):
    b = bytes(frame); pos = 7; out = []; base = 0
    while True:
        w = int.from_bytes(b[pos:pos + 4], "little"); pos += 4
        if w == 0: break
        sz = w & 0x7FFFFFFF
        if w >> 31: out.append((base, sz, 0, 0)); base += sz; pos += sz; continue
        p = pos; end = pos + sz; op = base
        while p < end:
            t = b[p]; p += 1; l = t >> 4
            if l == 15:
                while True:
                    x = b[p]; p += 1; l += x
                    if x != 255: break
            p += l
            if p >= end: out.append((op, l, 0, 0)); op += l; break
            o = b[p] | (b[p + 1] << 8); p += 2; m = t & 15
            if m == 15:
                while True:
                    x = b[p]; p += 1; m += x
                    if x != 255: break
            m += 4; out.append((op, l, m, o)); op += l + m
        base = op; pos = end
    return out
for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=0)):
    ref = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))
    eng = Engine(0); p = conduit.make_preferences(blockSizeID=kw["bsid"], blockMode=kw["indep"])
    src = torch.from_numpy(data).cuda(); frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
    eng.compress_async(src, frame, p); r = eng.result(); mine = frame[:r.size].cpu().numpy().tobytes(); eng.close()
    print(what, kw, "gpu %d liblz4 %d bytes: %.4f" % (len(mine), len(ref), len(mine) / len(ref)))
    for name, fr in (("gpu", mine), ("liblz4", ref)):
        s = seqs(fr); lit = sum(x[1] for x in s); mt = sum(x[2] for x in s); ns = len(s)
        hist = np.zeros(8, dtype=np.int64)
        for (op, l, m, o) in s:
            if l: hist[((op % 1024) * 8) // 1024] += l
        short = sum(1 for x in s if 0 < x[2] < 8); longm = sum(1 for x in s if x[2] >= 64)
        print("  %-6s sequences %d, literal bytes %d, match bytes %d, overhead %d; matches < 8 bytes %d, >= 64 bytes %d; literal bytes by eighth of a 1 KiB helping %s" % (
            name, ns, lit, mt, len(fr) - lit, short, longm, (hist * 100 // max(1, hist.sum())).tolist()))
