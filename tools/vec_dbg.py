"""Dev tool (GPU box): first wrong byte of a 64 KiB-block decode, and the sequences around it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
import oracle

def seqs(payload):
    i, op, out = 0, 0, []
    n = len(payload)
    while i < n:
        pos = i; t = payload[i]; i += 1
        lit = t >> 4
        if lit == 15:
            while True:
                b = payload[i]; i += 1; lit += b
                if b != 255: break
        p = i; i += lit
        if i >= n: out.append((pos, p, lit, 0, 0, op)); break
        off = payload[i] | (payload[i + 1] << 8); i += 2
        ml = t & 15
        if ml == 15:
            while True:
                b = payload[i]; i += 1; ml += b
                if b != 255: break
        ml += 4
        out.append((pos, p, lit, ml, off, op)); op += lit + ml
    return out

data = datagen.synth_text(1 << 20, 5)
fr = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(bsid=4, indep=1))
eng = Engine(0)
dev = torch.from_numpy(np.frombuffer(fr, dtype=np.uint8).copy()).cuda()
back = torch.zeros(len(data), dtype=torch.uint8, device="cuda")
eng.decompress_frame_async(dev, dev.numel(), back)
try: r = eng.result(); print("result", r.size)
except Exception as e: print("error", e)
got = back.cpu().numpy()
bad = np.nonzero(got != data)[0]
print("wrong bytes", len(bad), "first", bad[:10])
if len(bad):
    b = int(bad[0]) >> 16
    # walk to block b
    pos = 7
    for _ in range(b):
        w = int.from_bytes(fr[pos:pos + 4], "little"); pos += 4 + (w & 0x7FFFFFFF)
    w = int.from_bytes(fr[pos:pos + 4], "little"); payload = fr[pos + 4:pos + 4 + (w & 0x7FFFFFFF)]
    rel = int(bad[0]) & 0xFFFF
    ss = seqs(payload)
    for (tp, p, lit, ml, off, op) in ss:
        if op + lit + ml >= rel - 40 and op <= rel + 40:
            print("token@%d lit %d (src %d) ml %d off %d  out %d..%d (match at %d, source %d..%d)" % (tp, lit, p, ml, off, op, op + lit + ml, op + lit, op + lit - off, op + lit - off + ml))
    print("wrong at block %d offset %d: got %r want %r" % (b, rel, bytes(got[bad[0] - 8:bad[0] + 8]), bytes(data[bad[0] - 8:bad[0] + 8])))
    print("all wrong offsets in block:", [int(x) & 0xFFFF for x in bad[:40] if (int(x) >> 16) == b])
