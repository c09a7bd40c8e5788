#!/usr/bin/env python3
"""Development (GPU box): real text instead of the synthetic kind - the image's own Python sources, concatenated (up to 256 MiB), framed by the oracle (= liblz4's bytes) with
4 MiB and 64 KiB independent blocks and 64 KiB linked ones, decoded device-resident; and through this library's own encoder.  Bytes must match; prints ratio and GiB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine
want = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
buf = bytearray()
for root in ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include"):
    for dp, dn, fn in os.walk(root):
        for f in sorted(fn):
            if f.endswith((".py", ".h", ".hpp", ".txt", ".md", ".rst", ".json")):
                try: buf += open(os.path.join(dp, f), "rb").read()
                except OSError: pass
        if len(buf) >= want: break
    if len(buf) >= want: break
data = np.frombuffer(bytes(buf[:want]), dtype=np.uint8)
print("real text: %d MiB" % (data.size >> 20), flush=True)
eng = Engine(0); eng.set_timing(True)
src = torch.from_numpy(data.copy()).cuda()
for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=1), dict(bsid=4, indep=0)):
    fr = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))
    dev = torch.from_numpy(np.frombuffer(fr + bytes(64), dtype=np.uint8).copy()).cuda()
    back = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
    best = None
    for it in range(3):
        eng.decompress_frame_async(dev, len(fr), back); r = eng.result(); t = eng.get_timing()["decompress_total"]
        best = t if best is None or t < best else best
    ok = r.size == data.size and bool(torch.equal(back[:data.size], src))
    p = conduit.make_preferences(blockSizeID=kw["bsid"], blockMode=1 if kw["indep"] else 0)
    f2 = torch.empty(eng.frame_bound_inband(data.size, p), dtype=torch.uint8, device="cuda")
    eng.compress_async(src, f2, p, inband=True); rc = eng.result(); tc = eng.get_timing()["compress_total"]
    back.zero_(); eng.decompress_frame_async(f2, int(rc.size), back); rd = eng.result(); td = eng.get_timing()["decompress_total"]
    ok2 = rd.size == data.size and bool(torch.equal(back[:data.size], src))
    print("bsid %d %s: liblz4 ratio %.3f, foreign decode %.2f ms = %.1f GiB/s ok=%s path %s | own: ratio %.3f compress %.2f ms (%.1f GiB/s) decompress %.2f ms (%.1f GiB/s) ok=%s" % (
        kw["bsid"], "independent" if kw["indep"] else "linked", data.size / len(fr), best, data.size / 2**30 / (best / 1e3), ok, hex(int(r.flags) >> 12),
        data.size / int(rd.consumed), tc, data.size / 2**30 / (tc / 1e3), td, data.size / 2**30 / (td / 1e3), ok2), flush=True)
