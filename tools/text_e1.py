"""Dev tool (GPU box): pass E1 on text - compress time and ratio at 64 KiB and 4 MiB independent blocks, next to liblz4's ratio."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
import oracle
eng = Engine(0); eng.set_timing(True)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
host = datagen.synth_text(64 << 20, 99)
txt = torch.from_numpy(host).cuda().repeat(n // (64 << 20))
for bsid in (4, 7):
    p = conduit.make_preferences(blockSizeID=bsid, blockMode=1)
    frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        eng.compress_async(txt, frame, p); r = eng.result(); t = eng.get_timing()
    ref = len(oracle.conduit_compress(host[:16 << 20].tobytes(), oracle.mkprefs(bsid=bsid, indep=1)))
    back = torch.zeros_like(txt); eng.decompress_frame_async(frame, int(r.size), back); eng.result()
    print("text bsid %d: E1 %.2f ms per %d MiB (%.1f GiB/s), emit %.2f, ratio %.4f (liblz4 %.4f on 16 MiB), ok=%s" % (
        bsid, t["find_matches"], n >> 20, n / t["find_matches"] / 1e-3 / 2**30, t["emit"], n / r.size, (16 << 20) / ref, bool(torch.equal(back, txt))), flush=True)
