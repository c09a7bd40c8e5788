"""Dev tool (GPU box): text in the reference's default framing (64 KiB linked blocks): compress, and decompress from the stream alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
eng = Engine(0); eng.set_timing(True)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
txt = torch.from_numpy(datagen.synth_text(64 << 20, 99)).cuda().repeat(n // (64 << 20))
p = conduit.make_preferences(blockSizeID=4, blockMode=0)
frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda"); back = torch.empty_like(txt)
for _ in range(3):
    eng.compress_async(txt, frame, p, inband=True); r = eng.result(); tc = eng.get_timing()
    back.zero_(); eng.decompress_frame_async(frame, int(r.size), back); r2 = eng.result(); td = eng.get_timing()
tcomp = tc["find_matches"] + tc["layout"] + tc["emit"]; tdec = sum(v for k, v in td.items() if k in ("walk", "decode", "decode_parse", "decode_copy", "finish"))
print("text 64 KiB linked, %d MiB: ratio %.3f compress %.2f ms (%.1f GiB/s) decompress %s ok=%s path %s" % (n >> 20, n / r.size, tcomp, n / tcomp / 1e-3 / 2**30, {k: round(v, 2) for k, v in td.items() if v}, bool(torch.equal(back, txt)), hex(r2.flags >> 12)))
