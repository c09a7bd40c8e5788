#!/usr/bin/env python3
"""Dev tool (GPU box): per-kernel ms of one compress+decompress of synth50 (size in MiB as argv[1], default 4096)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096) << 20
bsid = int(sys.argv[2]) if len(sys.argv) > 2 else 7
src = synth50_device(n, 1234); eng = Engine(0)
p = conduit.make_preferences(blockSizeID=bsid, blockMode=1)
bs = 1 << (8 + 2 * bsid)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda"); nb = (n + bs - 1) // bs; table = eng.new_table(nb)
back = torch.empty_like(src)
eng.set_timing(True)
for it in range(3):
    eng.compress_async(src, frame, p, table); r = eng.result(); tc = eng.get_timing()
    back.zero_()
    eng.decompress_blocks_async(frame, frame.numel(), back, table, nb, p.frameInfo); r2 = eng.result(); td = eng.get_timing()
ok = bool(torch.equal(back, src))
print("ok", ok, "ratio %.4f" % (n / r.size), {k: round(v, 3) for k, v in {**{k: tc[k] for k in ("find_matches", "layout", "emit")}, **{k: td[k] for k in ("decode", "decode_parse", "decode_copy", "finish")}}.items()})
print("decode GB/s algo: %.0f" % ((n + r.size) / (td["decode"] * 1e-3) / 1e9 if td["decode"] else 0))
