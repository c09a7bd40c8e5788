#!/usr/bin/env python3
"""Development (GPU box): the foreign 4 GiB frame decoded 24 times: every timing slot per iteration (where does the spread come from?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 4 << 30
src = synth50_device(n, 1234, "cuda"); eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
eng.compress_async(src, frame, p); rc = eng.result()
back = torch.zeros(n, dtype=torch.uint8, device="cuda")
keys = None
for it in range(24):
    eng.decompress_frame_async(frame, int(rc.size), back); r = eng.result(); t = eng.get_timing()
    if keys is None: keys = [k for k, v in t.items() if v and v > 0.001 and not k.startswith("compress") and k not in ("find_matches", "layout", "emit", "xxh32")]
    print(" ".join("%s %.3f" % (k, t[k]) for k in keys), flush=True)
print("ok", bool(torch.equal(back, src)))
