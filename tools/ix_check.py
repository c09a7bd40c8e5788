#!/usr/bin/env python3
"""Dev tool (GPU box): indexed decode vs source on synth50 / structured inputs; per-kernel ms.  argv: MiB [bsid]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bsid = int(sys.argv[2]) if len(sys.argv) > 2 else 7
kind = sys.argv[3] if len(sys.argv) > 3 else "synth50"
n = mib << 20
eng = Engine(0)
p = conduit.make_preferences(blockSizeID=bsid, blockMode=0 if os.environ.get('LINKED') else 1)
bs = 1 << (8 + 2 * bsid)
def run(src, label):
    n = src.numel()
    frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda"); nb = (n + bs - 1) // bs
    table = eng.new_table(nb); index = torch.zeros(eng.index_size(n, p) * int(os.environ.get('IX_SCALE', '1')), dtype=torch.uint8, device='cuda')
    back = torch.empty_like(src)
    eng.set_timing(True)
    for it in range(3):
        eng.compress_async(src, frame, p, table, index); r = eng.result(); tc = eng.get_timing()
        back.zero_()
        eng.decompress_blocks_async(frame, frame.numel(), back, table, nb, p.frameInfo, index); r2 = eng.result(); td = eng.get_timing()
    ok = bool(torch.equal(back, src)) and r2.size == n
    hd = index[:32].cpu().numpy().view(np.uint32)
    print(label, "ok", ok, "ratio %.4f" % (n / r.size), "index", "usable" if hd[0] == 0x3258494C else "UNUSABLE", "seqs", int(hd[3]), "entries", int(hd[4]), {k: round(v, 3) for k, v in {**tc, **td}.items() if v > 0}, flush=True)
    if not ok:
        d = (back != src).nonzero()
        print("  first diff at", int(d[0]) if len(d) else None, "of", n, "n diffs", len(d), "size", r2.size)
    return ok
good = True
if kind == "synth50":
    good &= run(synth50_device(n, 1234), "synth50 %d MiB" % mib)
elif kind == "text":
    t = torch.from_numpy(datagen.synth_text(min(n, 64 << 20), 99)).cuda()
    good &= run(t.repeat(max(1, n // t.numel())), "text %d MiB" % mib)
else:
    for seed in range(int(kind)):
        a = np.frombuffer(datagen.structured(n, seed), dtype=np.uint8).copy()
        good &= run(torch.from_numpy(a).cuda(), "structured seed %d" % seed)
sys.exit(0 if good else 1)
