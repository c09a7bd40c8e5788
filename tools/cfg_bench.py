#!/usr/bin/env python3
"""Dev tool (GPU box): BASELINE configs[1] (decode-only, text, 64 KiB independent blocks) and configs[4]
(linked 64 KiB blocks) next to the independent-block path, device-resident, per-kernel HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device

eng = Engine(0); eng.set_timing(True)
def run(name, src, prefs, nb, reps=3):
    n = src.numel()
    frame = torch.empty(eng.frame_bound(n, prefs), dtype=torch.uint8, device="cuda"); table = eng.new_table(nb)
    back = torch.empty_like(src)
    for _ in range(reps):
        eng.compress_async(src, frame, prefs, table); r = eng.result(); tc = eng.get_timing()
        back.zero_()
        eng.decompress_blocks_async(frame, frame.numel(), back, table, nb, prefs.frameInfo); r2 = eng.result(); td = eng.get_timing()
    ok = bool(torch.equal(back, src))
    tcomp = tc["find_matches"] + tc["layout"] + tc["emit"]; tdec = td["decode"] + td["finish"]
    print("%-34s ok=%s ratio %.3f  compress %.2f ms (%.1f GiB/s)  decompress %.2f ms (%.1f GiB/s, %.0f GB/s algorithmic)" % (
        name, ok, n / r.size, tcomp, n / tcomp / 1e-3 / 2**30, tdec, n / tdec / 1e-3 / 2**30, (n + r.size) / tdec / 1e-3 / 1e9))

size = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
# cfg 2: enwik-style text, 64 KiB independent blocks (tile 64 MiB of generated text; tiles are >> 64 KiB apart so no cross-tile matches)
txt = torch.from_numpy(datagen.synth_text(64 << 20, 99)).cuda().repeat(size // (64 << 20))
run("cfg2 text 64KiB independent", txt, conduit.make_preferences(blockSizeID=4, blockMode=1), size >> 16)
run("     text 4MiB independent", txt, conduit.make_preferences(blockSizeID=7, blockMode=1), size >> 22)
s50 = synth50_device(size, 1234)
run("cfg3 synth50 4MiB independent", s50, conduit.make_preferences(blockSizeID=7, blockMode=1), size >> 22)
run("     synth50 64KiB independent", s50, conduit.make_preferences(blockSizeID=4, blockMode=1), size >> 16)
run("cfg5 synth50 64KiB linked", s50[:256 << 20], conduit.make_preferences(blockSizeID=4, blockMode=0), (256 << 20) >> 16, reps=2)
rnd = torch.randint(0, 256, (size,), dtype=torch.uint8, device="cuda")
run("     random 4MiB (all stored)", rnd, conduit.make_preferences(blockSizeID=7, blockMode=1), size >> 22)
