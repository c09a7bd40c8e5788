#!/usr/bin/env python3
"""Dev tool (GPU box): cycle counters of the fused decoder's workgroup 0 (LZ4F_MI355X_PROF=1; the parser's own stamps need a build with -DFZ_PROF)."""
import ctypes, os, sys
os.environ["LZ4F_MI355X_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import _ffi, conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096) << 20
from lz4_frame_conduit_amd import datagen
src = torch.from_numpy(datagen.synth_text(n, 5)).cuda() if "text" in sys.argv else synth50_device(n, 1234); eng = Engine(0)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda"); nb = n >> 22; table = eng.new_table(nb)
index = eng.new_index(n, p) if len(sys.argv) > 2 and sys.argv[2] == "ix" else None
eng.compress_async(src, frame, p, table, index); r = eng.result()
back = torch.empty_like(src)
eng.set_timing(True)
for _ in range(3):
    eng.decompress_blocks_async(frame, frame.numel(), back, table, nb, p.frameInfo, index); eng.result()
print("decode ms", eng.get_timing()["decode"], "ok", bool(torch.equal(back, src)))
buf = (ctypes.c_ulonglong * 128)()
print("rc", ctypes.CDLL(_ffi.LIB_PATH).lz4f_mi355x_debug_prof(buf))
print("parser", {"total": int(buf[0]), "ring_wait": int(buf[1]), "nseq": int(buf[2])})
for w in range(1, 8):
    print("copier", w, {k: int(buf[8 * w + i]) for i, k in enumerate(["wait_produced", "literals", "wait_chain", "matches", "slots", "dep", "drain"])})
