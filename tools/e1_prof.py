#!/usr/bin/env python3
"""Dev tool (GPU box): phase cycle counters of one E1 wave (chunk 1000); needs a build with -DE1_PROF (make CXXFLAGS_EXTRA=-DE1_PROF)."""
import ctypes, os, sys
os.environ["LZ4F_MI355X_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import _ffi, conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 4096 << 20
src = synth50_device(n, 1234); eng = Engine(0)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda"); table = eng.new_table(n >> 22)
eng.set_timing(True)
for _ in range(3):
    eng.compress_async(src, frame, p, table); r = eng.result()
print("find_matches ms", eng.get_timing()["find_matches"])
buf = (ctypes.c_ulonglong * 128)()
ctypes.CDLL(_ffi.LIB_PATH).lz4f_mi355x_debug_prof(buf)
names = ["total", "seed", "probe", "verify", "extend", "restart", "iterations", "iterations_with_candidates", "records", "clear_and_seed"]
print({k: int(buf[64 + i]) for i, k in enumerate(names)})
