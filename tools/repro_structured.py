#!/usr/bin/env python3
"""Dev tool (GPU box): decode a GPU-made frame of datagen.structured with each decoder variant, report first mismatch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from lz4_frame_conduit_amd import _ffi, conduit, datagen
seed = int(sys.argv[1]); size = int(sys.argv[2]); bsid = int(sys.argv[3]); indep = int(sys.argv[4])
L = _ffi.lib()
data = datagen.structured(size, 1000 + seed)
prefs = conduit.make_preferences(blockSizeID=bsid, blockMode=indep)
cap = L.lz4f_mi355x_compressFrameBound(len(data), ctypes.byref(prefs))
dst = ctypes.create_string_buffer(cap)
r = L.lz4f_mi355x_compressFrame(dst, cap, data, len(data), ctypes.byref(prefs))
frame = dst.raw[:r]
assert oracle.decompress_frame(frame, len(data) + 64)[0] == data
open("gpurun_out/repro_frame.lz4", "wb").write(frame)
out = ctypes.create_string_buffer(len(data) + 8); used = ctypes.c_size_t(0)
r2 = L.lz4f_mi355x_decompressFrame(out, len(data) + 8, frame, len(frame), ctypes.byref(used))
if L.LZ4F_isError(r2):
    print("mode", os.environ.get("LZ4F_MI355X_DECODE", "default"), "ERROR", L.LZ4F_getErrorName(r2), L.lz4f_mi355x_last_error()); sys.exit(0)
got = np.frombuffer(out.raw[:r2], dtype=np.uint8); want = np.frombuffer(data, dtype=np.uint8)
if r2 != len(data): print("size", r2, len(data))
n = min(len(got), len(want)); bad = np.nonzero(got[:n] != want[:n])[0]
print("mode", os.environ.get("LZ4F_MI355X_DECODE", "default"), "mismatches", len(bad), "first", bad[:8].tolist(), "last", bad[-3:].tolist() if len(bad) else [])
if len(bad):
    b = int(bad[0]); print("block", b >> (8 + 2 * bsid), "offset in block", b & ((1 << (8 + 2 * bsid)) - 1), "got", got[b:b+16].tolist(), "want", want[b:b+16].tolist())
    # runs of bad bytes
    runs = []; s = int(bad[0]); p = s
    for x in bad[1:]:
        x = int(x)
        if x != p + 1: runs.append((s, p - s + 1)); s = x
        p = x
    runs.append((s, p - s + 1)); print("runs", runs[:10], "n_runs", len(runs))
