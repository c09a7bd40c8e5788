#!/usr/bin/env python3
"""Development (GPU box): what ONE 64 KiB block costs on the device (kernel by kernel, HIP events) and through LZ4F_compressUpdate / LZ4F_decompress."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import _ffi, conduit, datagen
from lz4_frame_conduit_amd.device import Engine
L = _ffi.lib()
data = datagen.synth50(1 << 20, 3)
eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=4, blockMode=0)
src = torch.from_numpy(data[:131072].copy()).cuda()
frame = torch.empty(eng.frame_bound(131072, p), dtype=torch.uint8, device="cuda")
for n in (65536, 131072):
    for it in range(3):
        t0 = time.perf_counter(); eng.compress_async(src[:n], frame, p); r = eng.result(); t1 = time.perf_counter(); t = eng.get_timing()
    print("device-resident %d KiB linked: wall %.0f us; find_matches %.1f layout %.1f emit %.1f total %.1f us" % (n >> 10, (t1 - t0) * 1e6, t["find_matches"] * 1e3, t["layout"] * 1e3, t["emit"] * 1e3, t["compress_total"] * 1e3))
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    for it in range(3):
        t0 = time.perf_counter(); eng.decompress_frame_async(frame, int(r.size), back); r2 = eng.result(); t1 = time.perf_counter(); t = eng.get_timing()
    print("   decode: wall %.0f us; walk %.1f decode %.1f finish %.1f total %.1f us" % ((t1 - t0) * 1e6, t["walk"] * 1e3, t["decode"] * 1e3, t["finish"] * 1e3, t["decompress_total"] * 1e3))
# the streaming functions, a block per call
c = ctypes.c_void_p(); L.LZ4F_createCompressionContext(ctypes.byref(c), 100)
pp = conduit.make_preferences()
buf = ctypes.create_string_buffer(1 << 17)
L.LZ4F_compressBegin(c, buf, 1 << 17, ctypes.byref(pp))
raw = data.tobytes()
out = []
t0 = time.perf_counter()
for i in range(16):
    r = L.LZ4F_compressUpdate(c, buf, 1 << 17, raw[i * 65536:(i + 1) * 65536], 65536, None); out.append(buf.raw[:r])
t1 = time.perf_counter()
print("LZ4F_compressUpdate, 64 KiB per call: %.0f us per call" % ((t1 - t0) / 16 * 1e6))
