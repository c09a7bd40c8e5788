#!/usr/bin/env python3
"""Dev tool (GPU box): host-to-host rate of the C-ABI bulk calls (SURVEY 8d variant H: host buffers in, host buffers out,
PCIe copies and host-side framing included), with pageable buffers and with page-locked ones (lz4f_mi355x_host_alloc).
`python tools/host_bench.py [MiB] [blockSizeID]`; DATA=text, LINKED=1, DEVICES=n."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lz4_frame_conduit_amd import _ffi, conduit, datagen
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
bsid = int(sys.argv[2]) if len(sys.argv) > 2 else 7
L = _ffi.lib()
if os.environ.get("DEVICES"): assert not L.LZ4F_isError(L.lz4f_mi355x_use_devices(int(os.environ["DEVICES"])))
kind = os.environ.get('DATA', 'synth50')          # DATA=text: dense data (13-byte sequences)
if kind == 'text':
    t = datagen.synth_text(min(n, 64 << 20), 99)
    data = np.ascontiguousarray(np.tile(t, max(1, n // t.size)))
else:
    data = np.ascontiguousarray(datagen.synth50(n, 1234))
prefs = conduit.make_preferences(blockSizeID=bsid, blockMode=0 if os.environ.get('LINKED') else 1)
bound = L.lz4f_mi355x_compressFrameBound(n, ctypes.byref(prefs))
def pinned(size):
    p = L.lz4f_mi355x_host_alloc(size); assert p
    return np.ctypeslib.as_array((ctypes.c_uint8 * size).from_address(p)), p
def ptr(a): return a.ctypes.data_as(ctypes.c_void_p)
out = {"data": kind, "MiB": n >> 20, "block": 1 << (8 + 2 * bsid)}
for mode in ("pageable", "pinned"):
    if mode == "pageable": src = data; dst = np.empty(bound, dtype=np.uint8); back = np.empty(n + 8, dtype=np.uint8)
    else:
        src, p1 = pinned(n); src[:] = data; dst, p2 = pinned(bound); back, p3 = pinned(n + 8)
    best_c = best_d = 1e9; csize = 0
    for it in range(4):
        t0 = time.perf_counter()
        r = L.lz4f_mi355x_compressFrame(ptr(dst), bound, ptr(src), n, ctypes.byref(prefs))
        t1 = time.perf_counter()
        assert not L.LZ4F_isError(r), (L.LZ4F_getErrorName(r), L.lz4f_mi355x_last_error())
        csize = r
        used = ctypes.c_size_t(0)
        t2 = time.perf_counter()
        r2 = L.lz4f_mi355x_decompressFrame(ptr(back), back.size, ptr(dst), csize, ctypes.byref(used))
        t3 = time.perf_counter()
        assert not L.LZ4F_isError(r2), (L.LZ4F_getErrorName(r2), L.lz4f_mi355x_last_error())
        if it: best_c = min(best_c, t1 - t0); best_d = min(best_d, t3 - t2)
    ok = r2 == n and bool(np.array_equal(back[:n], data))
    g = n / 2**30
    out[mode] = {"ok": ok, "ratio": round(n / csize, 4), "compress_GiBs": round(g / best_c, 2), "decompress_GiBs": round(g / best_d, 2), "e2e_GiBs": round(g / (best_c + best_d), 2)}
print(out)
