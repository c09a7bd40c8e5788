#!/usr/bin/env python3
"""Dev tool (GPU box): single-byte mutations of a linked frame; the GPU verdict (ok / error) must match the oracle's,
and an accepted frame must decode to the same bytes.  Every call must return (no hang)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from lz4_frame_conduit_amd import _ffi, datagen
L = _ffi.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_mut = int(sys.argv[2]) if len(sys.argv) > 2 else 150
data = datagen.structured(1 << 20, 77) + datagen.synth50(1 << 20, 5).tobytes()
if os.environ.get('DATA') == 'text':                # dense data: the frame goes through the pointer-doubling kernels
    data = datagen.synth_text(2 << 20, 7).tobytes()
# INDEP=1: independent blocks (64 KiB: the wave-per-block decoder whose lanes find the tokens; BSID=7: big blocks, density probe)
frame = oracle.conduit_compress(data, oracle.mkprefs(bsid=int(os.environ.get("BSID", "4")), indep=1 if os.environ.get("INDEP") else 0))
cap = len(data) + 8
diff = 0; t0 = time.time()
for i in range(n_mut):
    pos = int(rng.integers(7, len(frame))); x = int(rng.integers(1, 256))
    bad = bytearray(frame); bad[pos] ^= x; bad = bytes(bad)
    try:
        want, _ = oracle.decompress_frame(bad, cap); ov = "ok"
    except oracle.OracleError as e:
        want, ov = None, str(e)
    dst = ctypes.create_string_buffer(cap); used = ctypes.c_size_t(0)
    r = L.lz4f_mi355x_decompressFrame(dst, cap, bad, len(bad), ctypes.byref(used))
    gv = "ok" if not L.LZ4F_isError(r) else L.LZ4F_getErrorName(r).decode()
    if (ov == "ok") != (gv == "ok") or (ov == "ok" and dst.raw[:r] != want):
        diff += 1; print("pos", pos, "xor", x, "oracle", ov, "gpu", gv)
print("mutations", n_mut, "verdict differences", diff, "seconds %.1f" % (time.time() - t0))
sys.exit(1 if diff else 0)
