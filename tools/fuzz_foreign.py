#!/usr/bin/env python3
"""Development (GPU box): single-byte mutations of a FOREIGN frame of big independent blocks with long sequences (liblz4's bytes of synth50 + a ragged tail: the
stretch finder, the self-feeding copy kernel) - or, with a fourth argument "text", of text (dense: the workgroup-per-block decoder of decode_relay.cuh; LZ4F_MI355X_DENSE_MODE=2 for the
wave-per-block one); the verdict must be the oracle's and an accepted frame must decode to the oracle's bytes.  argv: seed mutations [bsid] [text]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from lz4_frame_conduit_amd import _ffi, datagen
L = _ffi.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_mut = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bsid = int(sys.argv[3]) if len(sys.argv) > 3 else 7
data = np.concatenate([datagen.synth50(9 << 20, int(rng.integers(1 << 30))), rng.integers(0, 256, 70001, dtype=np.uint8)]).tobytes()
if len(sys.argv) > 4 and sys.argv[4] == "text": data = datagen.synth_text(9 << 20, int(rng.integers(1 << 30))).tobytes()
frame = oracle.conduit_compress(data, oracle.mkprefs(bsid=bsid, indep=1))
cap = len(data) + 8
diff = 0; t0 = time.time()
for i in range(n_mut):
    pos = int(rng.integers(7, len(frame))); x = int(rng.integers(1, 256))
    if i % 5 == 0: pos = 7 + int(rng.integers(0, 4000)) + int(rng.integers(0, 3)) * (len(frame) // 3)      # (some near block starts)
    bad = bytearray(frame); bad[pos] ^= x; bad = bytes(bad)
    try:
        want, _ = oracle.decompress_frame(bad, cap); ov = "ok"
    except oracle.OracleError as e:
        want, ov = None, str(e)
    dst = ctypes.create_string_buffer(cap); used = ctypes.c_size_t(0)
    t1 = time.time()
    r = L.lz4f_mi355x_decompressFrame(dst, cap, bad, len(bad), ctypes.byref(used))
    if time.time() - t1 > 0.3: print("slow: %.2f s" % (time.time() - t1), "pos", pos, "xor", x, "oracle", ov, "gpu", "ok" if not L.LZ4F_isError(r) else L.LZ4F_getErrorName(r).decode(), flush=True)
    gv = "ok" if not L.LZ4F_isError(r) else L.LZ4F_getErrorName(r).decode()
    if (ov == "ok") != (gv == "ok") or (ov == "ok" and dst.raw[:r] != want):
        diff += 1; print("pos", pos, "xor", x, "oracle", ov, "gpu", gv)
print("bsid", bsid, "mutations", n_mut, "verdict differences", diff, "seconds %.1f" % (time.time() - t0))
sys.exit(1 if diff else 0)
