#!/usr/bin/env python3
"""Development (GPU box): hand-made dense blocks (tests/test_gpu_parity.py: _dense_block - short sequences with overlapping matches, offset 65535, long runs and long
matches in between, tokens with length bytes) in frames of 1..5 blocks, through the workgroup-per-block decoder (decode_relay.cuh) and the wave-per-block one; the
oracle's bytes must come out.  argv: first seed, number of frames"""
import ctypes, os, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from lz4_frame_conduit_amd import _ffi
from test_gpu_parity import _dense_block
L = _ffi.lib()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
bad = 0; t0 = time.time()
for f in range(n_frames):
    rng = np.random.default_rng(seed0 + f)
    bsid = int(rng.integers(5, 8)); bs = {5: 256 << 10, 6: 1 << 20, 7: 4 << 20}[bsid]
    hdr = oracle.conduit_compress(b"", oracle.mkprefs(bsid=bsid, indep=1))[:7]
    fr, data = bytearray(hdr), bytearray()
    for b in range(int(rng.integers(1, 6))):
        n_out = bs if rng.integers(0, 3) else int(rng.integers(70000, bs))
        body, out = _dense_block(n_out, int(rng.integers(1 << 30)), bool(rng.integers(0, 2)) and n_out > 700000)
        fr += struct.pack("<I", len(body)) + body; data += out
    fr += struct.pack("<I", 0)
    want, used = oracle.decompress_frame(bytes(fr), len(data) + 8)
    assert want == bytes(data) and used == len(fr), "the generator"
    for mode in ("1", "2"):
        os.environ["LZ4F_MI355X_DENSE_MODE"] = mode
        L.lz4f_mi355x_release_engines()
        cap = len(data) + 8
        dst = ctypes.create_string_buffer(cap); usedc = ctypes.c_size_t(0)
        r = L.lz4f_mi355x_decompressFrame(dst, cap, bytes(fr), len(fr), ctypes.byref(usedc))
        ok = (not L.LZ4F_isError(r)) and dst.raw[:r] == bytes(data)
        if not ok: bad += 1; print("frame", f, "seed", seed0 + f, "bsid", bsid, "mode", mode, "FAILED", L.LZ4F_getErrorName(r).decode() if L.LZ4F_isError(r) else "bytes differ")
    print("frame", f, "bsid", bsid, "bytes", len(data), "ok so far" if not bad else "", "%.0f s" % (time.time() - t0), flush=True)
print("frames", n_frames, "failures", bad)
sys.exit(1 if bad else 0)
