#!/usr/bin/env python3
"""Development (GPU box): compressed size of library variants (tools/ab_build.sh) against liblz4 (the oracle) on inputs whose sequences are LONG -
the inputs pass E1 searches in its sparse mode (wider first stride): rows of other widths than the bench's, copies at unaligned positions and of
random lengths, structured records.      python3 tools/ratio_sparse.py [name ...]   (no names: base + every lib_*.so under build/ab)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "lz4_frame_conduit_amd", "build", "ab")
CHILD = r'''
import ctypes, os, sys, json
sys.path.insert(0, %r)
import numpy as np
import oracle
from lz4_frame_conduit_amd import _ffi, conduit, datagen
L = _ffi.lib()
def gpu(data, p):
    cap = L.lz4f_mi355x_compressFrameBound(len(data), ctypes.byref(p)); dst = ctypes.create_string_buffer(cap)
    r = L.lz4f_mi355x_compressFrame(dst, cap, data, len(data), ctypes.byref(p))
    assert not L.LZ4F_isError(r), (L.LZ4F_getErrorName(r), L.lz4f_mi355x_last_error())
    return dst.raw[:r]
def rows(n, width, seed):
    rng = np.random.default_rng(seed); a = rng.integers(0, 256, n, dtype=np.uint8).reshape(-1, width); r = a.shape[0]
    odd = np.arange(1, r, 2); back = rng.integers(1, max(2, min(60, 30000 // width)), odd.size) * 2 + 1
    src = np.maximum(odd - back, 0); src -= src %% 2; a[odd] = a[src]; return a.reshape(-1).tobytes()
def mix(n, seed, lit=(100, 2000), mat=(20, 600)):
    rng = np.random.default_rng(seed); out = bytearray(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes())
    while len(out) < n:
        out += rng.integers(0, 256, int(rng.integers(*lit)), dtype=np.uint8).tobytes()
        ln = int(rng.integers(*mat)); back = int(rng.integers(ln, min(len(out), 65000)))
        st = len(out) - back; out += out[st:st + ln]
    return bytes(out[:n])
def records(n, seed):
    rng = np.random.default_rng(seed); keys = [rng.integers(0, 256, 24, dtype=np.uint8).tobytes() for _ in range(200)]
    out = bytearray()
    while len(out) < n:
        out += keys[int(rng.integers(0, 200))] + rng.integers(0, 256, int(rng.integers(40, 400)), dtype=np.uint8).tobytes()
    return bytes(out[:n])
N = 8 << 20
inputs = {"synth50": datagen.synth50(N, 1234).tobytes(), "rows128": rows(N, 128, 1), "rows256": rows(N, 256, 2), "rows2048": rows(N, 2048, 3), "rows200": rows(200 * (N // 200), 200, 4),
          "mix": mix(N, 5), "mix_short": mix(N, 6, (30, 300), (8, 60)), "mix_long": mix(N, 7, (1000, 9000), (100, 5000)), "records": records(N, 8),
          "structured": datagen.structured(N, 5), "ints": datagen.ints_100000() * 8}
out = {}
for name, data in inputs.items():
    for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=0)):
        p = conduit.make_preferences(blockSizeID=kw["bsid"], blockMode=kw["indep"])
        f = gpu(data, p)
        ok = oracle.decompress_frame(f, cap=len(data) + 64)[0] == data
        ref = len(oracle.conduit_compress(data, oracle.mkprefs(**kw)))
        out[name + ("/4M" if kw["bsid"] == 7 else "/64kL")] = [round(len(f) / ref, 4), ok]
print("RS_RESULT " + json.dumps(out))
''' % ROOT
names = sys.argv[1:]
libs = {}
if not names:
    libs["base"] = os.path.join(ROOT, "lz4_frame_conduit_amd", "liblz4f_mi355x.so")
    for f in sorted(os.listdir(AB)) if os.path.isdir(AB) else []:
        if f.startswith("lib_") and f.endswith(".so"): libs[f[4:-3]] = os.path.join(AB, f)
else:
    for n in names: libs[n] = os.path.join(ROOT, "lz4_frame_conduit_amd", "liblz4f_mi355x.so") if n == "base" else os.path.join(AB, "lib_%s.so" % n)
rows_ = {}
for n, path in libs.items():
    env = dict(os.environ); env["LZ4F_MI355X_LIB"] = path
    try:
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=400)
    except subprocess.TimeoutExpired:
        print("%-10s TIMEOUT" % n, flush=True); break
    line = [l for l in r.stdout.splitlines() if l.startswith("RS_RESULT ")]
    if r.returncode != 0 or not line:
        print("%-10s FAILED rc=%d %s" % (n, r.returncode, r.stderr[-400:].replace("\n", " | ")), flush=True); continue
    rows_[n] = json.loads(line[0][10:])
keys = list(next(iter(rows_.values())).keys()) if rows_ else []
print("size / liblz4's size  " + "  ".join("%-9s" % n for n in rows_))
for k in keys:
    print("%-20s  " % k + "  ".join("%-9s" % ("%.4f%s" % (rows_[n][k][0], "" if rows_[n][k][1] else "!RT")) for n in rows_), flush=True)
