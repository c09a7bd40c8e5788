#!/usr/bin/env python3
"""Dev tool (GPU box): decode every golden frame with the bulk call and report which ones fail."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from lz4_frame_conduit_amd import _ffi, datagen
from conftest import golden_file
L = _ffi.lib()
g = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "golden.json")))
txt = datagen.synth_text(2 << 20, 99).tobytes(); s50 = datagen.synth50(8 << 20, 1234).tobytes()
named = {"hello20": datagen.hello20(), "empty": b"", "rep42": datagen.rep42(), "ints": datagen.ints_100000(), "hello100k": datagen.hello_100000(),
         "tiny12": b"abcdefghijkl", "tiny13": b"abcdabcdabcda", "random10m": datagen.random_bytes(10 << 20, 7).tobytes(),
         "synth50_8m": s50, "synth50_2m": s50[:2 << 20], "text_2m": txt, "text512k": txt[:512 << 10]}
for key, ent in g["frames"].items():
    frame = golden_file(ent["file"]) if "file" in ent else (bytes.fromhex(ent["hex"]) if "hex" in ent else oracle.conduit_compress(named[ent["input"]], oracle.mkprefs(**ent["prefs"])))
    want = named[ent["input"]]
    cap = len(want) + 8
    dst = ctypes.create_string_buffer(max(cap, 1)); used = ctypes.c_size_t(0)
    r = L.lz4f_mi355x_decompressFrame(dst, cap, frame, len(frame), ctypes.byref(used))
    if L.LZ4F_isError(r):
        print(key, ent.get("prefs"), "ERROR", L.LZ4F_getErrorName(r).decode())
        # compare with checksum off: find first mismatch using oracle output
    elif dst.raw[:r] != want:
        got = np.frombuffer(dst.raw[:r], dtype=np.uint8); w = np.frombuffer(want, dtype=np.uint8)
        bad = np.nonzero(got[:len(w)] != w[:len(got)])[0]
        print(key, ent.get("prefs"), "MISMATCH", len(bad), bad[:5].tolist())
print("done")
