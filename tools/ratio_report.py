#!/usr/bin/env python3
"""Dev tool (GPU box): compressed sizes of the HIP encoder next to the oracle (== liblz4 1.9.3) on the synthetic inputs."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from lz4_frame_conduit_amd import _ffi, conduit, datagen

L = _ffi.lib()
def gpu(data, p):
    cap = L.lz4f_mi355x_compressFrameBound(len(data), ctypes.byref(p)); dst = ctypes.create_string_buffer(cap)
    r = L.lz4f_mi355x_compressFrame(dst, cap, data, len(data), ctypes.byref(p))
    assert not L.LZ4F_isError(r), (L.LZ4F_getErrorName(r), L.lz4f_mi355x_last_error())
    return dst.raw[:r]
inputs = {"ints": datagen.ints_100000(), "hello100k": datagen.hello_100000(), "text_2m": datagen.synth_text(2 << 20, 99).tobytes(),
          "synth50_8m": datagen.synth50(8 << 20, 1234).tobytes(), "rep42": datagen.rep42()}
variants = [("", "")] if len(sys.argv) < 2 else [tuple(v.split(":")) for v in sys.argv[1:]]
for chunk, seed in variants:
  if chunk: os.environ["LZ4F_MI355X_CHUNK"] = chunk
  if seed: os.environ["LZ4F_MI355X_SEED"] = seed
  print("== chunk", chunk or "default", "seed", seed or "default")
  for name, data in inputs.items():
    for kw in (dict(bsid=4, indep=1), dict(bsid=7, indep=1), dict()):
      p = conduit.make_preferences(blockSizeID=kw.get("bsid", 0), blockMode=kw.get("indep", 0))
      f = gpu(data, p)
      ref = oracle.conduit_compress(data, oracle.mkprefs(**kw))
      ok = oracle.decompress_frame(f, cap=len(data) + 64)[0] == data
      print("%-11s %-22s gpu %9d  liblz4 %9d  gpu/liblz4 %.4f  ratio %.3f  roundtrip %s" % (name, kw, len(f), len(ref), len(f) / len(ref), len(data) / len(f), ok))
