#!/usr/bin/env python3
"""Development (GPU box): the self-feeding copy kernel on the headline decode (4 GiB synth50, in-band index): total / copy ms and the cycle counters of
workgroup 0 (its feeding wave: parse rounds, descriptor loads, resolve, waiting for ring room; its copier waves).  `foreign` as argument: the bare frame."""
import ctypes, os, sys
os.environ["LZ4F_MI355X_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import _ffi, conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = 4 << 30
src = synth50_device(n, 1234, "cuda"); eng = Engine(0); eng.set_timing(True)
p = conduit.make_preferences(blockSizeID=7, blockMode=1)
frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda")
inband = "foreign" not in sys.argv
eng.compress_async(src, frame, p, inband=inband); rc = eng.result()
back = torch.zeros(n, dtype=torch.uint8, device="cuda")
for it in range(3):
    if it == 2: ctypes.CDLL(_ffi.LIB_PATH).lz4f_mi355x_debug_prof((ctypes.c_ulonglong * 128)())      # (the grid-wide words: of the last decode only)
    eng.decompress_frame_async(frame, int(rc.size), back); r = eng.result(); t = eng.get_timing()
print("decode total %.3f ms (parse kernels %.3f, copy %.3f) ok=%s" % (t["decompress_total"], t["decode_parse"], t["decode_copy"], bool(torch.equal(back, src))))
buf = (ctypes.c_ulonglong * 128)()
print("rc", ctypes.CDLL(_ffi.LIB_PATH).lz4f_mi355x_debug_prof(buf))
print("feeder", {k: int(buf[i]) for i, k in enumerate(["total", "ring_wait", "nseq", "parse", "desc_load", "resolve", "rounds"])})
for w in range(1, 8):
    print("copier", w, {k: int(buf[8 * w + i]) for i, k in enumerate(["wait_produced", "literals", "wait_chain", "matches", "slots", "dep", "drain"])})
g = [int(buf[i]) for i in range(64, 76)]
print("grid: workgroup cycles max %d avg %d; feeder total max %d avg %d, ring wait avg %d, parse avg %d, resolve avg %d; starts spread %.1f us, ends spread %.1f us, first start to last end %.1f us" % (
    g[0], g[1] // 1024, g[2], g[3] // 1024, g[4] // 1024, g[9] // 1024, g[10] // 1024, (g[7] - g[6]) / 100.0, (g[8] - g[11]) / 100.0, (g[8] - g[6]) / 100.0))
c = [int(buf[i]) for i in range(76, 80)]
print("copier waves, average per wave: waiting for descriptors %d, literal phase %d, waiting for the chain %d, match phase %d cycles" % tuple(x // (1024 * 7) for x in c))
