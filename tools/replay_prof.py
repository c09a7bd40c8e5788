#!/usr/bin/env python3
"""Development (GPU box, under rocprofv3 --kernel-trace --stats): the reference's call pattern on 8 MiB - LZ4F_compressUpdate per 16 KiB slice,
LZ4F_decompress per slice - so that the per-block kernels and the gaps between them can be read from the trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lz4_frame_conduit_amd import conduit, datagen
data = datagen.synth50(8 << 20, 3).tobytes()
chunks = [data[i:i + 16384] for i in range(0, len(data), 16384)]
for it in range(2):
    t0 = time.perf_counter(); z = b"".join(conduit.compress(chunks)); t1 = time.perf_counter()
    zc = [z[i:i + 16384] for i in range(0, len(z), 16384)]
    back = b"".join(conduit.decompress(zc)); t2 = time.perf_counter()
    print("compress %.1f us per 64 KiB block, decompress %.1f us per block, ok=%s" % ((t1 - t0) / 128 * 1e6, (t2 - t1) / 128 * 1e6, back == data), flush=True)
