#!/usr/bin/env python3
"""Development (GPU box): text in 4 MiB independent blocks through the workgroup-per-block decoder (decode_relay.cuh) with a -DDB_PROF build
(tools/ab_build.sh dbprof "-DDB_PROF"; LZ4F_MI355X_LIB=.../lib_dbprof.so): where a producer wave's cycles go, per window."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import _ffi, datagen
from lz4_frame_conduit_amd.device import Engine
tile = datagen.synth_text(128 << 20, 99)
if len(sys.argv) > 1 and sys.argv[1] == "real":                           # the image's Python sources and headers (tools/real_text.py)
    b = bytearray()
    for root in ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include"):
        for dp, dn, fn in os.walk(root):
            for f in sorted(fn):
                if f.endswith((".py", ".h", ".hpp", ".txt", ".md", ".rst", ".json")):
                    try: b += open(os.path.join(dp, f), "rb").read()
                    except OSError: pass
            if len(b) >= (128 << 20): break
        if len(b) >= (128 << 20): break
    tile = np.frombuffer(bytes(b[:128 << 20]), dtype=np.uint8).copy()
fr = oracle.conduit_compress(tile.tobytes(), oracle.mkprefs(bsid=7, indep=1))
dev = torch.from_numpy(np.frombuffer(fr + bytes(64), dtype=np.uint8).copy()).cuda()
back = torch.zeros(len(tile) + 64, dtype=torch.uint8, device="cuda")
eng = Engine(0); eng.set_timing(True)
lib = ctypes.CDLL(_ffi.LIB_PATH); buf = (ctypes.c_ulonglong * 128)()
for it in range(3):
    if it == 2: lib.lz4f_mi355x_debug_prof(buf)
    eng.decompress_frame_async(dev, len(fr), back); r = eng.result(); t = eng.get_timing()
print("128 MiB of text, 4 MiB blocks: decompress %.3f ms ok=%s" % (t["decompress_total"], bool(r.size == len(tile) and bytes(back[:len(tile)].cpu().numpy().tobytes()) == tile.tobytes())))
print("rc", lib.lz4f_mi355x_debug_prof(buf))
z = [int(buf[96 + i]) for i in range(32)]
nw = max(z[6], 1)
print("windows %d (16 blocks), sequences taken one at a time %d; per window: matches copied at once %.2f, left to the finishing wave %.2f" % (z[6], z[10], z[7] / nw, z[9] / nw))
print("producer, cycles per window: wait for the turn %d, table entry + pass + window bytes %d, hops + sum %d, literals + wait for the finishing wave to be near + early matches %d, post %d" % (
    z[0] // nw, z[1] // nw, z[2] // nw, (z[3] + z[4]) // nw, z[5] // nw))
ng = max(z[18], 1); nf = max(z[22], 1)
print("speculators: groups %d (%.2f per window), cycles per group: waiting %d, working %d (%.1f two-token steps)" % (z[18], z[18] / nw, z[16] // ng, z[17] // ng, z[19] / ng))
print("finishing wave: windows %d, cycles per window: waiting for the slot %d, copying + publishing %d; byte-per-lane plans %.2f of the windows, listed matches %.2f per window" % (z[22], z[20] // nf, z[21] // nf, z[23] / nf, z[24] / nf))
print("finishing wave found the slot it had read ahead posted: %.2f of the windows; a producer posts %.2f windows ahead of the finishing wave" % (z[25] / nf, z[26] / nw))
