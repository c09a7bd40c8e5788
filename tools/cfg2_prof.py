#!/usr/bin/env python3
"""Development (GPU box): BASELINE configs[1]'s decode (text, 64 KiB independent blocks, liblz4's frames through the oracle) with a -DDB_PROF build
(tools/ab_build.sh dbprof "-DDB_PROF"; LZ4F_MI355X_LIB=.../lib_dbprof.so): where the lanes' path of the wave-per-block decoder spends its cycles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import _ffi, conduit, datagen
from lz4_frame_conduit_amd.device import Engine
tile = datagen.synth_text(64 << 20, 99)
fr = oracle.conduit_compress(tile.tobytes(), oracle.mkprefs(bsid=4, indep=1))
dev = torch.from_numpy(np.frombuffer(fr + bytes(64), dtype=np.uint8).copy()).cuda()
back = torch.zeros(len(tile) + 64, dtype=torch.uint8, device="cuda")
eng = Engine(0); eng.set_timing(True)
lib = ctypes.CDLL(_ffi.LIB_PATH); buf = (ctypes.c_ulonglong * 128)()
for it in range(3):
    if it == 2: lib.lz4f_mi355x_debug_prof(buf)
    eng.decompress_frame_async(dev, len(fr), back); r = eng.result(); t = eng.get_timing()
print("64 MiB of text: decode %.3f ms ok=%s" % (t["decode"], bool(r.size == len(tile) and bytes(back[:len(tile)].cpu().numpy().tobytes()) == tile.tobytes())))
print("rc", lib.lz4f_mi355x_debug_prof(buf))
z = [int(buf[96 + i]) for i in range(11)]
w = max(z[10], 1); nw = max(z[6], 1)
print("waves %d; per wave: windows %d, rounds %d, ordered matches %d, scalar sequences %d" % (z[10], z[6] // w, z[7] // w, z[8] // w, z[9] // w))
print("cycles per window: wait for the window's bytes %d, parse + prefix %d, literal store %d, rounds %d, ordered loop %d; scalar path per sequence %d (total per wave %d)" % (
    z[0] // nw, z[1] // nw, z[2] // nw, z[3] // nw, z[4] // nw, z[5] // max(z[9], 1), sum(z[:6]) // w))
