#!/usr/bin/env python3
"""Dev tool (GPU box): compress a small crafted input on the GPU and list the sequences of block 0."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from lz4_frame_conduit_amd import _ffi, conduit
def seqs(frame):
    pos = 7; w = int.from_bytes(frame[pos:pos+4],'little'); pos += 4
    csz = w & 0x7fffffff; blk = frame[pos:pos+csz]; out=[]; p=0; op=0
    while p < len(blk):
        t = blk[p]; p+=1; lit = t>>4
        if lit==15:
            while True:
                b=blk[p]; p+=1; lit+=b
                if b!=255: break
        p += lit; op += lit
        if p >= len(blk): out.append((op-lit, lit,0,0)); break
        off = blk[p] | blk[p+1]<<8; p+=2; m = t&15
        if m==15:
            while True:
                b=blk[p]; p+=1; m+=b
                if b!=255: break
        m+=4; out.append((op-lit, lit,m,off)); op += m
    return out
L = _ffi.lib()
rng = np.random.default_rng(5)
case = sys.argv[1] if len(sys.argv) > 1 else "p15"
pre = rng.integers(0, 256, 9000, dtype=np.uint8).tobytes()
if case == "p15":
    pat = pre[100:115]; data = pre + b"xyz" + pat * 400 + rng.integers(0, 256, 500, dtype=np.uint8).tobytes()
elif case == "p15fresh":
    pat = rng.integers(0, 256, 15, dtype=np.uint8).tobytes(); data = pre + pat * 400 + rng.integers(0, 256, 500, dtype=np.uint8).tobytes()
elif case == "p1":
    data = pre + b"\x07" * 6000 + rng.integers(0, 256, 500, dtype=np.uint8).tobytes()
prefs = conduit.make_preferences(blockSizeID=7, blockMode=1)
cap = L.lz4f_mi355x_compressFrameBound(len(data), ctypes.byref(prefs)); dst = ctypes.create_string_buffer(cap)
r = L.lz4f_mi355x_compressFrame(dst, cap, data, len(data), ctypes.byref(prefs)); frame = dst.raw[:r]
assert oracle.decompress_frame(frame, len(data) + 64)[0] == data
ref = oracle.conduit_compress(data, oracle.mkprefs(bsid=7, indep=1))
print(case, "len", len(data), "ours", len(frame), "liblz4", len(ref))
print("ours :", seqs(frame)[:14])
print("lz4  :", seqs(ref)[:8])
