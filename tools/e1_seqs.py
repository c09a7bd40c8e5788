#!/usr/bin/env python3
"""Dev tool (GPU box): sequence statistics of the GPU encoder next to liblz4's (oracle) on one input.
argv: kind (text|rep42|synth50|ints) bytes bsid linked"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
def blocks(frame):
    pos = 7
    while True:
        w = int.from_bytes(frame[pos:pos+4], 'little'); pos += 4
        if w == 0: return
        csz = w & 0x7fffffff
        yield (w >> 31), frame[pos:pos+csz]; pos += csz
def seqs(blk):
    out=[]; p=0; op=0
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
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
bsid = int(sys.argv[3]) if len(sys.argv) > 3 else 4
linked = int(sys.argv[4]) if len(sys.argv) > 4 else 0
data = {"text": lambda: datagen.synth_text(n, 99).tobytes(), "rep42": lambda: datagen.rep42(n), "synth50": lambda: datagen.synth50(n, 1234).tobytes(),
        "ints": lambda: datagen.ints_100000()[:n], "hello": lambda: datagen.hello_100000()[:n]}[kind]()
src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda(); eng = Engine(0)
p = conduit.make_preferences(blockSizeID=bsid, blockMode=0 if linked else 1)
frame = torch.empty(eng.frame_bound(len(data), p), dtype=torch.uint8, device="cuda")
eng.compress_async(src, frame, p); r = eng.result()
ours = frame[:r.size].cpu().numpy().tobytes()
ref = oracle.conduit_compress(data, oracle.mkprefs(bsid=bsid, indep=0 if linked else 1))
assert oracle.decompress_frame(ours, len(data) + 64)[0] == data
for name, f in (("ours", ours), ("lz4 ", ref)):
    ss = [s for raw, b in blocks(f) if not raw for s in seqs(b)]
    nm = sum(1 for s in ss if s[2]); lits = sum(s[1] for s in ss); mb = sum(s[2] for s in ss)
    offs = collections.Counter(min(15, (s[3]).bit_length()) for s in ss if s[2])
    print(name, "size", len(f), "seqs", len(ss), "matches", nm, "literal bytes", lits, "match bytes", mb, "mean mlen %.1f" % (mb / max(1, nm)),
          "offset bits", sorted(offs.items()))
    print("   first:", ss[:12])
