#!/usr/bin/env python3
"""Dev tool (GPU box): soak of the indexed decode path - random structured / synth50 / text inputs, random sizes and block
sizes, every frame decoded with and without the index and compared with the source.  argv: cases [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = Engine(0)
used = bad = 0
t0 = time.time()
for i in range(cases):
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(1, 40 << 20))
    if kind == 0: data = np.frombuffer(datagen.structured(n, int(rng.integers(1 << 30))), dtype=np.uint8).copy()
    elif kind == 1: data = datagen.synth50(max(1024, n & ~1023), int(rng.integers(1 << 30)))[:n]
    elif kind == 2: data = datagen.synth_text(n, int(rng.integers(1 << 30)))
    else:
        parts = []
        while sum(len(p) for p in parts) < n:
            m = int(rng.integers(1, 3 << 20))
            parts.append(rng.integers(0, 256, m, dtype=np.uint8) if rng.integers(0, 2) else np.frombuffer(datagen.structured(m, int(rng.integers(1 << 30))), dtype=np.uint8))
        data = np.concatenate(parts)[:n].copy()
    linked = int(rng.integers(0, 3)) == 0
    only = os.environ.get('SOAK_ONLY')
    bsid = int(rng.integers(4, 8)) if linked else int(rng.integers(5, 8))
    bck = int(rng.integers(0, 2))
    if os.environ.get('SOAK_BSID'): bsid = int(os.environ['SOAK_BSID'])
    p = conduit.make_preferences(blockSizeID=bsid, blockMode=0 if linked else 1, blockChecksum=bck)
    bs = 1 << (8 + 2 * bsid)
    if only is not None and int(only) != i:
        rng.choice([1, 1, 4, 16]); continue
    src = torch.from_numpy(data).cuda(); nb = (src.numel() + bs - 1) // bs
    frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
    table = eng.new_table(nb)
    index = torch.zeros(eng.index_size(src.numel(), p) * int(rng.choice([1, 1, 4, 16])), dtype=torch.uint8, device="cuda")
    eng.compress_async(src, frame, p, table, index); r = eng.result()
    used += int(index[:4].cpu().numpy().view(np.uint32)[0] == 0x3258494C)
    for ix in (index, None, index):
        back = torch.empty_like(src)
        eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo, ix); r2 = eng.result()
        if r2.size != src.numel() or not torch.equal(back, src):
            if only is not None:
                import oracle
                host = frame[:r.size].cpu().numpy().tobytes()
                try:
                    out, used = oracle.decompress_frame(host, cap=src.numel() + 64)
                    print("oracle decode of the GPU frame: used", used, "of", r.size, "equal to source:", out == data.tobytes())
                except Exception as ex:
                    print("oracle rejects the GPU frame:", ex)
                d = (back != src).nonzero().flatten()
                print("first diff", int(d[0]), "last", int(d[-1]), "count", int(d.numel()), "block", int(d[0]) // bs, "r2.size", r2.size)
                tb = table.cpu().numpy().view(np.uint8)[: nb * 24].reshape(nb, 24)
                bb = int(d[0]) // bs
                for q in range(max(0, bb - 2), min(nb, bb + 2)):
                    w = int(np.frombuffer(tb[q].tobytes()[16:20], dtype=np.uint32)[0]); so = int(np.frombuffer(tb[q].tobytes()[0:8], dtype=np.uint64)[0])
                    print("  block", q, "stored" if w >> 31 else "compressed", "csize", w & 0x7FFFFFFF, "src_off", so)
                # runs of differing bytes
                dd = d.cpu().numpy(); cuts = np.nonzero(np.diff(dd) > 64)[0]
                starts = np.concatenate(([dd[0]], dd[cuts + 1])); ends = np.concatenate((dd[cuts], [dd[-1]]))
                for a_, b_ in list(zip(starts, ends))[:8]: print("  diff run", int(a_) - bb * bs, "..", int(b_) - bb * bs, "(in block)")
                # sequences of the block around the first difference
                w = int(np.frombuffer(tb[bb].tobytes()[16:20], dtype=np.uint32)[0]); so = int(np.frombuffer(tb[bb].tobytes()[0:8], dtype=np.uint64)[0])
                pl = host[so: so + (w & 0x7FFFFFFF)]
                pos = 0; op = 0; k = 0; first = int(d[0]) - bb * bs
                while pos < len(pl):
                    t = pl[pos]; pos += 1; lit = t >> 4
                    if lit == 15:
                        while True:
                            v = pl[pos]; pos += 1; lit += v
                            if v != 255: break
                    p0 = pos; pos += lit
                    if pos >= len(pl):
                        if op + lit >= first - 200000: print('   seq', k, 'FINAL lit', lit, 'at out', op)
                        break
                    off = pl[pos] | (pl[pos + 1] << 8); pos += 2; ml = t & 15
                    if ml == 15:
                        while True:
                            v = pl[pos]; pos += 1; ml += v
                            if v != 255: break
                    ml += 4
                    if op + lit + ml >= first - 300 and op <= first + 6000 and k < 100000:
                        print('   seq', k, 'lit', lit, 'match', ml, 'off', off, 'out', op, 'match dst', op + lit, 'src', op + lit - off, flush=True)
                    op += lit + ml; k += 1
                import subprocess
            bad += 1; print("MISMATCH case", i, "kind", kind, "n", n, "bsid", bsid, "linked", linked, "indexed", ix is not None, flush=True)
    if i % 10 == 9: print("case", i + 1, "usable indexes", used, "bad", bad, "%.0f s" % (time.time() - t0), flush=True)
print("done: cases", cases, "usable indexes", used, "mismatches", bad)
sys.exit(1 if bad else 0)
