#!/usr/bin/env python3
"""Dev tool (GPU box): repeat compress -> decompress of synth50 on the device; on a mismatch say which side is wrong
(the oracle decodes the offending block's payload on the host)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = Engine(0)
p = conduit.make_preferences(blockSizeID=7, blockMode=1, blockChecksum=int(os.environ.get("BCK", "1")))
nb = n >> 22
bad_runs = 0
for it in range(iters):
    src = synth50_device(n, 4321 + it % 3)
    frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda"); table = eng.new_table(nb)
    eng.compress_async(src, frame, p, table); r = eng.result()
    for rep in range(3):
        back = torch.empty_like(src)
        eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo); r2 = eng.result()
        if r2.size == n and torch.equal(back, src): continue
        bad_runs += 1
        neq = (back != src).nonzero().flatten()
        first = int(neq[0]); blk = first >> 22
        print("iter", it, "rep", rep, "MISMATCH bytes", int(neq.numel()), "first", first, "block", blk, "offset in block", first & ((1 << 22) - 1), "last", int(neq[-1]))
        t = table.cpu().numpy().view(np.uint8)[:nb * 24].reshape(nb, 24)
        e = np.frombuffer(t[blk].tobytes(), dtype=np.uint64)
        src_off = int(e[0]); word = int(np.frombuffer(t[blk].tobytes()[16:20], dtype=np.uint32)[0])
        payload = frame[src_off:src_off + (word & 0x7fffffff)].cpu().numpy().tobytes()
        try:
            dec = oracle.decompress_block(payload, 1 << 22)
            ok = dec == src[blk << 22:(blk + 1) << 22].cpu().numpy().tobytes()
            print("   oracle decode of that block's payload matches the source:", ok, "-> the", "DECODER" if ok else "ENCODER", "is wrong")
        except Exception as ex:
            print("   oracle rejects the payload:", ex, "-> the ENCODER is wrong")
        break
print("bad runs", bad_runs, "of", iters)
