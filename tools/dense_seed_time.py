#!/usr/bin/env python3
"""Development (GPU box): decode time of the text frame tools/fuzz_foreign.py makes for a seed (9 MiB, 4 MiB independent blocks), both dense decoders.  argv: seeds..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import _ffi, datagen
from lz4_frame_conduit_amd.device import Engine
L = _ffi.lib()
for seed in [int(a) for a in sys.argv[1:]] or [5, 21]:
    rng = np.random.default_rng(seed)
    _ = np.concatenate([datagen.synth50(9 << 20, int(rng.integers(1 << 30))), rng.integers(0, 256, 70001, dtype=np.uint8)])
    data = datagen.synth_text(9 << 20, int(rng.integers(1 << 30)))
    fr = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(bsid=7, indep=1))
    dev = torch.from_numpy(np.frombuffer(fr + bytes(64), dtype=np.uint8).copy()).cuda()
    back = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    for mode in ("1", "2"):
        os.environ["LZ4F_MI355X_DENSE_MODE"] = mode
        L.lz4f_mi355x_release_engines()
        eng = Engine(0); eng.set_timing(True)
        best = None
        for it in range(3):
            eng.decompress_frame_async(dev, len(fr), back); r = eng.result(); t = eng.get_timing()["decompress_total"]
            best = t if best is None or t < best else best
        ok = r.size == len(data) and bytes(back[:len(data)].cpu().numpy().tobytes()) == data.tobytes()
        print("seed", seed, "ratio %.3f" % (len(data) / len(fr)), "mode", mode, "decompress %.2f ms" % best, "ok", ok, flush=True)
        eng.close()
