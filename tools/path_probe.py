"""Developer aid: which way a foreign frame goes through the decoder (result.flags >> 12), with the engine's PROF output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import datagen
from lz4_frame_conduit_amd.device import Engine

def main():
    eng = Engine(0)
    for name, data, kw in (("synth50 64K linked", datagen.synth50(16 << 20, 21), dict(bsid=4, indep=0)),
                           ("synth50 4M linked", datagen.synth50(16 << 20, 21), dict(bsid=7, indep=0)),
                           ("text 4M linked", datagen.synth_text(8 << 20, 22), dict(bsid=7, indep=0))):
        fr = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))
        dev = torch.from_numpy(np.frombuffer(fr, dtype=np.uint8).copy()).cuda()
        src = torch.from_numpy(data).cuda()
        back = torch.zeros_like(src)
        eng.decompress_frame_async(dev, dev.numel(), back)
        r = eng.result()
        print(name, "path", hex(r.flags >> 12), "ok", bool(torch.equal(back, src)), flush=True)
    eng.close()

main()
