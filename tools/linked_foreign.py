"""Dev tool (GPU box): a linked frame WITHOUT trailer or index (as liblz4 writes it) decoded from the stream alone: argv MiB bsid kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
eng = Engine(0); eng.set_timing(True)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
bsid = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = sys.argv[3] if len(sys.argv) > 3 else "synth50"
src = torch.from_numpy(datagen.synth_text(64 << 20, 99)).cuda().repeat(n // (64 << 20)) if kind == "text" else synth50_device(n, 1234)
p = conduit.make_preferences(blockSizeID=bsid, blockMode=0)
frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda"); back = torch.empty_like(src)
eng.compress_async(src, frame, p); r = eng.result()
for _ in range(3):
    back.zero_(); eng.decompress_frame_async(frame, int(r.size), back); r2 = eng.result(); td = eng.get_timing()
print("%s linked bsid %d, %d MiB, no index: decompress %s ok=%s path %s" % (kind, bsid, n >> 20, {k: round(v, 2) for k, v in td.items() if v and k not in ("find_matches", "layout", "emit")}, bool(torch.equal(back, src)), hex(r2.flags >> 12)))
