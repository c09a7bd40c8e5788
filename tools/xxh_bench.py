"""Dev tool (GPU box): block checksum and content checksum rates on the device path (4 MiB blocks of synth50)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4_frame_conduit_amd import conduit
from lz4_frame_conduit_amd.device import Engine, synth50_device
eng = Engine(0); eng.set_timing(True)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
src = synth50_device(n, 1234)
for name, kw in (("plain", {}), ("block checksums", dict(blockChecksum=1)), ("content checksum", dict(contentChecksum=1))):
    p = conduit.make_preferences(blockSizeID=7, blockMode=1, **kw)
    frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda"); back = torch.empty_like(src)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.compress_async(src, frame, p, inband=True); r = eng.result(); t1 = time.perf_counter()
        eng.decompress_frame_async(frame, int(r.size), back); r2 = eng.result(); t2 = time.perf_counter()
    print("%-18s compress %.2f ms  decompress %.2f ms  ok=%s" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, bool(torch.equal(back, src))), flush=True)
