#!/usr/bin/env python3
"""Development (GPU box): pass E1 / whole-compress time and ratio of library variants (tools/ab_build.sh), each in a child process.
    python3 tools/ab_run.py [name ...]        (no names: every lib_*.so under lz4_frame_conduit_amd/build/ab, plus the in-tree library as 'base')
Per variant: synth50 4 GiB in 4 MiB blocks (the headline input), synth50 1 GiB in 64 KiB linked blocks, text 1 GiB in 64 KiB and in 4 MiB blocks (AB_CASES adds real_4m, real_64k: 128 MiB of real text);
best of 3 find_matches / emit ms, ratio, and a device round trip through the in-tree decoder of that variant."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "lz4_frame_conduit_amd", "build", "ab")
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from lz4_frame_conduit_amd import conduit, datagen
from lz4_frame_conduit_amd.device import Engine, synth50_device
eng = Engine(0); eng.set_timing(True)
out = {}
cases = os.environ.get("AB_CASES", "s50_4m,s50_64kL,text_64k,text_4m").split(",")
def run(name, src, bsid, mode):
    n = src.numel()
    p = conduit.make_preferences(blockSizeID=bsid, blockMode=mode)
    frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda")
    back = torch.empty_like(src)
    best = None
    for it in range(4):
        eng.compress_async(src, frame, p, inband=True); r = eng.result(); t = eng.get_timing()
        if it and (best is None or t["find_matches"] < best[0]): best = (t["find_matches"], t["emit"], int(r.size))
    back.zero_()
    eng.decompress_frame_async(frame, int(r.size), back); r2 = eng.result()
    ok = bool(r2.size == n and torch.equal(back, src))
    out[name] = {"e1_ms": round(best[0], 3), "emit_ms": round(best[1], 3), "ratio": round(n / r2.consumed, 4), "ok": ok}
if "s50_4m" in cases or "s50_64kL" in cases:
    s = synth50_device(4 << 30, 1234, "cuda")
    if "s50_4m" in cases: run("s50_4m", s, 7, 1)
    if "s50_64kL" in cases: run("s50_64kL", s[:1 << 30], 4, 0)
    del s
if "text_64k" in cases or "text_4m" in cases:
    tx = torch.from_numpy(datagen.synth_text(64 << 20, 99)).cuda().repeat(16)
    if "text_64k" in cases: run("text_64k", tx, 4, 1)
    if "text_4m" in cases: run("text_4m", tx, 7, 1)
if "real_4m" in cases or "real_64k" in cases:                          # real text: the image's Python sources and headers (tools/real_text.py), 128 MiB
    import numpy as np
    buf = bytearray()
    for root in ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include"):
        for dp, dn, fn in os.walk(root):
            for f in sorted(fn):
                if f.endswith((".py", ".h", ".hpp", ".txt", ".md", ".rst", ".json")):
                    try: buf += open(os.path.join(dp, f), "rb").read()
                    except OSError: pass
            if len(buf) >= (128 << 20): break
        if len(buf) >= (128 << 20): break
    rt = torch.from_numpy(np.frombuffer(bytes(buf[:128 << 20]), dtype=np.uint8).copy()).cuda()
    if "real_4m" in cases: run("real_4m", rt, 7, 1)
    if "real_64k" in cases: run("real_64k", rt, 4, 1)
print("AB_RESULT " + json.dumps(out))
''' % ROOT
names = sys.argv[1:]
libs = {}
if not names:
    libs["base"] = os.path.join(ROOT, "lz4_frame_conduit_amd", "liblz4f_mi355x.so")
    for f in sorted(os.listdir(AB)) if os.path.isdir(AB) else []:
        if f.startswith("lib_") and f.endswith(".so"): libs[f[4:-3]] = os.path.join(AB, f)
else:
    for n in names: libs[n] = os.path.join(ROOT, "lz4_frame_conduit_amd", "liblz4f_mi355x.so") if n == "base" else os.path.join(AB, "lib_%s.so" % n)
for n, path in libs.items():
    env = dict(os.environ); env["LZ4F_MI355X_LIB"] = path
    try:
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        print("%-14s TIMEOUT" % n, flush=True); break                      # (a hung variant: no further GPU work in this call)
    line = [l for l in r.stdout.splitlines() if l.startswith("AB_RESULT ")]
    if r.returncode != 0 or not line:
        print("%-14s FAILED rc=%d %s" % (n, r.returncode, r.stderr[-300:].replace("\n", " | ")), flush=True); continue
    d = json.loads(line[0][10:])
    print("%-14s " % n + "  ".join("%s: e1 %.3f emit %.3f ratio %.4f%s" % (k, v["e1_ms"], v["emit_ms"], v["ratio"], "" if v["ok"] else " ROUNDTRIP-FAIL") for k, v in d.items()), flush=True)
