"""The oracle (the CPU restatement every parity test leans on) under AddressSanitizer + UBSan: golden frames, the malformed set and
fresh mutations go through `oracle/liborc_asan.so` in a child process (the sanitizer runtime has to be loaded first).  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import glob, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import oracle
from lz4_frame_conduit_amd import datagen
n = 0
# compress + decompress over shapes and framings, then single-byte mutations of every frame (verdict or bytes, never a report)
rng = np.random.default_rng(5)
inputs = [b"", b"hello world, hello world, hello world!", bytes(70000), datagen.synth50(1 << 18, 3).tobytes(), datagen.synth_text(200000, 4).tobytes(),
          rng.integers(0, 256, 100000, dtype=np.uint8).tobytes(), datagen.structured(150000, 9)]
for data in inputs:
    for kw in (dict(bsid=4, indep=0), dict(bsid=4, indep=1, bck=1, cck=1), dict(bsid=5, indep=1), dict(bsid=7, indep=0, cck=1)):
        frame = oracle.conduit_compress(data, oracle.mkprefs(**kw))
        out, used = oracle.decompress_frame(frame, cap=len(data) + 64)
        assert out == data and used == len(frame)
        for _ in range(25):
            bad = bytearray(frame); bad[int(rng.integers(0, len(bad)))] ^= int(rng.integers(1, 256))
            try: oracle.decompress_frame(bytes(bad), cap=len(data) + 64)
            except oracle.OracleError: pass
            n += 1
        for cut in (1, 5, len(frame) // 2):
            try: oracle.decompress_frame(frame[:max(0, len(frame) - cut)], cap=len(data) + 64)
            except oracle.OracleError: pass
# the committed liblz4 frames
for f in sorted(glob.glob(os.path.join(%(root)r, "tests", "golden", "*.lz4"))):
    raw = open(f, "rb").read()
    try: oracle.decompress_frame(raw, cap=64 << 20)
    except oracle.OracleError: pass
    n += 1
print("sanitized calls", n)
'''


def test_oracle_clean_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan for this gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc_asan.so"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               ORC_LIB=os.path.join(ROOT, "oracle", "liborc_asan.so"))
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized calls" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
