"""pytest configuration: markers + shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol export (no GPU needed).
`-m gpu`       : parity tests proper -- the HIP path through the C ABI vs the oracle.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def named_inputs():
    from lz4_frame_conduit_amd import datagen
    txt = datagen.synth_text(2 << 20, 99).tobytes()
    s50 = datagen.synth50(8 << 20, 1234).tobytes()
    return {
        "hello20": datagen.hello20(), "empty": b"", "rep42": datagen.rep42(), "ints": datagen.ints_100000(),
        "hello100k": datagen.hello_100000(), "tiny12": b"abcdefghijkl", "tiny13": b"abcdabcdabcda",
        "random10m": datagen.random_bytes(10 << 20, 7).tobytes(),
        "synth50_8m": s50, "synth50_2m": s50[:2 << 20], "text_2m": txt, "text512k": txt[:512 << 10],
    }


def golden_file(name):
    with open(os.path.join(GOLDEN_DIR, name), "rb") as f:
        return f.read()
