"""lz4_frame_conduit_amd -- MI355X-native LZ4 frame codec behind the Codec.Compression.LZ4.Conduit API.

The product is liblz4f_mi355x.so (HIP kernels for gfx950 + a C++ host layer, C ABI in
include/lz4f_mi355x.h).  This package is the thin Python face used by tests and bench.py:
`conduit` mirrors the reference's stream transformers, `device` drives the device-resident bulk
path on torch tensors.  (The directory is spelled with underscores so that Python can import it;
`lz4-frame-conduit_amd` is a symlink to it.)
"""
from . import _ffi  # noqa: F401
from ._ffi import build, lib  # noqa: F401
