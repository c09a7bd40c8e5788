"""oracle -- ctypes face of the CPU restatement in oracle/*.c.

TEST INFRASTRUCTURE ONLY (see oracle/orc.h): imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by lz4_frame_conduit_amd/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_size_t = ctypes.c_size_t


class FrameInfo(ctypes.Structure):
    _fields_ = [("blockSizeID", ctypes.c_uint32), ("blockMode", ctypes.c_uint32), ("contentChecksumFlag", ctypes.c_uint32),
                ("frameType", ctypes.c_uint32), ("contentSize", ctypes.c_uint64), ("dictID", ctypes.c_uint32),
                ("blockChecksumFlag", ctypes.c_uint32)]


class Prefs(ctypes.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int32), ("autoFlush", ctypes.c_uint32),
                ("favorDecSpeed", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 3)]


def mkprefs(bsid=0, indep=0, cck=0, bck=0, csize=0, dictid=0, autoflush=0) -> Prefs:
    p = Prefs()
    p.frameInfo.blockSizeID, p.frameInfo.blockMode = bsid, indep
    p.frameInfo.contentChecksumFlag, p.frameInfo.blockChecksumFlag = cck, bck
    p.frameInfo.contentSize, p.frameInfo.dictID = csize, dictid
    p.autoFlush = autoflush
    return p


def build(force: bool = False) -> str:
    """Compile oracle/*.c (gcc) -> oracle/liborc.so and oracle/orc_cpu_baseline."""
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_xxh32.c", "orc_lz4block.c", "orc_lz4frame.c", "orc.h", "orc_cpu_baseline.c")]
    exe = os.path.join(_HERE, "orc_cpu_baseline")
    stale = force or not (os.path.exists(so) and os.path.exists(exe)) or any(
        os.path.getmtime(s) > min(os.path.getmtime(so), os.path.getmtime(exe)) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # ORC_LIB: another build of the same sources (tests/test_oracle_sanitizers.py points it at the ASan/UBSan one)
        L = ctypes.CDLL(os.environ.get("ORC_LIB") or build())
        vp, u8p = ctypes.c_void_p, ctypes.c_char_p
        sig = {
            "orc_xxh32": (ctypes.c_uint32, [vp, c_size_t, ctypes.c_uint32]),
            "orc_lz4_compress_bound": (ctypes.c_int, [ctypes.c_int]),
            "orc_lz4_compress_default": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int]),
            "orc_lz4_decompress_safe": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, c_size_t]),
            "orc_lz4_count_sequences": (ctypes.c_int, [vp, ctypes.c_int]),
            "orc_is_error": (ctypes.c_uint, [c_size_t]),
            "orc_error_name": (ctypes.c_char_p, [c_size_t]),
            "orc_block_size": (c_size_t, [ctypes.c_uint32]),
            "orc_compress_bound": (c_size_t, [c_size_t, ctypes.POINTER(Prefs)]),
            "orc_write_header": (c_size_t, [vp, c_size_t, ctypes.POINTER(Prefs)]),
            "orc_conduit_compress": (c_size_t, [vp, c_size_t, ctypes.POINTER(Prefs), c_size_t, vp, c_size_t]),
            "orc_decompress_frame": (c_size_t, [vp, c_size_t, vp, c_size_t, ctypes.POINTER(c_size_t), ctypes.POINTER(FrameInfo)]),
        }
        for k, (r, a) in sig.items():
            f = getattr(L, k)
            f.restype, f.argtypes = r, a
        _LIB = L
    return _LIB


class OracleError(Exception):
    pass


def _buf(b):
    """bytes / bytearray / numpy uint8 array -> (ctypes pointer-able object, length)."""
    import numpy as np
    if isinstance(b, np.ndarray):
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return b.ctypes.data_as(ctypes.c_void_p), b.size, b
    b = bytes(b)
    return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p), len(b), b


def xxh32(data, seed: int = 0) -> int:
    p, n, keep = _buf(data)
    return lib().orc_xxh32(p, n, seed)


def compress_block(data, dst_cap: int | None = None) -> bytes:
    """LZ4_compress_default restated; b"" when it does not fit dst_cap."""
    p, n, keep = _buf(data)
    cap = lib().orc_lz4_compress_bound(n) if dst_cap is None else dst_cap
    out = ctypes.create_string_buffer(max(cap, 1))
    r = lib().orc_lz4_compress_default(p, out, n, cap)
    return out.raw[:r]


def decompress_block(payload, dst_cap: int, history: bytes = b"") -> bytes:
    """LZ4_decompress_safe(_usingDict) restated; raises OracleError on malformed input."""
    p, n, keep = _buf(payload)
    h = bytes(history)[-65536:]
    out = ctypes.create_string_buffer(len(h) + max(dst_cap, 1))
    ctypes.memmove(out, h, len(h))
    r = lib().orc_lz4_decompress_safe(p, ctypes.byref(out, len(h)), n, dst_cap, len(h))
    if r < 0:
        raise OracleError("block decode failed")
    return out.raw[len(h):len(h) + r]


def count_sequences(payload) -> int:
    p, n, keep = _buf(payload)
    return lib().orc_lz4_count_sequences(p, n)


def compress_bound(n: int, prefs: Prefs | None) -> int:
    return lib().orc_compress_bound(n, ctypes.byref(prefs) if prefs is not None else None)


def header_bytes(prefs: Prefs) -> bytes:
    out = ctypes.create_string_buffer(32)
    n = lib().orc_write_header(out, 32, ctypes.byref(prefs))
    return out.raw[:n]


def conduit_compress(data, prefs: Prefs | None = None, slice_: int = 16384) -> bytes:
    """The reference's compress conduit call pattern (Conduit.hsc:457-533) on the restatement."""
    p, n, keep = _buf(data)
    prefs = prefs if prefs is not None else mkprefs()
    bs = lib().orc_block_size(prefs.frameInfo.blockSizeID)
    cap = n + (n // bs + 2) * 8 + 64
    out = ctypes.create_string_buffer(cap)
    r = lib().orc_conduit_compress(p, n, ctypes.byref(prefs), slice_, out, cap)
    if lib().orc_is_error(r):
        raise OracleError(lib().orc_error_name(r).decode())
    return out.raw[:r]


def decompress_frame(frame, cap: int | None = None):
    """First frame of `frame` -> (output bytes, consumed).  Raises OracleError(name) like LZ4F."""
    p, n, keep = _buf(frame)
    if cap is None:
        cap = max(n * 300, 1 << 16)
    out = ctypes.create_string_buffer(cap)
    used = c_size_t(0)
    fi = FrameInfo()
    r = lib().orc_decompress_frame(p, n, out, cap, ctypes.byref(used), ctypes.byref(fi))
    if lib().orc_is_error(r):
        raise OracleError(lib().orc_error_name(r).decode())
    return out.raw[:r], used.value
