"""Python face of the conduits (tests and scripts).  The drivers themselves are the C++ mirror of
Codec.Compression.LZ4.Conduit in csrc/conduit.cpp; this file only feeds them chunks and collects what
they yield.  Names follow the reference (/root/reference/src/Codec/Compression/LZ4/Conduit.hsc:58-89).
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List, Optional

from . import _ffi
from ._ffi import FrameInfo, Preferences

# BlockSizeID / BlockMode / ... -- CTypes.hsc:48-150
LZ4F_default, LZ4F_max64KB, LZ4F_max256KB, LZ4F_max1MB, LZ4F_max4MB = 0, 4, 5, 6, 7
LZ4F_blockLinked, LZ4F_blockIndependent = 0, 1
LZ4F_noContentChecksum, LZ4F_contentChecksumEnabled = 0, 1
LZ4F_noBlockChecksum, LZ4F_blockChecksumEnabled = 0, 1


class Lz4FrameError(Exception):
    """What `throwString` raises in the reference (Conduit.hsc:160, :616, :689)."""


def lz4DefaultPreferences() -> Preferences:           # Conduit.hsc:248-263
    return Preferences()


def make_preferences(blockSizeID=LZ4F_default, blockMode=LZ4F_blockLinked, contentChecksum=0, blockChecksum=0,
                     contentSize=0, dictID=0, autoFlush=0) -> Preferences:
    p = Preferences()
    p.frameInfo.blockSizeID, p.frameInfo.blockMode = blockSizeID, blockMode
    p.frameInfo.contentChecksumFlag, p.frameInfo.blockChecksumFlag = contentChecksum, blockChecksum
    p.frameInfo.contentSize, p.frameInfo.dictID = contentSize, dictID
    p.autoFlush = autoFlush
    return p


def bsChunksOf(chunkSize: int, bs: bytes) -> List[bytes]:     # Conduit.hsc:428-433
    if chunkSize < 1:
        raise ValueError("bsChunksOf: chunkSize < 1: %d" % chunkSize)
    out = []
    while len(bs) > chunkSize:
        out.append(bs[:chunkSize]); bs = bs[chunkSize:]
    out.append(bs)
    return out


def _run(entry, chunks: Iterable[bytes], *lead):
    L = _ffi.lib()
    it = iter(chunks)
    keep = [None]
    out: List[bytes] = []

    def _await(_user, pdata):
        try:
            b = next(it)
        except StopIteration:
            pdata[0] = None
            return 0
        buf = ctypes.create_string_buffer(bytes(b), max(len(b), 1))
        keep[0] = buf
        pdata[0] = ctypes.cast(buf, ctypes.c_void_p).value
        return len(b)

    def _yield(_user, data, size):
        out.append(ctypes.string_at(data, size) if size else b"")

    err = ctypes.create_string_buffer(512)
    rc = getattr(L, entry)(*lead, _ffi.AWAIT_FN(_await), _ffi.YIELD_FN(_yield), None, err, 512)
    if rc != 0:
        raise Lz4FrameError(err.value.decode())
    return out


def _p(prefs: Optional[Preferences]):
    return ctypes.byref(prefs) if prefs is not None else None


def compress(chunks: Iterable[bytes]) -> List[bytes]:                                   # Conduit.hsc:336-337
    return _run("lz4f_mi355x_conduit_compress", chunks, 0, None)


def compressWithOutBufferSize(bufferSize: int, chunks: Iterable[bytes], prefs: Optional[Preferences] = None) -> List[bytes]:   # :457-533
    return _run("lz4f_mi355x_conduit_compress", chunks, bufferSize, _p(prefs))


def compressWithPreferences(prefs: Preferences, chunks: Iterable[bytes]) -> List[bytes]:
    return _run("lz4f_mi355x_conduit_compress", chunks, 0, _p(prefs))


def compressYieldImmediately(chunks: Iterable[bytes], prefs: Optional[Preferences] = None) -> List[bytes]:   # :364-425
    return _run("lz4f_mi355x_conduit_compress_yield_immediately", chunks, _p(prefs))


def decompress(chunks: Iterable[bytes]) -> List[bytes]:                                 # Conduit.hsc:598-701
    return _run("lz4f_mi355x_conduit_decompress", chunks)


def compressBatched(chunks: Iterable[bytes], prefs: Optional[Preferences] = None, batchBytes: int = 64 << 20, blockList: bool = False) -> List[bytes]:
    """blockList: the frame's block list follows it as a skippable frame (include/lz4f_mi355x.h: lz4f_mi355x_appendBlockList)."""
    return _run("lz4f_mi355x_conduit_compress_batched_listed" if blockList else "lz4f_mi355x_conduit_compress_batched", chunks, batchBytes, _p(prefs))


def appendBlockList(frame: bytes) -> bytes:
    """A finished LZ4 frame (any encoder's) + its block list as a skippable frame; host work only (no GPU)."""
    L = _ffi.lib()
    need = L.lz4f_mi355x_blockListSize(frame, len(frame))
    if L.LZ4F_isError(need):
        raise Lz4FrameError("lz4frame error: " + L.LZ4F_getErrorName(need).decode())
    buf = ctypes.create_string_buffer(frame, len(frame) + need)
    r = L.lz4f_mi355x_appendBlockList(buf, len(frame), len(frame) + need)
    if L.LZ4F_isError(r):
        raise Lz4FrameError("lz4frame error: " + L.LZ4F_getErrorName(r).decode())
    return buf.raw[:r]


def decompressBatched(chunks: Iterable[bytes], batchBytes: Optional[int] = None) -> List[bytes]:
    """Every frame of the stream through the bulk path in bounded memory: runs of whole blocks, `batchBytes` (default 256 MiB) at a time."""
    if batchBytes is None:
        return _run("lz4f_mi355x_conduit_decompress_batched", chunks)
    return _run("lz4f_mi355x_conduit_decompress_batched_bounded", chunks, batchBytes)
