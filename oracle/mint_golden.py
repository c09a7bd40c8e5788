#!/usr/bin/env python3
"""oracle/mint_golden.py -- mint the golden vectors under tests/golden/ (TEST INFRASTRUCTURE).

The reference (nh2/lz4-frame-conduit) holds no known-answer bytes: its tests pipe frames
through the external `lz4` CLI (/root/reference/test/Main.hs:27-36, :60-112) and check a
round trip (:114-119).  Its arithmetic is the lz4/lz4 C library, bundled as a git submodule
that is empty in /root/reference.  The same upstream library is installed in THIS container
as /usr/lib/x86_64-linux-gnu/liblz4.so.1 (v1.9.3), so this script drives it through ctypes
-- with the exact call pattern of the reference's conduits (Conduit.hsc:457-533, :598-701) --
and records what it produces.  The output (tests/golden/golden.json + a few .lz4 files) is
committed; the GPU box needs neither liblz4 nor /root/reference.

Run:  python oracle/mint_golden.py         (only here, never at test time)
"""
from __future__ import annotations

import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lz4_frame_conduit_amd import datagen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
L = ctypes.CDLL("/usr/lib/x86_64-linux-gnu/liblz4.so.1")
c_size_t, c_void_p = ctypes.c_size_t, ctypes.c_void_p


class FrameInfo(ctypes.Structure):
    _fields_ = [("blockSizeID", ctypes.c_uint), ("blockMode", ctypes.c_uint), ("contentChecksumFlag", ctypes.c_uint),
                ("frameType", ctypes.c_uint), ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                ("blockChecksumFlag", ctypes.c_uint)]


class Prefs(ctypes.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                ("favorDecSpeed", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]


for name, res, args in [
    ("LZ4F_isError", ctypes.c_uint, [c_size_t]),
    ("LZ4F_getErrorName", ctypes.c_char_p, [c_size_t]),
    ("LZ4F_createCompressionContext", c_size_t, [ctypes.POINTER(c_void_p), ctypes.c_uint]),
    ("LZ4F_freeCompressionContext", c_size_t, [c_void_p]),
    ("LZ4F_compressBegin", c_size_t, [c_void_p, c_void_p, c_size_t, ctypes.POINTER(Prefs)]),
    ("LZ4F_compressBound", c_size_t, [c_size_t, ctypes.POINTER(Prefs)]),
    ("LZ4F_compressUpdate", c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    ("LZ4F_compressEnd", c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p]),
    ("LZ4F_createDecompressionContext", c_size_t, [ctypes.POINTER(c_void_p), ctypes.c_uint]),
    ("LZ4F_freeDecompressionContext", c_size_t, [c_void_p]),
    ("LZ4F_getFrameInfo", c_size_t, [c_void_p, ctypes.POINTER(FrameInfo), c_void_p, ctypes.POINTER(c_size_t)]),
    ("LZ4F_decompress", c_size_t, [c_void_p, c_void_p, ctypes.POINTER(c_size_t), c_void_p, ctypes.POINTER(c_size_t), c_void_p]),
    ("LZ4_compress_default", ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int, ctypes.c_int]),
    ("LZ4_versionNumber", ctypes.c_int, []),
]:
    f = getattr(L, name)
    f.restype, f.argtypes = res, args


def mkprefs(bsid=0, indep=0, cck=0, bck=0, csize=0, dictid=0, autoflush=0) -> Prefs:
    p = Prefs()
    p.frameInfo.blockSizeID, p.frameInfo.blockMode = bsid, indep
    p.frameInfo.contentChecksumFlag, p.frameInfo.blockChecksumFlag = cck, bck
    p.frameInfo.contentSize, p.frameInfo.dictID = csize, dictid
    p.autoFlush = autoflush
    return p


PREF_SETS = {
    "default": dict(),                                  # Conduit.hsc:248-263: 64 KiB linked, no checksums
    "cli": dict(bsid=7, indep=1, cck=1),                # `lz4` CLI default (test/Main.hs:35)
    "cli_bck": dict(bsid=7, indep=1, cck=1, bck=1),
    "indep64k": dict(bsid=4, indep=1),
    "indep64k_bck": dict(bsid=4, indep=1, bck=1),
    "indep4m_bck": dict(bsid=7, indep=1, bck=1),        # bench configuration (cfg 3/4)
    "linked256k_cck": dict(bsid=5, indep=0, cck=1),
}


def chk(r):
    if L.LZ4F_isError(r):
        raise RuntimeError(L.LZ4F_getErrorName(r).decode())
    return r


def conduit_compress(data: bytes, prefs: Prefs, slice_: int = 16384, chunks=None) -> bytes:
    """Conduit.hsc:457-533 driven on liblz4: one out buffer of compressBound(19+16384); update per
    <=16 KiB slice; flush the buffer whenever remaining < bound; footer at EOF."""
    ctx = c_void_p()
    chk(L.LZ4F_createCompressionContext(ctypes.byref(ctx), 100))
    try:
        bound = chk(L.LZ4F_compressBound(19 + 16384, ctypes.byref(prefs)))
        bound_slice = max(bound, chk(L.LZ4F_compressBound(slice_, ctypes.byref(prefs))))
        size = bound_slice
        buf = ctypes.create_string_buffer(size)
        out = []
        used = chk(L.LZ4F_compressBegin(ctx, buf, size, ctypes.byref(prefs)))
        for chunk in (chunks if chunks is not None else [data]):
            for off in range(0, max(len(chunk), 1), slice_):
                piece = chunk[off:off + slice_]
                if size - used < bound_slice:
                    out.append(buf.raw[:used]); used = 0
                w = chk(L.LZ4F_compressUpdate(ctx, ctypes.byref(buf, used), size - used, piece, len(piece), None))
                used += w
        foot = chk(L.LZ4F_compressBound(0, ctypes.byref(prefs)))
        if size - used < foot:
            out.append(buf.raw[:used]); used = 0
        used += chk(L.LZ4F_compressEnd(ctx, ctypes.byref(buf, used), size - used, None))
        out.append(buf.raw[:used])
        return b"".join(out)
    finally:
        L.LZ4F_freeCompressionContext(ctx)


def lz4f_decompress_all(frame: bytes, dst_cap: int = 1 << 22):
    """Feed the whole frame; returns (output bytes | None, error name | None, consumed)."""
    ctx = c_void_p()
    chk(L.LZ4F_createDecompressionContext(ctypes.byref(ctx), 100))
    try:
        out = []
        pos = 0
        dst = ctypes.create_string_buffer(dst_cap)
        src = ctypes.create_string_buffer(frame, len(frame))
        while True:
            ds, ss = c_size_t(dst_cap), c_size_t(len(frame) - pos)
            r = L.LZ4F_decompress(ctx, dst, ctypes.byref(ds), ctypes.byref(src, pos), ctypes.byref(ss), None)
            if L.LZ4F_isError(r):
                return None, L.LZ4F_getErrorName(r).decode(), pos
            out.append(dst.raw[:ds.value]); pos += ss.value
            if r == 0:
                return b"".join(out), None, pos
            if ss.value == 0 and ds.value == 0:
                return b"".join(out), "TRUNCATED(hint=%d)" % r, pos
    finally:
        L.LZ4F_freeDecompressionContext(ctx)


def conduit_decompress_trace(frame: bytes, chunk: int):
    """Conduit.hsc:598-701 on liblz4: 5+2/10 header sniff, getFrameInfo, then per chunk
    LZ4F_decompress with dst = max(hint, 16 KiB).  Returns the per-call trace."""
    ctx = c_void_p()
    chk(L.LZ4F_createDecompressionContext(ctypes.byref(ctx), 100))
    trace = []
    try:
        flg = frame[4]
        hlen = 5 + (10 if flg & 8 else 2)
        hdr = frame[:hlen]
        fi = FrameInfo(); n = c_size_t(len(hdr))
        hint = chk(L.LZ4F_getFrameInfo(ctx, ctypes.byref(fi), hdr, ctypes.byref(n)))
        trace.append(["getFrameInfo", n.value, 0, hint])
        pos = hlen
        out = []
        while hint != 0 and pos < len(frame):
            bs = frame[pos:pos + chunk]; pos += len(bs)
            off = 0
            while True:
                cap = max(hint, 16384)
                dst = ctypes.create_string_buffer(cap)
                ds, ss = c_size_t(cap), c_size_t(len(bs) - off)
                piece = bs[off:]
                hint = chk(L.LZ4F_decompress(ctx, dst, ctypes.byref(ds), piece, ctypes.byref(ss), None))
                trace.append(["decompress", ss.value, ds.value, hint])
                out.append(dst.raw[:ds.value]); off += ss.value
                if off >= len(bs):
                    break
        return trace, b"".join(out)
    finally:
        L.LZ4F_freeDecompressionContext(ctx)


def walk_blocks(frame: bytes):
    flg = frame[4]
    pos = 7 + (8 if flg & 8 else 0) + (4 if flg & 1 else 0)
    bck = (flg >> 4) & 1
    sizes = []
    while True:
        h = int.from_bytes(frame[pos:pos + 4], "little"); pos += 4
        if h == 0:
            break
        sizes.append(h)
        pos += (h & 0x7FFFFFFF) + 4 * bck
    return sizes


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    G = {"liblz4_version": L.LZ4_versionNumber(), "numpy": np.__version__, "frames": {}, "headers": {}, "bounds": {},
         "malformed": [], "traces": {}, "blocks": {}}

    # --- G3 headers + compressBound values (rows a6/a7)
    for name, kw in {**PREF_SETS, "cli_bck_csize": dict(bsid=7, indep=1, cck=1, bck=1, csize=12345),
                     "cli_bck_csize_dict": dict(bsid=7, indep=1, cck=1, bck=1, csize=12345, dictid=0xDEADBEEF)}.items():
        p = mkprefs(**kw)
        ctx = c_void_p(); chk(L.LZ4F_createCompressionContext(ctypes.byref(ctx), 100))
        buf = ctypes.create_string_buffer(64)
        n = chk(L.LZ4F_compressBegin(ctx, buf, 64, ctypes.byref(p)))
        L.LZ4F_freeCompressionContext(ctx)
        G["headers"][name] = {"prefs": kw, "hex": buf.raw[:n].hex()}
        G["bounds"][name] = {str(s): chk(L.LZ4F_compressBound(s, ctypes.byref(p)))
                             for s in (0, 1, 16384, 19 + 16384, 65535, 65536, 65537, 4194304, 10 << 20)}
    G["bounds"]["NULL"] = {str(s): L.LZ4F_compressBound(s, None) for s in (0, 1, 65536, 4194304)}

    # --- G1,G2,G4,G5,G6 named inputs x preference sets, through the compress-conduit call pattern
    inputs = {"hello20": datagen.hello20(), "empty": b"", "rep42": datagen.rep42(), "ints": datagen.ints_100000(),
              "hello100k": datagen.hello_100000(), "tiny12": b"abcdefghijkl", "tiny13": b"abcdabcdabcda"}
    keep_bytes_limit = 4096
    keep_files = {("ints", "default"), ("rep42", "default")}
    for iname, data in inputs.items():
        for pname in ("default", "cli", "cli_bck", "indep64k_bck", "linked256k_cck"):
            frame = conduit_compress(data, mkprefs(**PREF_SETS[pname]))
            back, e, used = lz4f_decompress_all(frame)
            assert e is None and back == data and used == len(frame), (iname, pname, e)
            ent = {"input": iname, "prefs": PREF_SETS[pname], "input_len": len(data), "input_sha256": sha(data),
                   "frame_len": len(frame), "frame_sha256": sha(frame), "block_words": walk_blocks(frame)}
            if len(frame) <= keep_bytes_limit:
                ent["hex"] = frame.hex()
            if (iname, pname) in keep_files:
                fn = "%s_%s.lz4" % (iname, pname)
                open(os.path.join(OUT, fn), "wb").write(frame); ent["file"] = fn
            G["frames"]["%s/%s" % (iname, pname)] = ent

    # --- G10 slicing invariance (independent) + many-small-chunks input like test/Main.hs `prepare`
    data = datagen.ints_100000()
    f_a = conduit_compress(data, mkprefs(**PREF_SETS["indep64k_bck"]), slice_=16384)
    f_b = conduit_compress(data, mkprefs(**PREF_SETS["indep64k_bck"]), slice_=16384,
                           chunks=[data[i:i + 4093] for i in range(0, len(data), 4093)])
    f_c = conduit_compress(data, mkprefs(**PREF_SETS["indep64k_bck"]), slice_=1 << 20)
    assert f_a == f_b == f_c
    f_d = conduit_compress(data, mkprefs(), slice_=16384, chunks=[data[i:i + 4093] for i in range(0, len(data), 4093)])
    assert f_d == conduit_compress(data, mkprefs()), "linked mode: sub-block slices are slicing-invariant"
    G["slicing"] = {"indep_invariant": True, "linked_subblock_invariant": True,
                    "linked_65536_slices_frame_len": len(conduit_compress(data, mkprefs(), slice_=65536))}

    # --- G7 cfg 1: 10 MiB random through the default conduit
    rnd = datagen.random_bytes(10 << 20, 7).tobytes()
    frame = conduit_compress(rnd, mkprefs())
    assert len(frame) == 10486411
    G["frames"]["random10m/default"] = {"input": "random10m", "prefs": {}, "input_len": len(rnd), "input_sha256": sha(rnd),
                                        "frame_len": len(frame), "frame_sha256": sha(frame),
                                        "n_blocks": len(walk_blocks(frame)), "all_raw": all(w >> 31 for w in walk_blocks(frame))}

    # --- G8 block-level known answers: LZ4_compress_default sizes + hashes on synthetic blocks
    def block_kat(name, data, bs):
        ents = []
        for off in range(0, len(data), bs):
            blk = data[off:off + bs]
            cap = len(blk) + len(blk) // 255 + 16
            out = ctypes.create_string_buffer(cap)
            c = L.LZ4_compress_default(blk, out, len(blk), cap)
            ents.append([c, sha(out.raw[:c])[:16]])
        G["blocks"][name] = {"block_size": bs, "input_len": len(data), "input_sha256": sha(data), "csize_sha": ents}
    s50 = datagen.synth50(8 << 20, 1234).tobytes()
    block_kat("synth50_4m", s50, 4 << 20)
    block_kat("synth50_64k", s50[:2 << 20], 64 << 10)
    txt = datagen.synth_text(2 << 20, 99).tobytes()
    block_kat("text_64k", txt, 64 << 10)
    block_kat("text_4m", datagen.synth_text(4 << 20, 99).tobytes(), 4 << 20)
    # a linked 64 KiB frame of synth50 (cfg 5) and of text
    for nm, dat in (("synth50_2m", s50[:2 << 20]), ("text_2m", txt)):
        fr = conduit_compress(dat, mkprefs())
        G["frames"][nm + "/default"] = {"input": nm, "prefs": {}, "input_len": len(dat), "input_sha256": sha(dat),
                                        "frame_len": len(fr), "frame_sha256": sha(fr), "block_words": walk_blocks(fr)}
    # one real liblz4 text frame kept as a file (independent 64 KiB + block checksums), 512 KiB of text
    fr = conduit_compress(txt[:512 << 10], mkprefs(**PREF_SETS["indep64k_bck"]))
    open(os.path.join(OUT, "text512k_indep64k_bck.lz4"), "wb").write(fr)
    G["frames"]["text512k/indep64k_bck"] = {"input": "text512k", "prefs": PREF_SETS["indep64k_bck"], "input_len": 512 << 10,
                                            "input_sha256": sha(txt[:512 << 10]), "frame_len": len(fr), "frame_sha256": sha(fr),
                                            "block_words": walk_blocks(fr), "file": "text512k_indep64k_bck.lz4"}

    # --- G9 malformed frames: every single-byte mutation of two small frames -> liblz4 verdict
    base_inputs = {"m_bck_cck": (b"The quick brown fox jumps over the lazy dog. " * 3, dict(bsid=4, indep=1, cck=1, bck=1)),
                   "m_default": (b"The quick brown fox jumps over the lazy dog. " * 3, dict())}
    for nm, (dat, kw) in base_inputs.items():
        fr = conduit_compress(dat, mkprefs(**kw))
        G["frames"][nm] = {"input_hex": dat.hex(), "prefs": kw, "hex": fr.hex(), "frame_len": len(fr)}
        for pos in range(len(fr)):
            for x in (0x01, 0x80, 0xFF):
                mut = datagen.mutate(fr, pos, x)
                out, e, used = lz4f_decompress_all(mut, 1 << 16)
                G["malformed"].append({"base": nm, "pos": pos, "xor": x, "error": e,
                                       "out_sha256": sha(out)[:16] if out is not None else None,
                                       "consumed": used})
    # truncations and trailing garbage, skippable frame, dictID header
    fr = bytes.fromhex(G["frames"]["m_bck_cck"]["hex"])
    G["special"] = {}
    for cut in (0, 3, 5, 6, 7, 10, 11, len(fr) - 9, len(fr) - 5, len(fr) - 1):
        out, e, used = lz4f_decompress_all(fr[:cut], 1 << 16)
        G["special"]["truncate_%d" % cut] = {"error": e, "consumed": used, "out_len": None if out is None else len(out)}
    out, e, used = lz4f_decompress_all(fr + b"GARBAGE", 1 << 16)
    G["special"]["trailing"] = {"error": e, "consumed": used, "out_len": len(out)}
    skip = bytes.fromhex("5a2a4d18") + (5).to_bytes(4, "little") + b"12345"
    out, e, used = lz4f_decompress_all(skip + fr, 1 << 16)
    G["special"]["skippable_then_frame"] = {"error": e, "consumed": used, "out_len": len(out)}

    # --- decompress-conduit traces (H2): (srcConsumed, dstProduced, hint) per call
    for key, chunk in (("hello20/default", 1 << 20), ("rep42/default", 100), ("hello100k/cli", 1000),
                       ("ints/default", 32768), ("ints/cli", 65536), ("m_bck_cck", 7)):
        ent = G["frames"][key]
        fr = bytes.fromhex(ent["hex"]) if "hex" in ent else conduit_compress(inputs[ent["input"]], mkprefs(**ent["prefs"]))
        tr, out = conduit_decompress_trace(fr, chunk)
        G["traces"]["%s@%d" % (key, chunk)] = {"chunk": chunk, "calls": tr, "out_sha256": sha(out)}

    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(OUT, "golden.json"), os.path.getsize(os.path.join(OUT, "golden.json")), "bytes;",
          len(G["frames"]), "frames,", len(G["malformed"]), "malformed cases")


if __name__ == "__main__":
    main()
