"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/lz4f_mi355x.h declares,
and its host-side logic (bounds, header bytes, frame-info parsing, error names, conduit plumbing)
matches the golden vectors.  No compute call needs a GPU here; the one that would must fail loudly."""
import ctypes
import os
import re

import pytest

from lz4_frame_conduit_amd import _ffi, conduit
from lz4_frame_conduit_amd._ffi import FrameInfo, Preferences

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    _ffi.build()
    return _ffi.lib()


def prefs_of(kw):
    return conduit.make_preferences(blockSizeID=kw.get("bsid", 0), blockMode=kw.get("indep", 0), contentChecksum=kw.get("cck", 0),
                                    blockChecksum=kw.get("bck", 0), contentSize=kw.get("csize", 0), dictID=kw.get("dictid", 0))


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "lz4f_mi355x.h")).read()
    declared = set(re.findall(r"^LZ4F_MI355X_API[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 45
    raw = ctypes.CDLL(_ffi.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, missing
    assert declared == set(_ffi.DECLARED_SYMBOLS), declared ^ set(_ffi.DECLARED_SYMBOLS)


def test_struct_layouts():
    # CTypes.hsc Storable offsets
    assert ctypes.sizeof(FrameInfo) == 32 and ctypes.sizeof(Preferences) == 56
    assert [getattr(FrameInfo, f).offset for f, _ in FrameInfo._fields_] == [0, 4, 8, 12, 16, 24, 28]
    assert [getattr(Preferences, f).offset for f, _ in Preferences._fields_] == [0, 32, 36, 40, 44]


def test_error_names_and_codes(L):
    names = ["OK_NoError", "ERROR_GENERIC", "ERROR_maxBlockSize_invalid", "ERROR_blockMode_invalid", "ERROR_contentChecksumFlag_invalid",
             "ERROR_compressionLevel_invalid", "ERROR_headerVersion_wrong", "ERROR_blockChecksum_invalid", "ERROR_reservedFlag_set",
             "ERROR_allocation_failed", "ERROR_srcSize_tooLarge", "ERROR_dstMaxSize_tooSmall", "ERROR_frameHeader_incomplete",
             "ERROR_frameType_unknown", "ERROR_frameSize_wrong", "ERROR_srcPtr_wrong", "ERROR_decompressionFailed",
             "ERROR_headerChecksum_invalid", "ERROR_contentChecksum_invalid", "ERROR_frameDecoding_alreadyStarted"]
    for code, name in enumerate(names):
        v = (1 << 64) - code if code else 0
        if code:
            assert L.LZ4F_isError(v) and L.LZ4F_getErrorName(v).decode() == name
    assert not L.LZ4F_isError(0) and not L.LZ4F_isError(65544)
    assert L.LZ4F_getVersion() == 100


def test_compress_bound_matches_liblz4(L, golden):
    for name, table in golden["bounds"].items():
        p = None if name == "NULL" else ctypes.byref(prefs_of(golden["headers"][name]["prefs"]))
        for s, v in table.items():
            assert L.LZ4F_compressBound(int(s), p) == v, (name, s)
            assert L.lz4f_mi355x_compressBound(int(s), p) == v


def test_compress_begin_header_bytes(L, golden):
    for name, ent in golden["headers"].items():
        ctx = ctypes.c_void_p()
        assert L.LZ4F_createCompressionContext(ctypes.byref(ctx), 100) == 0
        buf = ctypes.create_string_buffer(64)
        n = L.LZ4F_compressBegin(ctx, buf, 64, ctypes.byref(prefs_of(ent["prefs"])))
        assert not L.LZ4F_isError(n) and buf.raw[:n].hex() == ent["hex"], name
        small = L.LZ4F_compressBegin(ctx, buf, 18, None)
        assert L.LZ4F_getErrorName(small) == b"ERROR_dstMaxSize_tooSmall"
        assert L.LZ4F_freeCompressionContext(ctx) == 0
    assert L.LZ4F_freeCompressionContext(None) == 0


def test_get_frame_info(L, golden):
    for name, ent in golden["headers"].items():
        hdr = bytes.fromhex(ent["hex"])
        d = ctypes.c_void_p()
        assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
        fi = FrameInfo(); n = ctypes.c_size_t(len(hdr))
        hint = L.LZ4F_getFrameInfo(d, ctypes.byref(fi), hdr, ctypes.byref(n))
        assert hint == 4 and n.value == len(hdr), name
        kw = ent["prefs"]
        assert fi.blockSizeID == (kw.get("bsid", 0) or 4) and fi.blockMode == kw.get("indep", 0)
        assert fi.contentChecksumFlag == kw.get("cck", 0) and fi.blockChecksumFlag == kw.get("bck", 0)
        assert fi.contentSize == kw.get("csize", 0) and fi.dictID == kw.get("dictid", 0)
        # a second call reports the stored info and consumes nothing
        n2 = ctypes.c_size_t(len(hdr))
        assert L.LZ4F_getFrameInfo(d, ctypes.byref(fi), hdr, ctypes.byref(n2)) == 4 and n2.value == 0
        assert L.LZ4F_freeDecompressionContext(d) == 0
    # the reference's 7/15-byte sniff on a dictID header: incomplete
    hdr = bytes.fromhex(golden["headers"]["cli_bck_csize_dict"]["hex"])
    for cut in (7, 15):
        d = ctypes.c_void_p(); L.LZ4F_createDecompressionContext(ctypes.byref(d), 100)
        fi = FrameInfo(); n = ctypes.c_size_t(cut)
        r = L.LZ4F_getFrameInfo(d, ctypes.byref(fi), hdr[:cut], ctypes.byref(n))
        assert L.LZ4F_getErrorName(r) == b"ERROR_frameHeader_incomplete"
        L.LZ4F_freeDecompressionContext(d)


def test_header_errors_without_gpu(L, golden):
    """Header-level verdicts of the malformed set need no block decode: same names as liblz4."""
    n = 0
    for m in golden["malformed"]:
        if m["error"] not in ("ERROR_frameType_unknown", "ERROR_headerChecksum_invalid", "ERROR_reservedFlag_set", "ERROR_headerVersion_wrong"):
            continue
        if m["pos"] >= 7:
            continue
        base = bytearray(bytes.fromhex(golden["frames"][m["base"]]["hex"])); base[m["pos"]] ^= m["xor"]
        d = ctypes.c_void_p(); L.LZ4F_createDecompressionContext(ctypes.byref(d), 100)
        dst = ctypes.create_string_buffer(1 << 16); ds = ctypes.c_size_t(1 << 16); ss = ctypes.c_size_t(len(base))
        r = L.LZ4F_decompress(d, dst, ctypes.byref(ds), bytes(base), ctypes.byref(ss), None)
        assert L.LZ4F_getErrorName(r).decode() == m["error"], m
        L.LZ4F_freeDecompressionContext(d)
        n += 1
    assert n >= 20


def test_bsChunksOf():
    # test/Main.hs:56-58
    assert conduit.bsChunksOf(3, b"abc123def4567") == [b"abc", b"123", b"def", b"456", b"7"]
    with pytest.raises(ValueError):
        conduit.bsChunksOf(0, b"x")


def test_decompress_conduit_protocol_errors(L):
    # Conduit.hsc:615-616
    with pytest.raises(conduit.Lz4FrameError, match="not enough bytes for header; expected 5, got 3"):
        conduit.decompress([b"\x04\x22", b"\x4d"])
    with pytest.raises(conduit.Lz4FrameError, match="lz4frame error: ERROR_frameType_unknown"):
        conduit.decompress([b"hello world, not a frame"])
    # header only, then EOF: Conduit.hsc:689
    with pytest.raises(conduit.Lz4FrameError, match="stream ended before EndMark"):
        conduit.decompress([bytes.fromhex("04224d184040c0")])


def test_empty_frame_needs_no_gpu(L, golden):
    # no block is ever formed for empty input: header + EndMark come from the host layer alone
    assert b"".join(conduit.compress([])).hex() == golden["frames"]["empty/default"]["hex"]
    p = conduit.make_preferences(blockSizeID=7, blockMode=1, contentChecksum=1)
    assert b"".join(conduit.compressWithPreferences(p, [b""])).hex() == golden["frames"]["empty/cli"]["hex"]
    assert b"".join(conduit.decompress([bytes.fromhex(golden["frames"]["empty/cli"]["hex"])])) == b""


def test_no_gpu_fails_loudly(L):
    """Without a device every block-level call must fail with an error code (never fall back to a CPU codec)."""
    if L.lz4f_mi355x_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(conduit.Lz4FrameError, match="lz4frame error: ERROR_GENERIC"):
        conduit.compress([b"x" * 70000])
    assert b"no usable HIP device" in L.lz4f_mi355x_last_error()
    h = ctypes.c_void_p()
    r = L.lz4f_mi355x_engine_create(ctypes.byref(h), 0, None, 0)
    assert L.LZ4F_isError(r)


def test_cli_builds_and_fails_loudly_without_gpu(L):
    """mi355x-lz4c (app/Main.hs equivalent) is an ordinary client of the C ABI: it must exist next to the
    library, print usage, and - without a device - exit non-zero with an error instead of emitting anything."""
    import os
    import subprocess
    cli = os.path.join(os.path.dirname(_ffi.LIB_PATH), "mi355x-lz4c")
    assert os.path.exists(cli)
    assert b"Usage: mi355x-lz4c" in subprocess.run([cli, "--help"], stdout=subprocess.PIPE, check=True, timeout=60).stdout
    assert subprocess.run([cli, "--bogus"], stderr=subprocess.PIPE, timeout=60).returncode == 2
    if L.lz4f_mi355x_device_count() > 0:
        pytest.skip("a GPU is present")
    p = subprocess.run([cli], input=b"x" * 100000, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 1 and b"ERROR_GENERIC" in p.stderr


def test_block_list_of_a_finished_frame_is_host_work(L):
    """lz4f_mi355x_appendBlockList (include/lz4f_mi355x.h) walks a frame's size words on the host and appends the skippable
    frame the device decoder looks for: no GPU involved, any encoder's frame.  Checked here: the layout (magic, size field,
    16-byte aligned list of size-word positions, footer in the last 32 bytes), that the frame in front is untouched and that
    an LZ4 reader (the oracle, and liblz4 when installed) still reads it and sees one skippable frame behind it; the GPU
    side of it (the device decoder using the list) is tests/test_gpu_parity.py::test_block_list_trailer_from_the_host_paths."""
    import struct
    import oracle
    from lz4_frame_conduit_amd import datagen
    data = datagen.structured(300000, 5)
    for kw in (dict(bsid=4, indep=1), dict(bsid=4, indep=0, bck=1, cck=1), dict(bsid=5, indep=1, cck=1)):
        frame = oracle.conduit_compress(data, oracle.mkprefs(**kw))
        listed = conduit.appendBlockList(frame)
        assert listed[:len(frame)] == frame and len(listed) > len(frame)
        F = len(frame)
        assert listed[F:F + 4] == bytes.fromhex("5e2a4d18") and struct.unpack_from("<I", listed, F + 4)[0] == len(listed) - F - 8
        seqs, ents, p0, p1, magic, n_blocks, total = struct.unpack_from("<6IQ", listed, len(listed) - 32)
        assert (seqs, ents, p0, p1) == (0, 0, 0, 0) and magic == 0x58495A4C and total == len(listed) - F
        list_at = (F + 8 + 15) & ~15
        at = struct.unpack_from("<%dQ" % n_blocks, listed, list_at)
        crc = 4 if kw.get("bck") else 0
        pos = 7
        for a in at:                                       # every entry is a size word, each where the one before ends
            assert a == pos
            pos += 4 + (struct.unpack_from("<I", frame, a)[0] & 0x7FFFFFFF) + crc
        assert struct.unpack_from("<I", frame, pos)[0] == 0 and n_blocks == -(-len(data) // (65536 << (2 * (kw["bsid"] - 4))))
        out, used = oracle.decompress_frame(listed)
        assert out == data and used == F
        out2, used2 = oracle.decompress_frame(listed[F:])
        assert out2 == b"" and used2 == len(listed) - F    # a skippable frame: nothing decoded, all of it consumed
    # a frame without blocks gets nothing; a frame that does not end where the caller says is refused; so is too little room
    empty = oracle.conduit_compress(b"", oracle.mkprefs())
    assert conduit.appendBlockList(empty) == empty
    frame = oracle.conduit_compress(data, oracle.mkprefs(bsid=4, indep=1))
    for bad in (frame + b"\0", frame[:-1]):
        r = L.lz4f_mi355x_blockListSize(bad, len(bad))
        assert L.LZ4F_isError(r)
    buf = ctypes.create_string_buffer(frame, len(frame) + 16)
    r = L.lz4f_mi355x_appendBlockList(buf, len(frame), len(frame) + 16)
    assert L.LZ4F_isError(r) and L.LZ4F_getErrorName(r) == b"ERROR_dstMaxSize_tooSmall" and buf.raw[:len(frame)] == frame
