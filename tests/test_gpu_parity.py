"""Parity tests proper: the HIP path, called through the C ABI, against the oracle and the golden
vectors.  Bar: bit-exact decode; encode round-trips bit-exact and its
ratio stays within the stated tolerance of LZ4_compress_default (the oracle is bit-exact with it)."""
import ctypes
import hashlib
import os
import sys

import numpy as np
import pytest

import oracle
from conftest import golden_file
from lz4_frame_conduit_amd import _ffi, conduit, datagen

pytestmark = pytest.mark.gpu

# encoder ratio tolerance vs LZ4_compress_default: compressed size may exceed liblz4's by at most this factor
RATIO_TOL = 1.05
sha = lambda b: hashlib.sha256(b).hexdigest()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    lib = _ffi.lib()
    assert lib.lz4f_mi355x_device_count() >= 1, "these tests need the MI355X"
    return lib


def prefs_of(kw):
    return conduit.make_preferences(blockSizeID=kw.get("bsid", 0), blockMode=kw.get("indep", 0), contentChecksum=kw.get("cck", 0),
                                    blockChecksum=kw.get("bck", 0), contentSize=kw.get("csize", 0), dictID=kw.get("dictid", 0))


def gpu_decompress_frame(L, frame: bytes, cap: int):
    dst = ctypes.create_string_buffer(max(cap, 1))
    used = ctypes.c_size_t(0)
    r = L.lz4f_mi355x_decompressFrame(dst, cap, frame, len(frame), ctypes.byref(used))
    if L.LZ4F_isError(r):
        raise RuntimeError(L.LZ4F_getErrorName(r).decode() + " | " + L.lz4f_mi355x_last_error().decode())
    return dst.raw[:r], used.value


def gpu_compress_frame(L, data: bytes, prefs) -> bytes:
    cap = L.lz4f_mi355x_compressFrameBound(len(data), ctypes.byref(prefs))
    dst = ctypes.create_string_buffer(cap)
    r = L.lz4f_mi355x_compressFrame(dst, cap, data, len(data), ctypes.byref(prefs))
    if L.LZ4F_isError(r):
        raise RuntimeError(L.LZ4F_getErrorName(r).decode() + " | " + L.lz4f_mi355x_last_error().decode())
    return dst.raw[:r]


PREF_SETS = {"default": {}, "cli": dict(bsid=7, indep=1, cck=1), "cli_bck": dict(bsid=7, indep=1, cck=1, bck=1),
             "indep64k_bck": dict(bsid=4, indep=1, bck=1), "linked256k_cck": dict(bsid=5, indep=0, cck=1)}
# result.flags >> 12 (include/lz4f_mi355x.h: LZ4F_MI355X_PATH_*)
PATH = dict(table=0x001, trailer=0x002, parallel_walk=0x004, indexed=0x008, self_index=0x010, doubling=0x020, hops=0x040, window=0x080, fused=0x100,
            wave_per_block=0x200, dropped=0x400, workgroup_per_block=0x800)
SMALL_INPUTS = ["hello20", "empty", "rep42", "ints", "hello100k", "tiny12", "tiny13"]


# ------------------------------------------------------------------------------------------------
def test_xxh32_kernel_matches_oracle(L):
    import torch
    from lz4_frame_conduit_amd.device import Engine
    rng = np.random.default_rng(5)
    lens = [0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 63, 64, 1023, 1024, 1025, 4096, 65535, 65536, 100001, 1 << 20]
    blob = rng.integers(0, 256, sum(lens) + 7, dtype=np.uint8)
    offs, pos = [], 3                                   # odd base offset: unaligned reads
    for n in lens:
        offs.append(pos); pos += n
    eng = Engine(0)
    got = eng.xxh32(torch.from_numpy(blob).cuda(), np.array(offs), np.array(lens))
    exp = [oracle.xxh32(blob[o:o + n]) for o, n in zip(offs, lens)]
    assert list(got) == exp
    eng.close()


def test_decode_liblz4_frames_bit_exact(L, golden, named_inputs):
    """Real liblz4 1.9.3 frames (committed files) and oracle frames pinned to liblz4 by hash."""
    for key in ("ints/default", "rep42/default", "text512k/indep64k_bck"):
        ent = golden["frames"][key]
        out, used = gpu_decompress_frame(L, golden_file(ent["file"]), ent["input_len"] + 8)
        assert used == ent["frame_len"], key
        assert out == named_inputs[ent["input"]], key
    for iname in SMALL_INPUTS:
        for pname, kw in PREF_SETS.items():
            ent = golden["frames"]["%s/%s" % (iname, pname)]
            frame = oracle.conduit_compress(named_inputs[iname], oracle.mkprefs(**kw))
            assert sha(frame) == ent["frame_sha256"]
            out, used = gpu_decompress_frame(L, frame, ent["input_len"] + 8)
            assert (out, used) == (named_inputs[iname], len(frame)), (iname, pname)


def test_decode_synthetic_blocks(L, golden, named_inputs):
    """cfg 2 / cfg 3 / cfg 5 shapes at oracle-friendly sizes: text at 64 KiB independent blocks,
    synth50 at 4 MiB independent blocks and at 64 KiB linked blocks."""
    cases = [("text_2m", dict(bsid=4, indep=1)), ("synth50_8m", dict(bsid=7, indep=1, bck=1)), ("synth50_2m", {}), ("text_2m", {}),
             ("synth50_8m", dict(bsid=6, indep=0))]
    for iname, kw in cases:
        data = named_inputs[iname]
        frame = oracle.conduit_compress(data, oracle.mkprefs(**kw))
        out, used = gpu_decompress_frame(L, frame, len(data))
        assert used == len(frame) and sha(out) == sha(data), (iname, kw)


def test_random_10mib_conduit_roundtrip(L, golden, named_inputs):
    """BASELINE configs[0]: 10 MiB random through compress .| decompress; every block stored raw."""
    data = named_inputs["random10m"]
    chunks = [data[i:i + 1000003] for i in range(0, len(data), 1000003)]
    frame = b"".join(conduit.compress(chunks))
    assert len(frame) == 10486411 and sha(frame) == golden["frames"]["random10m/default"]["frame_sha256"]
    back = b"".join(conduit.decompress([frame[i:i + 3000017] for i in range(0, len(frame), 3000017)]))
    assert back == data


def test_compress_roundtrip_and_ratio(L, golden, named_inputs):
    for iname in SMALL_INPUTS + ["text512k", "synth50_2m"]:
        data = named_inputs[iname]
        for pname, kw in PREF_SETS.items():
            frame = gpu_compress_frame(L, data, prefs_of(kw))
            out, used = oracle.decompress_frame(frame, cap=len(data) + 64)          # the oracle decodes what the GPU wrote
            assert out == data and used == len(frame), (iname, pname)
            out2, used2 = gpu_decompress_frame(L, frame, len(data) + 8)
            assert out2 == data and used2 == len(frame), (iname, pname)
            ref = len(oracle.conduit_compress(data, oracle.mkprefs(**kw)))
            if len(data) >= 65536:
                assert len(frame) <= ref * RATIO_TOL + 64, (iname, pname, len(frame), ref)
            assert frame[:4] == bytes.fromhex("04224d18")


def test_compress_4m_blocks_ratio_vs_liblz4(L, golden, named_inputs):
    """cfg 3 shape: synth50, 4 MiB independent blocks (+ block checksums): per-block payload sizes against
    LZ4_compress_default's (golden `blocks/synth50_4m`)."""
    data = named_inputs["synth50_8m"]
    frame = gpu_compress_frame(L, data, prefs_of(dict(bsid=7, indep=1, bck=1)))
    out, used = oracle.decompress_frame(frame, cap=len(data) + 64)
    assert out == data and used == len(frame)
    pos, sizes = 7, []
    while True:
        w = int.from_bytes(frame[pos:pos + 4], "little"); pos += 4
        if w == 0:
            break
        sizes.append(w & 0x7FFFFFFF); pos += (w & 0x7FFFFFFF) + 4
    ref = [c for c, _ in golden["blocks"]["synth50_4m"]["csize_sha"]]
    assert len(sizes) == len(ref) == 2
    for got, want in zip(sizes, ref):
        assert got <= want * RATIO_TOL, (got, want)
    nseq = oracle.count_sequences(frame[11:11 + sizes[0]])
    assert nseq > 1000


def test_streaming_api_matches_liblz4_traces(L, golden, named_inputs):
    """H2: the decompress conduit's exact call pattern; per call (srcConsumed, dstProduced, hint) == liblz4's."""
    for key, tr in golden["traces"].items():
        fkey, chunk = key.rsplit("@", 1)
        chunk = int(chunk)
        ent = golden["frames"][fkey]
        frame = bytes.fromhex(ent["hex"]) if "hex" in ent else oracle.conduit_compress(named_inputs[ent["input"]], oracle.mkprefs(**ent["prefs"]))
        d = ctypes.c_void_p(); assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
        hlen = 5 + (10 if frame[4] & 8 else 2)
        fi = _ffi.FrameInfo(); n = ctypes.c_size_t(hlen)
        hint = L.LZ4F_getFrameInfo(d, ctypes.byref(fi), frame[:hlen], ctypes.byref(n))
        calls = [["getFrameInfo", n.value, 0, hint]]
        pos, out = hlen, []
        while hint != 0 and pos < len(frame):
            bs = frame[pos:pos + chunk]; pos += len(bs); off = 0
            while True:
                cap = max(hint, 16384)
                dst = ctypes.create_string_buffer(cap); ds = ctypes.c_size_t(cap); ss = ctypes.c_size_t(len(bs) - off)
                piece = bs[off:]
                hint = L.LZ4F_decompress(d, dst, ctypes.byref(ds), piece, ctypes.byref(ss), None)
                assert not L.LZ4F_isError(hint), (key, L.LZ4F_getErrorName(hint), L.lz4f_mi355x_last_error())
                calls.append(["decompress", ss.value, ds.value, hint]); out.append(dst.raw[:ds.value]); off += ss.value
                if off >= len(bs):
                    break
        L.LZ4F_freeDecompressionContext(d)
        assert sha(b"".join(out)) == tr["out_sha256"], key
        assert calls == tr["calls"], key


def test_conduits_on_reference_test_inputs(L, golden, named_inputs):
    """test/Main.hs:60-119 restated: compress -> (oracle decodes, standing in for `lz4 -d`);
    liblz4 frames -> decompress; compress .| decompress identity on many-small-chunk inputs."""
    for iname in ["hello20", "ints", "hello100k", "rep42"]:
        data = named_inputs[iname]
        chunks = [data[i:i + 4093] for i in range(0, len(data), 4093)]      # many small ByteStrings, like `prepare`
        frame = b"".join(conduit.compress(chunks))
        assert frame[:7] == bytes.fromhex("04224d184040c0")
        out, used = oracle.decompress_frame(frame, cap=len(data) + 64)
        assert out == data
        assert b"".join(conduit.decompress([frame[i:i + 5000] for i in range(0, len(frame), 5000)])) == data
        assert b"".join(conduit.compressYieldImmediately(chunks))[:7] == frame[:7]
        f2 = b"".join(conduit.compressYieldImmediately([data]))
        assert oracle.decompress_frame(f2, cap=len(data) + 64)[0] == data
        # CLI-like frames from the oracle (== liblz4 bytes) through our decompress
        cli = oracle.conduit_compress(data, oracle.mkprefs(bsid=7, indep=1, cck=1))
        assert b"".join(conduit.decompress([cli])) == data
        assert b"".join(conduit.decompressBatched([cli[:100], cli[100:]])) == data
        fb = b"".join(conduit.compressBatched(chunks, conduit.make_preferences(blockSizeID=5, blockMode=1, blockChecksum=1), batchBytes=1 << 20))
        assert oracle.decompress_frame(fb, cap=len(data) + 64)[0] == data
    rng = np.random.default_rng(2024)                                        # the QuickCheck property
    for trial in range(40):
        n = int(rng.integers(0, 10000)); alpha = int(rng.choice([2, 16, 256]))
        s = rng.integers(0, alpha, n, dtype=np.uint8).tobytes()
        pieces = [s[i:i + 1000] for i in range(0, len(s), 1000)]
        assert b"".join(conduit.decompress(conduit.compress(pieces))) == s


def test_malformed_frames_same_verdict_as_liblz4(L, golden):
    agree = same = stricter = 0
    for m in golden["malformed"]:
        if m["xor"] != 0xFF and m["pos"] % 3:            # subsample: each call is a GPU round trip
            continue
        base = bytearray(bytes.fromhex(golden["frames"][m["base"]]["hex"])); base[m["pos"]] ^= m["xor"]
        d = ctypes.c_void_p(); L.LZ4F_createDecompressionContext(ctypes.byref(d), 100)
        pos, out, verdict = 0, [], None
        while True:
            dst = ctypes.create_string_buffer(1 << 16); ds = ctypes.c_size_t(1 << 16); ss = ctypes.c_size_t(len(base) - pos)
            r = L.LZ4F_decompress(d, dst, ctypes.byref(ds), bytes(base[pos:]), ctypes.byref(ss), None)
            if L.LZ4F_isError(r):
                verdict = L.LZ4F_getErrorName(r).decode(); break
            out.append(dst.raw[:ds.value]); pos += ss.value
            if r == 0:
                break
            if ss.value == 0 and ds.value == 0:
                verdict = "TRUNCATED(hint=%d)" % r; break
        L.LZ4F_freeDecompressionContext(d)
        exp = m["error"]
        if exp is None and verdict is not None:
            assert verdict in ("ERROR_GENERIC", "ERROR_decompressionFailed"), (m, verdict); stricter += 1
        elif exp is None:
            assert sha(b"".join(out))[:16] == m["out_sha256"] and pos == m["consumed"], m; same += 1
        else:
            assert verdict == exp, (m, verdict); agree += 1
    assert agree > 80 and same > 40 and stricter <= 12


def test_device_resident_path_and_block_table(L):
    """K variant: frame and output never leave HBM; the table from compress drives decompress; the
    walk kernel rebuilds the same table from the frame bytes."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    eng = Engine(0)
    shapes = [(datagen.synth50(16 << 20, 77), kw) for kw in (dict(bsid=7, indep=1, bck=1), dict(bsid=4, indep=1), dict(bsid=4, indep=0))]
    shapes += [(np.frombuffer(datagen.structured(9 << 20, 2000 + i), dtype=np.uint8).copy(), kw)
               for i, kw in enumerate((dict(bsid=7, indep=1), dict(bsid=7, indep=0, bck=1), dict(bsid=5, indep=0), dict(bsid=6, indep=1)))]
    for data, kw in shapes:
        src = torch.from_numpy(data).cuda()
        p = prefs_of(kw)
        cap = eng.frame_bound(src.numel(), p)
        frame = torch.empty(cap, dtype=torch.uint8, device="cuda")
        nb = (src.numel() + (1 << (8 + 2 * (kw["bsid"]))) - 1) >> (8 + 2 * kw["bsid"])
        table = eng.new_table(nb)
        eng.compress_async(src, frame, p, table)
        r = eng.result()
        assert r.n_blocks == nb and 0 < r.size <= cap
        host_frame = frame[:r.size].cpu().numpy().tobytes()
        out, used = oracle.decompress_frame(host_frame, cap=src.numel() + 64)
        assert used == r.size and out == data.tobytes(), kw
        back = torch.zeros_like(src)
        eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo)
        r2 = eng.result()
        assert r2.size == src.numel() and torch.equal(back, src), kw
        back.zero_()
        eng.decompress_frame_async(frame, r.size, back)                      # foreign-frame path: walk kernel
        r3 = eng.result()
        assert r3.size == src.numel() and r3.consumed == r.size and r3.n_blocks == nb and torch.equal(back, src), kw
    eng.close()


def test_full_size_properties_1gib(L):
    """Size-independent properties at a BASELINE-scale size (1 GiB per call, 4 MiB independent blocks):
    encode -> decode identity checked on the device, and the checksum of block checksums agrees with
    the same reduction over the decoded bytes."""
    import torch
    from lz4_frame_conduit_amd.device import Engine, synth50_device
    n = 1 << 30
    src = synth50_device(n, 4321)
    eng = Engine(0)
    p = prefs_of(dict(bsid=7, indep=1, bck=1))
    frame = torch.empty(eng.frame_bound(n, p), dtype=torch.uint8, device="cuda")
    nb = n >> 22
    table = eng.new_table(nb)
    eng.compress_async(src, frame, p, table)
    r = eng.result()
    ratio = n / r.size
    assert 1.7 < ratio < 2.2, ratio                      # liblz4: 1.944 on this recipe
    for rep in range(4):                                  # (repeated, into uninitialised memory: the hand-offs inside the decoder are timing-sensitive)
        back = torch.empty_like(src)
        eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo)
        r2 = eng.result()
        assert r2.size == n and torch.equal(back, src), rep
    offs = np.arange(nb, dtype=np.uint64) << 22
    lens = np.full(nb, 1 << 22, dtype=np.uint32)
    a, b = eng.xxh32(src, offs, lens), eng.xxh32(back, offs, lens)
    assert oracle.xxh32(a.tobytes()) == oracle.xxh32(b.tobytes())
    assert int(a[0]) == oracle.xxh32(src[:1 << 22].cpu().numpy())
    eng.close()


# ------------------------------------------------------------------------------------------------
# the command-line driver (app/Main.hs equivalent) exchanges frames with liblz4 in both directions
def test_cli_interop_with_liblz4_frames(L, golden, named_inputs, tmp_path):
    import os
    import subprocess
    cli = os.path.join(os.path.dirname(_ffi.LIB_PATH), "mi355x-lz4c")
    assert os.path.exists(cli), "csrc/Makefile builds it next to the library"
    run = lambda args, **kw: subprocess.run([cli] + args, check=True, timeout=120, **kw)
    # (1) frames written by liblz4 (committed golden files) -> `mi355x-lz4c -d`, file arguments and stdin/stdout
    for key in ("ints/default", "rep42/default", "text512k/indep64k_bck"):
        ent = golden["frames"][key]
        f_in = tmp_path / "in.lz4"; f_out = tmp_path / "out.bin"
        f_in.write_bytes(golden_file(ent["file"]))
        run(["-d", str(f_in), str(f_out)])
        assert sha(f_out.read_bytes()) == ent["input_sha256"], key
        piped = run(["-d", "--batch", "0"], input=golden_file(ent["file"]), stdout=subprocess.PIPE).stdout      # the reference's conduit verbatim
        assert sha(piped) == ent["input_sha256"], key
    # (2) `mi355x-lz4c` output is a frame the oracle (bit-exact with liblz4's decoder) accepts, for the CLI's option shapes
    data = named_inputs["synth50_2m"] + named_inputs["text512k"]
    f_src = tmp_path / "src.bin"; f_src.write_bytes(data)
    for opts in ([], ["--batch", "0"], ["-B7", "-BI", "--content-checksum"], ["-B5", "-BD", "--block-checksum"], ["-B4", "-BI", "--batch", "1"]):
        f_lz = tmp_path / "x.lz4"
        run(opts + [str(f_src), str(f_lz)])
        frame = f_lz.read_bytes()
        assert frame[:4] == bytes.fromhex("04224d18")
        assert oracle.decompress_frame(frame, len(data) + 8)[0] == data, opts
        assert len(frame) < len(data)
    # (3) errors are reported, not swallowed: a truncated frame fails with the reference's message
    bad = golden_file(golden["frames"]["ints/default"]["file"])[:-9]
    p = subprocess.run([cli, "-d"], input=bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 1 and b"mi355x-lz4c:" in p.stderr


# ------------------------------------------------------------------------------------------------
# decoder completeness around the hot loop -- skippable frames, concatenated frames, contentSize / dictID
# headers (the reference's own `decompress` cannot read a dictID header, Appendix C quirk 2), stored blocks.
def test_batched_decoder_walks_whole_streams(L, golden, named_inputs):
    text = named_inputs["text512k"]
    rnd = named_inputs["random10m"][:300000]                              # stored (raw) blocks
    skip = bytes.fromhex("502a4d18") + (5).to_bytes(4, "little") + b"hello"
    frames = {
        "plain": (oracle.conduit_compress(text, oracle.mkprefs(4, 1, 1, 0, 0, 0, 0), 16384), text),
        "csize+dictid": (oracle.conduit_compress(text, oracle.mkprefs(5, 0, 1, 1, len(text), 77, 0), 16384), text),
        "stored+dictid": (oracle.conduit_compress(rnd, oracle.mkprefs(4, 1, 0, 0, 0, 9, 0), 16384), rnd),
        "empty": (oracle.conduit_compress(b"", oracle.mkprefs(4, 1, 1, 0, 0, 0, 0), 16384), b""),
    }
    for name, (fr, want) in frames.items():
        assert oracle.decompress_frame(fr, len(want) + 8)[0] == want
    streams = {
        "single": ["csize+dictid"], "concat": ["plain", "stored+dictid", "csize+dictid"], "skip first": ["SKIP", "plain"],
        "skip between and last": ["plain", "SKIP", "empty", "stored+dictid", "SKIP"],
    }
    for label, parts in streams.items():
        stream = b"".join(skip if p == "SKIP" else frames[p][0] for p in parts)
        want = b"".join(b"" if p == "SKIP" else frames[p][1] for p in parts)
        for chunks in ([stream], [stream[:1], stream[1:5], stream[5:9], stream[9:70000], stream[70000:]]):
            assert b"".join(conduit.decompressBatched(chunks)) == want, label
    # many short blocks under a big blockSizeID (a flushing writer: 16 KiB blocks in a 4 MiB-block frame): the batched decoder
    # hands the output over slab by slab and must not size anything as blocks x maxBlockSize (24 MiB here would ask for 6 GiB)
    big = datagen.synth_text(24 << 20, 5).tobytes()
    flushed = oracle.conduit_compress(big, oracle.mkprefs(7, 1, 1, 0, 0, 0, 1), 16384)       # autoFlush: a block per 16 KiB slice
    pieces = list(conduit.decompressBatched([flushed]))
    assert len(pieces) >= 2 and sha(b"".join(pieces)) == sha(big)
    # the bulk calls over page-locked buffers, and over pageable ones, give the same frames' content
    import ctypes
    n = len(big)
    p = prefs_of(dict(bsid=6, indep=1, cck=1))
    cap = L.lz4f_mi355x_compressFrameBound(n, ctypes.byref(p))
    hp = [L.lz4f_mi355x_host_alloc(x) for x in (n, cap, n + 8)]
    assert all(hp)
    try:
        ctypes.memmove(hp[0], big, n)
        r = L.lz4f_mi355x_compressFrame(hp[1], cap, hp[0], n, ctypes.byref(p))
        assert not L.LZ4F_isError(r), L.LZ4F_getErrorName(r)
        fr = ctypes.string_at(hp[1], r)
        assert oracle.decompress_frame(fr, n + 8) == (big, len(fr))
        used = ctypes.c_size_t(0)
        r2 = L.lz4f_mi355x_decompressFrame(hp[2], n + 8, hp[1], r, ctypes.byref(used))
        assert r2 == n and used.value == r and ctypes.string_at(hp[2], n) == big
    finally:
        for x in hp: L.lz4f_mi355x_host_free(x)
    # what must still fail, with liblz4's names
    with pytest.raises(conduit.Lz4FrameError, match="ERROR_frameType_unknown"):
        conduit.decompressBatched([frames["plain"][0] + b"\x01\x02\x03\x04\x05\x06\x07\x08"])
    with pytest.raises(conduit.Lz4FrameError, match="stream ended before EndMark"):
        conduit.decompressBatched([frames["plain"][0][:-20]])
    bad = bytearray(frames["csize+dictid"][0]); bad[6] ^= 1                # contentSize field: header checksum catches it
    with pytest.raises(conduit.Lz4FrameError, match="ERROR_headerChecksum_invalid"):
        conduit.decompressBatched([bytes(bad)])
    wrong = oracle.conduit_compress(text, oracle.mkprefs(4, 1, 0, 0, len(text), 0, 0), 16384)
    short = oracle.conduit_compress(text[:-1], oracle.mkprefs(4, 1, 0, 0, 0, 0, 0), 16384)
    forged = wrong[:15] + short[7:]                                        # header promises one byte more than the blocks hold
    with pytest.raises(conduit.Lz4FrameError, match="ERROR_frameSize_wrong"):
        conduit.decompressBatched([forged])


def test_batched_decoder_is_memory_bounded(L, named_inputs):
    """decompressBatched must not gather the stream (the reference's decompress holds one max(hint, 16 KiB) buffer, Conduit.hsc:634-659):
    the conduit cuts runs of whole blocks out of the chunks as they arrive and hands them to lz4f_mi355x_fdec_blocks batch by batch.
    Output must start while most of the input has not been asked for yet; linked frames carry their 64 KiB of history from batch to
    batch; content checksum and contentSize are checked over all batches; every framing gives the oracle's bytes."""
    import ctypes
    rng = np.random.default_rng(3)
    data = np.concatenate([datagen.synth50(24 << 20, 9), rng.integers(0, 256, 3 << 20, dtype=np.uint8), np.frombuffer(datagen.synth_text(5 << 20, 2).tobytes(), dtype=np.uint8)]).tobytes()
    framings = [dict(bsid=4, indep=0, cck=1), dict(bsid=4, indep=1, bck=1), dict(bsid=7, indep=1, cck=1, csize=len(data)), dict(bsid=5, indep=0, bck=1, cck=1), dict(bsid=6, indep=1)]
    for kw in framings:
        fr = oracle.conduit_compress(data, oracle.mkprefs(**kw))
        stream = fr
        for batch, step in ((1 << 20, 300007), (8 << 20, 1 << 20), (None, 65536)):
            chunks = [stream[i:i + step] for i in range(0, len(stream), step)]
            assert sha(b"".join(conduit.decompressBatched(chunks, batch))) == sha(data), (kw, batch, step)
    # two frames and a skippable one, cut at awkward places
    a = oracle.conduit_compress(data[:5 << 20], oracle.mkprefs(bsid=4, indep=0, cck=1)); b = oracle.conduit_compress(data[5 << 20:9 << 20], oracle.mkprefs(bsid=5, indep=1))
    skip = bytes.fromhex("5e2a4d18") + (70001).to_bytes(4, "little") + bytes(70001)
    stream = a + skip + b
    for step in (1, 7, 4099):
        cuts = [stream[:step], stream[step:len(a) - 2], stream[len(a) - 2:len(a) + 9], stream[len(a) + 9:len(a) + len(skip) + 3], stream[len(a) + len(skip) + 3:]]
        assert b"".join(conduit.decompressBatched(cuts, 1 << 20)) == data[:9 << 20]
    # bounded: with 1 MiB batches the first output arrives before a tenth of the input has been asked for
    fr = oracle.conduit_compress(data, oracle.mkprefs(bsid=4, indep=0, cck=1))
    pulled, first_out_at, got = [0], [None], []
    it = iter(fr[i:i + 65536] for i in range(0, len(fr), 65536))
    keep = [None]
    def _await(_u, pdata):
        try: c = next(it)
        except StopIteration: pdata[0] = None; return 0
        keep[0] = ctypes.create_string_buffer(c, len(c)); pdata[0] = ctypes.cast(keep[0], ctypes.c_void_p).value; pulled[0] += len(c); return len(c)
    def _yield(_u, p, n):
        if first_out_at[0] is None: first_out_at[0] = pulled[0]
        got.append(ctypes.string_at(p, n))
    err = ctypes.create_string_buffer(512)
    rc = L.lz4f_mi355x_conduit_decompress_batched_bounded(1 << 20, _ffi.AWAIT_FN(_await), _ffi.YIELD_FN(_yield), None, err, 512)
    assert rc == 0, err.value
    assert b"".join(got) == data and first_out_at[0] is not None and first_out_at[0] < len(fr) // 10, (first_out_at[0], len(fr))
    # errors keep their names when the damage is in a later batch
    bad = bytearray(fr); bad[-2] ^= 1                                                # the content checksum
    with pytest.raises(conduit.Lz4FrameError, match="ERROR_contentChecksum_invalid"):
        conduit.decompressBatched([bytes(bad)], 1 << 20)
    with pytest.raises(conduit.Lz4FrameError, match="stream ended before EndMark"):
        conduit.decompressBatched([fr[:len(fr) * 2 // 3]], 1 << 20)
    forged = oracle.conduit_compress(data[:3 << 20], oracle.mkprefs(bsid=4, indep=1, csize=(3 << 20)))
    short = oracle.conduit_compress(data[:(3 << 20) - 1], oracle.mkprefs(bsid=4, indep=1))
    with pytest.raises(conduit.Lz4FrameError, match="ERROR_frameSize_wrong"):
        conduit.decompressBatched([forged[:15] + short[7:]], 1 << 20)


# ------------------------------------------------------------------------------------------------
# Sequence-shape fuzz: inputs built to hit every length class and overlap case of the copy paths (datagen.structured),
# both directions, every block size, linked and independent, against the oracle (bit-exact with liblz4): 72 cases.
def test_structured_inputs_both_directions(L):
    sizes = [70000, 300000, 1 << 20, (4 << 20) + 12345, 9 << 20]
    combos = [dict(bsid=4, indep=1), dict(bsid=4, indep=0), dict(bsid=5, indep=1, bck=1), dict(bsid=6, indep=0, cck=1), dict(bsid=7, indep=1), dict(bsid=7, indep=0)]
    n_cases = 0
    worst = (0.0, -1, "")
    for seed in range(12):
        data = datagen.structured(sizes[seed % len(sizes)], 1000 + seed)
        for kw in combos:
            ref = oracle.conduit_compress(data, oracle.mkprefs(**kw))
            out, used = gpu_decompress_frame(L, ref, len(data) + 8)                  # liblz4-identical frame -> GPU decode
            assert used == len(ref) and sha(out) == sha(data), (seed, kw, "decode")
            frame = gpu_compress_frame(L, data, prefs_of(kw))                         # GPU encode -> oracle decode, GPU decode
            assert oracle.decompress_frame(frame, len(data) + 64)[0] == data, (seed, kw, "encode/oracle")
            assert gpu_decompress_frame(L, frame, len(data) + 8)[0] == data, (seed, kw, "encode/gpu")
            worst = max(worst, (len(frame) / len(ref), seed, str(kw)))
            n_cases += 1
    assert n_cases == 72
    # these inputs are built to stress the copy paths, not to look like data: tiny alphabets, 4-byte matches and short-period
    # runs that begin anywhere favour liblz4's position-by-position search over 64 probes per step by sixteen waves that share
    # one table (where a periodic run begins, the waves in front of the hindmost one keep overwriting its few hot slots; which
    # wave gets there first differs from run to run, so does the size).  The ratio bar of the parity configs (RATIO_TOL) is
    # checked on their inputs above; here only a sanity bound.
    print("worst size ratio vs liblz4 on structured inputs: %.3f (seed %d, %s)" % worst)
    assert worst[0] <= 1.32, worst          # measured 1.228 .. 1.255 over runs (round 3, seed 11: a 9 MiB input of short-period runs), + 5 %


# ------------------------------------------------------------------------------------------------
# Streaming compress API under arbitrary call patterns: random slice sizes (0 .. several blocks), LZ4F_flush at random
# points, autoFlush on/off, every framing.  Each call must stay within LZ4F_compressBound of its slice, and the
# concatenated output must be a frame the oracle (= liblz4's decoder) and the GPU decode to the input.
def test_streaming_compress_random_call_patterns(L, named_inputs):
    rng = np.random.default_rng(4242)
    data_all = datagen.structured(3 << 20, 3001) + named_inputs["text512k"]
    framings = [dict(bsid=4, indep=0), dict(bsid=4, indep=1, bck=1), dict(bsid=5, indep=0, cck=1), dict(bsid=6, indep=1), dict(bsid=7, indep=0), dict(bsid=7, indep=1, cck=1, bck=1)]
    for trial in range(12):
        kw = dict(framings[trial % len(framings)])
        auto = int(trial % 3 == 1)
        n = int(rng.integers(1, len(data_all)))
        start = int(rng.integers(0, len(data_all) - n + 1))
        data = data_all[start:start + n]
        p = prefs_of(kw); p.autoFlush = auto
        c = ctypes.c_void_p(); assert L.LZ4F_createCompressionContext(ctypes.byref(c), 100) == 0
        out = []
        hdr = ctypes.create_string_buffer(32)
        r = L.LZ4F_compressBegin(c, hdr, 32, ctypes.byref(p)); assert not L.LZ4F_isError(r)
        out.append(hdr.raw[:r])
        pos = 0
        while pos < n:
            kind = int(rng.integers(0, 10))
            if kind == 0:
                k = 0
            elif kind <= 5:
                k = int(rng.integers(1, 70000))
            else:
                k = int(rng.integers(1, 3 << 20))
            k = min(k, n - pos)
            bound = L.LZ4F_compressBound(k, ctypes.byref(p))
            dst = ctypes.create_string_buffer(max(bound, 1))
            r = L.LZ4F_compressUpdate(c, dst, bound, data[pos:pos + k], k, None)
            assert not L.LZ4F_isError(r), (trial, L.LZ4F_getErrorName(r), L.lz4f_mi355x_last_error())
            assert r <= bound
            out.append(dst.raw[:r]); pos += k
            if int(rng.integers(0, 6)) == 0:
                fb = L.LZ4F_compressBound(0, ctypes.byref(p)); fd = ctypes.create_string_buffer(fb)
                r = L.LZ4F_flush(c, fd, fb, None); assert not L.LZ4F_isError(r) and r <= fb
                out.append(fd.raw[:r])
        eb = L.LZ4F_compressBound(0, ctypes.byref(p)); ed = ctypes.create_string_buffer(eb)
        r = L.LZ4F_compressEnd(c, ed, eb, None); assert not L.LZ4F_isError(r) and r <= eb
        out.append(ed.raw[:r])
        L.LZ4F_freeCompressionContext(c)
        frame = b"".join(out)
        got, used = oracle.decompress_frame(frame, n + 64)
        assert used == len(frame) and got == data, (trial, kw, auto)
        got2, used2 = gpu_decompress_frame(L, frame, n + 8)
        assert used2 == len(frame) and got2 == data, (trial, kw, auto)


# ------------------------------------------------------------------------------------------------
# Streaming decompress API under arbitrary call patterns: random source slice sizes and destination capacities (down
# to 1 byte), over frames with short (flushed) blocks, stored blocks, checksums, skippable and concatenated frames.
def test_streaming_decompress_random_call_patterns(L, named_inputs):
    rng = np.random.default_rng(777)
    text = named_inputs["text512k"]; rnd = named_inputs["random10m"][:200000]; st = datagen.structured(1 << 20, 3100)
    skip = bytes.fromhex("5a2a4d18") + (7).to_bytes(4, "little") + b"skipme!"
    def flushed(data, kw, slice_):                      # oracle frame with autoFlush: one (short) block per slice
        return oracle.conduit_compress(data, oracle.mkprefs(autoflush=1, **kw), slice_)
    streams = [
        (oracle.conduit_compress(st, oracle.mkprefs(bsid=4, indep=0, cck=1)), st),
        (oracle.conduit_compress(text, oracle.mkprefs(bsid=7, indep=1, bck=1, cck=1)), text),
        (flushed(text, dict(bsid=5, indep=0, cck=1), 10000), text),
        (flushed(st, dict(bsid=4, indep=1, bck=1), 50000), st),
        (oracle.conduit_compress(rnd, oracle.mkprefs(bsid=4, indep=1, csize=len(rnd))), rnd),
        (skip + oracle.conduit_compress(text, oracle.mkprefs(bsid=6, indep=1)) + skip + oracle.conduit_compress(rnd, oracle.mkprefs(bsid=4, indep=0)), text + rnd),
    ]
    for si, (stream, want) in enumerate(streams):
        for trial in range(2):
            d = ctypes.c_void_p(); assert L.LZ4F_createDecompressionContext(ctypes.byref(d), 100) == 0
            out = bytearray(); pos = 0; hint = 1; calls = 0
            small = trial == 1
            while pos < len(stream):
                k = int(rng.integers(1, 300 if small else 120000)); k = min(k, len(stream) - pos)
                cap = int(rng.integers(1, 500 if small else 300000))
                dst = ctypes.create_string_buffer(cap); ds = ctypes.c_size_t(cap); ss = ctypes.c_size_t(k)
                hint = L.LZ4F_decompress(d, dst, ctypes.byref(ds), stream[pos:pos + k], ctypes.byref(ss), None)
                assert not L.LZ4F_isError(hint), (si, trial, L.LZ4F_getErrorName(hint), L.lz4f_mi355x_last_error())
                assert ss.value <= k and ds.value <= cap
                out += dst.raw[:ds.value]; pos += ss.value; calls += 1
                assert calls < 2_000_000
            # drain what the context still holds (nothing left to feed)
            while hint != 0:
                dst = ctypes.create_string_buffer(65536); ds = ctypes.c_size_t(65536); ss = ctypes.c_size_t(0)
                hint = L.LZ4F_decompress(d, dst, ctypes.byref(ds), b"", ctypes.byref(ss), None)
                assert not L.LZ4F_isError(hint)
                if ds.value == 0: break
                out += dst.raw[:ds.value]
            L.LZ4F_freeDecompressionContext(d)
            assert hint == 0, (si, trial)
            assert bytes(out) == want, (si, trial, len(out), len(want))


# ------------------------------------------------------------------------------------------------
# Linked frames go through the windowed kernel (decode_linked.cuh); frames it hands back (short blocks made by
# autoFlush) go through the generic one.  Both must give liblz4's bytes; so must decoding a slab with history.
def test_linked_frames_windowed_and_fallback(L, named_inputs):
    st = datagen.structured(3 << 20, 3200)
    s50 = named_inputs["synth50_8m"][:5 << 20]
    cases = [
        (oracle.conduit_compress(s50, oracle.mkprefs(bsid=4, indep=0)), s50),                                  # the reference's default framing
        (oracle.conduit_compress(st, oracle.mkprefs(bsid=4, indep=0, bck=1, cck=1)), st),
        (oracle.conduit_compress(st, oracle.mkprefs(bsid=6, indep=0)), st),
        (oracle.conduit_compress(st, oracle.mkprefs(bsid=4, indep=0, autoflush=1), 50000), st),                # short blocks: fallback path
        (oracle.conduit_compress(named_inputs["random10m"][:400000] + st[:300000], oracle.mkprefs(bsid=4, indep=0)), named_inputs["random10m"][:400000] + st[:300000]),  # stored blocks inside
    ]
    import os
    # three decoders for a linked frame that arrives without an index: the one that indexes it itself (default), the window
    # kernel (what the former falls back to, and takes dense frames by itself), the single-workgroup generic kernel
    for env in ({}, {"LZ4F_MI355X_NO_SELFINDEX": "1"}, {"LZ4F_MI355X_NO_SELFINDEX": "1", "LZ4F_MI355X_NO_WINDOW": "1"}):
        os.environ.update(env)
        L.lz4f_mi355x_release_engines()                                       # (an engine reads its switches when it is made)
        try:
            for i, (frame, want) in enumerate(cases):
                out, used = gpu_decompress_frame(L, frame, len(want) + 8)
                assert used == len(frame) and sha(out) == sha(want), (i, env)
        finally:
            for k in env:
                os.environ.pop(k, None)
            L.lz4f_mi355x_release_engines()
    # malformed linked frames: same verdicts as the oracle
    base = cases[0][0]
    for pos in (200, 5000, 70000, 140000, len(base) // 2):
        bad = datagen.mutate(base, pos)
        try:
            want, _ = oracle.decompress_frame(bad, len(s50) + 8); verdict = "ok"
        except oracle.OracleError as e:
            want, verdict = None, str(e)
        try:
            got, _ = gpu_decompress_frame(L, bad, len(s50) + 8); gv = "ok"
        except RuntimeError as e:
            got, gv = None, str(e).split(" | ")[0]
        if verdict == "ok":
            assert gv == "ok" and got == want, (pos, gv)
        else:
            assert gv != "ok", (pos, verdict)


def _indexed_inputs():
    rng = np.random.default_rng(5)
    yield "synth50", datagen.synth50(12 << 20, 91)
    yield "synth50-ragged", datagen.synth50(9 << 20, 92)[: (9 << 20) - 12345]
    for i in range(4):
        yield "structured-%d" % i, np.frombuffer(datagen.structured((5 << 20) + 777 * i, 3100 + i), dtype=np.uint8).copy()
    yield "text", datagen.synth_text(6 << 20, 17)
    yield "zeros", np.zeros(5 << 20, dtype=np.uint8)
    yield "period-3", np.tile(np.frombuffer(b"abc", dtype=np.uint8), (4 << 20) // 3 + 1)[: 4 << 20].copy()
    mixed = np.concatenate([rng.integers(0, 256, 3 << 20, dtype=np.uint8), datagen.synth50(4 << 20, 93), rng.integers(0, 256, 1 << 20, dtype=np.uint8)])
    yield "stored-blocks-mixed", mixed
    yield "tiny", np.frombuffer(b"hello hello hello hello hello hello hello hello", dtype=np.uint8).copy()


@pytest.mark.parametrize("selffeed", [True, False])
def test_indexed_decode_same_bytes_as_generic(L, monkeypatch, selffeed):
    """The compressor's sequence index only tells the decoder where it may start parsing: with it (indexed kernels) and
    without it (generic kernels) the same frame gives the same bytes, for every block size the indexed path takes.
    Independent blocks go through the self-feeding copy kernel (k_copy_selffed: the workgroup's first wave parses and resolves);
    selffeed = False: through k_parse_indexed / k_resolve_direct / k_copy_indexed as linked and dense frames do."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    if not selffeed: monkeypatch.setenv("LZ4F_MI355X_NO_SELFFEED", "1")         # (switches are read when an engine is made)
    eng = Engine(0)
    used = 0
    for name, data in _indexed_inputs():
        for bsid, indep in ((5, 1), (6, 1), (7, 1), (4, 0), (7, 0)):          # linked frames too (the reference's default framing is 64 KiB linked)
            kw = dict(bsid=bsid, indep=indep, bck=1 if bsid == 6 else 0)
            src = torch.from_numpy(data).cuda()
            p = prefs_of(kw)
            bs = 1 << (8 + 2 * bsid)
            nb = (src.numel() + bs - 1) // bs
            frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
            table, index = eng.new_table(nb), eng.new_index(src.numel(), p)
            eng.compress_async(src, frame, p, table, index)
            r = eng.result()
            plain = torch.empty_like(frame)
            eng.compress_async(src, plain, p, eng.new_table(nb))
            rp = eng.result()
            # (pass E1's waves race for hash-table slots, so two compressions of one input may choose different - equally valid - matches:
            # the frames are compared through what they decode to; sizes are held against liblz4's in the ratio tests)
            chk = torch.zeros_like(src)
            eng.decompress_frame_async(plain, int(rp.size), chk)
            rc = eng.result()
            assert rc.size == src.numel() and torch.equal(chk, src), (name, kw)
            hd = index[:32].cpu().numpy().view(np.uint32)
            used += int(hd[0] == 0x3258494C)
            back = torch.zeros_like(src)
            eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo, index)
            r2 = eng.result()
            assert r2.size == src.numel() and torch.equal(back, src), (name, kw)
            back.zero_()
            eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo)
            r3 = eng.result()
            assert r3.size == src.numel() and torch.equal(back, src), (name, kw)
    assert used >= 35                                                         # (dense streams may legitimately get an unusable index)
    eng.close()


@pytest.mark.parametrize("how", ["doubling", "hops"])
def test_traced_decode_same_bytes(L, monkeypatch, how):
    """Dense frames (text) are decoded by tracing every output byte back to its literal instead of walking the chain: by
    pointer doubling (k_pd_init / k_pd_round, up to 1 GiB of output) or hop by hop (k_trace_copy).  Forced on here for
    every input and framing: same bytes as the source, and a wrong index is still noticed (the tracers validate what they
    follow and hand over to the generic kernels)."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    monkeypatch.setenv("LZ4F_MI355X_TRACE_ALWAYS", "1")
    if how == "hops":
        monkeypatch.setenv("LZ4F_MI355X_NO_DOUBLING", "1")
    eng = Engine(0)
    used = 0
    prev = None
    for name, data in _indexed_inputs():
        for bsid, indep in ((5, 1), (7, 1), (4, 0), (7, 0)):
            kw = dict(bsid=bsid, indep=indep)
            src = torch.from_numpy(data).cuda()
            p = prefs_of(kw)
            bs = 1 << (8 + 2 * bsid)
            nb = (src.numel() + bs - 1) // bs
            frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
            table, index = eng.new_table(nb), eng.new_index(src.numel(), p)
            eng.compress_async(src, frame, p, table, index)
            r = eng.result()
            used += int(index[:4].cpu().numpy().view(np.uint32)[0] == 0x3258494C)
            back = torch.zeros_like(src)
            eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo, index)
            r2 = eng.result()
            assert r2.size == src.numel() and torch.equal(back, src), (name, kw)
            if prev is not None and prev[0].numel() == index.numel():       # another stream's index for this frame
                back.zero_()
                eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo, prev[0])
                r3 = eng.result()
                assert r3.size == src.numel() and torch.equal(back, src), (name, kw, "foreign index")
            prev = (index,)
    assert used >= 28
    eng.close()


def test_linked_dense_frame_with_unusable_index(L):
    """Text has more sequences than an index of the recommended size has room for: the compressor marks it unusable.  For a
    linked frame that must not mean the window kernel (one chain, ~50 ms per MiB): the decoder makes its own index and
    decodes by pointer doubling.  Same bytes, and in a time the chain could not do."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    eng = Engine(0)
    eng.set_timing(True)
    data = np.tile(datagen.synth_text(4 << 20, 5), 4)
    src = torch.from_numpy(data).cuda()
    p = prefs_of(dict(bsid=4, indep=0))
    nb = src.numel() >> 16
    frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
    table, index = eng.new_table(nb), eng.new_index(src.numel(), p)
    eng.compress_async(src, frame, p, table, index)
    r = eng.result()
    assert int(index[:4].cpu().numpy().view(np.uint32)[0]) != 0x3258494C           # unusable, as expected for this density
    best = 1e9
    for _ in range(3):
        back = torch.zeros_like(src)
        eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo, index)
        r2 = eng.result()
        assert r2.size == src.numel() and torch.equal(back, src)
        best = min(best, eng.get_timing()["decode"])
    assert best < 200.0, best                                                       # (16 MiB through the window kernel: ~850 ms; here ~2)
    eng.close()


def test_indexed_decode_survives_wrong_indexes(L):
    """A stale, foreign, truncated or corrupted index must never change the output: the decoder notices and falls back."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    eng = Engine(0)
    rng = np.random.default_rng(11)
    a = np.frombuffer(datagen.structured(9 << 20, 4200), dtype=np.uint8).copy()
    b = datagen.synth50(9 << 20, 4201)
    p = prefs_of(dict(bsid=6, indep=1))
    nb = 9

    def pack(data):
        src = torch.from_numpy(data).cuda()
        frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
        table, index = eng.new_table(nb), eng.new_index(src.numel(), p)
        eng.compress_async(src, frame, p, table, index)
        return src, frame, table, index, eng.result().size

    sa, fa, ta, ia, na = pack(a)
    sb, fb, tb, ib, nb_sz = pack(b)
    assert int(ia[:4].cpu().numpy().view(np.uint32)[0]) == 0x3258494C and int(ib[:4].cpu().numpy().view(np.uint32)[0]) == 0x3258494C

    def check(src, frame, table, size, index):
        back = torch.zeros_like(src)
        eng.decompress_blocks_async(frame, size, back, table, nb, p.frameInfo, index)
        r = eng.result()
        assert r.size == src.numel() and torch.equal(back, src)

    check(sa, fa, ta, na, ib)                                                 # another stream's index, same geometry
    check(sb, fb, tb, nb_sz, ia)
    check(sa, fa, ta, na, torch.zeros_like(ia))                               # never written
    check(sa, fa, ta, na, ia[:64].clone())                                    # truncated behind the header
    check(sa, fa, ta, na, torch.from_numpy(rng.integers(0, 256, ia.numel(), dtype=np.uint8)).cuda())
    hd = ia[:32].cpu().numpy().view(np.uint32)
    n_entries = int(hd[4])
    fixed = 32 + nb * 16 + nb * int(hd[2]) * 8
    for trial in range(24):                                                   # one damaged word somewhere in the tables
        bad = ia.clone()
        words = bad[: fixed + n_entries * 16].view(torch.int32)
        at = int(rng.integers(8, words.numel()))
        words[at] = int(words[at].item()) ^ (1 << int(rng.integers(0, 24)))
        check(sa, fa, ta, na, bad)
    small = torch.zeros(fixed + 16, dtype=torch.uint8, device="cuda")         # too small for the stream: the compressor says so in the header
    src = torch.from_numpy(a).cuda()
    frame = torch.empty_like(fa)
    table = eng.new_table(nb)
    eng.compress_async(src, frame, p, table, small)
    r = eng.result()
    assert int(small[:4].cpu().numpy().view(np.uint32)[0]) == 0
    check(src, frame, table, r.size, small)
    # linked frames: the same with their own index, a foreign one and a damaged one
    pl = prefs_of(dict(bsid=6, indep=0))
    frame_l = torch.empty(eng.frame_bound(src.numel(), pl), dtype=torch.uint8, device="cuda")
    idx = eng.new_index(src.numel(), pl)
    eng.compress_async(src, frame_l, pl, table, idx)
    r = eng.result()
    back = torch.zeros_like(src)
    eng.decompress_blocks_async(frame_l, r.size, back, table, nb, pl.frameInfo, idx)
    assert eng.result().size == src.numel() and torch.equal(back, src)
    for other in (ib, torch.zeros_like(idx)):
        back.zero_()
        eng.decompress_blocks_async(frame_l, r.size, back, table, nb, pl.frameInfo, other)
        assert eng.result().size == src.numel() and torch.equal(back, src)
    sl = torch.from_numpy(b).cuda()                                            # synth50: this one takes the indexed kernels (block-parallel)
    eng.compress_async(sl, frame_l, pl, table, idx)
    r = eng.result()
    for trial in range(8):
        bad = idx.clone()
        hdw = bad[:32].cpu().numpy().view(np.uint32)
        words = bad[: 32 + nb * 16 + nb * int(hdw[2]) * 8 + int(hdw[4]) * 16].view(torch.int32)
        at = int(rng.integers(8, words.numel()))
        words[at] = int(words[at].item()) ^ (1 << int(rng.integers(0, 24)))
        back = torch.zeros_like(sl)
        eng.decompress_blocks_async(frame_l, r.size, back, table, nb, pl.frameInfo, bad)
        assert eng.result().size == sl.numel() and torch.equal(back, sl)
    eng.close()


def test_host_calls_across_staging_pieces(L):
    """The host-pointer calls move data through pinned staging in 32 MiB pieces (copy and DMA overlapped): sizes around the
    piece boundaries, a mixed stream, independent and linked frames; the oracle decodes what the GPU wrote."""
    rng = np.random.default_rng(21)
    for n in ((32 << 20) - 1, (32 << 20) + 1, (64 << 20) + 12345, (70 << 20) + 7):
        data = np.concatenate([datagen.synth50(24 << 20, 3), rng.integers(0, 256, 9 << 20, dtype=np.uint8),
                               np.frombuffer(datagen.structured(n - (33 << 20), 5), dtype=np.uint8)])[:n].tobytes()
        for kw in (dict(bsid=7, indep=1, cck=1), dict(bsid=4, indep=0)):
            frame = gpu_compress_frame(L, data, prefs_of(kw))
            out, used = gpu_decompress_frame(L, frame, len(data) + 8)
            assert used == len(frame) and out == data, (n, kw)
            if n == (32 << 20) + 1:
                ref, ref_used = oracle.decompress_frame(frame, cap=len(data) + 64)
                assert ref_used == len(frame) and ref == data, (n, kw)


@pytest.mark.gpu
def test_parallel_walk_of_device_frames(L, monkeypatch):
    """Device-resident frames without a block table (small blocks): the size words are found in parallel (k_walk_cand ...
    k_walk_verdict) and must give the table the serial walk gives - also when the payload is full of bytes that look like
    size words (stored blocks of small little-endian integers: false candidates, which are filtered out or send the frame
    to the serial walk), with a block checksum behind every block, and with garbage behind the frame."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    eng = Engine(0)
    monkeypatch.setenv("LZ4F_MI355X_SERIAL_WALK", "1")                                     # (switches are read when an engine is made)
    eng_serial = Engine(0)
    monkeypatch.delenv("LZ4F_MI355X_SERIAL_WALK")
    rng = np.random.default_rng(31)
    ints = rng.integers(0, 40000, 3 << 20, dtype=np.uint32).view(np.uint8)                 # 12 MiB, every word a plausible size word
    inputs = [("synth50", datagen.synth50(24 << 20, 5)), ("text", datagen.synth_text(20 << 20, 3)), ("small ints", ints),
              ("structured", np.frombuffer(datagen.structured(17 << 20, 77), dtype=np.uint8).copy())]
    for name, data in inputs:
        for kw in (dict(bsid=4, indep=1), dict(bsid=4, indep=1, bck=1), dict(bsid=5, indep=1), dict(bsid=4, indep=0)):
            ref = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))              # == liblz4's frame
            tail = rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()
            dev = torch.from_numpy(np.frombuffer(ref + tail, dtype=np.uint8).copy()).cuda()
            src = torch.from_numpy(data).cuda()
            results = []
            for serial in (False, True):
                en = eng_serial if serial else eng
                back = torch.zeros_like(src)
                en.decompress_frame_async(dev, dev.numel(), back)
                r = en.result()
                assert r.size == src.numel() and r.consumed == len(ref) and torch.equal(back, src), (name, kw, serial)
                assert bool((r.flags >> 12) & PATH["parallel_walk"]) == (not serial and len(ref) >= (1 << 20)), (name, kw, serial, hex(r.flags))
                results.append((int(r.n_blocks), int(r.consumed)))
            assert results[0] == results[1], (name, kw, results)
    eng.close(); eng_serial.close()


@pytest.mark.gpu
def test_inband_trailer_interop_and_robustness(L):
    """The in-band index: a skippable frame behind the LZ4 frame.  liblz4 (the oracle), the reference's decompress conduit and the
    batched conduit decode the stream to the input; this library's device decoder uses the trailer (no walk, indexed parse) and
    gives the same bytes - also when the trailer is damaged, truncated, or lies about the frame."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    eng = Engine(0)
    rng = np.random.default_rng(12)
    for name, data, kw in (("synth50", datagen.synth50(24 << 20, 9), dict(bsid=7, indep=1)), ("synth50/1M", datagen.synth50(8 << 20, 10), dict(bsid=6, indep=1, bck=1)),
                           ("text", datagen.synth_text(6 << 20, 4), dict(bsid=5, indep=1)), ("linked", datagen.synth50(4 << 20, 11), dict(bsid=4, indep=0))):
        src = torch.from_numpy(data).cuda()
        p = prefs_of(kw)
        frame = torch.empty(eng.frame_bound_inband(src.numel(), p), dtype=torch.uint8, device="cuda")
        eng.compress_async(src, frame, p, inband=True)
        r = eng.result()
        stream = frame[:r.size].cpu().numpy().tobytes()
        out, used = oracle.decompress_frame(stream, cap=len(data) + 64)                    # liblz4's decoder: the frame, then what is left is a skippable frame
        assert out == data.tobytes() and used < len(stream), name
        rest = stream[used:]
        assert rest[:4] == bytes.fromhex("5e2a4d18") and int.from_bytes(rest[4:8], "little") == len(rest) - 8, name
        assert b"".join(conduit.decompress([stream])) == data.tobytes(), name            # Conduit.hsc:598: stops at the EndMark
        assert b"".join(conduit.decompressBatched([stream[:100000], stream[100000:]])) == data.tobytes(), name
        back = torch.zeros_like(src)
        eng.decompress_frame_async(frame, int(r.size), back)
        r2 = eng.result()
        assert r2.size == src.numel() and r2.consumed == used and torch.equal(back, src), name
        path = int(r2.flags) >> 12                                                       # the trailer's list and its index were used - for a linked frame too (round 4:
        assert path & PATH["trailer"] and not path & PATH["dropped"], (name, hex(path))      # until then its decoder indexed it itself)
        if name != "text": assert path & PATH["indexed"], (name, hex(path))              # (a dense stream overflows the index it is given: its trailer is the block list alone)
        if name == "linked": assert not path & PATH["self_index"], hex(path)
        # damage: bytes of the block list, of the index, of the footer; a footer that names another block count; a cut trailer
        tr = len(stream) - used
        for trial in range(12):
            bad = bytearray(stream)
            if trial < 6:
                for _ in range(1 + trial): bad[used + int(rng.integers(8, tr))] ^= int(rng.integers(1, 256))
            elif trial == 6: bad[-12:-8] = (int.from_bytes(bad[-12:-8], "little") + 1).to_bytes(4, "little")
            elif trial == 7: bad[-8:] = (int.from_bytes(bad[-8:], "little") - 16).to_bytes(8, "little")
            elif trial == 8: bad = bad[:len(bad) - 5000] if tr > 6000 else bad[:-8]
            elif trial == 9: bad[used + 16:used + 16 + 64] = bytes(64)
            elif trial == 10: bad[-32:-28] = (1).to_bytes(4, "little")                     # "one sequence": the workspace it sizes is too small, the device notices
            else: bad[-32:-24] = b"\xf0\xff\xff\xff" * 2                                 # counts the index cannot hold
            dev = torch.zeros(len(bad) + 32, dtype=torch.uint8, device="cuda")
            dev[:len(bad)] = torch.from_numpy(np.frombuffer(bytes(bad), dtype=np.uint8).copy()).cuda()
            back.zero_()
            eng.decompress_frame_async(dev, len(bad), back)
            r3 = eng.result()
            assert r3.size == src.numel() and torch.equal(back, src), (name, trial)
    eng.close()


@pytest.mark.gpu
def test_encoder_record_pool_is_bounded_and_overflow_is_graceful(L):
    """The match finder's record lists are allocated from a pool as the tiles are merged (csrc/encode.cuh: rec_ctl / rec_offs): sized for a
    sequence per 5.3 input bytes by default (1.5 bytes of workspace per input byte instead of the worst case's 2) and by the caller's word
    where the data is known (LZ4F_MI355X_RECS_PER_TILE=1024: 0.13 bytes per byte, enough for the bench input eight times over).  With the default pool neither the
    bench input nor dense text nor the structured generators touch its end (result.flags bit 9 clear); with a pool made too small on purpose
    (LZ4F_MI355X_RECS_PER_TILE) the tiles that find it empty go out as literals: flag set, frame still valid and still the input, only
    bigger - never a wrong byte, never a write outside the pool."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    POOL_SHORT = 0x200
    text = datagen.synth_text(24 << 20, 3)
    s50 = datagen.synth50(32 << 20, 4)
    stru = np.frombuffer(datagen.structured(16 << 20, 8), dtype=np.uint8)
    rep = np.frombuffer((b"abcdefg" * 10)[:64] * (16 << 14), dtype=np.uint8)             # short period: a sequence per few bytes where the parse is dense

    def run(eng, data, kw):
        src = torch.from_numpy(data.copy()).cuda()
        p = prefs_of(kw)
        frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
        eng.compress_async(src, frame, p)
        r = eng.result()
        stream = frame[:r.size].cpu().numpy().tobytes()
        out, used = oracle.decompress_frame(stream, cap=len(data) + 64)
        assert out == data.tobytes() and used == len(stream)
        return r.size, r.flags

    eng = Engine(0)
    sizes = {}
    for name, data in (("text", text), ("s50", s50), ("structured", stru), ("period", rep)):
        for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=0)):
            size, flags = run(eng, data, kw)
            assert not (flags & POOL_SHORT), (name, kw)
            sizes[(name, kw["bsid"])] = size
    ws4g = L.lz4f_mi355x_dev_workspace_size(4 << 30, ctypes.byref(prefs_of(dict(bsid=7, indep=1))))
    assert ws4g < 6.6e9, ws4g                                        # (round 2: 8.6 GB)
    eng.close()
    os.environ["LZ4F_MI355X_RECS_PER_TILE"] = "1024"                # a record per 64 bytes: text needs nine times that, the bench input a thirtieth
    try:
        small = Engine(0)
    finally:
        del os.environ["LZ4F_MI355X_RECS_PER_TILE"]
    for name, data in (("text", text), ("s50", s50)):
        for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=0)):
            size, flags = run(small, data, kw)
            if name == "text":
                assert (flags & POOL_SHORT) and size > sizes[(name, kw["bsid"])], (name, kw)
            else:
                assert not (flags & POOL_SHORT) and abs(size - sizes[(name, kw["bsid"])]) < 0.01 * size, (name, kw)
    small.close()


@pytest.mark.gpu
def test_block_list_trailer_from_the_host_paths(L, tmp_path):
    """The block list made on the HOST (lz4f_mi355x_appendBlockList over any finished frame; compressBatched(blockList=True); `mi355x-lz4c
    --index`): the device decoder finds the size words through it (PATH trailer set, no walk of any kind), the bytes are the input's, every
    other reader skips the extra frame, and a list that lies is only a hint that fails its check."""
    import subprocess
    import torch
    from lz4_frame_conduit_amd.device import Engine
    eng = Engine(0)
    P = PATH
    walks = P["parallel_walk"] | P["table"]

    def dev_decode(stream: bytes, n_out: int):
        dev = torch.zeros(len(stream) + 32, dtype=torch.uint8, device="cuda")
        dev[:len(stream)] = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
        back = torch.zeros(n_out + 64, dtype=torch.uint8, device="cuda")
        eng.decompress_frame_async(dev, len(stream), back)
        r = eng.result()
        return back[:r.size].cpu().numpy().tobytes(), r

    data = datagen.synth50(24 << 20, 21)
    raw = data.tobytes()
    # (1) a foreign frame (the oracle's = liblz4's bytes), listed after the fact on the host
    for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=1, bck=1), dict(bsid=4, indep=0, cck=1)):
        frame = oracle.conduit_compress(raw, oracle.mkprefs(**kw))
        plain_out, plain_r = dev_decode(frame, len(raw))
        assert plain_out == raw and not ((plain_r.flags >> 12) & P["trailer"])
        listed = conduit.appendBlockList(frame)
        out, r = dev_decode(listed, len(raw))
        path = r.flags >> 12
        assert out == raw and r.consumed == len(frame), kw
        assert (path & P["trailer"]) and not (path & walks), (kw, hex(path))
        # a lying list: one entry moved, the count changed - the check fails, the frame is walked, the bytes are still right
        n_blocks = int.from_bytes(listed[-12:-8], "little")
        list_at = (len(frame) + 8 + 15) & ~15
        for trial in range(3):
            bad = bytearray(listed)
            if trial == 0: bad[list_at + 8 * (n_blocks // 2)] ^= 0x10
            elif trial == 1: bad[-12:-8] = (n_blocks - 1).to_bytes(4, "little")
            else: bad[list_at:list_at + 8] = (3).to_bytes(8, "little")
            out, r = dev_decode(bytes(bad), len(raw))
            assert out == raw and r.consumed == len(frame), (kw, trial)
    # (2) the batched conduit with the list, and the readers that do not know it
    p = conduit.make_preferences(blockSizeID=7, blockMode=1)
    stream = b"".join(conduit.compressBatched([raw[:5_000_000], raw[5_000_000:]], p, batchBytes=8 << 20, blockList=True))
    out, used = oracle.decompress_frame(stream, cap=len(raw) + 64)
    assert out == raw and stream[used:used + 4] == bytes.fromhex("5e2a4d18")
    assert conduit.appendBlockList(stream[:used]) == stream                   # the same bytes the host walk of the finished frame gives
    assert b"".join(conduit.decompressBatched([stream])) == raw and b"".join(conduit.decompress([stream])) == raw
    out, r = dev_decode(stream, len(raw))
    assert out == raw and ((r.flags >> 12) & P["trailer"]) and not ((r.flags >> 12) & walks)
    # (3) the command-line driver
    cli = os.path.join(os.path.dirname(_ffi.LIB_PATH), "mi355x-lz4c")
    fin, fz, fback = tmp_path / "in.bin", tmp_path / "in.lz4", tmp_path / "back.bin"
    fin.write_bytes(raw[:9_000_000])
    subprocess.run([cli, str(fin), str(fz), "-B6", "-BI", "--index", "--batch", "4"], check=True, timeout=300)
    z = fz.read_bytes()
    out, used = oracle.decompress_frame(z, cap=9_000_000 + 64)
    assert out == raw[:9_000_000] and used < len(z) and conduit.appendBlockList(z[:used]) == z
    subprocess.run([cli, "-d", str(fz), str(fback)], check=True, timeout=300)
    assert fback.read_bytes() == raw[:9_000_000]
    out, r = dev_decode(z, 9_000_000)
    assert out == raw[:9_000_000] and ((r.flags >> 12) & P["trailer"])
    eng.close()


@pytest.mark.gpu
def test_soak_and_fuzz_slices():
    """A slice of the development soak (tools/soak_indexed.py: random shapes compressed, then decoded with, without and again with the
    index) and of the mutation fuzz (tools/fuzz_linked.py: single-byte mutations, verdicts and bytes against the oracle) on every run
    of the GPU tests: the encoder's waves race for hash slots, so intermittent failures are the realistic kind, and a fixed case list
    does not find them.  Seeds change from run to run (printed on failure); each tool runs in a child process with a time limit."""
    import subprocess, time
    seed = int(time.time()) % 100000
    env = dict(os.environ)
    jobs = [([sys.executable, os.path.join(ROOT, "tools", "soak_indexed.py"), "14", str(seed)], {}),
            ([sys.executable, os.path.join(ROOT, "tools", "fuzz_linked.py"), str(seed), "50"], {}),                                   # linked 64 KiB (self-index, window kernel)
            ([sys.executable, os.path.join(ROOT, "tools", "fuzz_linked.py"), str(seed + 1), "40"], {"INDEP": "1", "BSID": "7"}),      # big independent blocks (density probe, stretches)
            ([sys.executable, os.path.join(ROOT, "tools", "fuzz_linked.py"), str(seed + 2), "40"], {"INDEP": "1", "BSID": "4"}),      # small independent blocks (lanes find the tokens)
            ([sys.executable, os.path.join(ROOT, "tools", "fuzz_linked.py"), str(seed + 3), "40"], {"DATA": "text"})]                 # dense linked frame (pointer doubling)
    for cmd, extra in jobs:
        e = dict(env); e.update(extra)
        r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (cmd[1:], extra, r.stdout[-600:], r.stderr[-300:])


@pytest.mark.gpu
def test_caller_block_table_is_validated(L):
    """lz4f_mi355x_dev_decompressBlocks takes the caller's block table.  Its entries are decoded concurrently, so the device checks them
    against each other before anything runs: in order, inside the frame and the output buffer, no two sharing frame or output bytes.
    A table that fails is refused (ERROR_GENERIC, first_bad_block) and the output is not touched."""
    import torch
    from lz4_frame_conduit_amd.device import Engine, DeviceCodecError
    from lz4_frame_conduit_amd._ffi import Block
    eng = Engine(0)
    data = datagen.synth50(20 << 20, 9)
    src = torch.from_numpy(data).cuda()
    p = prefs_of(dict(bsid=7, indep=1))
    nb = 5
    frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
    table = eng.new_table(nb)
    eng.compress_async(src, frame, p, table)
    r = eng.result()
    back = torch.zeros_like(src)
    eng.decompress_blocks_async(frame, r.size, back, table, nb, p.frameInfo); assert eng.result().size == src.numel() and torch.equal(back, src)
    host = table.cpu().numpy().copy()
    ent = np.frombuffer(host.tobytes(), dtype=np.dtype([("src_off", "<u8"), ("dst_off", "<u8"), ("word", "<u4"), ("dst_size", "<u4")]))[:nb + 1].copy()
    def attempt(mut):
        e2 = ent.copy(); mut(e2)
        t2 = torch.from_numpy(np.frombuffer(e2.tobytes(), dtype=np.uint8).copy()).cuda()
        out = torch.full_like(src, 0x5A)
        eng.decompress_blocks_async(frame, r.size, out, t2, nb, p.frameInfo)
        with pytest.raises(DeviceCodecError): eng.result()
        assert bool((out == 0x5A).all()), "a refused table must not have written anything"
    def swap(e): e[[1, 2]] = e[[2, 1]]
    def same_out(e): e["dst_off"][3] = e["dst_off"][2]
    def overlap_out(e): e["dst_off"][2] -= 4096
    def overlap_in(e): e["src_off"][3] = e["src_off"][2] + 100
    def past_frame(e): e["src_off"][4] = int(r.size) + (1 << 20)
    def past_out(e): e["dst_off"][4] = src.numel() - 100
    def huge_word(e): e["word"][1] = (8 << 20)
    for m in (swap, same_out, overlap_out, overlap_in, past_frame, past_out, huge_word): attempt(m)
    eng.close()


@pytest.mark.gpu
def test_deterministic_encoder_mode(L):
    """liblz4 maps equal input to equal bytes; the default encoder here does not (sixteen waves race for hash slots: sizes differ by
    ~1e-5 between runs, every frame valid).  lz4f_mi355x_engine_set_deterministic / LZ4F_MI355X_DETERMINISTIC=1: one wave per
    workgroup parses in order - the same bytes every time, on fresh engines too, within the ratio tolerance, decodable by liblz4."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    inputs = {"synth50": datagen.synth50(24 << 20, 3), "text": datagen.synth_text(12 << 20, 4), "structured": np.frombuffer(datagen.structured(6 << 20, 8), dtype=np.uint8)}
    for name, data in inputs.items():
        src = torch.from_numpy(data.copy()).cuda()
        for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=0)):
            p = prefs_of(kw)
            seen = set()
            for attempt in range(3):
                eng = Engine(0)
                eng.set_deterministic(True)
                frame = torch.empty(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
                for rep in range(2):
                    frame.zero_()
                    eng.compress_async(src, frame, p)
                    r = eng.result()
                    seen.add(sha(frame[:r.size].cpu().numpy().tobytes()))
                host = frame[:r.size].cpu().numpy().tobytes()
                eng.close()
            assert len(seen) == 1, (name, kw, len(seen))
            out, used = oracle.decompress_frame(host, cap=len(data) + 64)
            assert used == len(host) and out == data.tobytes(), (name, kw)
            ref = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))
            assert len(host) <= len(ref) * (1.32 if name == "structured" else RATIO_TOL), (name, kw, len(host), len(ref))      # (structured: the stress inputs' own bound, see test_structured_inputs_both_directions)
    # a record pool set too small for dense input (LZ4F_MI355X_RECS_PER_TILE) must not make the bytes a matter of which tile asked first:
    # the deterministic mode sizes the pool for the worst case itself
    import os
    os.environ["LZ4F_MI355X_RECS_PER_TILE"] = "512"
    try:
        data = inputs["text"]; src = torch.from_numpy(data.copy()).cuda(); p = prefs_of(dict(bsid=7, indep=1)); seen = set()
        for attempt in range(3):
            eng = Engine(0); eng.set_deterministic(True)
            frame = torch.zeros(eng.frame_bound(src.numel(), p), dtype=torch.uint8, device="cuda")
            eng.compress_async(src, frame, p); r = eng.result()
            assert not (int(r.flags) & 0x200), "tiles went out as literals in deterministic mode"
            seen.add(sha(frame[:r.size].cpu().numpy().tobytes())); eng.close()
        assert len(seen) == 1
    finally:
        os.environ.pop("LZ4F_MI355X_RECS_PER_TILE", None)
    # the twelve LZ4F_* functions (the reference's call pattern, Conduit.hsc:457-533): their engines read LZ4F_MI355X_DETERMINISTIC when they are made
    os.environ["LZ4F_MI355X_DETERMINISTIC"] = "1"
    L.lz4f_mi355x_release_engines()
    try:
        data = inputs["synth50"][:3 << 20].tobytes() + inputs["text"][:1 << 20].tobytes()
        chunks = [data[i:i + 16384] for i in range(0, len(data), 16384)]
        a = b"".join(conduit.compress(chunks)); b = b"".join(conduit.compress(chunks))
        assert a == b
        out, used = oracle.decompress_frame(a, cap=len(data) + 64)
        assert used == len(a) and out == data
    finally:
        os.environ.pop("LZ4F_MI355X_DETERMINISTIC", None)
        L.lz4f_mi355x_release_engines()


def _short_sequences(n: int, seed: int) -> np.ndarray:
    """34 random bytes, then a copy of 30 bytes from 0.1..3 KiB back (inside what liblz4's 4096-entry table still knows), and so on: ~30 payload bytes per sequence - sparse by the density probe's
    rule (>= 24), and more than a thousand sequences in a 32 KiB stretch of payload: the self-feeding copy kernel takes such a stretch in pieces."""
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 256, n, dtype=np.uint8)
    pos = 65536
    while pos + 64 <= n:
        back = int(rng.integers(100, 3000))
        out[pos + 34:pos + 64] = out[pos + 34 - back:pos + 64 - back]
        pos += 64
    return out


@pytest.mark.gpu
def test_foreign_big_independent_blocks_stretch_parallel(L):
    """Frames of big independent blocks that come without an index - what LZ4F_compressFrame / `lz4 -B7` write and the reference's
    decompress tests feed (test/Main.hs:33-36) - are cut into stretches by the decoder itself (decode_spx.cuh: lanes that start at
    guessed tokens, stitched where the true chain lands exactly on a lane's start) and go through the indexed kernels.  liblz4's
    bytes in, the input out, for long sequences, runs, stored blocks inside, short last blocks, block checksums, 256 KiB .. 4 MiB
    blocks; the same frames through the generic decoders (LZ4F_MI355X_NO_SPX); device-resident calls report the path; and
    single-byte mutations get the oracle's verdicts (whoever decodes them in the end)."""
    import torch
    from lz4_frame_conduit_amd.device import Engine, DeviceCodecError
    rng = np.random.default_rng(11)
    s50 = datagen.synth50(13 << 20, 77)
    noise = rng.integers(0, 256, 5 << 20, dtype=np.uint8)
    inputs = {
        "synth50+tail": np.concatenate([s50, noise[:70001]]),
        "stored inside": np.concatenate([s50[:5 << 20], noise[:(4 << 20) + 333], s50[5 << 20:9 << 20]]),
        "runs": np.concatenate([np.zeros(3 << 20, dtype=np.uint8), s50[:2 << 20], np.full(2500000, 7, dtype=np.uint8)]),
        "structured": np.frombuffer(datagen.structured(6 << 20, 5), dtype=np.uint8),
        "ints": np.frombuffer(datagen.ints_100000() * 9, dtype=np.uint8),
        "one short block": s50[:300000],
        "tiny": s50[:13],
        "short sequences": _short_sequences(6 << 20, 21),
    }
    frames = []
    for name, data in inputs.items():
        for kw in (dict(bsid=7, indep=1), dict(bsid=6, indep=1, bck=1), dict(bsid=5, indep=1, cck=1)):
            frames.append((name, kw, data, oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))))
    import os
    # (LZ4F_MI355X_FEED_ROUND=24: the copy kernel's first wave parses rounds of 24 sequences, so that every run of a foreign frame goes in pieces)
    for env in ({}, {"LZ4F_MI355X_FEED_ROUND": "24"}, {"LZ4F_MI355X_NO_SELFFEED": "1"}, {"LZ4F_MI355X_NO_SPX": "1"}):
        os.environ.update(env)
        L.lz4f_mi355x_release_engines()
        try:
            eng = Engine(0)
            for name, kw, data, fr in frames:
                out, used = gpu_decompress_frame(L, fr, len(data) + 8)                       # host-pointer call (slabs)
                assert used == len(fr) and out == data.tobytes(), (name, kw, env)
                dev = torch.from_numpy(np.frombuffer(fr + bytes(32), dtype=np.uint8).copy()).cuda()
                back = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
                eng.decompress_frame_async(dev, len(fr), back)                               # device-resident call
                r = eng.result()
                assert r.size == len(data) and r.consumed == len(fr) and back[:len(data)].cpu().numpy().tobytes() == data.tobytes(), (name, kw, env)
                path = int(r.flags) >> 12
                if name in ("synth50+tail", "stored inside") and kw["bsid"] >= 6:
                    want = 0 if "LZ4F_MI355X_NO_SPX" in env else PATH["self_index"] | PATH["indexed"]
                    assert path & (PATH["self_index"] | PATH["indexed"] | PATH["dropped"]) == want, (name, kw, env, hex(path))
            eng.close()
        finally:
            for k in env: os.environ.pop(k, None)
            L.lz4f_mi355x_release_engines()
    # mutations: the verdict (and, when accepted, the bytes) of the oracle
    name, kw, data, base = frames[0]
    cap = len(data) + 8
    diff = []
    for i in range(60):
        pos = int(rng.integers(7, len(base))); x = int(rng.integers(1, 256))
        if i < 12: pos = 7 + int(rng.integers(0, 3000)) + (i % 3) * 2150000                 # some near the starts of the first blocks
        bad = bytearray(base); bad[pos] ^= x; bad = bytes(bad)
        try:
            want, _ = oracle.decompress_frame(bad, cap); ov = "ok"
        except oracle.OracleError as e:
            want, ov = None, str(e)
        try:
            got, _ = gpu_decompress_frame(L, bad, cap); gv = "ok"
        except RuntimeError as e:
            got, gv = None, str(e).split(" | ")[0]
        if (ov == "ok") != (gv == "ok") or (ov == "ok" and got != want): diff.append((pos, x, ov, gv))
    assert not diff, diff[:5]


def _lz4_seq(lit: bytes, mlen: int, off: int) -> bytes:
    """One LZ4 sequence (token, literal length bytes, literals, offset, match length bytes); mlen == 0: the block's last, literals only."""
    def ext(v):
        out = bytearray()
        while v >= 255: out.append(255); v -= 255
        out.append(v); return bytes(out)
    ll, ml = len(lit), (mlen - 4 if mlen else 0)
    b = bytearray([(min(ll, 15) << 4) | min(ml, 15)])
    if ll >= 15: b += ext(ll - 15)
    b += lit
    if mlen:
        b += bytes([off & 255, off >> 8])
        if ml >= 15: b += ext(ml - 15)
    return bytes(b)


def _dense_block(n_out: int, seed: int, specials: bool):
    """A hand-made LZ4 block of short sequences (dense by the density probe's rule) and what it decodes to: 1..9 literals and a match of 4..18 bytes from
    anywhere in the last 64 KiB - offsets 1..3 (overlapping), 65535, and with `specials` now and then what the lanes cannot take: literal runs of 15..400 and
    of 70000 bytes, matches of 19..2000 bytes and of 200000 bytes, a token whose sequence leaves its 64-byte window."""
    rng = np.random.default_rng(seed)
    out = bytearray(rng.integers(0, 256, 64, dtype=np.uint8).tobytes())
    first = bytes(out)
    seqs = []
    pending_lit = first
    while len(out) < n_out - (300000 if specials else 4096):
        r = int(rng.integers(0, 1000))
        mlen = int(rng.integers(4, 19))
        hi = min(len(out), 65535)
        off = int(rng.integers(1, hi + 1))
        if r < 30: off = int(rng.integers(1, 4))
        elif r < 60: off = hi
        elif r < 200: off = int(rng.integers(1, min(hi, 300) + 1))
        if specials and r in (500, 501, 502): mlen = int(rng.integers(19, 2000))
        if specials and r == 503 and len(out) > (1 << 20): mlen = 200000
        seqs.append((pending_lit, mlen, off))
        src = len(out) - off
        for i in range(mlen): out.append(out[src + i])
        nl = int(rng.integers(1, 10))
        if specials and r in (600, 601, 602, 603): nl = int(rng.integers(15, 400))
        if specials and r == 604: nl = 70000
        if r in (700, 701): nl = 0
        pending_lit = rng.integers(0, 256, nl, dtype=np.uint8).tobytes()
        out += pending_lit
    tail = rng.integers(0, 256, 40, dtype=np.uint8).tobytes()
    out += tail
    body = b"".join(_lz4_seq(l, m, o) for l, m, o in seqs) + _lz4_seq(pending_lit + tail, 0, 0)
    return body, bytes(out)


@pytest.mark.gpu
def test_dense_payloads_in_big_independent_blocks(L):
    """Text-like payloads in big independent blocks without an index (what `lz4 -c` writes of text, test/Main.hs:33-36): the density probe sends them to the
    workgroup-per-block decoder whose waves take the payload's 64-byte windows in turn (decode_relay.cuh; LZ4F_MI355X_DENSE_MODE=1: a wave per block with an LDS
    ring, decode_ring.cuh; 2: a wave per block).  liblz4's bytes of text, and hand-made blocks with every kind of sequence (overlapping matches, offset 65535,
    long runs and long matches in between, tokens with length bytes), a stored block between them, block and content checksums; truncated room; and single-byte
    mutations get the oracle's verdicts."""
    import os, struct, torch
    from lz4_frame_conduit_amd.device import Engine
    rng = np.random.default_rng(5)
    frames = []
    text = datagen.synth_text(9 << 20, 31)
    for kw in (dict(bsid=7, indep=1), dict(bsid=6, indep=1, bck=1), dict(bsid=5, indep=1, cck=1)):
        frames.append(("text", kw, text.tobytes(), oracle.conduit_compress(text.tobytes(), oracle.mkprefs(**kw))))
    # hand-made: header of an empty frame with the same preferences, then blocks of our own
    hdr = oracle.conduit_compress(b"", oracle.mkprefs(bsid=7, indep=1))[:7]
    blocks = [_dense_block(4 << 20, 1, False), _dense_block(3 << 20, 2, True), _dense_block(4 << 20, 3, True), _dense_block(70000, 4, False)]
    noise = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    fr, data = bytearray(hdr), bytearray()
    for i, (body, out) in enumerate(blocks):
        fr += struct.pack("<I", len(body)) + body; data += out
        if i == 1: fr += struct.pack("<I", len(noise) | 0x80000000) + noise; data += noise
    fr += struct.pack("<I", 0)
    want, used = oracle.decompress_frame(bytes(fr), len(data) + 8)
    assert want == bytes(data) and used == len(fr)
    frames.append(("hand-made", dict(bsid=7, indep=1), bytes(data), bytes(fr)))
    for mode in ("0", "1", "2"):
        os.environ["LZ4F_MI355X_DENSE_MODE"] = mode
        L.lz4f_mi355x_release_engines()
        try:
            eng = Engine(0)
            for name, kw, data_b, frb in frames:
                out, used = gpu_decompress_frame(L, frb, len(data_b) + 8)
                assert used == len(frb) and out == data_b, (name, kw, mode)
                dev = torch.from_numpy(np.frombuffer(frb + bytes(32), dtype=np.uint8).copy()).cuda()
                room = len(data_b) + 16 if name == "text" else 5 * (4 << 20) + 16            # (short blocks in the middle: the device call wants every block's full room)
                back = torch.zeros(room, dtype=torch.uint8, device="cuda")
                eng.decompress_frame_async(dev, len(frb), back)
                r = eng.result()
                assert r.size == len(data_b) and r.consumed == len(frb) and back[:len(data_b)].cpu().numpy().tobytes() == data_b, (name, kw, mode)
                assert (int(r.flags) >> 12) & PATH["wave_per_block"], (name, kw, mode, hex(int(r.flags) >> 12))
                assert bool((int(r.flags) >> 12) & PATH["workgroup_per_block"]) == (mode != "2"), (name, kw, mode, hex(int(r.flags) >> 12))
                if name == "text": assert int(back[len(data_b):].max()) == 0, (name, kw, mode)
                for short in (1, 5000):                                                    # room that ends inside the last block
                    with pytest.raises(RuntimeError):
                        gpu_decompress_frame(L, frb, len(data_b) - short)
            eng.close()
            # mutations: the verdict (and, when accepted, the bytes) of the oracle
            for name, kw, data_b, base in (frames[0], frames[3]):
                cap = len(data_b) + 8
                diff = []
                for i in range(40):
                    pos = int(rng.integers(7, len(base))); x = int(rng.integers(1, 256))
                    bad = bytearray(base); bad[pos] ^= x; bad = bytes(bad)
                    try:
                        want, _ = oracle.decompress_frame(bad, cap); ov = "ok"
                    except oracle.OracleError as e:
                        want, ov = None, str(e)
                    try:
                        got, _ = gpu_decompress_frame(L, bad, cap); gv = "ok"
                    except RuntimeError as e:
                        got, gv = None, str(e).split(" | ")[0]
                    if (ov == "ok") != (gv == "ok") or (ov == "ok" and got != want): diff.append((pos, x, ov, gv))
                assert not diff, (name, mode, diff[:5])
        finally:
            os.environ.pop("LZ4F_MI355X_DENSE_MODE", None)
            L.lz4f_mi355x_release_engines()


def _real_text(n: int) -> bytes:
    """Real text instead of the synthetic kind: the image's own Python sources and headers, concatenated (the GPU box runs the same image)."""
    import os
    buf = bytearray()
    for root in ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include"):
        for dp, dn, fn in os.walk(root):
            for f in sorted(fn):
                if f.endswith((".py", ".h", ".hpp", ".txt", ".md", ".rst", ".json")):
                    try: buf += open(os.path.join(dp, f), "rb").read()
                    except OSError: pass
            if len(buf) >= n: break
        if len(buf) >= n: break
    return bytes(buf[:n])


@pytest.mark.gpu
def test_real_text_both_directions_and_ratio(L):
    """Real text (source code: longer matches and nearer repeats than the synthetic Zipf text, three times its ratio): what this library writes of it the oracle
    decodes, what liblz4 writes of it this library decodes - 4 MiB and 64 KiB independent blocks, 64 KiB linked - and the size stays within REAL_TEXT_RATIO_TOL of
    liblz4's.  That tolerance is what round 4 measured (11.5 % bigger; tools/real_text.py, NOTES_r4.md), not what is wanted: the test is there so that the number is
    looked at and does not get worse unseen."""
    REAL_TEXT_RATIO_TOL = 1.14
    data = _real_text(24 << 20)
    if len(data) < (8 << 20): pytest.skip("no text to speak of in this image")
    for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=1), dict(bsid=4, indep=0)):
        mine = gpu_compress_frame(L, data, prefs_of(kw))
        out, used = oracle.decompress_frame(mine, cap=len(data) + 64)
        assert out == data and used == len(mine), kw
        ref = oracle.conduit_compress(data, oracle.mkprefs(**kw))
        assert len(mine) <= len(ref) * REAL_TEXT_RATIO_TOL, (kw, len(mine), len(ref), len(mine) / len(ref))
        for fr in (ref, mine):
            back, used = gpu_decompress_frame(L, fr, len(data) + 8)
            assert back == data and used == len(fr), kw


@pytest.mark.gpu
def test_foreign_frames_never_write_past_capacity(L):
    """Foreign frames of big independent blocks go through the stretch-parallel self-index (decode_spx.cuh), whose scan learns what the
    PAYLOADS claim to decode to.  Those claims must never size a write: an output buffer one byte short, and a block whose last match
    makes it decode to block_size + k, are errors (liblz4: dstMaxSize_tooSmall / decompressionFailed) and the bytes behind the
    caller's capacity stay untouched - through the device-resident call and the host-pointer call."""
    import torch
    from lz4_frame_conduit_amd.device import Engine, DeviceCodecError
    rng = np.random.default_rng(5)
    GUARD = 1 << 20
    eng = Engine(0)

    def device_decode(frame: bytes, cap: int):
        dev = torch.from_numpy(np.frombuffer(frame + bytes(32), dtype=np.uint8).copy()).cuda()
        back = torch.full((cap + GUARD,), 0xA5, dtype=torch.uint8, device="cuda")
        try:
            eng.decompress_frame_async(dev, len(frame), back[:cap]); r = eng.result(); verdict = "ok"
        except DeviceCodecError as e:
            r, verdict = None, str(e)
        torch.cuda.synchronize()
        assert bool((back[cap:] == 0xA5).all()), "bytes behind the caller's capacity were written"
        return r, verdict, back

    # (1) liblz4's framing (-B7, -B6, -B5 independent) of long sequences, capacity one byte (and one block) short
    data = np.concatenate([datagen.synth50(9 << 20, 3), rng.integers(0, 256, 123457, dtype=np.uint8)])
    for kw in (dict(bsid=7, indep=1), dict(bsid=6, indep=1, bck=1), dict(bsid=5, indep=1)):
        fr = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))
        r, verdict, back = device_decode(fr, len(data))
        assert verdict == "ok" and r.size == len(data) and back[:len(data)].cpu().numpy().tobytes() == data.tobytes(), (kw, verdict)
        for short in (1, 70000, (4 << 20) + 5):
            r, verdict, _ = device_decode(fr, len(data) - short)
            assert verdict != "ok" and "dstMaxSize_tooSmall" in verdict, (kw, short, verdict)
            with pytest.raises(RuntimeError, match="dstMaxSize_tooSmall"):
                gpu_decompress_frame(L, fr, len(data) - short)

    # (2) a hand-made frame: four 256 KiB blocks of 1 KiB sequences (600 literals + a 424-byte copy of them), block `odd` has one match
    # that is k bytes too long - it decodes to 256 KiB + k
    bs = 256 << 10
    def block(extra: int) -> bytes:
        out = bytearray()
        nseq = bs // 1024
        for q in range(nseq - 1):
            out += _lz4_seq(rng.integers(0, 256, 600, dtype=np.uint8).tobytes(), 424 + (extra if q == nseq // 2 else 0), 600)
        out += _lz4_seq(rng.integers(0, 256, 600, dtype=np.uint8).tobytes(), 412, 600)
        out += _lz4_seq(rng.integers(0, 256, 12, dtype=np.uint8).tobytes(), 0, 0)
        return bytes(out)
    header = oracle.conduit_compress(b"", oracle.mkprefs(bsid=5, indep=1))[:7]
    def frame_of(blocks):
        f = bytearray(header)
        for b in blocks: f += len(b).to_bytes(4, "little") + b
        return bytes(f) + bytes(4)
    good = frame_of([block(0) for _ in range(4)])
    want, _ = oracle.decompress_frame(good, 4 * bs + 8)
    assert len(want) == 4 * bs
    r, verdict, back = device_decode(good, 4 * bs)
    assert verdict == "ok" and r.size == 4 * bs and back[:4 * bs].cpu().numpy().tobytes() == want
    for odd, k in ((1, 1), (1, 300), (3, 70000), (0, 5), (2, (4 << 20))):
        bad = frame_of([block(k if i == odd else 0) for i in range(4)])
        with pytest.raises(oracle.OracleError):
            oracle.decompress_frame(bad, 4 * bs + (8 << 20))
        for cap in (4 * bs, 4 * bs + k):
            r, verdict, _ = device_decode(bad, cap)
            assert verdict != "ok", (odd, k, cap)
        with pytest.raises(RuntimeError):
            gpu_decompress_frame(L, bad, 4 * bs + k)
    eng.close()


@pytest.mark.gpu
def test_streams_beyond_32_bits_device_resident(L):
    """The reference tests "more than 32-bit many Bytes" (test/Main.hs:80-84, :107-111: 5 GiB of 0x2A through compress and through
    decompress).  Here: 5 GiB of 0x2A and 5 GiB of synth50, each through ONE device-resident call each way - positions in the input,
    in the output and (for the stored case) in the frame all cross 2^32 inside a call - in the `lz4` tool's framing (4 MiB independent
    blocks), with and without the in-band trailer, and in the conduit's default framing (64 KiB linked blocks); the device compares
    all 5 GiB with the input, and the oracle decodes the frame's first blocks."""
    import torch
    from lz4_frame_conduit_amd.device import Engine, synth50_device
    n = 5 << 30
    eng = Engine(0)
    for kind in ("rep42", "synth50", "random"):
        if kind == "rep42": src = torch.full((n,), 42, dtype=torch.uint8, device="cuda")
        elif kind == "synth50": src = synth50_device(n, 4321, "cuda")
        else: src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")                  # stored blocks: the FRAME is beyond 2^32 too
        back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        for kw, inband in ((dict(bsid=7, indep=1), True), (dict(bsid=7, indep=1), False), (dict(), True)):
            if kind == "random" and not kw: continue
            p = prefs_of(kw)
            frame = torch.empty(eng.frame_bound_inband(n, p), dtype=torch.uint8, device="cuda")
            eng.compress_async(src, frame, p, inband=inband); r = eng.result()
            assert r.consumed == n and r.size > 0, (kind, kw, inband)
            back.fill_(0xA5)
            eng.decompress_frame_async(frame, int(r.size), back[:n]); r2 = eng.result()
            assert r2.size == n, (kind, kw, inband, r2.size)
            assert bool(torch.equal(back[:n], src)) and bool((back[n:] == 0xA5).all()), (kind, kw, inband)
            if kind == "random": assert r.size > (1 << 32) + (1 << 30)
            # the oracle on the frame's head: the first blocks decode to the input's first bytes
            head = frame[:min(int(r.size), 9 << 20)].cpu().numpy().tobytes()
            bs = 4 << 20 if kw else 64 << 10
            pos, got = 7, b""
            while len(got) < 2 * bs and pos + 4 <= len(head):
                w = int.from_bytes(head[pos:pos + 4], "little"); pos += 4
                if w == 0: break
                csz = w & 0x7FFFFFFF
                if pos + csz > len(head): break
                got += head[pos:pos + csz] if w >> 31 else oracle.decompress_block(head[pos:pos + csz], bs, got[-65536:] if not kw else b"")
                pos += csz
            assert len(got) >= bs and got == src[:len(got)].cpu().numpy().tobytes(), (kind, kw, inband)
            del frame
        del src, back
    eng.close()


@pytest.mark.gpu
def test_decode_path_by_input_class(L):
    """Which kernels a decompress call launches is a function of the call alone - the arguments, the frame's header and trailer,
    the payload's density where the frame has to be indexed here, the switches the engine was made with - and is reported in
    result.flags (LZ4F_MI355X_PATH_*).  One case per input class,
    run in two different orders on one engine and on a fresh engine: the same path every time, and the source's bytes."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    P = PATH
    s50, text = datagen.synth50(16 << 20, 21), datagen.synth_text(8 << 20, 22)
    def foreign(data, **kw): return oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))
    cases = {}      # name -> (kind, data, prefs kw, must have, must not have)
    # (big independent blocks without an index: long sequences -> the decoder cuts the blocks into stretches itself (decode_spx.cuh) and
    # runs the indexed kernels; dense payloads (a probe on the device decides) -> fused workgroups AND the wave-per-block decoder are
    # launched and one of them runs - text goes to the lanes)
    cases["foreign 4M independent"] = ("foreign", s50, dict(bsid=7, indep=1), P["self_index"] | P["indexed"] | P["fused"], P["table"] | P["trailer"] | P["parallel_walk"] | P["window"] | P["dropped"] | P["wave_per_block"])
    cases["foreign 1M independent"] = ("foreign", s50, dict(bsid=6, indep=1), P["self_index"] | P["indexed"] | P["fused"], P["table"] | P["trailer"] | P["parallel_walk"] | P["window"] | P["dropped"] | P["wave_per_block"])
    cases["foreign 4M independent text"] = ("foreign", text, dict(bsid=7, indep=1), P["fused"] | P["wave_per_block"], P["table"] | P["trailer"] | P["parallel_walk"] | P["indexed"] | P["window"])
    cases["foreign 64K independent"] = ("foreign", s50, dict(bsid=4, indep=1), P["parallel_walk"] | P["wave_per_block"], P["table"] | P["trailer"] | P["indexed"] | P["fused"])
    cases["foreign 64K independent, short"] = ("foreign", s50[:300000], dict(bsid=4, indep=1), P["wave_per_block"], P["parallel_walk"] | P["trailer"] | P["indexed"] | P["fused"])
    cases["foreign 64K linked"] = ("foreign", s50, dict(bsid=4, indep=0), P["self_index"] | P["indexed"] | P["window"] | P["fused"], P["table"] | P["trailer"] | P["wave_per_block"] | P["dropped"])
    cases["foreign 4M linked text"] = ("foreign", text, dict(bsid=7, indep=0), P["self_index"] | P["indexed"] | P["doubling"], P["table"] | P["trailer"] | P["hops"])
    cases["in-band 4M"] = ("inband", s50, dict(bsid=7, indep=1), P["trailer"] | P["indexed"] | P["fused"], P["table"] | P["parallel_walk"] | P["self_index"] | P["doubling"] | P["dropped"])
    cases["in-band 64K"] = ("inband", s50, dict(bsid=4, indep=1), P["trailer"] | P["wave_per_block"], P["table"] | P["parallel_walk"] | P["indexed"] | P["fused"])
    # (text has more sequences than an index of the recommended size holds: the compressor marks it unusable, the trailer carries the block list alone)
    cases["in-band 4M text"] = ("inband", text, dict(bsid=7, indep=1), P["trailer"] | P["fused"] | P["wave_per_block"], P["table"] | P["indexed"] | P["hops"] | P["self_index"] | P["parallel_walk"])
    cases["table + index 4M"] = ("indexed", s50, dict(bsid=7, indep=1), P["table"] | P["indexed"] | P["fused"], P["trailer"] | P["parallel_walk"] | P["self_index"] | P["dropped"])
    cases["table 4M"] = ("table", s50, dict(bsid=7, indep=1), P["table"] | P["self_index"] | P["indexed"] | P["fused"], P["trailer"] | P["parallel_walk"] | P["dropped"])
    made = {}
    def prepare(eng, name):
        kind, data, kw, _, _ = cases[name]
        src = torch.from_numpy(data).cuda()
        p = prefs_of(kw)
        if kind == "foreign":
            fr = foreign(data, **kw)
            return dict(src=src, frame=torch.from_numpy(np.frombuffer(fr, dtype=np.uint8).copy()).cuda(), size=len(fr))
        bs = 1 << (8 + 2 * kw["bsid"]); nb = (src.numel() + bs - 1) // bs
        frame = torch.empty(eng.frame_bound_inband(src.numel(), p), dtype=torch.uint8, device="cuda")
        if kind == "inband":
            eng.compress_async(src, frame, p, inband=True)
            return dict(src=src, frame=frame, size=int(eng.result().size))
        table, index = eng.new_table(nb), eng.new_index(src.numel(), p)
        eng.compress_async(src, frame, p, table, index)
        return dict(src=src, frame=frame, size=int(eng.result().size), table=table, index=index if kind == "indexed" else None, nb=nb, info=p.frameInfo)
    def run(eng, name):
        m = made[name]
        back = torch.zeros_like(m["src"])
        if "table" in m: eng.decompress_blocks_async(m["frame"], m["size"], back, m["table"], m["nb"], m["info"], m["index"])
        else: eng.decompress_frame_async(m["frame"], m["size"], back)
        r = eng.result()
        assert r.size == m["src"].numel() and torch.equal(back, m["src"]), name
        return int(r.flags) >> 12
    eng = Engine(0)
    for name in cases: made[name] = prepare(eng, name)
    names = list(cases)
    seen = {}
    for order in (names, names[::-1], names[3:] + names[:3]):
        for name in order:
            path = run(eng, name)
            _, _, _, must, must_not = cases[name]
            assert path & must == must and path & must_not == 0, (name, hex(path), hex(must), hex(must_not))
            assert seen.setdefault(name, path) == path, (name, hex(path), hex(seen[name]))       # not a function of what ran before
    eng.close()
    for name in names[::2]:                                                     # nor of the engine's age
        fresh = Engine(0)
        assert run(fresh, name) == seen[name], name
        fresh.close()


@pytest.mark.gpu
def test_linked_wait_budget_expires(L, monkeypatch):
    """A workgroup of a linked frame that waits for the one in front has a budget derived from the frame's size (engine.hip:
    wait_ticks).  With a budget of one tick every wait expires at once: the indexed kernels give up, say so
    (LZ4F_MI355X_PATH_INDEX_DROPPED), and the kernels behind them still produce liblz4's bytes."""
    import torch
    from lz4_frame_conduit_amd.device import Engine
    monkeypatch.setenv("LZ4F_MI355X_WAIT_TICKS", "1")
    eng = Engine(0)
    monkeypatch.delenv("LZ4F_MI355X_WAIT_TICKS")
    ref_eng = Engine(0)
    dropped = 0
    # a second block that is done in no time but needs the first one's last bytes (its own tail repeated: one match into the block
    # in front, then matches that read what that one will write): its workgroup has to wait for the first block's
    d0 = datagen.synth50(4 << 20, 24).tobytes()
    waits = d0 + d0[-2000:] * ((4 << 20) // 2000)
    for data, kw in ((waits, dict(bsid=7, indep=0)), (datagen.structured(24 << 20, 500), dict(bsid=7, indep=0)), (datagen.structured(6 << 20, 501), dict(bsid=4, indep=0)),
                     (datagen.synth50(12 << 20, 23).tobytes(), dict(bsid=6, indep=0))):
        fr = oracle.conduit_compress(data, oracle.mkprefs(**kw))
        dev = torch.from_numpy(np.frombuffer(fr, dtype=np.uint8).copy()).cuda()
        want = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        for en in (eng, ref_eng):
            back = torch.zeros_like(want)
            en.decompress_frame_async(dev, dev.numel(), back)
            r = en.result()
            assert r.size == want.numel() and torch.equal(back, want), kw
            if en is eng: dropped += ((r.flags >> 12) & PATH["dropped"]) != 0
            else: assert ((r.flags >> 12) & PATH["dropped"]) == 0, kw
    assert dropped >= 1
    eng.close(); ref_eng.close()


@pytest.mark.gpu
def test_device_path_content_checksum(L):
    """Frames with a content checksum on the device-pointer path (Frame format "Content checksum"; liblz4 writes it in
    LZ4F_compressEnd and checks it at the end of LZ4F_decompress).  Compress: the oracle (liblz4) accepts the frame, which it
    would not with a wrong word.  Decompress: liblz4's frames decode; a flipped checksum bit, or a flipped bit in a stored
    block (nothing else would notice), gives ERROR_contentChecksum_invalid.  Also with block checksums (the 4-wave kernel for
    few blocks, the wave-per-block one for many) and with the in-band trailer behind the checksum word."""
    import torch
    from lz4_frame_conduit_amd.device import Engine, DeviceCodecError
    eng = Engine(0)
    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, 3 << 20, dtype=np.uint8)
    inputs = [("synth50", np.concatenate([datagen.synth50(9 << 20, 31), noise[:12345]])), ("noise", noise), ("empty", np.zeros(0, dtype=np.uint8)), ("tiny", noise[:13].copy()),
              ("text", datagen.synth_text(5 << 20, 6))]
    for name, data in inputs:
        for kw in (dict(bsid=7, indep=1, cck=1), dict(bsid=4, indep=1, cck=1, bck=1), dict(bsid=6, indep=0, cck=1), dict(bsid=7, indep=1, cck=1, bck=1)):
            p = prefs_of(kw)
            src = torch.from_numpy(data).cuda() if len(data) else torch.zeros(0, dtype=torch.uint8, device="cuda")
            for inband in (False, True):
                if inband and len(data) == 0: continue
                frame = torch.empty(eng.frame_bound_inband(len(data), p) + 64, dtype=torch.uint8, device="cuda")
                eng.compress_async(src, frame, p, inband=inband)
                r = eng.result()
                stream = frame[:r.size].cpu().numpy().tobytes()
                out, used = oracle.decompress_frame(stream, cap=len(data) + 64)          # liblz4 checks the word
                assert out == data.tobytes() and (used == len(stream)) == (not inband), (name, kw, inband)
                back = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
                eng.decompress_frame_async(frame, int(r.size), back)
                r2 = eng.result()
                assert r2.size == len(data) and r2.consumed == used and back[:len(data)].cpu().numpy().tobytes() == data.tobytes(), (name, kw, inband)
            # liblz4's own frame, then the same frame with a wrong checksum word
            ref = oracle.conduit_compress(data.tobytes(), oracle.mkprefs(**kw))
            for flip in (None, len(ref) - 2):
                bad = bytearray(ref)
                if flip is not None: bad[flip] ^= 0x10
                dev = torch.from_numpy(np.frombuffer(bytes(bad) + bytes(32), dtype=np.uint8).copy()).cuda()
                back = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
                eng.decompress_frame_async(dev, len(bad), back)
                if flip is None:
                    r3 = eng.result()
                    assert r3.size == len(data) and r3.consumed == len(ref) and back[:len(data)].cpu().numpy().tobytes() == data.tobytes(), (name, kw)
                else:
                    with pytest.raises(DeviceCodecError, match="contentChecksum_invalid"): eng.result()
    # a flipped bit inside a stored block of a frame without block checksums: only the content checksum can tell
    ref = bytearray(oracle.conduit_compress(noise.tobytes(), oracle.mkprefs(bsid=5, indep=1, cck=1)))
    ref[len(ref) // 2 + 1000] ^= 1                                           # (inside block 6's stored payload)
    dev = torch.from_numpy(np.frombuffer(bytes(ref), dtype=np.uint8).copy()).cuda()
    back = torch.zeros(len(noise) + 16, dtype=torch.uint8, device="cuda")
    eng.decompress_frame_async(dev, len(ref), back)
    with pytest.raises(DeviceCodecError, match="contentChecksum_invalid"): eng.result()
    eng.close()


@pytest.mark.gpu
def test_bulk_calls_over_several_gpus():
    """lz4f_mi355x_use_devices: the host-pointer bulk calls deal their slabs over that many GPUs (blocks of an independent-block
    frame need nothing from each other: no collective, the host puts the output in order).  Runs in a child process with
    LZ4F_MI355X_LOGICAL_DEVICES=3, so that on a one-GPU box the deal over devices, the turn order of the slabs and the engine sets
    per device run exactly as they would on three GPUs (logical device d = physical d mod visible); with several GPUs visible the
    same code spreads over them.  liblz4 (the oracle) must decode the frames, the library must give the input back."""
    import subprocess
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import oracle
from lz4_frame_conduit_amd import _ffi, datagen
sys.path.insert(0, %r)
from test_gpu_parity import gpu_compress_frame, gpu_decompress_frame, prefs_of
L = _ffi.lib()
have = torch.cuda.device_count()
assert L.lz4f_mi355x_device_count() == have >= 1
n = max(3, have)
assert L.LZ4F_isError(L.lz4f_mi355x_use_devices(0)) and L.LZ4F_isError(L.lz4f_mi355x_use_devices(n + 1))
assert L.lz4f_mi355x_use_devices(1) == 0
data = datagen.synth50(300 << 20, 41).tobytes()
one = {}
for devs in (1, n, 2):
    assert L.lz4f_mi355x_use_devices(devs) == 0
    for kw in (dict(bsid=7, indep=1), dict(bsid=4, indep=1, bck=1), dict(bsid=4, indep=0)):
        frame = gpu_compress_frame(L, data, prefs_of(kw))
        out, used = oracle.decompress_frame(frame, cap=len(data) + 64)
        assert used == len(frame) and out == data, (devs, kw)
        back, used2 = gpu_decompress_frame(L, frame, len(data) + 8)
        assert used2 == len(frame) and back == data, (devs, kw)
        one.setdefault(str(kw), len(frame))
        assert abs(len(frame) - one[str(kw)]) <= one[str(kw)] // 200, (devs, kw, len(frame), one[str(kw)])      # (same slabs, same blocks: the size can only differ by the encoder's races)
L.lz4f_mi355x_use_devices(1)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ); env["LZ4F_MI355X_LOGICAL_DEVICES"] = "3"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-300:], r.stderr[-1500:])
