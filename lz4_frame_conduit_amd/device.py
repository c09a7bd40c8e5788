"""Device-resident bulk path on torch tensors (torch = device memory + streams only).

`Engine` wraps lz4f_mi355x_engine; tensors are uint8 CUDA(HIP) tensors whose data_ptr() goes straight
into the C ABI.  Work is enqueued on torch's current stream, so torch.cuda.Event timings bracket it.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _ffi
from ._ffi import Block, FrameInfo, Preferences, Result


class DeviceCodecError(Exception):
    pass


def _chk(L, r):
    if L.LZ4F_isError(r):
        raise DeviceCodecError("%s (%s)" % (L.LZ4F_getErrorName(r).decode(), L.lz4f_mi355x_last_error().decode()))
    return r


class Engine:
    def __init__(self, device: int = 0, stream: "torch.cuda.Stream | None" = None):
        self.L = _ffi.lib()
        self.device = device
        torch.cuda.set_device(device)
        self.stream = stream if stream is not None else torch.cuda.current_stream(device)
        h = ctypes.c_void_p()
        _chk(self.L, self.L.lz4f_mi355x_engine_create(ctypes.byref(h), device, ctypes.c_void_p(self.stream.cuda_stream), 1))
        self.h = h
        self._res = torch.zeros(32, dtype=torch.uint8, device="cuda:%d" % device)
        self._res_host = torch.zeros(32, dtype=torch.uint8).pin_memory()      # the result record comes back by DMA into page-locked memory: one wait, no staging

    def close(self):
        if self.h:
            self.L.lz4f_mi355x_engine_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    TIMING_SLOTS = ("find_matches", "layout", "emit", "xxh32_write", "walk", "xxh32_verify", "decode", "finish", "decode_parse", "decode_copy", "compress_total", "decompress_total")

    def set_deterministic(self, on: bool):
        """Equal input -> equal bytes (one wave per workgroup parses, in order); about a tenth of the match finder's speed."""
        _chk(self.L, self.L.lz4f_mi355x_engine_set_deterministic(self.h, 1 if on else 0))

    def set_timing(self, on: bool):
        _chk(self.L, self.L.lz4f_mi355x_engine_set_timing(self.h, 1 if on else 0))

    def get_timing(self) -> dict:
        ms = (ctypes.c_float * len(self.TIMING_SLOTS))()
        _chk(self.L, self.L.lz4f_mi355x_engine_get_timing_n(self.h, ms, len(self.TIMING_SLOTS)))
        return dict(zip(self.TIMING_SLOTS, [float(x) for x in ms]))

    # -- helpers
    def _result(self) -> Result:
        with torch.cuda.stream(self.stream):
            self._res_host.copy_(self._res, non_blocking=True)
        self.stream.synchronize()
        return Result.from_buffer_copy(self._res_host.numpy().tobytes())

    def frame_bound(self, n: int, prefs: Preferences) -> int:
        return _chk(self.L, self.L.lz4f_mi355x_compressFrameBound(n, ctypes.byref(prefs)))

    INBAND = (1 << 64) - 1

    def frame_bound_inband(self, n: int, prefs: Preferences) -> int:
        """Room for a frame with its trailer (block list + sequence index in a skippable frame behind it)."""
        return self.frame_bound(n, prefs) + int(self.L.lz4f_mi355x_trailer_bound(n, ctypes.byref(prefs)))

    def compress_async(self, src: torch.Tensor, dst: torch.Tensor, prefs: Preferences, table: "torch.Tensor | None" = None,
                       index: "torch.Tensor | None" = None, inband: bool = False):
        """Enqueue src -> one frame in dst.  Returns nothing; call result() after a sync.
        With `index` (new_index) the compressor also leaves its sequence index there for decompress_blocks_async.
        With `inband` the index and the block list go into the stream itself (a skippable frame behind the LZ4 frame, counted in
        result().size): decompress_frame_async finds them there."""
        assert src.dtype == torch.uint8 and dst.dtype == torch.uint8 and src.is_cuda and dst.is_cuda
        if inband:
            _chk(self.L, self.L.lz4f_mi355x_dev_compressFrameIndexed(self.h, dst.data_ptr(), dst.numel(), src.data_ptr(), src.numel(), ctypes.byref(prefs),
                                                                    self._res.data_ptr(), table.data_ptr() if table is not None else None, None, self.INBAND))
            return
        if index is not None:
            assert table is not None and index.dtype == torch.uint8 and index.is_cuda
            _chk(self.L, self.L.lz4f_mi355x_dev_compressFrameIndexed(self.h, dst.data_ptr(), dst.numel(), src.data_ptr(), src.numel(), ctypes.byref(prefs),
                                                                    self._res.data_ptr(), table.data_ptr(), index.data_ptr(), index.numel()))
            return
        _chk(self.L, self.L.lz4f_mi355x_dev_compressFrame(self.h, dst.data_ptr(), dst.numel(), src.data_ptr(), src.numel(), ctypes.byref(prefs),
                                                         self._res.data_ptr(), table.data_ptr() if table is not None else None))

    def decompress_blocks_async(self, frame: torch.Tensor, frame_len: int, dst: torch.Tensor, table: torch.Tensor, n_blocks: int, info: FrameInfo,
                                index: "torch.Tensor | None" = None):
        if index is not None:
            _chk(self.L, self.L.lz4f_mi355x_dev_decompressBlocksIndexed(self.h, dst.data_ptr(), dst.numel(), frame.data_ptr(), frame_len, table.data_ptr(),
                                                                       n_blocks, ctypes.byref(info), index.data_ptr(), index.numel(), self._res.data_ptr()))
            return
        _chk(self.L, self.L.lz4f_mi355x_dev_decompressBlocks(self.h, dst.data_ptr(), dst.numel(), frame.data_ptr(), frame_len, table.data_ptr(),
                                                            n_blocks, ctypes.byref(info), self._res.data_ptr()))

    def decompress_frame_async(self, frame: torch.Tensor, frame_len: int, dst: torch.Tensor):
        _chk(self.L, self.L.lz4f_mi355x_dev_decompressFrame(self.h, dst.data_ptr(), dst.numel(), frame.data_ptr(), frame_len, self._res.data_ptr()))

    def result(self) -> Result:
        r = self._result()                                   # (waits for the stream: everything enqueued so far, then the record's copy)
        if r.status != 0:
            raise DeviceCodecError("%s at block %d" % (self.L.LZ4F_getErrorName((1 << 64) - r.status).decode(), r.first_bad_block))
        return r

    def new_table(self, n_blocks: int) -> torch.Tensor:
        return torch.zeros((n_blocks + 1) * ctypes.sizeof(Block), dtype=torch.uint8, device="cuda:%d" % self.device)

    def index_size(self, n: int, prefs: Preferences) -> int:
        return int(self.L.lz4f_mi355x_dev_index_size(n, ctypes.byref(prefs)))

    def new_index(self, n: int, prefs: Preferences) -> torch.Tensor:
        return torch.zeros(self.index_size(n, prefs), dtype=torch.uint8, device="cuda:%d" % self.device)

    def xxh32(self, base: torch.Tensor, offs: np.ndarray, lens: np.ndarray) -> np.ndarray:
        dev = "cuda:%d" % self.device
        o = torch.from_numpy(np.ascontiguousarray(offs, dtype=np.uint64).view(np.int64)).to(dev)
        l = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.uint32).view(np.int32)).to(dev)
        out = torch.zeros(len(lens), dtype=torch.int32, device=dev)
        _chk(self.L, self.L.lz4f_mi355x_dev_xxh32(self.h, base.data_ptr(), o.data_ptr(), l.data_ptr(), len(lens), out.data_ptr()))
        self.stream.synchronize()
        return out.cpu().numpy().view(np.uint32)


def synth50_device(n: int, seed: int, device: str = "cuda:0") -> torch.Tensor:
    """synth50 recipe (datagen.synth50) generated in HBM with torch's generator: 512-byte rows, even rows
    random, odd row r = copy of even row r-(2k+1), k in [1,60).  Same structure as the numpy version,
    different RNG stream (the numpy one is the canonical input of the parity tests)."""
    assert n % 1024 == 0
    g = torch.Generator(device=device); g.manual_seed(seed)
    rows = n // 512
    a = torch.randint(0, 256, (rows, 512), dtype=torch.uint8, device=device, generator=g)
    odd = torch.arange(1, rows, 2, device=device)
    back = torch.randint(1, 60, (odd.numel(),), device=device, generator=g) * 2 + 1
    src = torch.clamp(odd - back, min=0)
    src = src - (src % 2)
    a[odd] = a[src]
    return a.reshape(-1)
