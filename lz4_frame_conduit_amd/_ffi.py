"""ctypes binding of liblz4f_mi355x.so (the C ABI in include/lz4f_mi355x.h).

Fails loudly when the library is missing: there is no Python / CPU fallback for the codec.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblz4f_mi355x.so")
if os.environ.get("LZ4F_MI355X_LIB"):         # development: a variant build of the same library (tools/ab_build.sh), for A-B measurements
    LIB_PATH = os.environ["LZ4F_MI355X_LIB"]

c_size_t, c_void_p = ctypes.c_size_t, ctypes.c_void_p


class FrameInfo(ctypes.Structure):          # LZ4F_frameInfo_t -- CTypes.hsc:155-199
    _fields_ = [("blockSizeID", ctypes.c_uint32), ("blockMode", ctypes.c_uint32), ("contentChecksumFlag", ctypes.c_uint32),
                ("frameType", ctypes.c_uint32), ("contentSize", ctypes.c_uint64), ("dictID", ctypes.c_uint32),
                ("blockChecksumFlag", ctypes.c_uint32)]


class Preferences(ctypes.Structure):        # LZ4F_preferences_t -- CTypes.hsc:202-232
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int32), ("autoFlush", ctypes.c_uint32),
                ("favorDecSpeed", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 3)]


class Result(ctypes.Structure):             # lz4f_mi355x_result
    _fields_ = [("size", ctypes.c_uint64), ("consumed", ctypes.c_uint64), ("status", ctypes.c_uint32), ("n_blocks", ctypes.c_uint32),
                ("first_bad_block", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


class Block(ctypes.Structure):              # lz4f_mi355x_block
    _fields_ = [("src_off", ctypes.c_uint64), ("dst_off", ctypes.c_uint64), ("word", ctypes.c_uint32), ("dst_size", ctypes.c_uint32)]


AWAIT_FN = ctypes.CFUNCTYPE(c_size_t, c_void_p, ctypes.POINTER(c_void_p))
YIELD_FN = ctypes.CFUNCTYPE(None, c_void_p, c_void_p, c_size_t)

PP = ctypes.POINTER(Preferences)
_SIGS = {
    # PART 1
    "LZ4F_isError": (ctypes.c_uint, [c_size_t]), "LZ4F_getErrorName": (ctypes.c_char_p, [c_size_t]), "LZ4F_getVersion": (ctypes.c_uint, []),
    "LZ4F_createCompressionContext": (c_size_t, [ctypes.POINTER(c_void_p), ctypes.c_uint]), "LZ4F_freeCompressionContext": (c_size_t, [c_void_p]),
    "LZ4F_compressBegin": (c_size_t, [c_void_p, c_void_p, c_size_t, PP]), "LZ4F_compressBound": (c_size_t, [c_size_t, PP]),
    "LZ4F_compressUpdate": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "LZ4F_flush": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p]), "LZ4F_compressEnd": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "LZ4F_createDecompressionContext": (c_size_t, [ctypes.POINTER(c_void_p), ctypes.c_uint]), "LZ4F_freeDecompressionContext": (c_size_t, [c_void_p]),
    "LZ4F_resetDecompressionContext": (None, [c_void_p]), "LZ4F_headerSize": (c_size_t, [c_void_p, c_size_t]),
    "LZ4F_getFrameInfo": (c_size_t, [c_void_p, ctypes.POINTER(FrameInfo), c_void_p, ctypes.POINTER(c_size_t)]),
    "LZ4F_decompress": (c_size_t, [c_void_p, c_void_p, ctypes.POINTER(c_size_t), c_void_p, ctypes.POINTER(c_size_t), c_void_p]),
    "haskell_lz4_freeCompressionContext": (None, [ctypes.POINTER(c_void_p)]), "haskell_lz4_freeDecompressionContext": (None, [ctypes.POINTER(c_void_p)]),
    # prefixed aliases
    "lz4f_mi355x_isError": (ctypes.c_uint, [c_size_t]), "lz4f_mi355x_getErrorName": (ctypes.c_char_p, [c_size_t]),
    "lz4f_mi355x_createCompressionContext": (c_size_t, [ctypes.POINTER(c_void_p), ctypes.c_uint]), "lz4f_mi355x_freeCompressionContext": (c_size_t, [c_void_p]),
    "lz4f_mi355x_compressBegin": (c_size_t, [c_void_p, c_void_p, c_size_t, PP]), "lz4f_mi355x_compressBound": (c_size_t, [c_size_t, PP]),
    "lz4f_mi355x_compressUpdate": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "lz4f_mi355x_flush": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p]), "lz4f_mi355x_compressEnd": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "lz4f_mi355x_createDecompressionContext": (c_size_t, [ctypes.POINTER(c_void_p), ctypes.c_uint]), "lz4f_mi355x_freeDecompressionContext": (c_size_t, [c_void_p]),
    "lz4f_mi355x_getFrameInfo": (c_size_t, [c_void_p, ctypes.POINTER(FrameInfo), c_void_p, ctypes.POINTER(c_size_t)]),
    "lz4f_mi355x_decompress": (c_size_t, [c_void_p, c_void_p, ctypes.POINTER(c_size_t), c_void_p, ctypes.POINTER(c_size_t), c_void_p]),
    # PART 2
    "lz4f_mi355x_last_error": (ctypes.c_char_p, []), "lz4f_mi355x_device_count": (ctypes.c_int, []), "lz4f_mi355x_set_device": (c_size_t, [ctypes.c_int]),
    "lz4f_mi355x_release_engines": (None, []),
    "lz4f_mi355x_decompressFrameTo": (c_size_t, [YIELD_FN, c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_size_t)]),
    "lz4f_mi355x_use_devices": (c_size_t, [ctypes.c_int]),
    "lz4f_mi355x_host_alloc": (c_void_p, [c_size_t]), "lz4f_mi355x_host_free": (None, [c_void_p]),
    "lz4f_mi355x_compressFrameBound": (c_size_t, [c_size_t, PP]),
    "lz4f_mi355x_compressFrame": (c_size_t, [c_void_p, c_size_t, c_void_p, c_size_t, PP]),
    "lz4f_mi355x_decompressFrame": (c_size_t, [c_void_p, c_size_t, c_void_p, c_size_t, ctypes.POINTER(c_size_t)]),
    "lz4f_mi355x_engine_create": (c_size_t, [ctypes.POINTER(c_void_p), ctypes.c_int, c_void_p, ctypes.c_int]), "lz4f_mi355x_engine_free": (c_size_t, [c_void_p]),
    "lz4f_mi355x_engine_stream": (c_void_p, [c_void_p]),
    "lz4f_mi355x_engine_set_deterministic": (c_size_t, [c_void_p, ctypes.c_int]),
    "lz4f_mi355x_engine_set_timing": (c_size_t, [c_void_p, ctypes.c_int]), "lz4f_mi355x_engine_get_timing": (c_size_t, [c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "lz4f_mi355x_engine_get_timing_n": (c_size_t, [c_void_p, ctypes.POINTER(ctypes.c_float), c_size_t]), "lz4f_mi355x_dev_workspace_size": (c_size_t, [c_size_t, PP]),
    "lz4f_mi355x_dev_compressFrame": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, PP, c_void_p, c_void_p]),
    "lz4f_mi355x_dev_decompressFrame": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "lz4f_mi355x_dev_index_size": (c_size_t, [c_size_t, PP]),
    "lz4f_mi355x_trailer_bound": (c_size_t, [c_size_t, PP]),
    "lz4f_mi355x_dev_compressFrameIndexed": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, PP, c_void_p, c_void_p, c_void_p, c_size_t]),
    "lz4f_mi355x_dev_decompressBlocksIndexed": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, ctypes.c_uint32, ctypes.POINTER(FrameInfo), c_void_p, c_size_t, c_void_p]),
    "lz4f_mi355x_dev_decompressBlocks": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, ctypes.c_uint32, ctypes.POINTER(FrameInfo), c_void_p]),
    "lz4f_mi355x_dev_xxh32": (c_size_t, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_uint32, c_void_p]),
    "lz4f_mi355x_conduit_compress": (ctypes.c_int, [c_size_t, PP, AWAIT_FN, YIELD_FN, c_void_p, ctypes.c_char_p, c_size_t]),
    "lz4f_mi355x_conduit_compress_yield_immediately": (ctypes.c_int, [PP, AWAIT_FN, YIELD_FN, c_void_p, ctypes.c_char_p, c_size_t]),
    "lz4f_mi355x_conduit_decompress": (ctypes.c_int, [AWAIT_FN, YIELD_FN, c_void_p, ctypes.c_char_p, c_size_t]),
    "lz4f_mi355x_conduit_compress_batched": (ctypes.c_int, [c_size_t, PP, AWAIT_FN, YIELD_FN, c_void_p, ctypes.c_char_p, c_size_t]),
    "lz4f_mi355x_conduit_compress_batched_listed": (ctypes.c_int, [c_size_t, PP, AWAIT_FN, YIELD_FN, c_void_p, ctypes.c_char_p, c_size_t]),
    "lz4f_mi355x_blockListSize": (c_size_t, [c_void_p, c_size_t]), "lz4f_mi355x_appendBlockList": (c_size_t, [c_void_p, c_size_t, c_size_t]),
    "lz4f_mi355x_conduit_decompress_batched": (ctypes.c_int, [AWAIT_FN, YIELD_FN, c_void_p, ctypes.c_char_p, c_size_t]),
    "lz4f_mi355x_conduit_decompress_batched_bounded": (ctypes.c_int, [c_size_t, AWAIT_FN, YIELD_FN, c_void_p, ctypes.c_char_p, c_size_t]),
    "lz4f_mi355x_fdec_create": (c_size_t, [ctypes.POINTER(c_void_p), c_void_p, c_size_t, ctypes.POINTER(FrameInfo)]),
    "lz4f_mi355x_fdec_blocks": (c_size_t, [c_void_p, YIELD_FN, c_void_p, c_void_p, c_size_t]),
    "lz4f_mi355x_fdec_end": (c_size_t, [c_void_p, c_void_p, c_size_t]), "lz4f_mi355x_fdec_free": (None, [c_void_p]),
}
DECLARED_SYMBOLS = tuple(_SIGS)

_LIB = None


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> lz4_frame_conduit_amd/liblz4f_mi355x.so (in-tree)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


def _share_hip_runtime_with_torch() -> None:
    """PyTorch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, the same soname as
    /opt/rocm's).  Two HIP runtimes in one process fight over the device ("no ROCm-capable device"),
    so when torch is installed its copy is mapped first and liblz4f_mi355x.so binds to it by soname.
    A non-Python host (the Haskell conduit) simply gets /opt/rocm's runtime through the RUNPATH."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(p):
                ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("liblz4f_mi355x.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)          # AttributeError = a declared symbol is not exported
            f.restype, f.argtypes = res, args
        _LIB = L
    return _LIB
