"""Multi-GPU bookkeeping for independent-block frames: blocks are dealt
round-robin to ranks, every rank encodes/decodes its own blocks with no data-path collective; the only
things exchanged are the per-block payload sizes (4 B x nblocks) and, at the end, max-over-ranks time.
Works with any torch.distributed backend (nccl = RCCL on the GPU box, gloo in the CPU tests)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def block_owner(i: int, world: int) -> int:
    return i % world


def my_blocks(n_blocks: int, rank: int, world: int) -> range:
    return range(rank, n_blocks, world)


def barrier_all(device: "torch.device | None" = None) -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: "torch.device | str" = "cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: "torch.device | str" = "cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def exchange_block_words(local: Dict[int, int], n_blocks: int, device: "torch.device | str" = "cpu") -> List[int]:
    """Every rank contributes the size words of the blocks it owns; all ranks get the full table.
    (One all-reduce of n_blocks int64 -- 8 B x nblocks, the only cross-device data of the compress path.)"""
    t = torch.zeros(n_blocks, dtype=torch.int64, device=device)
    for i, w in local.items():
        t[i] = w
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def frame_offsets(header_size: int, words: Sequence[int], block_checksum: bool) -> List[int]:
    """Byte offset of every block's size word in the assembled frame (exclusive scan)."""
    offs, pos = [], header_size
    for w in words:
        offs.append(pos)
        pos += 4 + (w & 0x7FFFFFFF) + (4 if block_checksum else 0)
    offs.append(pos)          # EndMark position
    return offs


def assemble_frame(header: bytes, words: Sequence[int], payloads: Sequence[bytes], checksums: "Sequence[int] | None" = None) -> bytes:
    """In-order emit on the host: header, then (size word, payload, [checksum]) per block, then EndMark."""
    out = [header]
    for i, (w, p) in enumerate(zip(words, payloads)):
        assert len(p) == (w & 0x7FFFFFFF)
        out.append(int(w).to_bytes(4, "little")); out.append(p)
        if checksums is not None:
            out.append(int(checksums[i]).to_bytes(4, "little"))
    out.append(b"\x00\x00\x00\x00")
    return b"".join(out)


def contiguous_blocks(n_blocks: int, rank: int, world: int) -> "tuple[int, int]":
    """[lo, hi): rank's contiguous run of blocks (runs of K blocks per GPU keep the copies large)."""
    per, extra = divmod(n_blocks, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def all_gather_sizes(mine: int, device: "torch.device | str" = "cpu") -> List[int]:
    """Every rank's byte count, on every rank (8 B x world: with the block words the only numbers that cross ranks)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(mine)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=device)
    t[dist.get_rank()] = int(mine)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
