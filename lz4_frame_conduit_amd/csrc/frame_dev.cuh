// frame_dev.cuh -- device-side frame plumbing around the block kernels:
//   k_xxh32_blocks   XXH32 of each block payload (row a5; block checksums, Frame format "BlockChecksum")
//   k_walk_frame     header validation + the serial walk over the block size words (row a3's
//                    getBlockHeader stage) -> block table
//   k_decode_blocks  one wave per table entry (decode.cuh) / one wave for a whole linked frame
//   k_finish_decode  totals, first error, and the (rare) compaction when a non-final block is short
#pragma once
#include "common.cuh"
#include "decode.cuh"
#include "decode_2k.cuh"
#include "decode_fused.cuh"
#include "decode_linked.cuh"
#include "encode.cuh"
#include "decode_indexed.cuh"

namespace lz4f {

// ------------------------------- XXH32 ----------------------------------------------------------
// XXH32 has four serial accumulator chains (acc = rotl(acc + w*P2, 13) * P1) per 16-byte stripe and no
// way to combine partial states, so one payload is one dependent chain.  Mapping: one wave per
// block; the 64 lanes fetch 1 KiB per step (16 B per lane, coalesced) and do the w*P2 products in
// parallel; the four chains then run on the scalar unit, fed by v_readlane.
__device__ __forceinline__ uint32_t xxh32_finish(uint32_t h, const uint8_t* p, uint32_t rem)
{
    while (rem >= 4) { h = rotl32(h + ld32(p) * XP3, 17) * XP4; p += 4; rem -= 4; }
    while (rem > 0) { h = rotl32(h + (*p) * XP5, 11) * XP1; p++; rem--; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint32_t wave_xxh32(const uint8_t* __restrict__ p, uint32_t len)
{
    const uint32_t lane = lane_id();
    uint32_t h;
    uint32_t done = 0;
    if (len >= 16) {
        uint32_t v1 = XP1 + XP2, v2 = XP2, v3 = 0, v4 = 0u - XP1;
        const uint32_t nstripes = len >> 4;
        for (uint32_t s0 = 0; s0 < nstripes; s0 += WAVE) {
            const uint32_t n = (nstripes - s0 < WAVE) ? nstripes - s0 : WAVE;     // uniform
            uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
            if (lane < n) {
                const b16_ua w = *(const b16_ua*)(p + (size_t)(s0 + lane) * 16);
                m0 = w.a * XP2; m1 = w.b * XP2; m2 = w.c * XP2; m3 = w.d * XP2;
            }
            if (n == WAVE) {
#pragma unroll
                for (int i = 0; i < WAVE; i++) {
                    v1 = rotl32(v1 + __builtin_amdgcn_readlane(m0, i), 13) * XP1;
                    v2 = rotl32(v2 + __builtin_amdgcn_readlane(m1, i), 13) * XP1;
                    v3 = rotl32(v3 + __builtin_amdgcn_readlane(m2, i), 13) * XP1;
                    v4 = rotl32(v4 + __builtin_amdgcn_readlane(m3, i), 13) * XP1;
                }
            } else {
                for (uint32_t i = 0; i < n; i++) {
                    v1 = rotl32(v1 + __builtin_amdgcn_readlane(m0, i), 13) * XP1;
                    v2 = rotl32(v2 + __builtin_amdgcn_readlane(m1, i), 13) * XP1;
                    v3 = rotl32(v3 + __builtin_amdgcn_readlane(m2, i), 13) * XP1;
                    v4 = rotl32(v4 + __builtin_amdgcn_readlane(m3, i), 13) * XP1;
                }
            }
        }
        done = nstripes << 4;
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = XP5;
    }
    h += len;
    return xxh32_finish(h, p + done, len - done);
}

// off/len given explicitly (lz4f_mi355x_dev_xxh32)
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_xxh32_ranges(const uint8_t* __restrict__ base, const uint64_t* __restrict__ off,
                                                                    const uint32_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    if (b >= n) return;
    const uint32_t h = wave_xxh32(base + off[b], len[b]);
    if (lane_id() == 0) out[b] = h;
}

// mode 0: write the checksum word after each payload of the block table (compress)
// mode 1: compare with the stored word; on mismatch flag the block (decompress)
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_xxh32_blocks(uint8_t* frame, BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                                    uint32_t n_max, uint32_t mode, uint32_t* __restrict__ bad)
{
    const uint32_t b = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    const uint32_t n = res ? res->n_blocks : n_max;
    if (b >= n || b >= n_max) return;
    if (res && res->status != ST_OK) return;
    const uint64_t off = table[b].src_off;
    const uint32_t len = table[b].word & 0x7FFFFFFFu;
    const uint32_t h = wave_xxh32(frame + off, len);
    if (lane_id() == 0) {
        uint8_t* c = frame + off + len;
        if (mode == 0) { c[0] = (uint8_t)h; c[1] = (uint8_t)(h >> 8); c[2] = (uint8_t)(h >> 16); c[3] = (uint8_t)(h >> 24); }
        else {
            const uint32_t stored = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            if (stored != h) atomicMin(bad, b);
        }
    }
}

// ------------------------------- frame walk -----------------------------------------------------
__device__ __forceinline__ uint32_t rd32_any(const uint8_t* p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ uint32_t xxh32_small(const uint8_t* p, uint32_t len)   // len < 16: header descriptor
{
    uint32_t h = XP5 + len;
    while (len >= 4) { h = rotl32(h + rd32_any(p) * XP3, 17) * XP4; p += 4; len -= 4; }
    while (len > 0) { h = rotl32(h + (*p) * XP5, 11) * XP1; p++; len--; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}

// Single thread: the walk is a pointer chase (each size word's position depends on all earlier
// ones).  Same validation order as the oracle's orc_decompress_frame.
__global__ void k_walk_frame(const uint8_t* __restrict__ frame, uint64_t frame_cap, uint64_t dst_cap,
                             BlockOut* __restrict__ table, uint32_t table_cap, ResultRec* __restrict__ res)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ResultRec r; r.size = 0; r.consumed = 0; r.status = ST_OK; r.n_blocks = 0; r.first_bad_block = 0xFFFFFFFFu; r.flags = 0;
    auto fail = [&](uint32_t st) { r.status = st; *res = r; };
    if (frame_cap < 7) return fail(12);                              // frameHeader_incomplete
    const uint32_t magic = rd32_any(frame);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                      // skippable frame: no output
        if (frame_cap < 8) return fail(12);
        const uint64_t sz = rd32_any(frame + 4);
        if (frame_cap < 8 + sz) return fail(12);
        r.consumed = 8 + sz; r.flags = 0x100; *res = r; return;
    }
    if (magic != 0x184D2204u) return fail(13);                       // frameType_unknown
    const uint32_t flg = frame[4];
    if ((flg >> 1) & 1) return fail(8);                              // reservedFlag_set
    if (((flg >> 6) & 3) != 1) return fail(6);                       // headerVersion_wrong
    const uint32_t hsize = 7 + (((flg >> 3) & 1) ? 8 : 0) + ((flg & 1) ? 4 : 0);
    if (frame_cap < hsize) return fail(12);
    const uint32_t bd = frame[5];
    const uint32_t bsid = (bd >> 4) & 7;
    if ((bd >> 7) & 1) return fail(8);
    if (bsid < 4) return fail(ST_MAXBLOCK);
    if (bd & 15) return fail(8);
    if (((xxh32_small(frame + 4, hsize - 5) >> 8) & 0xFF) != frame[hsize - 1]) return fail(17);   // headerChecksum_invalid
    r.flags = flg;
    const uint32_t bs = 1u << (8 + 2 * bsid);
    const uint32_t bck = (flg >> 4) & 1;
    uint64_t content = 0;
    if ((flg >> 3) & 1) content = (uint64_t)rd32_any(frame + 6) | ((uint64_t)rd32_any(frame + 10) << 32);
    uint64_t pos = hsize, out = 0;
    uint32_t n = 0;
    for (;;) {
        if (frame_cap - pos < 4) return fail(12);
        const uint32_t w = rd32_any(frame + pos);
        pos += 4;
        if (w == 0) break;
        const uint32_t csz = w & 0x7FFFFFFFu;
        if (csz > bs) return fail(ST_MAXBLOCK);
        if (frame_cap - pos < (uint64_t)csz + 4 * bck) return fail(12);
        if (n >= table_cap) return fail(ST_DSTSMALL);
        if (out >= dst_cap) return fail(ST_DSTSMALL);
        table[n].src_off = pos; table[n].dst_off = out; table[n].word = w;
        table[n].dst_size = (uint32_t)((dst_cap - out < bs) ? dst_cap - out : bs);      // capacity; decode overwrites
        out += bs;                                                  // provisional: full blocks (fixed up after decode)
        pos += (uint64_t)csz + 4 * bck;
        n++;
    }
    if ((flg >> 2) & 1) { if (frame_cap - pos < 4) return fail(12); pos += 4; }   // content checksum: skipped, see header
    r.n_blocks = n; r.consumed = pos; r.size = content;             // size = declared content size until decode fills it
    *res = r;
}

// ------------------------------- block decode ---------------------------------------------------
__global__ void k_init_result(ResultRec* res, uint32_t n_blocks, uint32_t flags)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        res->size = 0; res->consumed = 0; res->status = ST_OK; res->n_blocks = n_blocks; res->first_bad_block = 0xFFFFFFFFu; res->flags = flags;
    }
}
__global__ void k_set_block(BlockOut* t, BlockOut e) { if (threadIdx.x == 0 && blockIdx.x == 0) *t = e; }

// `hist0`: valid bytes directly in front of dst (streaming API: the previous blocks' last 64 KiB).
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_decode_blocks(const uint8_t* __restrict__ frame, uint8_t* dst, uint64_t dst_cap,
                                                                     BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                                     uint32_t n_max, uint32_t linked, uint32_t block_size, uint64_t hist0)
{
    const uint32_t w = uni(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6));
    if (res->status != ST_OK) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t lane = lane_id();
    if (!linked) {
        if (w >= n) return;
        const BlockOut e = table[w];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        int32_t got;
        if (e.word >> 31) {
            if (csz > e.dst_size) got = -2;
            else { wave_copy_disjoint(dst + e.dst_off, frame + e.src_off, csz); got = (int32_t)csz; }
        } else {
            got = wave_decode_block(frame + e.src_off, csz, dst + e.dst_off, e.dst_size, 0);
        }
        if (lane == 0) table[w].dst_size = (uint32_t)got;            // negative = failed
        return;
    }
    // linked frame: one wave, blocks in order, output packed (each block sees the bytes before it)
    if (w != 0) return;
    uint64_t out = 0;
    for (uint32_t b = 0; b < n; b++) {
        const BlockOut e = table[b];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        const uint32_t room = (uint32_t)((dst_cap - out < block_size) ? dst_cap - out : block_size);
        int32_t got;
        if (e.word >> 31) {
            if (csz > room) got = -2;
            else { wave_copy_disjoint(dst + out, frame + e.src_off, csz); got = (int32_t)csz; }
        } else {
            got = wave_decode_block(frame + e.src_off, csz, dst + out, room, out + hist0);
        }
        if (lane == 0) { table[b].dst_off = out; table[b].dst_size = (uint32_t)got; }
        if (got < 0) {
            for (uint32_t k = b + 1 + lane; k < n; k += WAVE) table[k].dst_size = 0;   // not decoded
            break;
        }
        out += (uint32_t)got;
    }
}

// totals + status; compacts the output if a non-final block of an independent frame decoded short
__global__ __launch_bounds__(64) void k_finish_decode(uint8_t* dst, BlockOut* __restrict__ table, ResultRec* res, uint32_t n_max,
                                                      uint32_t linked, uint32_t block_size, const uint32_t* __restrict__ bad_ck)
{
    if (res->status != ST_OK) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t lane = lane_id();
    // first failed block
    uint32_t first_bad = 0xFFFFFFFFu, bad_kind = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += WAVE) {
        const uint32_t b = b0 + lane;
        const bool bad = b < n && (int32_t)table[b].dst_size < 0;
        const uint64_t m = __ballot(bad);
        if (m) { const uint32_t f = (uint32_t)__builtin_ctzll(m); first_bad = b0 + f; bad_kind = __builtin_amdgcn_readlane(b < n ? table[b].dst_size : 0u, f); break; }
    }
    const uint32_t ck = bad_ck ? *bad_ck : 0xFFFFFFFFu;
    if (ck != 0xFFFFFFFFu && ck <= first_bad) {
        if (lane == 0) { res->status = ST_BLOCKCK; res->first_bad_block = ck; }
        return;
    }
    if (first_bad != 0xFFFFFFFFu) {
        if (lane == 0) { res->status = (bad_kind == (uint32_t)-2) ? ST_DSTSMALL : ST_GENERIC; res->first_bad_block = first_bad; }
        return;
    }
    // usual case, checked 64 blocks per step: every block already sits where its predecessors end (all blocks but
    // the last decoded to full size), so nothing has to move and the total is a sum
    uint64_t out = 0;
    bool in_place = true;
    for (uint32_t b0 = 0; b0 < n; b0 += WAVE) {
        const uint32_t b = b0 + lane;
        const uint32_t sz = b < n ? table[b].dst_size : 0u;
        const uint64_t at = b < n ? table[b].dst_off : 0ull;
        uint64_t incl = sz;
        for (uint32_t d = 1; d < WAVE; d <<= 1) { const uint64_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (!linked && __ballot(b < n && at != out + incl - sz)) { in_place = false; break; }
        out += __shfl(incl, WAVE - 1);
    }
    if (!in_place) {
        out = 0;
        for (uint32_t b = 0; b < n; b++) {          // uniform, serial: a rare path (a non-final block of an independent frame decoded short)
            const uint64_t at = table[b].dst_off; const uint32_t sz = table[b].dst_size;
            if (at != out) {
                for (uint32_t o = 0; o < sz; o += 1024) {            // forward move to a lower address
                    const uint32_t m = (sz - o < 1024) ? sz - o : 1024;
                    uint8_t v = 0; b16_ua v16;
                    const uint32_t i = lane * 16;
                    const bool full = i + 16 <= m;
                    if (full) v16 = *(const b16_ua*)(dst + at + o + i);
                    if (full) *(b16_ua*)(dst + out + o + i) = v16;
                    for (uint32_t j = (m & ~15u) + lane; j < m; j += WAVE) { v = dst[at + o + j]; dst[out + o + j] = v; }
                }
                if (lane == 0) table[b].dst_off = out;
            }
            out += sz;
        }
    }
    if (lane == 0) {
        const uint64_t declared = res->size;
        res->size = out;
        if ((res->flags >> 3) & 1) { if (declared != out) res->status = 14; }   // frameSize_wrong
    }
}

}  // namespace lz4f
