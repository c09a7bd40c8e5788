// common.cuh -- shared device helpers for the gfx950 (CDNA4, wave64) LZ4 frame kernels.
//
// Everything here is byte/integer work bound by HBM and by dependent-load latency; there is no
// matrix math anywhere on this path, so no MFMA.  Wave-uniform values are kept in SGPRs
// (readfirstlane) so that the sequence parse runs on the scalar unit and the 64 lanes are spent on
// the copies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

namespace lz4f {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32;      // (a pointer that keeps its address space: ds_* instructions, not flat_* - which count as memory operations too)
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef uint32_t u32_ua __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) b16_ua { uint32_t a, b, c, d; };   // 16 bytes, any alignment
struct __attribute__((aligned(4))) w3_a4 { uint32_t a, b, c; };               // 3 dwords, dword aligned

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// status codes written by kernels (values of LZ4F_errorCodes that can arise on the device)
enum : uint32_t { ST_OK = 0, ST_GENERIC = 1, ST_MAXBLOCK = 2, ST_BLOCKCK = 7, ST_DSTSMALL = 11, ST_DECOMP = 16 };

// ---- records shared with the host (include/lz4f_mi355x.h) ----
struct BlockOut {            // mirrors lz4f_mi355x_block
    uint64_t src_off, dst_off;
    uint32_t word, dst_size;
};
struct ResultRec {           // mirrors lz4f_mi355x_result
    uint64_t size, consumed;
    uint32_t status, n_blocks, first_bad_block, flags;
};

// sequence descriptor the decoders hand from their parse to their copies: x = lit_src | off[7:0] << 24, y = lit_len | off[15:8] << 24,
// z = dst, w = match_len (0: last sequence).  Positions and lengths are < 2^23 because a block holds at most 4 MiB.
struct SeqDesc { uint32_t x, y, z, w; };

// ---- XXH32 constants ----
constexpr uint32_t XP1 = 2654435761u, XP2 = 2246822519u, XP3 = 3266489917u, XP4 = 668265263u, XP5 = 374761393u;
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// ---- wave-cooperative copies (all arguments wave-uniform) --------------------------------------
// Non-overlapping copy of len bytes, any alignment.  16 B per lane (1 KiB per wave instruction);
// the ragged tail is one more 16-byte access ending exactly at len (rewrites identical bytes).
__device__ __forceinline__ void wave_copy_disjoint(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t len)
{
    const uint32_t lane = lane_id();
    if (len <= 16) {
        if (lane < len) dst[lane] = src[lane];
        return;
    }
    const uint32_t nfull = len >> 4, tail = len & 15;
    for (uint32_t base = 0; base < nfull + (tail ? 1u : 0u); base += WAVE) {
        const uint32_t c = base + lane;
        if (c < nfull) {
            *(b16_ua*)(dst + (size_t)c * 16) = *(const b16_ua*)(src + (size_t)c * 16);
        } else if (c == nfull && tail) {
            *(b16_ua*)(dst + len - 16) = *(const b16_ua*)(src + len - 16);
        }
    }
}

// One-round (len <= 1024) disjoint copy split into its load and its store, so that a wave can keep several
// copies in flight: issue the loads of 4 copies, then their stores (a lone load->store pair per wave leaves
// only ~16 KiB in flight per CU, half of what HBM latency x bandwidth needs).
struct Piece { uint32_t a, b, c, d; };          // plain registers (a packed/unaligned struct here is kept in scratch by hipcc)
__device__ __forceinline__ void piece_load(Piece& p, const uint8_t* __restrict__ src, uint32_t len)
{
    const uint32_t lane = lane_id(), nfull = len >> 4, tail = len & 15;
    p.a = p.b = p.c = p.d = 0;
    if (len < 16) { if (lane < len) p.a = src[lane]; }
    else {
        const bool body = lane < nfull, last = lane == nfull && tail;
        if (body || last) {
            const b16_ua t = *(const b16_ua*)(src + (body ? lane * 16 : len - 16));
            p.a = t.a; p.b = t.b; p.c = t.c; p.d = t.d;
        }
    }
}
__device__ __forceinline__ void piece_store(const Piece& p, uint8_t* __restrict__ dst, uint32_t len)
{
    const uint32_t lane = lane_id(), nfull = len >> 4, tail = len & 15;
    if (len < 16) { if (lane < len) dst[lane] = (uint8_t)p.a; }
    else {
        const bool body = lane < nfull, last = lane == nfull && tail;
        if (body || last) *(b16_ua*)(dst + (body ? lane * 16 : len - 16)) = b16_ua{p.a, p.b, p.c, p.d};
    }
}

// ---- copy pipeline primitives ------------------------------------------------------------------------------------------
// A copy job is one round (16..1024 bytes) of a non-overlapping copy.  job_load is UNCONDITIONAL: every lane loads
// 16 bytes (lanes without a piece read `safe`), so the load is never behind a branch and hipcc can count it exactly
// (loads behind any branch make it fall back to s_waitcnt vmcnt(0), which serialises load->store->load).  Callers
// ping-pong two register sets (no register moves between stages) so that the loads of the next jobs are in flight
// while the current ones are stored.
struct v4u_ua { uint32_t a, b, c, d; } __attribute__((packed, aligned(1)));
struct CopyJob { const uint8_t* s; uint8_t* d; uint32_t n; };          // n in [16, 1024], or 0 = no job (wave-uniform)
__device__ __forceinline__ void job_load(Piece& p, const CopyJob& j, const uint8_t* safe)
{
    const uint32_t lane = lane_id(), nfull = j.n >> 4, tail = j.n & 15;
    const uint8_t* a = safe;
    if (lane < nfull) a = j.s + lane * 16; else if (lane == nfull && tail) a = j.s + j.n - 16;
    const v4u_ua t = *(const v4u_ua*)a;
    p.a = t.a; p.b = t.b; p.c = t.c; p.d = t.d;
}
__device__ __forceinline__ void job_store(const CopyJob& j, const Piece& p)
{
    const uint32_t lane = lane_id(), nfull = j.n >> 4, tail = j.n & 15;
    if (lane < nfull || (lane == nfull && tail)) {
        uint8_t* a = (lane < nfull) ? j.d + lane * 16 : j.d + j.n - 16;
        *(v4u_ua*)a = v4u_ua{p.a, p.b, p.c, p.d};
    }
}

// LZ4 match copy: dst[i] = dst[i - offset] for i in [0,len), increasing i (forward overlap
// semantics).  dst - offset .. dst is already written by THIS wave
// (vector memory operations of one wave are performed in issue order), or by earlier kernels.
__device__ __forceinline__ void wave_copy_match(uint8_t* dst, uint32_t offset, uint32_t len)
{
    const uint32_t lane = lane_id();
    const uint8_t* src = dst - offset;
    if (offset >= len) {                       // no self-overlap at all
        wave_copy_disjoint(dst, src, len);
        return;
    }
    uint32_t done = 0;
    if (offset < 64) {
        // short period: replicate the pattern bytewise until >= 1 KiB of it exists (or the match ends)
        uint32_t idx = lane % offset;                 // (done + lane) % offset, kept incrementally
        const uint32_t inc = WAVE % offset;
        const uint32_t k = (1024 + offset - 1) / offset;          // widen to offset*k >= 1024 afterwards
        const uint32_t need = (k - 1) * offset;
        while (done < len && done < need) {
            if (done + lane < len) dst[done + lane] = src[idx];
            idx += inc; if (idx >= offset) idx -= offset;
            done += WAVE;
        }
        if (done >= len) return;
        offset *= k;                                  // dst[i] = dst[i - k*offset] holds once (k-1)*offset bytes exist
        src = dst - offset;
    }
    // period >= 64: rounds of R = 16*floor(offset/16) <= 1024 bytes, 16 B per lane; a round never
    // reads what it writes because R <= offset.
    const uint32_t R = (offset >= 1024) ? 1024u : (offset & ~15u);
    const uint32_t lanes_r = R >> 4;
    while (done < len) {
        const uint32_t left = len - done;
        if (left >= R) {
            if (lane < lanes_r) *(b16_ua*)(dst + done + lane * 16) = *(const b16_ua*)(src + done + lane * 16);
            done += R;
        } else {
            // last partial round: left < R <= offset, so it is a disjoint copy
            wave_copy_disjoint(dst + done, src + done, left);
            done = len;
        }
    }
}

// Wave-wide inclusive scans on the DPP path (row shifts + row broadcasts: no LDS crossbar, 6 instructions each).
__device__ __forceinline__ uint32_t dpp_incl_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t dpp_incl_scan_max(uint32_t v)     // (identity 0)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
    return v;
}

// the same, exclusive: what the lanes below me have (lane 0: 0)
__device__ __forceinline__ uint32_t dpp_excl_scan_max(uint32_t v)
{
    const uint32_t incl = dpp_incl_scan_max(v);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}

}  // namespace lz4f
