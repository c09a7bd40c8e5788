// decode_linked.cuh -- decode of a LINKED-block frame (the reference's default framing: 64 KiB dependent blocks,
// Conduit.hsc:248-263) by one workgroup, with the 64 KiB sliding window in LDS.
//
// A linked frame is one chain: block k's matches reach into the last 64 KiB of blocks < k, so the blocks cannot go to
// different workgroups.  decode_fused.cuh walks such a frame block by block with one parser wave, and every match copy
// is a round trip to MALL/HBM (~3 us loaded, six in flight): ~1300 cycles per match, 0.8 GiB/s.  Here instead (3.5 GiB/s;
// four parsers + three literal waves measured slower - the eight waves of the one CU share its scalar unit):
//   * the last 128 KiB of output live in an LDS ring (`win`); a match copy is ds_read_b128 -> ds_write_b128 (+ the
//     global store of the same registers), ~150 cycles, and LDS operations of a wave are performed in order, so
//     dependent matches need no waiting at all;
//   * block starts are entry points that are known without parsing, so three PARSER waves work on three consecutive
//     blocks at a time (each a scalar state machine like decode_fused's, descriptors with block-relative positions
//     through its own LDS ring);
//   * one CHAIN wave replays all matches in stream order out of the window;
//   * four LITERAL waves copy the literal runs payload -> output and into the window (run k of a slot -> wave k mod 4),
//     up to 64 KiB ahead of the chain (a ring of 128 KiB: writing position x overwrites x - 128 KiB, which no match at or beyond
//     `next_match_dst` can still want as long as x <= next_match_dst + 64 KiB).
// Every wave walks every slot of every block in order, so block bases and the end of the frame need
// no extra hand-off: the last sequence of a block is the one without a match.  Hand-offs are LDS words, no barriers.
// Restriction: every block but the last must decode to exactly the block size (true for frames written without
// LZ4F_flush / autoFlush); otherwise the kernel sets `*fallback` and leaves the frame to decode_fused.cuh.
#pragma once
#include "common.cuh"
#include "decode.cuh"
#include "decode_fused.cuh"

namespace lz4f {

constexpr int      LK_PARSERS = 3, LK_LITS = 4;
constexpr int      LK_WAVES = LK_PARSERS + 1 + LK_LITS;       // 8
constexpr uint32_t LK_WIN = 131072;                            // bytes of output kept in LDS (power of two)
constexpr uint32_t LK_BIAS = 65536;                            // window position of output byte 0 (history sits below it)
constexpr uint32_t LK_STAGE = 2048, LK_OVER = 576, LK_RING = 4;
constexpr uint32_t LK_NONE = 0xFFFFFFFFu;
constexpr uint32_t LK_SPIN_MAX = 1u << 24;                     // polls of one hand-off before the kernel gives up (a hang would take the GPU with it)

struct alignas(16) LkRing {                                    // parser p -> consumers
    uint4    desc[LK_RING][64];
    uint8_t  stage[2][LK_STAGE + LK_OVER];
    uint32_t count[LK_RING];                                   // descriptors in the slot
    uint32_t block[LK_RING];                                   // frame block the slot belongs to
    uint32_t lit_prog[LK_RING][LK_LITS];                       // literal wave w has finished this many of ITS runs of the slot (runs w, w+LK_LITS, ...)
    uint32_t produced;                                         // slots published
    uint32_t done[1 + LK_LITS];                                // slots each consumer wave is through with (chain, literal waves)
    uint32_t pad[2];
};
struct alignas(16) LkShared {
    uint8_t  win[LK_WIN];
    LkRing   ring[LK_PARSERS];
    uint32_t next_match_dst;                                   // absolute output position of the match the chain wave is at
    int32_t  status;                                           // < 0: stop (-1 malformed, -2 output too small, -3 not for this kernel)
    uint32_t bad_block;
    uint32_t why;                                              // (debug) which check stopped the kernel
};
static_assert(sizeof(LkShared) <= 163840, "one workgroup per CU");

typedef uint32_t lk_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lk_widx(uint32_t pos) { return (pos + LK_BIAS) & (LK_WIN - 1); }
// 16 bytes at any byte address of the window (gfx950 LDS takes unaligned b128 accesses; tools/probe/lds_unaligned.hip)
__device__ __forceinline__ lk_v4 lk_win_read16(const uint8_t* win, uint32_t idx)
{
    lk_v4 v;
    const uint32_t a = (uint32_t)(uintptr_t)(lptr_t)win + idx;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lk_win_write16(uint8_t* win, uint32_t idx, lk_v4 v)
{
    const uint32_t a = (uint32_t)(uintptr_t)(lptr_t)win + idx;
    asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory");
}

__device__ __forceinline__ void lk_stage_issue(uint8_t* slot, const uint8_t* __restrict__ in, uint32_t csize, uint32_t s)
{
    const uint32_t lane = lane_id();
    const uint32_t base = s * LK_STAGE;
    if (base >= csize) return;
    const uint32_t end = (base + LK_STAGE + LK_OVER < csize) ? base + LK_STAGE + LK_OVER : csize;
    const uint32_t span = end - base;
    for (uint32_t piece = 0; piece < span; piece += 1024) {
        const uint32_t o = piece + lane * 16;
        if (o + 16 <= span) __builtin_amdgcn_global_load_lds((gptr_t)(in + base + o), (lptr_t)(slot + piece), 16, 0, 0);
    }
    const uint32_t tail0 = span & ~15u;
    if (lane < span - tail0) slot[tail0 + lane] = in[base + tail0 + lane];
}

// ---------------- parser wave p: blocks p, p + LK_PARSERS, ... ----------------
// Descriptors are block-relative: x = literal source (offset in the block's payload) | offset low byte << 24,
// y = literal length | offset high byte << 24, z = output position of the literal run inside the block, w = match length.
__device__ __forceinline__ void lk_parser(LkShared& sh, LkRing& rg, uint32_t p, const uint8_t* __restrict__ frame, const BlockOut* __restrict__ table,
                                          uint32_t n, uint32_t block_size, uint64_t dst_cap)
{
    const uint32_t lane = lane_id();
    uint8_t* stages = &rg.stage[0][0];
    uint32_t slot_no = 0;                                                     // slots this parser has published
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    auto publish = [&](uint32_t count, uint32_t block) -> bool {
        for (uint32_t spin = 0;; spin++) {                                   // the slot being overwritten must be finished by every consumer
            if ((int32_t)lds_peek((const uint32_t*)&sh.status) < 0) return false;
            if (spin > LK_SPIN_MAX) { sh.bad_block = block; sh.why = 1; sh.status = -1; return false; }
            uint32_t lo = lds_peek(&rg.done[0]);
            for (int c = 1; c <= LK_LITS; c++) { const uint32_t v = lds_peek(&rg.done[c]); lo = v < lo ? v : lo; }
            if (slot_no < lo + LK_RING) break;
            __builtin_amdgcn_s_sleep(32);
        }
        const uint32_t i = slot_no % LK_RING;
        rg.desc[i][lane] = uint4{d0, d1, d2, d3};
        rg.count[i] = count; rg.block[i] = block;
        if (lane < LK_LITS) rg.lit_prog[i][lane] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot_no++;
        lds_poke(&rg.produced, slot_no);
        return true;
    };
    for (uint32_t b = p; b < n; b += LK_PARSERS) {
        const BlockOut e = table[b];
        const uint32_t csize = e.word & 0x7FFFFFFFu;
        const uint64_t base_out = (uint64_t)b * block_size;                  // (checked by the consumers)
        const uint32_t cap = (uint32_t)((dst_cap > base_out) ? ((dst_cap - base_out < block_size) ? dst_cap - base_out : block_size) : 0);
        if (e.word >> 31) {                                                  // stored block: one literal-only "sequence"
            if (csize > cap) { if (lane == 0) { sh.bad_block = b; sh.why = 2; sh.status = -2; } return; }
            d0 = 0; d1 = csize; d2 = 0; d3 = 0;
            if (!publish(1, b)) return;
            continue;
        }
        const uint8_t* in = frame + e.src_off;
        uint32_t nseq = 0, status = csize == 0 ? 1u : 0u, op = 0;
        int32_t  cur = -1;
        if (!status) lk_stage_issue(stages, in, csize, 0);
        auto need = [&](uint32_t qq) {
            const int32_t s = (int32_t)(qq / LK_STAGE);
            if (s == cur) return;
#pragma unroll 1
            for (int32_t it = (s == cur + 1) ? 1 : 0; it < 2; it++) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lk_stage_issue(stages + (uint32_t)((s + it) & 1) * (LK_STAGE + LK_OVER), in, csize, (uint32_t)(s + it));
            }
            cur = s;
        };
        lk_v4 win = {0u, 0u, 0u, 0u};
        uint32_t wb = 0xFFFFFC00u;
        auto reload = [&](uint32_t qq) {
            need(qq);
            const uint32_t s = qq / LK_STAGE;
            const uint32_t o = (qq - s * LK_STAGE) & ~7u;
            const uint32_t addr = (uint32_t)(uintptr_t)(lptr_t)(stages + (s & 1) * (LK_STAGE + LK_OVER)) + o + 8u * lane;
            asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(win) : "v"(addr) : "memory");
            wb = qq & ~7u;
        };
        auto fetch = [&](uint32_t qq) -> uint64_t {
            uint32_t rel = qq - wb;
            if (rel >= 504u) { reload(qq); rel = qq - wb; }
            const uint32_t l = rel >> 3, sh8 = (rel & 7u) * 8u;
            const uint64_t lo = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.x, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.y, l) << 32);
            const uint64_t hi = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.z, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.w, l) << 32);
            return (lo >> sh8) | ((hi << 1) << (63u - sh8));
        };
        auto ext_slow = [&](uint32_t pos, uint32_t& after) -> uint32_t {
            uint32_t add = 0;
            for (;;) {
                if (pos >= csize || add > 0x7FFF0000u) { status = 1; after = pos; return add; }
                const uint32_t v = uni((uint32_t)in[pos]);
                add += v; pos++;
                if (v != 255) { after = pos; return add; }
            }
        };
        uint32_t pos = 0;
        uint32_t fin = status;
        while (fin == 0) {
            const uint64_t w = fetch(pos);
            const uint32_t token = (uint32_t)w & 0xFF;
            uint32_t lit = token >> 4;
            uint32_t pl = pos + 1;
            uint32_t bad = 0;
            {
                const uint64_t x = w >> 8;
                const uint32_t f = (uint32_t)__builtin_ctzll(~x);
                const uint32_t k = f >> 3;
                const uint32_t ext = 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
                const bool is15 = lit == 15;
                lit = is15 ? 15u + ext : lit;
                pl = is15 ? pos + 2 + k : pl;
                if (is15 && k == 7) { lit = 15u + ext_slow(pos + 1, pl); bad |= status; }
            }
            bad |= pl > csize ? 1u : 0u;
            const uint32_t in_left = csize - pl;
            const uint32_t lit_src = pl;
            const bool is_last = lit > in_left || lit + 8 > in_left || lit + 12 > cap - op;
            uint32_t mlen = 0, off = 0, npos = pos;
            if (!is_last) {
                const uint32_t qo = pl + lit;
                const uint64_t w2 = fetch(qo);
                off = (uint32_t)w2 & 0xFFFF;
                bad |= off == 0 ? 1u : 0u;                                   // (offset reach is checked by the chain wave: it knows absolute positions)
                const uint64_t x = w2 >> 16;
                const uint32_t f = (uint32_t)__builtin_ctzll(~x);
                const uint32_t k = f >> 3;
                const uint32_t ext = 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
                const bool is15 = (token & 15) == 15;
                mlen = is15 ? 15u + ext : (token & 15);
                npos = is15 ? qo + 3 + k : qo + 2;
                if (is15 && k == 6) { mlen = 15u + ext_slow(qo + 2, npos); bad |= status; }
                bad |= (is15 && npos + 4 >= csize) ? 1u : 0u;
                mlen += 4;
                bad |= mlen + 5 > cap - (op + lit) ? 1u : 0u;
            } else {
                bad |= lit != in_left ? 1u : 0u;
                bad |= lit > cap - op ? 1u : 0u;
            }
            fin |= bad;
            if (fin == 0) {
                const uint32_t slot = nseq & 63;
                const bool mine = lane == slot;
                d0 = mine ? (lit_src | ((off & 0xFFu) << 24)) : d0;
                d1 = mine ? (lit | ((off >> 8) << 24)) : d1;
                d2 = mine ? op : d2;
                d3 = mine ? mlen : d3;
                nseq++;
                op += lit + mlen;
                if (slot == 63 || is_last) { if (!publish(slot + 1, b)) return; }
                pos = npos;
                fin |= is_last ? 2u : (pos >= csize ? 1u : 0u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (pending stage prefetch of this block)
        if (fin & 1u) { if (lane == 0) { sh.bad_block = b; sh.why = 3; sh.status = -1; } return; }
    }
}

// ---------------- what every consumer does: walk all slots of all blocks in order ----------------
struct LkCursor {
    uint32_t n0, n1, n2, n3;         // next slot number per parser ring (scalars: a dynamically indexed array would live in scratch)
    uint32_t block;                  // current block
    uint64_t base;                   // absolute output position of its first byte
    uint32_t gslot;                  // slots seen so far (all rings)
    __device__ __forceinline__ uint32_t next(uint32_t p) const { return p == 0 ? n0 : (p == 1 ? n1 : (p == 2 ? n2 : n3)); }
    __device__ __forceinline__ uint32_t bump(uint32_t p) { if (p == 0) return ++n0; if (p == 1) return ++n1; if (p == 2) return ++n2; return ++n3; }
};
static_assert(LK_PARSERS <= 4, "LkCursor has four counters");
// waits for the next slot; false = stop (error elsewhere).  The slot stays valid until the caller bumps its `done` word.
__device__ __forceinline__ bool lk_next_slot(LkShared& sh, LkCursor& c, uint32_t& p, uint32_t& idx, uint32_t& count)
{
    p = c.block % LK_PARSERS;
    LkRing& rg = sh.ring[p];
    for (uint32_t spin = 0; lds_peek(&rg.produced) <= c.next(p); spin++) {
        if ((int32_t)lds_peek((const uint32_t*)&sh.status) < 0) return false;
        if (spin > LK_SPIN_MAX) { sh.bad_block = c.block; sh.why = 4; sh.status = -1; return false; }
        __builtin_amdgcn_s_sleep(8);
    }
    idx = c.next(p) % LK_RING;
    count = uni(rg.count[idx]);
    return true;
}

// ---------------- literal wave ----------------
__device__ __forceinline__ void lk_literals(LkShared& sh, uint32_t me /* 0..LK_LITS-1 */, const uint8_t* __restrict__ frame, const BlockOut* __restrict__ table,
                                            uint8_t* out, uint32_t n, uint32_t block_size, uint64_t dst_cap)
{
    const uint32_t lane = lane_id();
    LkCursor c; c.n0 = c.n1 = c.n2 = c.n3 = 0; c.block = 0; c.base = 0; c.gslot = 0;
    while (c.block < n) {
        uint32_t p, idx, count;
        if (!lk_next_slot(sh, c, p, idx, count)) return;
        LkRing& rg = sh.ring[p];
        const uint4 d = rg.desc[idx][lane];
        const uint32_t vsrc = d.x & 0xFFFFFFu, vlen = d.y & 0xFFFFFFu, vdst = d.z, vml = d.w;
        const uint8_t* in = frame + table[c.block].src_off;
        const uint8_t* safe = frame;                                          // >= 16 readable bytes for idle lanes
        {
            // The literal runs of a slot are dealt round-robin to the literal waves (run k -> wave k % LK_LITS), so all of
            // them work on the block the chain is at.  Pieces of <= 1 KiB go through two ping-pong register sets (two loads
            // each): four payload reads in flight per wave.  A piece goes to the output and - if it can still be a match
            // source - into the window; when the last piece of this wave's j-th run is stored, lit_prog[me] becomes j
            struct LJob { const uint8_t* s; uint64_t d; uint32_t n, wi, fin, rend; };   // n == 0 && fin == 0: none; fin: bit0 keep in window, bits 1.. = k+1 when last piece of run k; rend: where the run ends
            uint32_t k = me, at = 0;
            bool stop = false;
            auto next_job = [&]() -> LJob {
                for (;;) {
                    if (k >= count || stop) return LJob{safe, 0, 0, 0, 0, 0};
                    const uint32_t len = __builtin_amdgcn_readlane(vlen, k);
                    const uint64_t dabs = c.base + __builtin_amdgcn_readlane(vdst, k);
                    const uint8_t* sp = in + __builtin_amdgcn_readlane(vsrc, k);
                    if (at == 0) {
                        if (dabs + len > dst_cap) { if (lane == 0) { sh.bad_block = c.block; sh.why = 5; sh.status = -2; } stop = true; continue; }
                        // the window may be written up to 64 KiB beyond the match the chain wave is at.  Not yet: hand out
                        // nothing, so that the runs already loaded get stored (the chain may be waiting for exactly those)
                        if ((uint64_t)lds_peek(&sh.next_match_dst) + 65536 < dabs + len) return LJob{safe, 0, 0, 0, 0, 0};
                        if (len == 0) { k += LK_LITS; return LJob{safe, 0, 0, 0, (k / LK_LITS) << 1, 0}; }   // nothing to copy, but the count must still advance in order
                    }
                    uint32_t nn = len - at; if (nn > 1024) nn = 1024;
                    const uint32_t wi = lk_widx((uint32_t)dabs + at);
                    if (nn > LK_WIN - wi) nn = LK_WIN - wi;                   // never across the end of the ring
                    const uint32_t keep = (len - at <= 65536 + 1024) ? 1u : 0u;   // only the last 64 KiB of a run can ever be a match source
                    LJob j{sp + at, dabs + at, nn, wi, keep, (uint32_t)(dabs + len)};
                    at += nn;
                    if (at >= len) { k += LK_LITS; j.fin |= (k / LK_LITS) << 1; at = 0; }
                    return j;
                }
            };
            auto jload = [&](Piece& pc, const LJob& j) {
                const uint32_t nfull = j.n >> 4, tail = j.n & 15;
                if (j.n >= 16 || j.n == 0) {
                    const uint8_t* a = safe;
                    if (lane < nfull) a = j.s + lane * 16; else if (lane == nfull && tail && j.n) a = j.s + j.n - 16;
                    const v4u_ua t = *(const v4u_ua*)a;
                    pc.a = t.a; pc.b = t.b; pc.c = t.c; pc.d = t.d;
                } else {                                                       // short piece: one byte per lane, nothing read beyond it
                    uint8_t v = 0;
                    if (lane < j.n) v = j.s[lane];
                    pc.a = v;
                }
            };
            auto jstore = [&](const LJob& j, const Piece& pc) {
                if (j.n >= 16) {
                    const uint32_t nfull = j.n >> 4, tail = j.n & 15;
                    if (lane < nfull || (lane == nfull && tail)) {
                        const uint32_t o = lane < nfull ? lane * 16 : j.n - 16;
                        *(v4u_ua*)(out + j.d + o) = v4u_ua{pc.a, pc.b, pc.c, pc.d};
                        // Into the window only what can still be a match source: the last 64 KiB of the run, to the byte.  The
                        // window is a ring of 128 KiB and the other literal waves may already be writing the 64 KiB beyond the
                        // match the chain is at - for a run that ends there, anything older than 64 KiB would land on exactly
                        // those positions (a stored 256 KiB block in front of a block that starts with literals, and the
                        // newer bytes lost the race: found by tools/soak_indexed.py).
                        if (j.fin & 1u) {
                            const uint32_t age = j.rend - ((uint32_t)j.d + o);           // from my first byte to the end of the run
                            if (age <= 65536u) lk_win_write16(sh.win, j.wi + o, lk_v4{pc.a, pc.b, pc.c, pc.d});
                            else if (age < 65536u + 16u) {                               // straddles the line: the younger bytes, one by one
                                const uint32_t first = age - 65536u;
#pragma unroll
                                for (uint32_t i = 1; i < 16; i++) {
                                    const uint32_t w = i < 4 ? pc.a : i < 8 ? pc.b : i < 12 ? pc.c : pc.d;
                                    if (i >= first) sh.win[j.wi + o + i] = (uint8_t)(w >> ((i & 3u) * 8u));
                                }
                            }
                        }
                    }
                } else if (lane < j.n) {
                    out[j.d + lane] = (uint8_t)pc.a;
                    sh.win[j.wi + lane] = (uint8_t)pc.a;
                }
                if (j.fin >> 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lds_poke(&rg.lit_prog[idx][me], j.fin >> 1); }
            };
            const LJob none{safe, 0, 0, 0, 0, 0};
            auto valid = [](const LJob& j) { return (j.n | j.fin) != 0; };
            for (uint32_t spin = 0; k < count && !stop;) {
                // (a set's second job is only asked for when the first exists: the throttle may open between two calls, and
                // a job handed out behind a "none" would never be stored)
                LJob a0 = next_job(), a1 = valid(a0) ? next_job() : none, b0, b1;
                if (!valid(a0)) {                                             // run k must wait for the chain wave; nothing of ours is in flight
                    if ((int32_t)lds_peek((const uint32_t*)&sh.status) < 0) return;
                    if (++spin > LK_SPIN_MAX) { if (lane == 0) { sh.bad_block = c.block; sh.why = 6; sh.status = -1; } return; }
                    __builtin_amdgcn_s_sleep(8);
                    continue;
                }
                spin = 0;
                Piece pa0, pa1, pb0, pb1;
                jload(pa0, a0); jload(pa1, a1);
                for (;;) {
                    b0 = valid(a1) ? next_job() : none; b1 = valid(b0) ? next_job() : none;
                    jload(pb0, b0); jload(pb1, b1);
                    jstore(a0, pa0); jstore(a1, pa1);
                    if (!valid(b0)) break;
                    a0 = valid(b1) ? next_job() : none; a1 = valid(a0) ? next_job() : none;
                    jload(pa0, a0); jload(pa1, a1);
                    jstore(b0, pb0); jstore(b1, pb1);
                    if (!valid(a0)) break;
                }
            }
            if (stop) return;
        }
        // block bookkeeping: the sequence without a match ends its block
        const uint32_t lastm = __builtin_amdgcn_readlane(vml, count - 1);
        if (lastm == 0) {
            const uint32_t bsz = __builtin_amdgcn_readlane(vdst, count - 1) + __builtin_amdgcn_readlane(vlen, count - 1);
            c.base += bsz; c.block++;
        }
        c.gslot++;
        lds_poke(&rg.done[1 + me], c.bump(p));
    }
}

// ---------------- chain wave: all matches, stream order, out of the window ----------------
__device__ __forceinline__ lk_v4 lk_win_read16_nowait(const uint8_t* win, uint32_t idx)
{
    lk_v4 v;
    const uint32_t a = (uint32_t)(uintptr_t)(lptr_t)win + idx;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lk_chain(LkShared& sh, uint8_t* out, BlockOut* __restrict__ table, uint32_t n, uint32_t block_size, uint64_t dst_cap,
                                         uint64_t hist0, uint32_t* __restrict__ fallback)
{
    const uint32_t lane = lane_id();
    LkCursor c; c.n0 = c.n1 = c.n2 = c.n3 = 0; c.block = 0; c.base = 0; c.gslot = 0;
    uint8_t* win = sh.win;
    while (c.block < n) {
        uint32_t p, idx, count;
        if (!lk_next_slot(sh, c, p, idx, count)) return;
        LkRing& rg = sh.ring[p];
        if (c.base != (uint64_t)c.block * block_size) {                       // an earlier block was short: not this kernel's case
            if (lane == 0) { *fallback = 1u; sh.why = 7; sh.status = -3; }
            return;
        }
        const uint4 d = rg.desc[idx][lane];
        const uint32_t vlen = d.y & 0xFFFFFFu, vdst = d.z, vml = d.w, voff = (d.x >> 24) | ((d.y >> 24) << 8);
        // per lane, once per slot: absolute destination of my match and the two rules that need absolute positions
        const uint64_t dabs64 = c.base + vdst + vlen;
        const uint32_t vdm = (uint32_t)dabs64;
        const bool has = lane < count && vml != 0;
        const uint64_t bad_reach = __ballot(has && (uint64_t)voff > dabs64 + hist0);
        const uint64_t bad_room = __ballot(has && dabs64 + vml > dst_cap);
        if (bad_reach | bad_room) {
            const uint64_t first = (bad_reach | bad_room) & (0 - (bad_reach | bad_room));
            if (lane == 0) { sh.bad_block = c.block; sh.why = 8; sh.status = (bad_reach & first) ? -1 : -2; }
            return;
        }
        // wait for literal run k (wave k % LK_LITS must have finished k / LK_LITS + 1 of its runs); false = stop
        auto lit_ready = [&](uint32_t k) -> bool {
            const uint32_t w = k % LK_LITS, need = k / LK_LITS;
            for (uint32_t spin = 0; lds_peek(&rg.lit_prog[idx][w]) <= need; spin++) {
                if ((int32_t)lds_peek((const uint32_t*)&sh.status) < 0) return false;
                if (spin > LK_SPIN_MAX) { sh.bad_block = c.block; sh.why = 10; sh.status = -1; return false; }
                __builtin_amdgcn_s_sleep(1);
            }
            return true;
        };
        // one match, any shape: copy [dm - o, dm - o + m) -> [dm, dm + m) inside the window and to the output.  LDS operations
        // of one wave are performed in order, so a later read sees an earlier write: only genuine self-overlap needs care.
        auto copy_any = [&](uint32_t dm, uint32_t o, uint32_t m) {
            uint8_t* og = out + dm;                                            // (absolute positions are below 4 GiB here: the launcher checks)
            uint32_t done = 0, off = o;
            if (off < 16) {
                // short period: bytewise until at least 1 KiB of the pattern exists, then treat it as a long period
                const uint32_t kmul = (1024 + off - 1) / off, need = (kmul - 1) * off;
                uint32_t ph = lane % off; const uint32_t inc = WAVE % off;
                while (done < m && done < need) {
                    if (done + lane < m) {
                        const uint8_t v = win[lk_widx(dm - off + ph)];
                        win[lk_widx(dm + done + lane)] = v; og[done + lane] = v;
                    }
                    ph += inc; if (ph >= off) ph -= off;
                    done += WAVE;
                }
                if (done > m) done = m;
                off *= kmul;
            }
            while (done < m) {
                uint32_t nn = m - done; if (nn > 1024) nn = 1024;
                if (nn > off) nn = off & ~15u;                                 // a round never reads what it writes (off >= 16 here)
                const uint32_t si = lk_widx(dm + done - off), di = lk_widx(dm + done);
                if (nn > LK_WIN - si) nn = LK_WIN - si;
                if (nn > LK_WIN - di) nn = LK_WIN - di;
                if (nn >= 16) {
                    const uint32_t nfull = nn >> 4, tail = nn & 15;
                    const bool act = lane < nfull || (lane == nfull && tail);
                    const uint32_t oo = lane < nfull ? lane * 16 : nn - 16;
                    lk_v4 v = {0u, 0u, 0u, 0u};
                    if (act) v = lk_win_read16(win, si + oo);
                    if (act) { lk_win_write16(win, di + oo, v); *(v4u_ua*)(og + done + oo) = v4u_ua{v.x, v.y, v.z, v.w}; }
                } else if (lane < nn) {
                    const uint8_t v = win[si + lane];
                    win[di + lane] = v; og[done + lane] = v;
                }
                done += nn;
            }
        };
        // Per lane, once per slot: is my match of the usual shape (16..1024 bytes, no self-overlap, not across the ring seam),
        // and does its source end at or below the destination of the match three places earlier (then it can share an LDS
        // round trip with up to three predecessors, whatever the grouping).
        const uint32_t vsi = lk_widx(vdm - voff), vdi = lk_widx(vdm);
        const bool usual = has && vml >= 16 && vml <= 1024 && voff >= vml && vsi + vml <= LK_WIN && vdi + vml <= LK_WIN;
        const uint32_t dm3 = __shfl_up(vdm, 3);
        const uint64_t usual_m = __ballot(usual);
        const uint64_t join_m = __ballot(usual && lane >= 3 && vdm - voff + vml <= dm3) & usual_m;
        // If the whole slot lies within 64 KiB of output, its literal runs can all be in the window before its first match:
        // one wait per slot instead of one per match.
        const uint32_t first_dst = (uint32_t)c.base + __builtin_amdgcn_readlane(vdst, 0);
        const uint32_t last_end = (uint32_t)c.base + __builtin_amdgcn_readlane(vdst, count - 1) + __builtin_amdgcn_readlane(vlen, count - 1);
        const bool whole = last_end - first_dst <= 65536;
        if (whole) {
            lds_poke(&sh.next_match_dst, first_dst);
            for (uint32_t j = 0; j < LK_LITS && j < count; j++) if (!lit_ready(count - 1 - j)) return;     // the last run of each literal wave
        }
        uint32_t k = 0;
        while (k < count) {
            const uint32_t m0 = __builtin_amdgcn_readlane(vml, k);
            if (m0 == 0) { k++; continue; }
            const uint32_t dm0 = __builtin_amdgcn_readlane(vdm, k);
            lds_poke(&sh.next_match_dst, dm0);
            if (!whole && !lit_ready(k)) return;
            if (!((usual_m >> k) & 1)) { copy_any(dm0, __builtin_amdgcn_readlane(voff, k), m0); k++; continue; }
            // matches k .. k+nb-1 share one LDS round trip: the following ones must be `usual`, joinable, and - when literals
            // are waited for one by one - their runs must have landed
            uint32_t nb = 1;
            {
                const uint64_t run = ~((usual_m & join_m) >> (k + 1));        // first zero bit after k ends the group
                const uint32_t can = run ? (uint32_t)__builtin_ctzll(run) : 63u;
                nb += can < 3 ? can : 3;
                if (k + nb > count) nb = count - k;
                if (!whole) nb = 1;
            }
            uint32_t ra[4], pc[4], dmv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t kk = (uint32_t)j < nb ? k + j : k;
                const uint32_t m = __builtin_amdgcn_readlane(vml, kk);
                const uint32_t nfull = m >> 4, tail = m & 15;
                const bool act = (uint32_t)j < nb && (lane < nfull || (lane == nfull && tail));
                pc[j] = act ? (lane < nfull ? lane * 16 : m - 16) : 0xFFFFFFFFu;            // my piece of match j, or none
                ra[j] = act ? __builtin_amdgcn_readlane(vsi, kk) + pc[j] : 0u;
                dmv[j] = __builtin_amdgcn_readlane(vdm, kk);
            }
            // (all four reads are issued by every lane, idle ones at window index 0: a read behind a branch would let the
            // compiler merge its result with another value BEFORE the wait below - it does not know the read is asynchronous)
            const lk_v4 v0 = lk_win_read16_nowait(win, ra[0]), v1 = lk_win_read16_nowait(win, ra[1]);
            const lk_v4 v2 = lk_win_read16_nowait(win, ra[2]), v3 = lk_win_read16_nowait(win, ra[3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (pc[0] != 0xFFFFFFFFu) { lk_win_write16(win, lk_widx(dmv[0]) + pc[0], v0); *(v4u_ua*)(out + dmv[0] + pc[0]) = v4u_ua{v0.x, v0.y, v0.z, v0.w}; }
            if (pc[1] != 0xFFFFFFFFu) { lk_win_write16(win, lk_widx(dmv[1]) + pc[1], v1); *(v4u_ua*)(out + dmv[1] + pc[1]) = v4u_ua{v1.x, v1.y, v1.z, v1.w}; }
            if (pc[2] != 0xFFFFFFFFu) { lk_win_write16(win, lk_widx(dmv[2]) + pc[2], v2); *(v4u_ua*)(out + dmv[2] + pc[2]) = v4u_ua{v2.x, v2.y, v2.z, v2.w}; }
            if (pc[3] != 0xFFFFFFFFu) { lk_win_write16(win, lk_widx(dmv[3]) + pc[3], v3); *(v4u_ua*)(out + dmv[3] + pc[3]) = v4u_ua{v3.x, v3.y, v3.z, v3.w}; }
            k += nb;
        }
        const uint32_t lastm = __builtin_amdgcn_readlane(vml, count - 1);
        if (lastm == 0) {
            const uint32_t lk = count - 1;
            lds_poke(&sh.next_match_dst, (uint32_t)(c.base + __builtin_amdgcn_readlane(vdst, lk) + __builtin_amdgcn_readlane(vlen, lk)));
            if (!lit_ready(lk)) return;                                        // the block's final literals
            const uint32_t bsz = __builtin_amdgcn_readlane(vdst, lk) + __builtin_amdgcn_readlane(vlen, lk);
            if (lane == 0) { table[c.block].dst_off = c.base; table[c.block].dst_size = bsz; }
            c.base += bsz; c.block++;
        }
        c.gslot++;
        lds_poke(&rg.done[0], c.bump(p));
    }
}

__global__ __launch_bounds__(64 * LK_WAVES) void k_decode_linked(const uint8_t* __restrict__ frame, uint8_t* dst, uint64_t dst_cap,
                                                                 BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                                 uint32_t n_max, uint32_t block_size, uint64_t hist0, uint32_t* __restrict__ fallback,
                                                                 const uint32_t* __restrict__ only_if = nullptr)
{   // fallback[0]: 1 = frame left to the generic kernel; fallback[1..2] (debug): which check stopped this kernel, at which block
    // only_if: launched behind the indexed kernels (decode_indexed.cuh), runs only if they gave the frame up
    __shared__ LkShared sh;
    if (res->status != ST_OK) return;
    if (only_if && *only_if == 0) { if (threadIdx.x == 0) *fallback = 0u; return; }
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t tid = threadIdx.x, wave = uni(tid >> 6);
    if (tid == 0) { sh.next_match_dst = 0; sh.status = 0; sh.bad_block = LK_NONE; *fallback = 0u; sh.why = 0; }
    if (tid < LK_PARSERS) {
        LkRing& rg = sh.ring[tid];
        rg.produced = 0;
        for (int c = 0; c <= LK_LITS; c++) rg.done[c] = 0;
        for (uint32_t i = 0; i < LK_RING; i++) { rg.count[i] = 0; rg.block[i] = 0; for (int w = 0; w < LK_LITS; w++) rg.lit_prog[i][w] = 0; }
    }
    // the history in front of this slab of the frame (at most 64 KiB matter) goes into the window below position 0
    const uint32_t h = hist0 > 65535 ? 65535u : (uint32_t)hist0;
    for (uint32_t i = tid; i < h; i += 64 * LK_WAVES) sh.win[lk_widx(0u - h + i)] = dst[(int64_t)i - (int64_t)h];
    __syncthreads();
    if (n == 0) return;
    if (wave < LK_PARSERS) {
        __builtin_amdgcn_s_setprio(2);
        lk_parser(sh, sh.ring[wave], wave, frame, table, n, block_size, dst_cap);
        __builtin_amdgcn_s_setprio(0);
    } else if (wave == LK_PARSERS) {
        __builtin_amdgcn_s_setprio(3);                                       // the chain wave is the critical path of a linked frame
        lk_chain(sh, dst, table, n, block_size, dst_cap, hist0, fallback);
        __builtin_amdgcn_s_setprio(0);
    } else {
        lk_literals(sh, wave - LK_PARSERS - 1, frame, table, dst, n, block_size, dst_cap);
    }
    __syncthreads();
    const int32_t st = (int32_t)uni((uint32_t)sh.status);
    if (tid == 0) { fallback[1] = st < 0 ? sh.why : 0u; fallback[2] = sh.bad_block; }
    if (st < 0 && st != -3) {
        uint32_t bb = uni(sh.bad_block); if (bb >= n) bb = 0;
        for (uint32_t k = bb + tid; k < n; k += 64 * LK_WAVES) table[k].dst_size = (k == bb) ? (uint32_t)st : 0u;
    }
}

}  // namespace lz4f
