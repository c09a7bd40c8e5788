// decode_fused.cuh -- fused LZ4 block decode for large blocks: one WORKGROUP (8 waves) per frame block,
// parse and copy overlapped.
//
// The sequence parse of a block is one serial dependent chain (~0.6 us per sequence on one wave); the copies are
// where the bytes are.  Running them back to back (a parse kernel, then a copy kernel) costs parse + copy; here they overlap, so a block
// costs max(parse, copy) ~ parse:
//   wave 0        PARSER.  Wave-uniform (SALU) state machine.  The payload streams through two 8 KiB LDS stages
//                 filled by direct-to-LDS loads (global_load_lds_dwordx4) one stage ahead, one 8-byte LDS read per
//                 sequence, length bytes decoded with ctz(~window).  64 descriptors are gathered in VGPR lanes and
//                 written as one slot of an LDS ring; `produced` is bumped.
//   waves 1..7    COPIERS.  Wave c takes ring slots c-1, c+6, ... : (1) literals of the slot, HBM -> HBM, 16 B/lane,
//                 4 copies in flight; (2) waits until the matches of the previous slot are complete (`match_done`),
//                 then replays its 64 matches in stream order, again up to 4 in flight when their sources cannot
//                 touch each other's destinations; (3) s_waitcnt vmcnt(0), bumps `match_done`.
// Waves of a workgroup are co-resident by construction, so the hand-offs are plain LDS words polled with s_sleep --
// no workgroup barrier inside the loop, hence the parser never waits for a copier except when the ring is full.
// Literal copies run ahead freely; match phases form a chain slot -> slot, which for 64-sequence slots is well
// below the parse time.  LDS: 8 KiB ring + 16.1 KiB stages ~ 24.3 KiB per workgroup: 4 workgroups (32 waves) per CU.
#pragma once
#include "common.cuh"
#include "decode.cuh"

namespace lz4f {

// Two shapes of the same workgroup.  Big blocks (one block per workgroup slot of the machine): 8 waves, 4 workgroups
// per CU.  Blocks of 1 MiB and less (more blocks than slots): 4 waves and a third of the LDS, 8 workgroups per CU -
// three copiers keep up with one parser, and twice as many blocks are in flight.
template <int W> struct FzCfg;
template <> struct FzCfg<8> { static constexpr int WAVES = 8; static constexpr uint32_t STAGE = 8192, RING = 8, LDS_BUDGET = 40960; };
template <> struct FzCfg<4> { static constexpr int WAVES = 4; static constexpr uint32_t STAGE = 2048, RING = 4, LDS_BUDGET = 20480; };
// the shapes of the self-feeding copy kernel (decode_indexed.cuh: k_copy_selffed): no payload stages (the area holds a ring of output positions
// instead), and a descriptor ring twice as deep - its first wave parses in rounds, and the copiers live off the ring meanwhile
struct FzCfgS8 { static constexpr int WAVES = 8; static constexpr uint32_t STAGE = 4096 - 576, RING = 16, LDS_BUDGET = 40960; };
struct FzCfgS4 { static constexpr int WAVES = 4; static constexpr uint32_t STAGE = 2048, RING = 8, LDS_BUDGET = 20480; };
constexpr uint32_t FZ_OVER = 576;                // >= the parser's 520-byte register window + 8
#ifndef FZ_FED_ROUNDS
#define FZ_FED_ROUNDS 1                          // gather rounds per register set in the fed copiers (two sets in flight)
#endif
#ifndef FZ_STORE_SC1          // how the fed copiers store output: 0 non-temporal, 1 sc1, 2 sc0 sc1 (write-through, dropped from L2), 3 plain.  Measured (round 3, copy kernel of the bench decode): 1.38 / 1.56 / 1.56 / 1.57 ms
#define FZ_STORE_SC1 0
#endif
#ifndef FZ_FED_OCC
#define FZ_FED_OCC 8
#endif
constexpr int      FZ_MATCH_SET = 3;
constexpr uint32_t FZ_SRC_BIAS = 1u << 22;       // direct matches: payload position relative to the block's payload + this (== IX_SRC_BIAS)
constexpr uint32_t FZ_PEND = 32;                // linked, fed: matches set aside until the block in front is done
// How long a workgroup of a linked frame waits for the one in front before it gives up (-> generic decoder): a safety net (they
// are dispatched in order) that has to outlast a whole dense frame whose groups run one after the other.  The host derives it from
// the frame's output size (engine.hip: fz_wait_budget - half a second plus 20 ticks of the 100 MHz clock per byte, a fifth of the
// slowest chain measured) and hands it to the kernel, which keeps it in LDS (FzShared::wait_lo/hi).
__device__ __forceinline__ bool fz_wait_expired(uint64_t t0, uint64_t budget) { return __builtin_amdgcn_s_memrealtime() - t0 > budget; }

template <class C>
struct alignas(16) FzShared {
    uint4    ring[C::RING][64];
    uint8_t  stage[2][C::STAGE + FZ_OVER];
    uint4    gt[C::WAVES - 1][2][64];           // per copier wave: the gather tables of its slot (see the literal phase)
    uint32_t produced;                          // slots the parser has published
    uint32_t total_slots;                       // valid once `finished` is set
    uint32_t last_count;                        // descriptors in the last slot
    uint32_t per_slot;                          // parser: descriptors in every other slot (64, or 8 for a small block)
    uint32_t finished;                          // parser is done (ok or not)
    uint32_t match_done;                        // slots that are complete, in slot order (advanced over slot_done by whoever finishes)
    uint32_t slot_done[C::RING];                // [slot % RING] == slot + 1: that slot's copies are all in memory
    uint32_t slot_cnt[C::RING];                 // fed descriptors: how many of them the slot holds (small blocks are cut into shorter slots)
    uint32_t pend_n, prev_ready, own_front;     // own_front: bytes of output in front of this block that are this workgroup's own (a group of linked blocks)
    uint32_t wait_lo;                           // (with wait_hi) the budget of a wait for the workgroup in front, in ticks of the 100 MHz clock
    // linked frames through the index: matches waiting for the block in front (see fz_copier)
    uint32_t pend_dst[FZ_PEND], pend_len[FZ_PEND], pend_off[FZ_PEND];
    int32_t  status;                            // < 0: malformed block
    uint32_t out_size;
    uint32_t wait_hi;
};
static_assert(sizeof(FzShared<FzCfg<8>>) <= FzCfg<8>::LDS_BUDGET && sizeof(FzShared<FzCfg<4>>) <= FzCfg<4>::LDS_BUDGET, "workgroups per CU");
static_assert(sizeof(FzShared<FzCfgS8>) <= FzCfgS8::LDS_BUDGET && sizeof(FzShared<FzCfgS4>) <= FzCfgS4::LDS_BUDGET, "workgroups per CU");

__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void lds_poke(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }

template <class C>
__device__ __forceinline__ void fz_stage_issue(uint8_t* slot, const uint8_t* __restrict__ in, uint32_t csize, uint32_t s)
{
    const uint32_t lane = lane_id();
    const uint32_t base = s * C::STAGE;
    if (base >= csize) return;
    const uint32_t end = (base + C::STAGE + FZ_OVER < csize) ? base + C::STAGE + FZ_OVER : csize;
    const uint32_t span = end - base;
    for (uint32_t piece = 0; piece < span; piece += 1024) {
        const uint32_t o = piece + lane * 16;
        if (o + 16 <= span) __builtin_amdgcn_global_load_lds((gptr_t)(in + base + o), (lptr_t)(slot + piece), 16, 0, 0);
    }
    const uint32_t tail0 = span & ~15u;
    if (lane < span - tail0) slot[tail0 + lane] = in[base + tail0 + lane];
}

// ---------------- parser wave ----------------
template <class C>
__device__ __forceinline__ void fz_parser(FzShared<C>& sh, const uint8_t* __restrict__ in, uint32_t csize, uint32_t cap, uint64_t hist,
                                          unsigned long long* prof)
{
#ifdef FZ_PROF
    const unsigned long long t_begin = clock64(); unsigned long long t_ring = 0;
#endif
    const uint32_t lane = lane_id();
    uint8_t* stages = &sh.stage[0][0];
    uint32_t nseq = 0, status = 0;
    uint32_t op = 0;
    int32_t  cur = -1;
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;

    if (csize == 0) status = 1;
    else fz_stage_issue<C>(stages, in, csize, 0);
    auto need = [&](uint32_t qq) {
        const int32_t s = (int32_t)(qq / C::STAGE);
        if (s == cur) return;
        // next stage in sequence: its DMA is in flight -> wait, then start the one after.  A jump further ahead (a very
        // long literal run) first has to start its own stage.
#pragma unroll 1
        for (int32_t it = (s == cur + 1) ? 1 : 0; it < 2; it++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fz_stage_issue<C>(stages + (uint32_t)((s + it) & 1) * (C::STAGE + FZ_OVER), in, csize, (uint32_t)(s + it));
        }
        cur = s;
    };
    // Register window: lane l holds the 16 payload bytes at wb + 8*l, so any 8-byte read inside [wb, wb+512) is four
    // v_readlane + a funnel shift, no LDS round trip.  A miss (the jump over a long literal run; none for dozens of
    // short sequences) reloads the window at the wanted position with one ds_read2_b64.
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    v4u_t win = {0u, 0u, 0u, 0u};
    uint32_t wb = 0xFFFFFC00u;                       // nothing loaded yet: every position misses
    auto reload = [&](uint32_t qq) {
        need(qq);
        const uint32_t s = qq / C::STAGE;
        const uint32_t o = (qq - s * C::STAGE) & ~7u;
        const uint32_t addr = (uint32_t)(uintptr_t)(lptr_t)(stages + (s & 1) * (C::STAGE + FZ_OVER)) + o + 8u * lane;
        // inline asm: keeps hipcc from draining vmcnt (the pending stage prefetch) before this LDS read
        asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(win) : "v"(addr) : "memory");
        wb = qq & ~7u;
    };
    auto fetch = [&](uint32_t qq) -> uint64_t {
        uint32_t rel = qq - wb;
        if (rel >= 504u) { reload(qq); rel = qq - wb; }                    // lanes 0..62 serve reads at rel 0..503
        const uint32_t l = rel >> 3, sh8 = (rel & 7u) * 8u;
        const uint64_t lo = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.x, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.y, l) << 32);
        const uint64_t hi = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.z, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.w, l) << 32);
        return (lo >> sh8) | ((hi << 1) << (63u - sh8));
    };
    // length bytes that do not end inside the 8-byte read that found them (runs of >= 6 x 0xFF): rare, byte by byte
    auto ext_slow = [&](uint32_t pos, uint32_t& after) -> uint32_t {
        uint32_t add = 0;
        for (;;) {
            if (pos >= csize || add > 0x7FFF0000u) { status = 1; after = pos; return add; }
            const uint32_t b = uni((uint32_t)in[pos]);
            add += b; pos++;
            if (b != 255) { after = pos; return add; }
        }
    };
    // The serial part of a sequence is only what the NEXT token's position depends on - the two lengths - and it is written for
    // the scalar unit, which a wave reaches once every ~4.7 cycles whatever the instruction (tools/probe/chain_rates.hip): every
    // instruction on the chain is 2 ns per sequence.  So the chain records four numbers per sequence (literal source, literal
    // length, output position, match length: one v_writelane each) and leaves the rest to `finish_slot`, where the 64 lanes do
    // it for 64 sequences at once: fetch the offset (2 bytes behind the literals, straight from the payload in L2), apply the
    // output-side rules, pack the descriptor.  Input-side rules stay on the chain: they are what keeps it inside the payload.
    uint32_t v_src = 0, v_lit = 0, v_op = 0, v_ml = 0;
    // descriptors per ring slot: 64 - or 8 for a small block (under 96 KiB of payload: the streaming API's 64 KiB block is 64 sequences of the
    // bench input), whose one slot of 64 would reach ONE copier wave, and only when the whole block is parsed: parse 40 us, then copy 25
    const uint32_t plog = csize < (96u << 10) ? 3u : 6u, pmask = (1u << plog) - 1u;
    if (lane == 0) sh.per_slot = pmask + 1;                                   // (read by the copiers with the first slot: a wave's LDS operations are performed in order)
    const uint32_t hist_reach = hist > 65536 ? 65536u : (uint32_t)hist;     // offsets are <= 65535: more history is not distinguishable
    // rules that need the offset / the output position; returns false if a sequence of the slot breaks one
    auto finish_slot = [&](uint32_t count) -> bool {
        uint32_t off = 0;
        bool bad = false;
        if (lane < count) {
            const uint64_t room = (uint64_t)cap - v_op;                          // (op <= cap is kept by the chain)
            if (v_ml) {
                const uint8_t* q = in + v_src + v_lit;                           // (the chain checked lit + 8 <= bytes left)
                off = (uint32_t)q[0] | ((uint32_t)q[1] << 8);
                bad = off == 0 || off > v_op + v_lit + hist_reach || (uint64_t)v_lit + v_ml + 5 > room;
            } else bad = v_lit > room;
        }
        d0 = v_src | ((off & 0xFFu) << 24);
        d1 = v_lit | ((off >> 8) << 24);
        d2 = v_op;
        d3 = v_ml;
        return __ballot(bad) == 0;
    };
    // publish the gathered descriptors as ring slot number `slot_idx`
    auto publish = [&](uint32_t slot_idx) {
#ifdef FZ_PROF
        const unsigned long long tr = clock64();
#endif
        while (slot_idx >= lds_peek(&sh.match_done) + C::RING) __builtin_amdgcn_s_sleep(8);    // ring full: wait for the oldest slot
#ifdef FZ_PROF
        t_ring += clock64() - tr;
#endif
        sh.ring[slot_idx % C::RING][lane] = uint4{d0, d1, d2, d3};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_poke(&sh.produced, slot_idx + 1);
    };
    // the first 8 bytes of the window lane that holds `qq` (enough for a token; `fetch` gives 8 bytes from qq on)
    auto fetch_byte = [&](uint32_t qq) -> uint32_t {
        uint32_t rel = qq - wb;
        if (rel >= 504u) { reload(qq); rel = qq - wb; }
        const uint32_t l = rel >> 3, sh8 = (rel & 7u) * 8u;
        const uint64_t lo = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.x, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.y, l) << 32);
        return (uint32_t)(lo >> sh8) & 0xFFu;
    };

    uint32_t pos = 0;
    uint32_t fin = status;                                                  // bit 0 error, bit 1 last sequence done
    while (fin == 0) {
        // ---- token, literal length ----
        const uint32_t token = fetch_byte(pos);
        uint32_t lit = token >> 4;
        uint32_t p = pos + 1;                                                // first literal byte
        uint32_t bad = 0;
        if (lit == 15) {
            const uint64_t x = fetch(pos) >> 8;                              // 7 candidate length bytes, top byte 0 (never 0xFF)
            const uint32_t f = (uint32_t)__builtin_ctzll(~x);
            const uint32_t k = f >> 3;
            lit = 15u + 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
            p = pos + 2 + k;
            if (k == 7) { lit = 15u + ext_slow(pos + 1, p); bad |= status; }
        }
        bad |= p > csize ? 1u : 0u;
        const uint32_t in_left = csize - p;
        const bool is_last = lit > in_left || lit + 8 > in_left || lit + 12 > cap - op;   // lit < 2^31
        uint32_t mlen = 0, npos = pos;
        if (!is_last) {
            // ---- match length (the offset itself is finish_slot's) ----
            const uint32_t qo = p + lit;
            mlen = token & 15;
            npos = qo + 2;
            if (mlen == 15) {
                const uint64_t x = fetch(qo) >> 16;                          // 6 candidate length bytes
                const uint32_t f = (uint32_t)__builtin_ctzll(~x);
                const uint32_t k = f >> 3;
                mlen = 15u + 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
                npos = qo + 3 + k;
                if (k == 6) { mlen = 15u + ext_slow(qo + 2, npos); bad |= status; }
                bad |= npos + 4 >= csize ? 1u : 0u;
            }
            mlen += 4;
            bad |= mlen > cap - (op + lit) ? 1u : 0u;                         // (keeps op <= cap; the exact rule is finish_slot's)
        } else {
            bad |= lit != in_left ? 1u : 0u;
            bad |= lit > cap - op ? 1u : 0u;
        }
        fin |= bad;
        if (fin == 0) {                                                      // record it
            const uint32_t slot = nseq & pmask;
            // (gfx9: one scalar operand per VALU instruction, so the lane number travels in M0 - which is the compiler's, hence kept)
            uint32_t keep;
            asm("s_mov_b32 %4, m0\n\ts_mov_b32 m0, %5\n\tv_writelane_b32 %0, %6, m0\n\tv_writelane_b32 %1, %7, m0\n\tv_writelane_b32 %2, %8, m0\n\tv_writelane_b32 %3, %9, m0\n\ts_mov_b32 m0, %4"
                : "+v"(v_src), "+v"(v_lit), "+v"(v_op), "+v"(v_ml), "=&s"(keep) : "s"(slot), "s"(p), "s"(lit), "s"(op), "s"(mlen));
            nseq++;
            op += lit + mlen;
            if (slot == pmask) {
                if (finish_slot(pmask + 1)) publish((nseq >> plog) - 1);
                else { fin |= 1u; nseq -= pmask + 1; }                        // (nothing of a slot with a bad sequence is published)
            }
            pos = npos;
            fin |= is_last ? 2u : (pos >= csize ? 1u : 0u);
        }
    }
    if ((fin & 1u) == 0 && (nseq & pmask) && !finish_slot(nseq & pmask)) fin |= 1u;    // the partial last slot
    if (fin & 1u) status = 1;
    // final bookkeeping.  Order matters for the partial last slot: its size must be readable by whoever sees it
    // published, so totals and `finished` are written BEFORE `produced` is bumped for it (LDS ops of one wave are
    // performed in order; a copier reads `produced` first, then `finished`).
    const uint32_t full = nseq >> plog, part = (status == 0) ? (nseq & pmask) : 0;
    if (part) {
        while (full >= lds_peek(&sh.match_done) + C::RING) __builtin_amdgcn_s_sleep(8);
        sh.ring[full % C::RING][lane] = uint4{d0, d1, d2, d3};
    }
    sh.total_slots = status ? lds_peek(&sh.produced) : full + (part ? 1u : 0u);       // (all lanes store the same values)
    sh.last_count = part ? part : pmask + 1;
    sh.out_size = op;
    sh.status = status ? -1 : 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_poke(&sh.finished, 1u);
    if (part) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lds_poke(&sh.produced, full + 1); }
#ifdef FZ_PROF          // (the cycle stamps of the parser: a build switch, because they cost the loop six scalar registers)
    if (prof && blockIdx.x == 0 && lane == 0) { prof[0] = clock64() - t_begin; prof[1] = t_ring; prof[2] = nseq; }
#endif
}

// ---------------- feeder wave (indexed decode, decode_indexed.cuh) ----------------
// The block's descriptors already exist in HBM (k_parse_indexed wrote them, one lane per index entry): wave 0 only moves
// them into the ring, 64 at a time, and checks what needs output positions -- they must tile [0, size) without gaps,
// stay inside the block's capacity, and every match must start at or behind the first byte of the block.
// Descriptor word 3 carries, besides the match length, the DIRECT flag (bit 31): the match's bytes are known to sit in
// the payload at the 23-bit position that then replaces the offset, so the copiers treat it like a literal run.
template <class C>
__device__ __forceinline__ void fz_feeder(FzShared<C>& sh, const SeqDesc* __restrict__ desc, const uint32_t* __restrict__ dsrc, uint32_t nseq,
                                          uint32_t csize, uint32_t cap, uint64_t hist, uint64_t pay_before)
{   // hist: output bytes in front of the block that a match may reach (linked frames); pay_before: frame bytes in front of the payload
    const uint32_t lane = lane_id();
    // descriptors per slot: 64, fewer for a block so small that whole slots would leave copier waves without work
    // (a 64 KiB block of 1 KiB sequences is ONE slot of 64: one busy wave per workgroup)
    uint32_t per = (nseq + (C::WAVES - 1) - 1) / (C::WAVES - 1);
    per = per > 64 ? 64u : (per < 8 ? 8u : per);
    const uint32_t nslots = (nseq + per - 1) / per;
    uint32_t expect = 0, status = nseq ? 0u : 1u, published = 0;
    uint4 nxt = {0u, 0u, 0u, 0u};
    uint32_t nxt_src = 0xFFFFFFFFu;
    if (lane < per && lane < nseq) { nxt = ((const uint4*)desc)[lane]; if (dsrc) nxt_src = dsrc[lane]; }
    for (uint32_t slot = 0; slot < nslots && !status; slot++) {
        const uint32_t first = slot * per;
        const uint32_t count = (nseq - first < per) ? nseq - first : per;
        uint4 d = nxt;
        if (nxt_src < (1u << 23) && !(d.w >> 31) && (d.w & 0xFFFFFFu))      // k_resolve_direct found the match's bytes in the payload
            d = uint4{(d.x & 0xFFFFFFu) | ((nxt_src & 0xFFu) << 24), (d.y & 0xFFFFFFu) | (((nxt_src >> 8) & 0xFFu) << 24), d.z,
                      (d.w & 0xFFFFFFu) | 0x80000000u | ((nxt_src >> 16) << 24)};
        const uint32_t nx = first + per + lane;
        nxt = uint4{0u, 0u, 0u, 0u}; nxt_src = 0xFFFFFFFFu;
        if (lane < per && nx < nseq) { nxt = ((const uint4*)desc)[nx]; if (dsrc) nxt_src = dsrc[nx]; }    // next slot's descriptors travel while this one is checked and published
        const uint32_t p = d.x & 0xFFFFFFu, lit = d.y & 0xFFFFFFu, op = d.z, ml = d.w & 0xFFFFFFu;
        const bool direct = (d.w >> 31) != 0;
        const uint32_t f24 = (d.x >> 24) | ((d.y >> 24) << 8) | (((d.w >> 24) & 0x7Fu) << 16);
        const uint64_t dm = (uint64_t)op + lit, end = dm + ml;
        const uint32_t prev_end = __shfl_up((uint32_t)end, 1);
        const bool last = first + lane + 1 == nseq;
        bool bad = op != (lane == 0 ? expect : prev_end) || (uint64_t)p + lit > csize || end > cap;
        if (last) bad |= ml != 0 || direct;
        else {
            bad |= ml < 4 || end + 5 > cap;
            const int64_t rel = (int64_t)f24 - (int64_t)FZ_SRC_BIAS;             // direct: payload position relative to this block's payload
            bad |= direct ? (rel + (int64_t)ml > (int64_t)csize || -rel > (int64_t)pay_before) : (f24 == 0 || f24 > 65535u || f24 > dm + hist);
        }
        if (__ballot(lane < count && bad)) { status = 1; break; }
        expect = __shfl((uint32_t)end, (int)count - 1);
        while (slot >= lds_peek(&sh.match_done) + C::RING && (int32_t)lds_peek((const uint32_t*)&sh.status) >= 0) __builtin_amdgcn_s_sleep(8);   // ring full: wait for the oldest slot
        if ((int32_t)lds_peek((const uint32_t*)&sh.status) < 0) { status = 1; break; }          // (a copier gave up: previous block of a linked frame failed)
        sh.ring[slot % C::RING][lane] = d;
        sh.slot_cnt[slot % C::RING] = count;
        if (slot + 1 == nslots) break;                                       // the last slot is published after the totals (see fz_parser)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_poke(&sh.produced, slot + 1);
        published = slot + 1;
    }
    sh.total_slots = status ? published : nslots;
    sh.last_count = status ? 64u : nseq - (nslots - 1) * per;
    sh.out_size = expect;
    sh.status = status ? -1 : 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_poke(&sh.finished, 1u);
    if (!status) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lds_poke(&sh.produced, nslots); }
}

// ---------------- copier waves ----------------
template <class C, bool FED = false>
__device__ __forceinline__ void fz_copier(FzShared<C>& sh, const uint8_t* __restrict__ in, uint8_t* out, uint32_t cw /* 0..6 */,
                                          const uint8_t* safe /* 16 readable bytes */, unsigned long long* prof, const uint32_t* prev_done = nullptr)
{
    const uint32_t lane = lane_id();
    unsigned long long t_wait_p = 0, t_lit = 0, t_wait_m = 0, t_match = 0, n_slots = 0, t_dep = 0, t_drain = 0;
    auto dump = [&]() { if (prof && FED && lane == 0) { atomicAdd(prof + 76, t_wait_p); atomicAdd(prof + 77, t_lit); atomicAdd(prof + 78, t_wait_m); atomicAdd(prof + 79, t_match); }
                        if (prof && blockIdx.x == 0 && lane == 0) { unsigned long long* o = prof + 8 * (cw + 1); o[0] = t_wait_p; o[1] = t_lit; o[2] = t_wait_m; o[3] = t_match; o[4] = n_slots; o[5] = t_dep; o[6] = t_drain; } };
    for (uint32_t slot = cw;; slot += C::WAVES - 1) {
        // wait for the slot (or for the end of the block)
        const unsigned long long c0 = clock64();
        for (;;) {
            if (lds_peek(&sh.produced) > slot) break;
            // the parser sets `finished` BEFORE it publishes a partial last slot: "finished and not yet produced" is not the
            // end for that slot - total_slots (written before `finished`) says whether it is still to come
            if (lds_peek(&sh.finished) && slot >= lds_peek(&sh.total_slots)) { dump(); return; }
            __builtin_amdgcn_s_sleep(4);
        }
        const unsigned long long c1 = clock64(); t_wait_p += c1 - c0; n_slots++;
        __builtin_amdgcn_s_setprio(1);                                       // copying beats polling
        const uint4 d = sh.ring[slot % C::RING][lane];
        // descriptors in this slot: 64, except a partial last slot -- which is published only after `finished`
        uint32_t count = FED ? 64u : lds_peek(&sh.per_slot);                // (parser: 64 per slot, 8 for a small block)
        if (FED) count = lds_peek(&sh.slot_cnt[slot % C::RING]);
        else if (lds_peek(&sh.finished) && slot + 1 == lds_peek(&sh.total_slots)) count = lds_peek(&sh.last_count);
        count = uni(count);
        const uint32_t vsrc = d.x & 0xFFFFFFu, vlen = d.y & 0xFFFFFFu, vdst = d.z, vml = FED ? (d.w & 0xFFFFFFu) : d.w, voff = (d.x >> 24) | ((d.y >> 24) << 8);
        // fed descriptors: a DIRECT match is one more copy out of the payload (position in the bits of the offset + 7 more)
        const bool vdirect = FED && lane < count && (d.w >> 31) != 0;
        const uint32_t vmsrc = (voff | (((d.w >> 24) & 0x7Fu) << 16)) - FZ_SRC_BIAS;       // signed: may point into the payload of the block before

        // ---- (1) literal runs (and direct matches): no dependencies.  A gather with no scalar work per run -- a CU has ONE
        // scalar unit, and a copy loop that spends ~100 scalar instructions per run is bound by it.  The runs of the slot are
        // cut into 16-byte units; a prefix sum over the lanes gives every run its first unit; lane l of round r takes unit
        // 64 r + l, finds the run it belongs to with a six-step binary search over the 64 prefix values (LDS), and moves its
        // 16 bytes (the last unit of a run overlaps the one before).  Runs shorter than 16 bytes go the same way byte by byte.
        {
            uint4* T0 = sh.gt[cw][0];
            uint4* T1 = sh.gt[cw][1];
            const uint32_t L = lane < count ? vlen : 0u, M = vdirect ? vml : 0u;
            auto scan = [&](uint32_t v, uint32_t& total) -> uint32_t {              // exclusive prefix sum over the wave
                uint32_t incl = v;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t t = __shfl_up(incl, sft); if ((int)lane >= sft) incl += t; }
                total = __builtin_amdgcn_readlane(incl, 63);
                return incl - v;
            };
            auto find = [&](uint32_t u) -> uint32_t {                               // last run whose first unit is <= u
                uint32_t j = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) { const uint32_t c = j + step; if (T0[c].x <= u) j = c; }
                return j;
            };
            // pass A: runs of >= 16 bytes
            uint32_t total;
            {
                const uint32_t uL = L >= 16 ? (L + 15) >> 4 : 0u, uM = M >= 16 ? (M + 15) >> 4 : 0u;
                const uint32_t P = scan(uL + uM, total);
                if (total) {
                    T0[lane] = uint4{P, uL, L, M};
                    T1[lane] = uint4{vsrc, vmsrc, vdst, 0u};
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
            }
            if (total) {
                auto ld = [&](Piece& pc, uint32_t u, uint32_t& dof, bool& act) {
                    const uint32_t j = find(u);
                    const uint4 e0 = T0[j], e1 = T1[j];
                    uint32_t r = u - e0.x;
                    const bool isM = r >= e0.y;
                    r -= isM ? e0.y : 0u;
                    const uint32_t len = isM ? e0.w : e0.z;
                    uint32_t off = r * 16u;
                    off = (off + 16u > len) ? len - 16u : off;
                    act = u < total;
                    dof = e1.z + (isM ? e0.z : 0u) + off;
                    const uint8_t* a = act ? in + (int32_t)((isM ? e1.y : e1.x) + off) : safe;
                    const v4u_ua t = *(const v4u_ua*)a;
                    pc.a = t.a; pc.b = t.b; pc.c = t.c; pc.d = t.d;
                };
                // fed descriptors: what is written here is not read again (matches come out of the payload), so the stores
                // bypass the caches and leave them to the payload, which is read twice
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef u32x4 u32x4_ua __attribute__((aligned(1)));
                auto st = [&](const Piece& pc, uint32_t dof, bool act) {
                    if (!act) return;
#if FZ_STORE_SC1 == 1
                    if (FED) { const u32x4 v = u32x4{pc.a, pc.b, pc.c, pc.d}; uint8_t* q = out + dof; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(q), "v"(v) : "memory"); }
#elif FZ_STORE_SC1 == 2
                    if (FED) { const u32x4 v = u32x4{pc.a, pc.b, pc.c, pc.d}; uint8_t* q = out + dof; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(q), "v"(v) : "memory"); }
#elif FZ_STORE_SC1 == 3
                    if (FED) *(v4u_ua*)(out + dof) = v4u_ua{pc.a, pc.b, pc.c, pc.d};
#else
                    if (FED) __builtin_nontemporal_store(u32x4{pc.a, pc.b, pc.c, pc.d}, (u32x4_ua*)(out + dof));
#endif
                    else *(v4u_ua*)(out + dof) = v4u_ua{pc.a, pc.b, pc.c, pc.d};
                };
                // ping-pong of G rounds each: loads are in flight while the previous set is stored
                constexpr int G = FED ? FZ_FED_ROUNDS : 1;
                Piece pa[G], pb[G];
                uint32_t da[G], db[G];
                bool xa[G], xb[G];
                uint32_t base = 0;
#pragma unroll
                for (int i = 0; i < G; i++) ld(pa[i], base + 64 * i + lane, da[i], xa[i]);
                base += 64 * G;
                for (;;) {
                    const bool more_b = base < total;
#pragma unroll
                    for (int i = 0; i < G; i++) ld(pb[i], base + 64 * i + lane, db[i], xb[i]);
                    base += 64 * G;
#pragma unroll
                    for (int i = 0; i < G; i++) st(pa[i], da[i], xa[i]);
                    if (!more_b) break;
                    const bool more_a = base < total;
#pragma unroll
                    for (int i = 0; i < G; i++) ld(pa[i], base + 64 * i + lane, da[i], xa[i]);
                    base += 64 * G;
#pragma unroll
                    for (int i = 0; i < G; i++) st(pb[i], db[i], xb[i]);
                    if (!more_a) break;
                }
            }
            // pass B: runs of 1..15 bytes, a lane per byte
            uint32_t total_b;
            {
                const uint32_t bL = L < 16 ? L : 0u, bM = M < 16 ? M : 0u;
                const uint32_t P = scan(bL + bM, total_b);
                if (total_b) {
                    T0[lane] = uint4{P, bL, L, M};
                    T1[lane] = uint4{vsrc, vmsrc, vdst, 0u};
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
            }
            for (uint32_t base = 0; base < total_b; base += 128) {
                uint8_t v[2]; uint32_t dof[2]; bool act[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t u = base + 64 * i + lane;
                    const uint32_t j = find(u);
                    const uint4 e0 = T0[j], e1 = T1[j];
                    uint32_t r = u - e0.x;
                    const bool isM = r >= e0.y;
                    r -= isM ? e0.y : 0u;
                    act[i] = u < total_b;
                    dof[i] = e1.z + (isM ? e0.z : 0u) + r;
                    const uint8_t* a = act[i] ? in + (int32_t)((isM ? e1.y : e1.x) + r) : safe;
                    v[i] = *a;
                }
#pragma unroll
                for (int i = 0; i < 2; i++) if (act[i]) out[dof[i]] = v[i];
            }
        }
        // ---- (2) matches, in stream order, after every earlier slot's matches ----
        const unsigned long long c2 = clock64(); t_lit += c2 - c1;
        __builtin_amdgcn_s_setprio(0);
        // (a slot without a match on the chain - all direct, or literals only - neither waits for its predecessors nor is waited for)
        const uint64_t on_chain = __ballot(lane < count && vml != 0 && !vdirect);
        if (on_chain) {
            while (lds_peek(&sh.match_done) < slot) {
                if ((int32_t)lds_peek((const uint32_t*)&sh.status) < 0) { dump(); return; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __builtin_amdgcn_sched_barrier(0); const unsigned long long c3 = clock64(); t_wait_m += c3 - c2; __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2);                                       // the match phases form the second serial chain of a block
        if (on_chain) {
            // Out-of-order replay.  A match may be copied as soon as every EARLIER match of this slot whose destination
            // overlaps its source has had its store issued (one wave's vector memory operations are performed in issue
            // order; earlier slots are complete, this slot's literals were written by this wave).  Later matches never
            // write below an earlier match's source, so nothing else orders them.  Each lane holds one match and the
            // bit mask `dep` of the earlier ones it reads from; `unstored` is wave-uniform; the lowest ready matches
            // fill two ping-pong register sets of NJ copies.  With in-order issue a match whose source is a few KiB back
            // stalled everything behind it for a memory round trip (~10 % of the matches at six in flight).
            constexpr int NJ = FZ_MATCH_SET;
            const uint32_t mdm = vdst + vlen;                                   // my match's destination, source = mdm - voff
            // signed: a source may start in the history in front of a linked block (negative) and end inside the block
            const int32_t ms0 = (int32_t)mdm - (int32_t)voff, ms1 = ms0 + (int32_t)vml;
            const bool has = lane < count && vml != 0 && !vdirect;             // direct matches went out with the literals
            const uint64_t fastmask = __ballot(has && vml >= 16 && vml <= 1024 && voff >= vml);   // one non-overlapping round
            // (only over the matches that are on the chain: a slot whose matches are all direct costs a ballot here)
            uint64_t todo = __ballot(has), unstored = todo;
            uint32_t dep_lo = 0, dep_hi = 0;
            for (uint64_t mm = todo; mm; mm &= mm - 1) {
                const uint32_t k = (uint32_t)__builtin_ctzll(mm);
                const uint32_t dk = __builtin_amdgcn_readlane(mdm, k), mk = __builtin_amdgcn_readlane(vml, k);
                const bool dep = k < lane && ms0 < (int32_t)(dk + mk) && ms1 > (int32_t)dk;
                if (k < 32) dep_lo |= dep ? (1u << k) : 0u; else dep_hi |= dep ? (1u << (k - 32)) : 0u;
            }
            __builtin_amdgcn_sched_barrier(0); t_dep += clock64() - c3; __builtin_amdgcn_sched_barrier(0);
            if (FED && prev_done && lds_peek(&sh.prev_ready) && __ballot(has && ms0 < 0)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (another wave saw the done word)
            if (FED && prev_done && !lds_peek(&sh.prev_ready)) {
                // Linked frame, and the block in front (another workgroup's) may not be finished.  A match that reads it - or reads
                // what such a match will write, directly or through others - is SET ASIDE: its destination goes on a short list,
                // every later chain match is checked against that list, and the list is replayed in order when the block in
                // front is far enough (k_copy_indexed, after this block's main pass).  Everything else runs through; waiting here instead would stop the
                // whole block after a ring's worth of slots (measured: 9.5 instead of ~3 ms for 4 GiB).
                const uint32_t np = lds_peek(&sh.pend_n);
                bool defer = has && ms0 < -(int32_t)lds_peek(&sh.own_front);      // reads another workgroup's output
                for (uint32_t q = 0; q < np; q++) {
                    const int32_t lo = (int32_t)sh.pend_dst[q], hi = lo + (int32_t)sh.pend_len[q];
                    defer |= has && ms0 < hi && ms1 > lo;
                }
                uint64_t dmask = __ballot(defer);
                while (dmask) {                                               // closure inside the slot
                    const bool more = has && !defer && (((dep_lo & (uint32_t)dmask) | (dep_hi & (uint32_t)(dmask >> 32))) != 0);
                    const uint64_t add = __ballot(more);
                    if (!add) break;
                    defer |= more; dmask |= add;
                }
                if (dmask) {
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(dmask);
                    if (np + cnt > FZ_PEND) {
                        // no room on the list: wait for the block in front after all (workgroups are dispatched in block order, so it
                        // is running or finished; the poll has a budget), replay what is on the list, go on without one
                        uint32_t v = 0;                                       // (1 = all of it done, 2 = failed; 3 = only its main pass: not enough here)
                        const uint64_t budget = (uint64_t)lds_peek(&sh.wait_lo) | ((uint64_t)lds_peek(&sh.wait_hi) << 32);
                        for (const uint64_t t0 = __builtin_amdgcn_s_memrealtime();;) {
                            v = __hip_atomic_load(prev_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (v == 1u || v == 2u || fz_wait_expired(t0, budget)) break;
                            __builtin_amdgcn_s_sleep(8);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (v != 1u) { lds_poke((uint32_t*)&sh.status, 0xFFFFFFFFu); dump(); return; }
                        for (uint32_t q = 0; q < np; q++) wave_copy_match(out + (int32_t)sh.pend_dst[q], sh.pend_off[q], sh.pend_len[q]);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        lds_poke(&sh.pend_n, 0u);
                        lds_poke(&sh.prev_ready, 1u);
                    } else {
                        if (defer) {
                            const uint32_t at = np + (uint32_t)__builtin_popcountll(dmask & ((1ull << lane) - 1ull));
                            sh.pend_dst[at] = mdm; sh.pend_len[at] = vml; sh.pend_off[at] = voff;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        lds_poke(&sh.pend_n, np + cnt);
                        todo &= ~dmask; unstored &= ~dmask;                   // not part of this slot's replay
                    }
                }
            }
            auto ready = [&]() -> uint64_t {
                return __ballot(has && ((dep_lo & (uint32_t)unstored) | (dep_hi & (uint32_t)(unstored >> 32))) == 0) & todo;
            };
            // take the lowest ready single-round matches into a register set
            auto select = [&](CopyJob (&J)[NJ], uint64_t rf) -> uint64_t {
                uint64_t taken = 0;
#pragma unroll
                for (int i = 0; i < NJ; i++) {
                    J[i] = CopyJob{safe, out, 0};
                    if (rf) {
                        const uint32_t c = (uint32_t)__builtin_ctzll(rf);
                        rf &= rf - 1; taken |= 1ull << c;
                        const uint32_t dm = __builtin_amdgcn_readlane(mdm, c);
                        J[i] = CopyJob{out + dm - __builtin_amdgcn_readlane(voff, c), out + dm, (uint32_t)__builtin_amdgcn_readlane(vml, c)};
                    }
                }
                todo &= ~taken;
                return taken;
            };
            while (todo) {
                uint64_t r = ready();
                if ((r & fastmask) == 0) {
                    // nothing pipelinable is ready (and nothing is in flight): the lowest ready match is an overlapping,
                    // long or tiny one -- generic copy, its stores are issued before anything that follows
                    const uint32_t c = (uint32_t)__builtin_ctzll(r);
                    wave_copy_match(out + __builtin_amdgcn_readlane(mdm, c), __builtin_amdgcn_readlane(voff, c), __builtin_amdgcn_readlane(vml, c));
                    todo &= ~(1ull << c); unstored &= ~(1ull << c);
                    continue;
                }
                CopyJob A[NJ], B[NJ];
                Piece PA[NJ], PB[NJ];
                uint64_t ma = select(A, r & fastmask), mb;
#pragma unroll
                for (int i = 0; i < NJ; i++) job_load(PA[i], A[i], safe);
                for (;;) {
                    mb = select(B, ready() & fastmask);                          // (set A still counts as unstored)
#pragma unroll
                    for (int i = 0; i < NJ; i++) job_load(PB[i], B[i], safe);
#pragma unroll
                    for (int i = 0; i < NJ; i++) job_store(A[i], PA[i]);
                    unstored &= ~ma;
                    if (mb == 0) break;
                    ma = select(A, ready() & fastmask);
#pragma unroll
                    for (int i = 0; i < NJ; i++) job_load(PA[i], A[i], safe);
#pragma unroll
                    for (int i = 0; i < NJ; i++) job_store(B[i], PB[i]);
                    unstored &= ~mb;
                    if (ma == 0) break;
                }
            }
        }
        // ---- (3) everything this slot wrote is in memory: let the next slot's matches go ----
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long c5 = clock64();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        t_drain += clock64() - c5;
        __builtin_amdgcn_sched_barrier(0);
        // this slot is done; `match_done` moves over every finished slot from the oldest unfinished one on.  Whoever finishes does
        // that (flag first, then look: in LDS's one order of operations either I see the predecessor's bump or it sees my flag),
        // so nobody waits for the order - a wave whose slot is not the oldest just leaves its flag and takes its next slot.
        lds_poke(&sh.slot_done[slot % C::RING], slot + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (;;) {
            const uint32_t md = lds_peek(&sh.match_done);
            if (lds_peek(&sh.slot_done[md % C::RING]) != md + 1) break;
            if (lane == 0) atomicCAS(&sh.match_done, md, md + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_setprio(0);
        t_match += clock64() - c3;
    }
}

// Decode one compressed block with the whole workgroup; returns decoded size or -1 (same value in all threads).
template <class C, bool FED = false>
__device__ __forceinline__ int32_t fz_decode_block(FzShared<C>& sh, const uint8_t* __restrict__ in, uint32_t csize, uint8_t* out, uint32_t cap, uint64_t hist,
                                                   const uint8_t* safe, unsigned long long* prof, const SeqDesc* __restrict__ fed = nullptr, uint32_t nfed = 0,
                                                   const uint32_t* __restrict__ fed_src = nullptr, uint64_t pay_before = 0, const uint32_t* prev_done = nullptr,
                                                   uint32_t own_front = 0, bool first_of_group = true)
{   // first_of_group == false: a later block of a group of linked blocks this workgroup decodes one after the other - the list of
    // set-aside matches is carried over (positions rebased by the caller), and sources in front of the block are this workgroup's own
    const uint32_t wave = uni(threadIdx.x >> 6);
    __syncthreads();                                                     // previous block's LDS state is dead
    if (threadIdx.x == 0) { sh.produced = 0; sh.total_slots = 0; sh.last_count = 64; sh.finished = 0; sh.match_done = 0; sh.status = 0; sh.out_size = 0; }
    if (threadIdx.x < C::RING) sh.slot_done[threadIdx.x] = 0;
    if (threadIdx.x == 0) { if (first_of_group) { sh.pend_n = 0; sh.prev_ready = 0; } sh.own_front = own_front; }
    __syncthreads();
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);                                   // the serial chain: win issue arbitration against the 7 copier waves of this SIMD
        if (FED) fz_feeder<C>(sh, fed, fed_src, nfed, csize, cap, hist, pay_before);
        else fz_parser<C>(sh, in, csize, cap, hist, prof);
        __builtin_amdgcn_s_setprio(0);
    }
    else fz_copier<C, FED>(sh, in, out, wave - 1, safe, prof, prev_done);
    __syncthreads();                                                     // all copies of this block are issued and complete
    const int32_t st = (int32_t)uni((uint32_t)sh.status);
    const uint32_t osz = uni(sh.out_size);
    return st < 0 ? -1 : (int32_t)osz;
}

template <class C>
__global__ __launch_bounds__(64 * C::WAVES, 8) void k_decode_blocks_fused(const uint8_t* __restrict__ frame, uint8_t* dst, uint64_t dst_cap,
                                                                          BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                                          uint32_t n_max, uint32_t linked, uint32_t block_size, uint64_t hist0,
                                                                          unsigned long long* prof, const uint32_t* __restrict__ only_if)
{
    __shared__ FzShared<C> sh;
    if (res->status != ST_OK) return;
    if (only_if && *only_if == 0) return;                                // (fallback launch behind decode_linked.cuh: nothing to do)
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t tid = threadIdx.x;
    if (linked && blockIdx.x != 0) return;
    uint32_t b = linked ? 0u : blockIdx.x;
    const uint32_t b_end = linked ? n : (b < n ? b + 1 : b);
    uint64_t out = 0;
    for (; b < b_end; b++) {
        const BlockOut e = table[b];
        const uint32_t csz = e.word & 0x7FFFFFFFu;
        const uint64_t at = linked ? out : e.dst_off;
        const uint32_t room = linked ? (uint32_t)((dst_cap - out < block_size) ? dst_cap - out : block_size) : e.dst_size;
        int32_t got;
        if (e.word >> 31) {                                              // stored block: all waves copy a slice
            if (csz > room) got = -2;
            else {
                const uint32_t per = (((csz + C::WAVES - 1) / C::WAVES) + 15) & ~15u;
                const uint32_t a = (tid >> 6) * per;
                if (a < csz) wave_copy_disjoint(dst + at + a, frame + e.src_off + a, (csz - a < per) ? csz - a : per);
                got = (int32_t)csz;
            }
        } else {
            got = fz_decode_block<C>(sh, frame + e.src_off, csz, dst + at, room, linked ? out + hist0 : 0, frame, prof);
        }
        if (tid == 0) { table[b].dst_off = at; table[b].dst_size = (uint32_t)got; }
        if (got < 0) {
            if (linked) for (uint32_t k = b + 1 + tid; k < n; k += 64 * C::WAVES) table[k].dst_size = 0;
            break;
        }
        out += (uint32_t)got;
        __syncthreads();
    }
}

}  // namespace lz4f
