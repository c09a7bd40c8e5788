// encode_solo.cuh -- pass E1 for the DETERMINISTIC mode (lz4f_mi355x_engine_set_deterministic, LZ4F_MI355X_DETERMINISTIC): equal input -> equal bytes.
// (replaces the inner block loop of LZ4F_compressUpdate, /root/reference/src/Codec/Compression/LZ4/Conduit.hsc:311, as encode.cuh does.)
//
// The workgroup-shared search of encode.cuh is fast because sixteen waves share one table and one window - and not reproducible for
// the same reason: what a probe finds depends on which wave got where first ("Deterministic parse").  Until round 4
// the deterministic mode ran that kernel with one wave per workgroup parsing in order: 10x slower.  This is the search this library
// had before the shared one (round 1): ONE WAVE per 64 KiB chunk with a hash table of its own (3840 x {u16 position, u8 tag}, 11.25 KiB
// of LDS: 13-14 waves per CU), the chunk parsed in order - 64 positions per probe step, two steps in flight, the stride growing after
// misses, candidates verified and extended against the input in memory (through L2: the window is not in LDS here).  Nothing is
// shared between waves, so the records are a function of the input alone.  The history in front of a chunk (64 KiB, or back to the
// block's start) is seeded into the table sparsely, its last KiB densely, so matches reach back across chunk boundaries.
// Same record and ChunkInfo format as pass E1 of encode.cuh: passes S and E2 do not know the difference.
#pragma once
#include "encode.cuh"

namespace lz4f {

constexpr uint32_t SOLO_HASH_LOG = 12, SOLO_HASH_SIZE = (15u << SOLO_HASH_LOG) >> 4;      // 15/16 of 4096 entries: what fits 14 waves into a CU
constexpr uint32_t SOLO_TAG_BITS = 8, SOLO_TAG_MASK = 0xFFu;
constexpr uint32_t SOLO_SEED_STRIDE = 4, SOLO_SEED_DENSE = 1024;
#ifndef SOLO_STEP_LONG
#define SOLO_STEP_LONG 4
#endif
constexpr uint32_t SOLO_LONG_MATCH = 192;
__device__ __forceinline__ uint32_t solo_slot(uint32_t hv) { return ((hv >> (32 - SOLO_HASH_LOG)) * 15u) >> 4; }

// grid: one wave per chunk (blockDim = 64 * WAVES_PER_WG)
template <int WAVES_PER_WG>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_find_matches_solo(const uint8_t* __restrict__ src, EncGeom g,
                                                                         ChunkInfo* __restrict__ info, uint64_t* __restrict__ recs)
{
    __shared__ uint16_t s_table[WAVES_PER_WG][SOLO_HASH_SIZE];
    __shared__ uint8_t s_tag[WAVES_PER_WG][SOLO_HASH_SIZE];     // 8 more hash bits per entry: filters false candidates without touching memory
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t chunk = uni(blockIdx.x * WAVES_PER_WG + wave);
    if (chunk >= g.n_chunks) return;
    uint16_t* table = s_table[wave];
    uint8_t* tags = s_tag[wave];

    const uint32_t blk = chunk / g.chunks_per_block, cib = chunk % g.chunks_per_block;
    const uint64_t bstart = g.first_off + (uint64_t)blk * g.block_size;
    const uint64_t bend_abs = (bstart + g.block_size < g.src_size) ? bstart + g.block_size : g.src_size;
    const uint64_t cs_abs = bstart + (uint64_t)cib * g.chunk_size;
    ChunkInfo* ci = info + chunk;
    if (cs_abs >= bend_abs) {               // chunk beyond a short last block
        if (lane == 0) { ci->nrec = 0; ci->first_lit = 0; ci->tail_lit = 0; ci->body_size = 0; }
        return;
    }
    const uint64_t ce_abs = (cs_abs + g.chunk_size < bend_abs) ? cs_abs + g.chunk_size : bend_abs;
    const uint64_t low_abs = g.linked ? 0 : bstart;                  // matches may not start before this
    const uint32_t back = (uint32_t)((cs_abs - low_abs < 65536u) ? (cs_abs - low_abs) : 65536u);
    const uint8_t* base = src + (cs_abs - back);                     // position 0
    const uint8_t* rd_end = src + g.src_size;                        // nothing is read at or beyond this
    const uint32_t cs = back, ce = back + (uint32_t)(ce_abs - cs_abs);
    const uint32_t bend = back + (uint32_t)(bend_abs - cs_abs);
    // (this chunk's list: a place of its own in the pool - the deterministic mode sizes the pool for the worst case, engine.hip)
    const uint64_t rec_at = (uint64_t)chunk * g.max_rec_per_chunk;
    const bool rec_room = rec_at + g.max_rec_per_chunk <= g.rec_pool;
    uint64_t* rec = rec_pool_of(recs, g) + (rec_room ? rec_at : 0);
    if (lane == 0) rec_offs(recs)[chunk] = (uint32_t)(rec_room ? rec_at : 0);

    // clear + pre-seed the table with the history in front of the chunk.  A 4096-entry table cannot hold
    // 64 KiB of positions, and later inserts win: seed the whole window sparsely (every SEED_STRIDE-th position,
    // roughly the density the skip-accelerated search itself leaves behind), then the last SEED_DENSE bytes densely.
    for (uint32_t i = lane; i < SOLO_HASH_SIZE / 2; i += WAVE) ((uint32_t*)table)[i] = 0;
    for (uint32_t i = lane; i < SOLO_HASH_SIZE * sizeof(uint8_t) / 4; i += WAVE) ((uint32_t*)tags)[i] = 0;
    if (back >= 4) {
        const uint32_t dense_from = back > SOLO_SEED_DENSE ? back - SOLO_SEED_DENSE : 0;
        const uint32_t ss = SOLO_SEED_STRIDE;
        constexpr int SEED_IN_FLIGHT = 16;                                           // loads in flight per wave (registers are plentiful: LDS bounds the occupancy)
        uint32_t q = 0;
        if (ss == 4) {
            // every 4th position: 16 bytes per lane hold four of them, a wave-load covers 1 KiB (a quarter of the load instructions)
            const uint32_t span = dense_from & ~1023u;
            for (; q < span; q += SEED_IN_FLIGHT * 1024) {
                uint4 vv[SEED_IN_FLIGHT];
#pragma unroll
                for (int u = 0; u < SEED_IN_FLIGHT; u++) { const uint32_t p = q + u * 1024 + lane * 16; const b16_ua t = *(const b16_ua*)(base + (p < span ? p : 0u)); vv[u] = uint4{t.a, t.b, t.c, t.d}; }
#pragma unroll
                for (int u = 0; u < SEED_IN_FLIGHT; u++) {
                    const uint32_t p = q + u * 1024 + lane * 16;
                    if (p < span) {
                        const uint32_t w[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
                        for (int k = 0; k < 4; k++) { const uint32_t hv = w[k] * 2654435761u; table[solo_slot(hv)] = (uint16_t)(p + 4 * k); tags[solo_slot(hv)] = (uint8_t)(hv >> (32 - SOLO_TAG_BITS - SOLO_HASH_LOG)); }
                    }
                }
            }
            q = span;
        }
        for (; q < dense_from; q += SEED_IN_FLIGHT * WAVE * ss) {
            uint32_t pp[SEED_IN_FLIGHT], vv[SEED_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < SEED_IN_FLIGHT; u++) { pp[u] = q + (u * WAVE + lane) * ss; vv[u] = ld32(base + (pp[u] < dense_from ? pp[u] : 0u)); }
#pragma unroll
            for (int u = 0; u < SEED_IN_FLIGHT; u++) if (pp[u] < dense_from) { const uint32_t hv = vv[u] * 2654435761u; table[solo_slot(hv)] = (uint16_t)pp[u]; tags[solo_slot(hv)] = (uint8_t)(hv >> (32 - SOLO_TAG_BITS - SOLO_HASH_LOG)); }
        }
        for (uint32_t q = dense_from; q + 4 <= back; q += SEED_IN_FLIGHT * WAVE) {       // (inserted in position order: later ones win)
            uint32_t vv[SEED_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < SEED_IN_FLIGHT; u++) { const uint32_t p = q + u * WAVE + lane; vv[u] = ld32(base + (p + 4 <= back ? p : 0u)); }
#pragma unroll
            for (int u = 0; u < SEED_IN_FLIGHT; u++) {
                const uint32_t p = q + u * WAVE + lane;
                if (p + 4 <= back) { const uint32_t hv = vv[u] * 2654435761u; table[solo_slot(hv)] = (uint16_t)p; tags[solo_slot(hv)] = (uint8_t)(hv >> (32 - SOLO_TAG_BITS - SOLO_HASH_LOG)); }
            }
        }
    }

    uint32_t nrec = 0, first_lit = 0, body = 0;
    uint32_t anchor = cs;
    // a match may start at p iff p + 4 <= ce and p + MFLIMIT <= bend; it may end at min(ce, bend - LASTLIT).
    // Blocks shorter than MFLIMIT+1 bytes are literals only (Appendix A.2).
    const uint32_t blen = (uint32_t)(bend_abs - bstart), clen = ce - cs;
    bool searchable = blen >= MFLIMIT + 1 && clen >= MINMATCH && rec_room;
    uint32_t last_start = 0, end_lim = 0;
    if (searchable) {
        last_start = (ce - MINMATCH < bend - MFLIMIT) ? ce - MINMATCH : bend - MFLIMIT;
        end_lim = (ce < bend - LASTLIT) ? ce : bend - LASTLIT;
        searchable = last_start >= cs;
    }
    if (searchable) {
        // The search is latency-bound (a stream load, a table probe and a candidate gather per step), so two probe
        // steps are kept in flight: step B = "the step after A if A finds nothing" is probed and inserted
        // speculatively while A's candidate gather is still outstanding, and the stream loads run one more step
        // ahead.  When A hits, B's table inserts (and A's beyond the hit) are rolled back from the values they
        // overwrote, so the table evolves exactly as in the one-step-at-a-time formulation.
        const uint32_t hmul = 2654435761u;
        auto stream = [&](uint32_t ipx, uint32_t stepx) -> uint32_t {           // my 4 bytes of the step at (ipx, stepx)
            const uint32_t px = ipx + lane * stepx;
            return ld32(base + (px <= last_start ? px : last_start));       // unconditional load (clamped): no branch, no early wait
        };
        uint32_t ip = cs, step = 1;
        // stream queue: my 4 bytes for the next six steps along the all-miss path (A, B and two more iterations),
        // so that the sequential input is always at least two iterations (~2 us) ahead of the probes
        uint32_t s0, s1, s2, s3, s4, s5, ipN, stepN;
        auto fill_queue = [&]() {
            uint32_t i_ = ip, st_ = step;
            s0 = stream(i_, st_); i_ += WAVE * st_; st_++;
            s1 = stream(i_, st_); i_ += WAVE * st_; st_++;
            s2 = stream(i_, st_); i_ += WAVE * st_; st_++;
            s3 = stream(i_, st_); i_ += WAVE * st_; st_++;
            s4 = stream(i_, st_); i_ += WAVE * st_; st_++;
            s5 = stream(i_, st_); i_ += WAVE * st_; st_++;
            ipN = i_; stepN = st_;
        };
        fill_queue();
        while (ip <= last_start) {
            const uint32_t seqA = s0;
            // ---- probe A ----
            const uint32_t pA = ip + lane * step;
            const bool actA = pA <= last_start;
            const uint32_t hvA = seqA * hmul, hA = solo_slot(hvA), tgA = (hvA >> (32 - SOLO_TAG_BITS - SOLO_HASH_LOG)) & SOLO_TAG_MASK;
            uint32_t eA = 0, tA = SOLO_TAG_MASK + 1;
            if (actA) { eA = table[hA]; tA = tags[hA]; table[hA] = (uint16_t)pA; tags[hA] = (uint8_t)tgA; }
            const uint32_t dA = (pA - eA) & 0xFFFFu;
            const bool okA = actA && tA == tgA && dA != 0 && dA <= pA;            // same 20 hash bits: worth a look at the bytes
            const uint32_t candA = pA - dA;
            // ---- speculative probe B (next step if A misses) + stream prefetch for the step after B ----
            const uint32_t ipB = ip + WAVE * step, stepB = step + 1;
            const uint32_t seqB = s1;
            const uint32_t pB = ipB + lane * stepB;
            const bool actB = pB <= last_start;
            const uint32_t hvB = seqB * hmul, hB = solo_slot(hvB), tgB = (hvB >> (32 - SOLO_TAG_BITS - SOLO_HASH_LOG)) & SOLO_TAG_MASK;
            uint32_t eB = 0, tB = SOLO_TAG_MASK + 1;
            if (actB) { eB = table[hB]; tB = tags[hB]; table[hB] = (uint16_t)pB; tags[hB] = (uint8_t)tgB; }
            const uint32_t dB = (pB - eB) & 0xFFFFu;
            const bool okB = actB && tB == tgB && dB != 0 && dB <= pB;
            const uint32_t candB = pB - dB;
            const uint32_t ipC = ipB + WAVE * stepB, stepC = stepB + 1;
            // A candidate is looked at in memory only when its lane passed the 20-bit tag filter (in literal regions almost
            // every step skips memory altogether), and then verification and extension are ONE round trip: the wave loads
            // the 64 bytes before and the 512 bytes from the probe position itself, on both sides; the candidate is a match
            // iff the first four bytes agree.  (A separate 4-byte gather first would cost a second trip on every match.)
            uint64_t cA = __ballot(okA), cB = __ballot(okB);
            bool hitB = false, found = false;
            uint32_t L = 0, mp = 0, mc = 0, dist = 0, room = 0;
            const uint32_t kb = lane + 1;
            uint8_t bb0 = 0, bb1 = 1;
            uint64_t x0 = 0;
            uint32_t a0 = 0;
            while (cA | cB) {
                hitB = cA == 0;
                if (!hitB) { L = (uint32_t)__builtin_ctzll(cA); cA &= cA - 1; } else { L = (uint32_t)__builtin_ctzll(cB); cB &= cB - 1; }
                mp = __builtin_amdgcn_readlane(hitB ? pB : pA, L);
                mc = __builtin_amdgcn_readlane(hitB ? candB : candA, L);
                dist = mp - mc;
                room = mp - anchor; if (mc < room) room = mc;
                bb0 = 0; bb1 = 1;
                if (kb <= room) { bb0 = base[mp - kb]; bb1 = base[mc - kb]; }
                a0 = mp + lane * 8;
                x0 = 0;
                if (a0 < end_lim) x0 = ld64_guard(base + a0, rd_end) ^ ld64_guard(base + (a0 - dist), rd_end);
                if (__builtin_amdgcn_readlane((uint32_t)x0, 0) == 0) { found = true; break; }       // mp + 4 <= end_lim always
            }
            if (!found) {                                                           // both steps missed: advance two steps, top up the queue
                ip = ipC; step = stepC;
                s0 = s2; s1 = s3; s2 = s4; s3 = s5;
                s4 = stream(ipN, stepN); ipN += WAVE * stepN; stepN++;
                s5 = stream(ipN, stepN); ipN += WAVE * stepN; stepN++;
                continue;
            }
            // roll back the inserts the greedy parse does not make: lanes beyond the hit (and all of B when A hit).  Several
            // lanes of a step can share a table slot (periodic data): a lane beyond the hit restoring "its" old value would
            // also wipe the insert of a lane up to the hit, so those are written again afterwards.
            if (!hitB) {
                if (actB) { table[hB] = (uint16_t)eB; tags[hB] = (uint8_t)tB; }
                if (actA && lane > L) { table[hA] = (uint16_t)eA; tags[hA] = (uint8_t)tA; }
                if (actA && lane <= L) { table[hA] = (uint16_t)pA; tags[hA] = (uint8_t)tgA; }
            } else {
                if (actB && lane > L) { table[hB] = (uint16_t)eB; tags[hB] = (uint8_t)tB; }
                if (actB && lane <= L) { table[hB] = (uint16_t)pB; tags[hB] = (uint8_t)tgB; }
            }
            uint32_t mlen = 0;
            {
                // backward
                const uint64_t ne = __ballot(!(kb <= room && bb0 == bb1));
                uint32_t nb = ne ? (uint32_t)__builtin_ctzll(ne) : WAVE;
                if (nb == WAVE && room > WAVE) {                                   // rare: more than 64 bytes backwards
                    uint32_t r2 = room - WAVE, m2 = mp - WAVE, c2 = mc - WAVE;
                    while (r2) {
                        const bool in = kb <= r2;
                        const bool eq = in && base[m2 - kb] == base[c2 - kb];
                        const uint64_t ne2 = __ballot(!eq);
                        const uint32_t n2 = ne2 ? (uint32_t)__builtin_ctzll(ne2) : WAVE;
                        nb += n2; m2 -= n2; c2 -= n2; r2 -= n2;
                        if (n2 < WAVE) break;
                    }
                }
                mp -= nb; mc -= nb; mlen += nb;
                // forward: the first 512 bytes came with the verification; longer matches go on below
                uint32_t g0 = 0;
                if (a0 < end_lim) { g0 = x0 ? (uint32_t)(__builtin_ctzll(x0) >> 3) : 8; const uint32_t r = end_lim - a0; if (g0 > r) g0 = r; }
                const uint64_t stop0 = __ballot(g0 < 8);
                bool more = false;
                if (stop0) { const uint32_t f = (uint32_t)__builtin_ctzll(stop0); mlen += f * 8 + __builtin_amdgcn_readlane(g0, f); }
                else { mlen += WAVE * 8; more = true; }
                while (more) {                                                      // long matches: keep going, 512 bytes per round
                    // (one round per trip: hipcc waits for each of these guarded loads separately, so a second round fetched
                    // "for free" cost two more round trips - and a match of exactly 512 bytes, the end of the first window, is common)
                    const uint32_t b0 = mp + mlen + lane * 8;
                    uint64_t y0 = 0;
                    if (b0 < end_lim) y0 = ld64_guard(base + b0, rd_end) ^ ld64_guard(base + (b0 - dist), rd_end);
                    uint32_t h0 = 0;
                    if (b0 < end_lim) { h0 = y0 ? (uint32_t)(__builtin_ctzll(y0) >> 3) : 8; const uint32_t r = end_lim - b0; if (h0 > r) h0 = r; }
                    const uint64_t s0 = __ballot(h0 < 8);
                    if (s0) { const uint32_t f = (uint32_t)__builtin_ctzll(s0); mlen += f * 8 + __builtin_amdgcn_readlane(h0, f); break; }
                    mlen += WAVE * 8;
                }
            }
            // append the sequence record (every lane stores the same 8 bytes: no lane-predicated branch in this loop)
            const uint32_t lit = mp - anchor;
            rec[nrec] = pack_rec(lit, mlen, dist);
            if (nrec == 0) first_lit = lit;
            body += seq_size(lit, mlen);
            nrec++;
            anchor = ip = mp + mlen;
            // (behind a long match the search goes on at stride SOLO_STEP_LONG, as the shared search does where sequences are long: what follows a 512-byte copy of
            // the bench input is 512 random bytes - strides 1..6 took three rounds of two steps to get across them, 4 and 5 take one; a match found a few bytes late gets
            // its start back from the backward extension)
            step = (SOLO_STEP_LONG > 1 && mlen >= SOLO_LONG_MATCH) ? (uint32_t)SOLO_STEP_LONG : 1u;
            if (nrec >= g.max_rec_per_chunk) break;          // cannot happen with chunk/4+1 slots; belt and braces
            // like the CPU encoder, also index ip-2; its bytes are requested together with the refilled stream queue
            const bool ins2 = ip >= 2 + cs && ip + 2 <= ce;
            const uint32_t q2 = ins2 ? ip - 2 : cs;
            const uint32_t v2 = ld32(base + q2);
            fill_queue();
            if (ins2) { const uint32_t hv = v2 * hmul; table[solo_slot(hv)] = (uint16_t)q2; tags[solo_slot(hv)] = (uint8_t)(hv >> (32 - SOLO_TAG_BITS - SOLO_HASH_LOG)); }
        }
    }
    if (lane == 0) { ci->nrec = nrec; ci->first_lit = first_lit; ci->tail_lit = ce - anchor; ci->body_size = body; }
}

// ------------------------------- pass S --------------------------------------------------------

}  // namespace lz4f
