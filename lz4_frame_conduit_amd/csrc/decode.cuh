// decode.cuh -- LZ4 block decode on gfx950.
//
// Replaces the inner block loop of LZ4F_decompress (called at
// /root/reference/src/Codec/Compression/LZ4/Conduit.hsc:591): per frame block, parse the
// sequence stream and perform the literal copies and the overlap-safe match copies.
//
// v1 mapping: one wavefront per frame block.  The sequence parse is a serial dependent chain, so
// it runs on the scalar unit (wave-uniform values, scalar loads from the read-only payload);
// every copy is spread over the 64 lanes, 16 B per lane (1 KiB per wave instruction), straight
// HBM -> HBM.  Match sources are bytes this same wave wrote earlier; a wave's vector memory
// operations are performed in issue order, so no LDS window is needed for correctness.
// Same accept/reject rules as the oracle (oracle/orc_lz4block.c: orc_lz4_decompress_safe).
#pragma once
#include "common.cuh"

namespace lz4f {

// 8 payload bytes starting at byte offset `ip` (little endian), read as aligned dwords so that
// the loads are scalar; never touches a dword at or beyond `lim4` (= payload end rounded up to 4).
__device__ __forceinline__ uint64_t fetch8(const uint8_t* __restrict__ in_al, uint32_t ip, uint32_t lim4)
{
    // in_al is the payload base rounded DOWN to 4; ip already includes the base misalignment
    uint32_t a0 = ip & ~3u;
    uint32_t a1 = a0 + 4, a2 = a0 + 8;
    const uint32_t last = lim4 - 4;
    a0 = a0 < last ? a0 : last; a1 = a1 < last ? a1 : last; a2 = a2 < last ? a2 : last;
    // three scalar loads (K$), one wait: the payload is read-only for the whole launch, so the scalar
    // cache is coherent with it; hipcc would otherwise issue vector loads + v_readfirstlane here
    uint32_t w0, w1, w2;
    asm volatile("s_load_dword %0, %3, %4\n\ts_load_dword %1, %3, %5\n\ts_load_dword %2, %3, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(w0), "=&s"(w1), "=&s"(w2)
                 : "s"(in_al), "s"(a0), "s"(a1), "s"(a2)
                 : "memory");
    const uint32_t sh = (ip & 3u) * 8;
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    uint64_t v = lo >> sh;
    if (sh) v |= (uint64_t)w2 << (64 - sh);
    return v;
}

// Decode one compressed block with one wave.  Returns decoded size, or -1 on malformed input.
//   in      : payload (csize bytes), any alignment, read-only for the whole launch
//   out     : where this block's bytes go; `hist` valid bytes lie directly in front of it
//   cap     : bytes available at out (maxBlockSize, or what is left of the destination)
__device__ __forceinline__ int32_t wave_decode_block(const uint8_t* __restrict__ in, uint32_t csize,
                                                     uint8_t* out, uint32_t cap, uint64_t hist)
{
    if (csize == 0) return -1;
    const uint32_t mis = (uint32_t)((uintptr_t)in & 3u);
    const uint8_t* __restrict__ in_al = in - mis;
    uint32_t ip = mis;                         // byte cursor relative to in_al
    const uint32_t iend = mis + csize;
    const uint32_t lim4 = (iend + 3u) & ~3u;
    uint32_t op = 0;
    const uint32_t oend = cap;

    for (;;) {
        // ---- token + literal length -------------------------------------------------------
        uint64_t w = fetch8(in_al, ip, lim4);
        const uint32_t token = (uint32_t)w & 0xFF;
        ip += 1;
        uint32_t lit = token >> 4;
        if (lit == 15) {
            w >>= 8;
            uint32_t avail = 7;
            for (;;) {
                if (ip >= iend) return -1;
                if (avail == 0) { w = fetch8(in_al, ip, lim4); avail = 8; }
                const uint32_t s = (uint32_t)w & 0xFF;
                w >>= 8; avail--; ip++;
                lit += s;
                if (s != 255) break;
                if (lit > 0x7FFFFFFFu - 255u) return -1;
            }
        }
        // ---- literals ---------------------------------------------------------------------
        // (oend - op) and (iend - ip) cannot underflow: op <= oend, ip <= iend are loop invariants
        if (ip > iend) return -1;
        const uint32_t in_left = iend - ip, out_left = oend - op;
        if ((uint64_t)lit + 12 > out_left || (uint64_t)lit + 8 > in_left) {
            // must be the last sequence: literals end exactly at the payload end
            if (lit != in_left || lit > out_left) return -1;
            wave_copy_disjoint(out + op, in_al + ip, lit);
            op += lit;
            return (int32_t)op;
        }
        wave_copy_disjoint(out + op, in_al + ip, lit);
        ip += lit; op += lit;

        // ---- match ------------------------------------------------------------------------
        w = fetch8(in_al, ip, lim4);
        const uint32_t offset = (uint32_t)w & 0xFFFF;
        ip += 2;
        if (offset == 0) return -1;
        if ((uint64_t)offset > (uint64_t)op + hist) return -1;
        uint32_t mlen = token & 15;
        if (mlen == 15) {
            w >>= 16;
            uint32_t avail = 6;
            for (;;) {
                if (ip >= iend) return -1;
                if (avail == 0) { w = fetch8(in_al, ip, lim4); avail = 8; }
                const uint32_t s = (uint32_t)w & 0xFF;
                w >>= 8; avail--; ip++;
                mlen += s;
                if (ip + 4 >= iend) return -1;
                if (s != 255) break;
                if (mlen > 0x7FFFFFFFu - 255u) return -1;
            }
        }
        mlen += 4;
        if ((uint64_t)mlen + 5 > (uint64_t)(oend - op)) return -1;       // last 5 bytes must be literals
        wave_copy_match(out + op, offset, mlen);
        op += mlen;
    }
}

// ---- independent small blocks: the lanes look for the tokens ----
// The version above is written for the scalar unit, and that is what a text-like stream (13-byte sequences: 5000 per 64 KiB
// block, 82 million per GiB) runs out of: a CU has ONE scalar unit, one instruction per cycle for all its waves (a wave gets a
// turn every ~4.7 cycles, tools/probe/chain_rates.hip), and ~140 scalar instructions per sequence is 20 ms per GiB however
// many waves there are.  (On top of that every sequence waits for memory three times.)  So here the 64 lanes do the per-
// sequence work, a window of 64 payload bytes at a time, lane l looking at byte pos + l:
//   1. every lane reads "its" byte as if it were a token: literal length, match length, and where the next token would be
//      (lane + 3 + literals) - valid for the common token without length extension bytes that ends inside the window;
//   2. the scalar unit only hops from token to token (one v_readlane per hop) and collects the mask of real tokens;
//   3. the real tokens' lanes fetch their offset from the lane it sits in (ds_bpermute), a prefix sum gives every token its
//      place in the output, a running maximum tells every byte lane which token it belongs to - and ALL literal bytes of the
//      window go out with one store instruction, each lane its own byte;
//   4. the matches follow in order, a byte per lane (three v_readlane and a load->store pair per match: the one memory
//      round trip left per sequence; same-wave accesses are performed in order, so a match sees what was stored before it).
// Tokens with extension bytes, runs that leave the window, and the block's last ~100 bytes go through `one_sequence`, which
// is the scalar decoder with the payload in a register window (lane l keeps the 16 bytes at wb + 8*l: token, lengths and
// offset are four v_readlane and a funnel shift instead of scalar loads; short literal runs come out of the window by
// ds_bpermute; the next window is on its way while this one is parsed).  Same accept/reject rules as wave_decode_block.
//   readable: bytes that may be read from `in` on (the frame's end), >= csize
#ifdef DB_PROF      // development: cycle stamps of the lanes' path, summed over the grid's first waves (tools/cfg2_prof.py)
__device__ unsigned long long g_dbprof[32];
#define DBP(...) __VA_ARGS__
#else
#define DBP(...)
#endif
template <bool VEC>
__device__ __forceinline__ int32_t wave_decode_block_win(const uint8_t* __restrict__ in, uint32_t csize, uint64_t readable,
                                                         uint8_t* out, uint32_t cap, uint32_t* expand /* 64 words of LDS, this wave's */)
{
    if (csize == 0) return -1;
    const uint32_t lane = lane_id();
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    typedef v4u_t v4u_ua __attribute__((aligned(1)));
    auto load_win = [&](uint32_t base) -> v4u_t {
        const uint64_t a = (uint64_t)base + 8u * lane;
        v4u_t v = {0u, 0u, 0u, 0u};
        if (a + 16 <= readable) v = *(const v4u_ua*)(in + a);
        else if (a < readable) {                                             // the frame's last bytes: one by one
            uint32_t t[4] = {0u, 0u, 0u, 0u};
            for (uint32_t i = 0; i < 16 && a + i < readable; i++) t[i >> 2] |= (uint32_t)in[a + i] << ((i & 3u) * 8u);
            v = v4u_t{t[0], t[1], t[2], t[3]};
        }
        return v;
    };
    v4u_t win = {0u, 0u, 0u, 0u}, nxt = {0u, 0u, 0u, 0u};
    constexpr uint32_t NONE = 0x80000000u;                                    // (no payload position comes within 504 of it)
    uint32_t wb = 0u - 512u, nb = NONE;                                      // nothing loaded (a first read near 0 counts as a step over the edge); nb: base of the window in `nxt`, if one is on its way
    auto ensure = [&](uint32_t qq) {
        if (qq - wb < 504u) return;                                          // lanes 0..62 serve reads at rel 0..503
        if (qq - nb < 504u) {                                                // the stream went on where it was: the prefetched window
            win = nxt; wb = nb;
            nb = wb + 496u; nxt = load_win(nb);
        } else {                                                             // a jump (long literal run, or the lanes' path ran ahead): load it now
            const bool near = qq - wb < 504u + 64u;                          // look ahead again only after a step just over the edge, not a stride
            wb = qq & ~7u; win = load_win(wb);
            nb = NONE;
            if (near) { nb = wb + 496u; nxt = load_win(nb); }
        }
    };
    auto fetch = [&](uint32_t qq) -> uint64_t {                              // 8 payload bytes from qq on
        ensure(qq);
        const uint32_t rel = qq - wb, l = rel >> 3, sh8 = (rel & 7u) * 8u;
        const uint64_t lo = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.x, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.y, l) << 32);
        const uint64_t hi = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.z, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.w, l) << 32);
        return (lo >> sh8) | ((hi << 1) << (63u - sh8));
    };
    auto ext_slow = [&](uint32_t at, uint32_t& after, bool& bad) -> uint32_t {   // length bytes that run on beyond one read
        uint32_t add = 0;
        for (;;) {
            if (at >= csize || add > 0x7FFF0000u) { bad = true; after = at; return add; }
            const uint32_t b = uni((uint32_t)in[at]);
            add += b; at++;
            if (b != 255) { after = at; return add; }
        }
    };
    uint32_t pos = 0, op = 0;
    DBP(unsigned long long a_top = 0, a_parse = 0, a_lit = 0, a_rounds = 0, a_ord = 0, a_seq = 0, n_win = 0, n_round = 0, n_ord = 0, n_seq = 0;)
    uint32_t dnext = 0, dnext_pos = NONE;                                    // the lanes' path: the next window's bytes, asked for ahead of time
    // one sequence on the scalar unit.  0: go on, 1: that was the last one, -1: malformed
    auto one_sequence = [&]() -> int {
        uint64_t w = fetch(pos);
        const uint32_t token = (uint32_t)w & 0xFF;
        uint32_t lit = token >> 4, p = pos + 1;
        bool bad = false;
        if (lit == 15) {
            const uint64_t x = w >> 8;                                       // 7 candidate length bytes, top byte 0 (never 0xFF)
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), k = f >> 3;
            lit = 15u + 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
            p = pos + 2 + k;
            if (k == 7) lit = 15u + ext_slow(pos + 1, p, bad);
        }
        if (bad || p > csize) return -1;
        const uint32_t in_left = csize - p, out_left = cap - op;
        const bool is_last = (uint64_t)lit + 12 > out_left || (uint64_t)lit + 8 > in_left;
        if (is_last && (lit != in_left || lit > out_left)) return -1;
        if (lit) {
            const uint32_t rel0 = p - wb;
            if (lit <= WAVE && rel0 + lit <= 504u) {                         // inside the window: lane i takes byte p + i out of it
                const uint32_t rel = rel0 + lane, l4 = (rel >> 3) << 2;
                const uint32_t xa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)l4, (int)win.x), ya = (uint32_t)__builtin_amdgcn_ds_bpermute((int)l4, (int)win.y);
                const uint32_t word = (rel & 4u) ? ya : xa;
                if (lane < lit) out[op + lane] = (uint8_t)(word >> ((rel & 3u) * 8u));
            } else wave_copy_disjoint(out + op, in + p, lit);
            op += lit;
        }
        if (is_last) return 1;
        const uint32_t qo = p + lit;
        w = fetch(qo);
        const uint32_t offset = (uint32_t)w & 0xFFFF;
        uint32_t mlen = token & 15, npos = qo + 2;
        if (mlen == 15) {
            const uint64_t x = w >> 16;                                      // 6 candidate length bytes
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), k = f >> 3;
            mlen = 15u + 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
            npos = qo + 3 + k;
            if (k == 6) mlen = 15u + ext_slow(qo + 2, npos, bad);
            if (npos + 4 >= csize) bad = true;
        }
        mlen += 4;
        if (bad || offset == 0 || offset > op || (uint64_t)mlen + 5 > (uint64_t)(cap - op)) return -1;   // (last 5 bytes must be literals)
        if (mlen <= WAVE) {                                                  // the usual short match: a byte per lane
            uint32_t idx = lane;
            if (offset < mlen) idx = lane % offset;                          // (an overlapping one repeats its period)
            if (lane < mlen) { const uint8_t b = out[op - offset + idx]; out[op + lane] = b; }
        } else wave_copy_match(out + op, offset, mlen);
        op += mlen;
        pos = npos;
        return 0;
    };
    for (;;) {
        // ---- the lanes' path: a window of 64 payload bytes that starts at a token, away from the block's end ----
        // (a token here costs >= 3 payload bytes and its match gives <= 18 output bytes, the literals are the window's own: a window
        // never makes more than 64 + 21 * 18 = 442; with 96 payload bytes and 1 KiB of room left none of its sequences can be the
        // last one or run into the end-of-block rules)
        if (VEC && pos <= csize && csize - pos >= 96u && cap - op >= 1024u) {
            typedef uint32_t u32_ua1 __attribute__((aligned(1)));
            DBP(const unsigned long long y0 = clock64();)
            const uint32_t d = dnext_pos == pos ? dnext : *(const u32_ua1*)(in + pos + lane);
            DBP(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long y1 = clock64(); a_top += y1 - y0;)
            // (a literal length of 15..269 - one extension byte - is still a lane's own business: the byte is in its dword.  Longer
            // ones, and tokens with match-length bytes, go to one_sequence.)
            const uint32_t t = d & 0xFFu, litn = t >> 4, ml = t & 15u, e1 = (d >> 8) & 0xFFu;
            const uint32_t hdr = litn == 15u ? 2u : 1u, lit = litn == 15u ? 15u + e1 : litn;      // bytes in front of the literals; literals
            const bool easy = ml != 15u && !(litn == 15u && e1 == 255u) && lane + hdr + lit + 2u <= 64u;
            const uint32_t nx = easy ? lane + hdr + lit + 2u : 255u;
            // the serial part: one hop per token (v_readlane, s_bitset1, two moves, compare, branch).  A token is marked before its
            // end is known; the last one is taken back if it does not end inside the window.
            uint64_t mask = 0;
            uint32_t s = 0, sp = 0, n;
            do {
                n = (uint32_t)__builtin_amdgcn_readlane((int)nx, (int)s);
                asm("s_bitset1_b64 %0, %1" : "+s"(mask) : "s"(s));
                sp = s; s = n;
            } while (n < 64u);
            if (n > 64u) { mask &= ~(1ull << sp); s = sp; }
            if (mask) {
                // the next window's bytes are asked for as soon as it is known where it starts: they travel while this window's
                // literals and matches are stored (memory operations of a wave return in order: by the time a match's bytes are
                // there, so are these)
                dnext_pos = pos + s;
                if (csize - dnext_pos >= 96u) dnext = *(const u32_ua1*)(in + dnext_pos + lane); else dnext_pos = NONE;
                const bool is_tok = (mask >> lane) & 1ull;
                const uint32_t mlen = ml + 4u;
                const uint32_t d2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + hdr + lit) << 2), (int)d);   // the dword of the lane the offset starts in
                const uint32_t off = d2 & 0xFFFFu;
                // one prefix sum for two spaces: the window's output bytes (low half) and its match bytes (high half; <= 672 and 378)
                const uint32_t tout = is_tok ? lit + mlen : 0u, mcnt = is_tok ? mlen : 0u;
                const uint32_t both = dpp_incl_scan_add(tout | (mcnt << 16));
                const uint32_t incl = both & 0xFFFFu, ex = incl - tout, mincl = both >> 16, mex = mincl - mcnt;
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                DBP(const unsigned long long y2 = clock64(); a_parse += y2 - y1; n_win++;)
                const uint32_t mdst = op + ex + lit;                         // where the token's match goes
                if (__ballot(is_tok && (off == 0u || off > mdst))) return -1;
                // literals: byte lane l belongs to the nearest token at or below it
                const uint32_t g1 = dpp_incl_scan_max(is_tok ? lane + 1u : 0u);      // (lane 0 is a token: never 0)
                const uint32_t pg = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((g1 - 1u) << 2), (int)((ex << 8) | (lit << 1) | (hdr - 1u)));
                const uint32_t r = lane - g1 - (pg & 1u);                    // byte number within the token's literal run (wraps for the token's own bytes)
                if (lane >= g1 + (pg & 1u) && r < ((pg >> 1) & 127u)) out[op + (pg >> 8) + r] = (uint8_t)d;
                // matches.  Most of them read what earlier windows wrote: those go together, 64 match bytes of the window per round, a
                // lane per byte - a prefix sum places every match in that space, its lane leaves its number at the start of its run
                // (`expand`, 64 words of LDS), a running maximum spreads it over the run.  A match whose source reaches into this
                // window's own matches waits for the ordered loop behind.
                DBP(const unsigned long long y3 = clock64(); a_lit += y3 - y2;)
                const uint32_t mtotal = (uint32_t)__builtin_amdgcn_readlane((int)mincl, 63);
                const uint32_t md0 = (uint32_t)__builtin_amdgcn_readlane((int)mdst, 0);          // the window's first match: everything in front of it is stored or on its way
                const bool indep = is_tok && mdst - off + mlen <= md0;                          // (implies off >= mlen: no overlap with itself either)
                const uint32_t pa_mine = mdst | (mlen << 24);                             // (a block is at most 4 MiB: 22 bits)
                volatile lds_u32* const xp = (volatile lds_u32*)expand;        // (volatile: see below; typed as LDS - through a generic pointer these were flat_* operations, each with a wait for every store in flight)
                uint32_t base = 0;
                while (base < mtotal) {                                       // rounds of up to 64 match bytes, cut between matches
                    const bool fits = is_tok && mex >= base && mincl <= base + WAVE;
                    const uint64_t fm = __ballot(fits);                      // (never empty: a match here is at most 18 bytes)
                    if (!fm) break;                                          // (... and if it ever were, the ordered loop behind takes what is left)
                    const uint32_t nbase = (uint32_t)__builtin_amdgcn_readlane((int)mincl, 63 - (int)__builtin_clzll(fm));
                    // to the compiler a lane that stores nothing in between reads back its own 0 - the other lanes' stores are not
                    // in its picture, hence volatile; same-wave LDS accesses are performed in order, so nothing else is needed
                    xp[lane] = 0u;
                    if (fits && indep) xp[mex - base] = lane + 1u;
                    const uint32_t k1 = dpp_incl_scan_max(xp[lane]);                           // 0: no match of the round covers byte j
                    const uint32_t kk = ((k1 ? k1 : 1u) - 1u) << 2;
                    const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)kk, (int)pa_mine),
                                   pb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)kk, (int)(off | ((mex - base) << 16)));
                    const uint32_t bi = lane - (pb >> 16), bd = pa & 0xFFFFFFu;                // my byte within the match, the match's destination
                    if (k1 && bi < (pa >> 24)) { const uint8_t b = out[bd - (pb & 0xFFFFu) + bi]; out[bd + bi] = b; }
                    base = nbase;
                    DBP(n_round++;)
                }
                DBP(const unsigned long long y4 = clock64(); a_rounds += y4 - y3;)
                const bool together = indep && mex < base;                      // (copied by the rounds above)
                uint64_t m = __ballot(is_tok && !together);                                    // the others, in order
                while (m) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t dk = (uint32_t)__builtin_amdgcn_readlane((int)mdst, (int)k), ok = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)k),
                                   lk = (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)k);
                    uint32_t idx = lane;
                    if (ok < lk) idx = lane % ok;                            // (overlapping: repeats its period; lk <= 18)
                    if (lane < lk) { const uint8_t b = out[dk - ok + idx]; out[dk + lane] = b; }
                    DBP(n_ord++;)
                }
                DBP(a_ord += clock64() - y4;)
                op += total;
                pos += s;
                continue;
            }
        }
        DBP(const unsigned long long x0 = clock64();)
        const int r = one_sequence();
        DBP(a_seq += clock64() - x0; n_seq++;)
        DBP(if (r != 0 && lane == 0 && blockIdx.x < 64) { atomicAdd(&g_dbprof[0], a_top); atomicAdd(&g_dbprof[1], a_parse); atomicAdd(&g_dbprof[2], a_lit); atomicAdd(&g_dbprof[3], a_rounds); atomicAdd(&g_dbprof[4], a_ord); atomicAdd(&g_dbprof[5], a_seq); atomicAdd(&g_dbprof[6], n_win); atomicAdd(&g_dbprof[7], n_round); atomicAdd(&g_dbprof[8], n_ord); atomicAdd(&g_dbprof[9], n_seq); atomicAdd(&g_dbprof[10], 1ull); })
        if (r < 0) return -1;
        if (r > 0) return (int32_t)op;
    }
}

// Table-driven block decode: wave w of the grid takes block w.
//   word bit31 set  -> stored block: plain copy
//   otherwise       -> LZ4 sequences
// linked != 0: one wave walks all blocks in order (each may reference the 64 KiB before it).
struct DecodeArgs {
    const uint8_t* frame;          // device frame bytes
    uint8_t*       dst;            // device output
    uint64_t       dst_cap;
    uint32_t       block_size;     // maxBlockSize of the frame
    uint32_t       linked;
};

}  // namespace lz4f
