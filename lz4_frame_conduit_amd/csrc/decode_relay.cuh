// decode_relay.cuh -- dense payloads in few big independent blocks: a workgroup per block, its waves taking the block's 64-byte payload
// windows in turn (the inner block loop of LZ4F_decompress behind
// /root/reference/src/Codec/Compression/LZ4/Conduit.hsc:591 for what `lz4 -c` writes of text, test/Main.hs:33-36).
//
// Such a block is one chain of ~13-byte sequences from its first byte to its last and comes without an index; a GiB in 4 MiB blocks is
// 256 of them.  One wave per block (decode.cuh: wave_decode_block_win, the lanes find the tokens of a window, the matches follow a
// byte per lane) runs ~600 instructions per window one after the other - 3600 cycles per 120 bytes of output whether the bytes come
// from memory or from LDS (a wave with a 64 KiB ring in LDS was tried: 14.7 -> 18.9 GiB/s for the GiB).  Very little of that is a chain from
// window to window: where the next window starts and where its output goes, and the few matches that read what the windows just
// before them wrote.  The workgroup is built around those two chains; everything else is spread over its waves:
//   - speculator waves: for EVERY payload position of a 64-byte group, what a window that started there would be (its length E and its
//     output T; lane c walks the tokens from byte c on, two at a time, by the producers' rules) - into a table in LDS.
//   - the walker wave: the chain of positions is then one LDS read per window: pos += E, op += T; the windows go on a list in LDS.
//   - producer waves take the windows off the list in rotation and do them: hop from token to token, prefix sums, literals, and the
//     matches whose source lies below `done_op` (everything in front of it is final), a byte per lane - looking at `done_op` again and
//     again while the finishing wave is still a few windows away.  The output lives in a ring of 128 KiB in LDS (block position mod
//     131072): nothing a producer does touches memory.  What is left of the window's matches goes into a slot for the finishing wave:
//     those that read nothing another of them writes as a byte-per-lane plan, the others as a list behind it.
//   - the finishing wave takes the slots in window order - the block's one chain of copies - and moves `done_op` along: a plan is one
//     read and one write whatever its number of matches (75 % of the windows have only that), a listed match ~600 cycles.
//   - the service wave stages the payload (a KiB at a time, 4 KiB ring in LDS, kept 2 KiB ahead of the walker and 1 KiB behind it) and
//     writes the ring out to memory behind `done_op`, 16 bytes per lane.
// Tokens the lanes cannot take (match lengths over 219, literal-length bytes beyond the first, sequences that do not fit into 64 bytes,
// the block's last ~100 bytes) are done one sequence at a time by the producer that gets them (the walker waits for its answer), after the windows before are finished;
// copies longer than a wave go straight to memory (after the service wave has caught up) and are mirrored into the ring.  Same
// accept/reject rules as wave_decode_block_win / the oracle (oracle/orc_lz4block.c: orc_lz4_decompress_safe).
// Every wait is a poll of LDS that also looks at `stop` and gives up after some millions of polls (a block that hangs is reported as failed).
// Measured (tools/text_big_blocks.py, tools/relay_prof.py; NOTES_r4.md): 1 GiB of text in 4 MiB blocks 13-14.7 -> 52-53 GiB/s; a workgroup
// has a CU to itself, so from ~3 blocks per CU on the wave-per-block decoder wins again (engine.hip picks by block count).
#pragma once
#include "decode.cuh"

namespace lz4f {

typedef __attribute__((address_space(3))) unsigned long long lds_u64;

constexpr uint32_t RL_QW = 16u;            // windows the walker may be ahead of the finishing wave
constexpr uint32_t RL_Q = 8u;              // windows posted to the finishing wave and not finished yet, at most

template <int W>
struct RelayLds {
    alignas(16) uint8_t ring[131072];              // output byte at block position p: ring[p & 0x1FFFF]
    alignas(16) uint8_t stage[4096];               // payload byte q: stage[q & 4095] for st_lo <= q < st_end
    alignas(16) uint32_t et[2048];                 // et[q & 2047] = ((q >> 11) + 1) << 19 | T << 7 | E: a window that starts at payload byte q is E bytes long and makes T bytes (E = 0: not for the lanes)
    alignas(16) unsigned long long wq[RL_QW][2];   // the walker's list of windows: [0] = (op | tag << 24) << 32 | pos | tag << 24 (one 64-bit write), [1] = 1 << 31 (a sequence to be done alone) | T << 7 | E
    alignas(8) unsigned long long mail;            // ... and where such a sequence left off, back to the walker (same packing)
    alignas(8) uint32_t st_lo; uint32_t st_end;    // (read as one 64-bit word)
    uint32_t done_op;                              // output below this is final (and in the ring)
    uint32_t flushed;                              // output below this is in memory
    uint32_t pos_hint;                             // where the payload is being read
    uint32_t stop;                                 // != 0: the block is finished (or failed)
    uint32_t flush_req;                            // a producer waits for flushed == done_op
    uint32_t pad;
    uint32_t fin_count;                            // windows the finishing wave is done with
    alignas(8) unsigned long long slot[RL_Q][24];  // window k's matches that could not go at once, for the finishing wave: [0] = op_end << 32 | tag << 16 | count (listed ones; bit 15: there is a plan), [1 + j] = off << 32 | mlen << 24 | dst
    alignas(8) unsigned long long plan[RL_Q][64];  // ... and (count & 0x8000) those of them that read nothing another one writes, byte by byte, a lane each: src << 32 | dst (dst = 0xFFFFFFFF: nothing)
    uint32_t expand[W][64];
};
constexpr uint32_t RL_MASK = 131071u;
constexpr uint32_t RL_SPIN_CAP = 1u << 22;
constexpr uint32_t RL_EXT_MAX = 200u;      // the lanes take tokens with one match-length byte up to this (matches of up to 219 bytes)
#ifndef RL_SLEEP
#define RL_SLEEP 8      // s_sleep between the polls that are not on a chain (x 64 cycles)
#endif
#ifndef RL_D
#define RL_D 6          // a producer's first look at done_op: when the finishing wave is this many windows behind its own, or fewer (<= RL_Q)
#endif
#ifndef RL_DPOST
#define RL_DPOST 4      // ... and its last
#endif
#ifndef RELAY_W
#define RELAY_W 9      // producer waves
#endif
#ifndef RELAY_S
#define RELAY_S 3      // speculator waves
#endif

#define RL_V32(x) (*(volatile lds_u32*)&(x))
#define RL_V64(x) (*(volatile lds_u64*)&(x))
#define RL_FENCE() asm volatile("" ::: "memory")     // (the compiler keeps ring / stage accesses on their side of a poll or a publish; the LDS performs a wave's accesses in order)

// ---- the service wave ----
template <int W>
__device__ __forceinline__ void relay_service(const uint8_t* __restrict__ in, uint32_t csize, uint64_t readable, uint8_t* out, RelayLds<W>* lds)
{
    const uint32_t lane = lane_id();
    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    typedef v4u_t v4u_ua __attribute__((aligned(1)));
    typedef __attribute__((address_space(3))) v4u_t lds_v4u;
    lds_u8* const ring = (lds_u8*)lds->ring;
    lds_u8* const stage = (lds_u8*)lds->stage;
    const uint32_t lim = ((csize + 128u + 1023u) & ~1023u);                  // nothing is read beyond this
    uint32_t st_lo = 0, st_end = 0, idle = 0;
    for (;;) {
        const uint32_t stop = uni(RL_V32(lds->stop));
        bool worked = false;
        // -- payload: keep [hint - 1 KiB, hint + 2 KiB) staged (the walker is up to 16 windows ahead of the producers' reads) --
        const uint32_t hint = uni(RL_V32(lds->pos_hint));
        if (!stop) {
            if (hint >= st_end + 1024u || hint < st_lo) {                    // the reader jumped (a long literal run)
                st_lo = st_end = hint & ~1023u;
                if (lane == 0) RL_V64(lds->st_lo) = ((unsigned long long)st_end << 32) | st_lo;
            }
            if (st_end < hint + 2048u && st_end < lim) {                      // (the chunk this one replaces ends 3 KiB in front of it: a KiB behind `hint`, where the producers may still be reading)
                const uint64_t a = (uint64_t)st_end + 16u * lane;
                v4u_t v = {0u, 0u, 0u, 0u};
                if (a + 16 <= readable) v = *(const v4u_ua*)(in + a);
                else if (a < readable) {                                     // the frame's last bytes: one by one
                    uint32_t t[4] = {0u, 0u, 0u, 0u};
                    for (uint32_t i = 0; i < 16 && a + i < readable; i++) t[i >> 2] |= (uint32_t)in[a + i] << ((i & 3u) * 8u);
                    v = v4u_t{t[0], t[1], t[2], t[3]};
                }
                *(lds_v4u*)(stage + ((st_end & 4095u) + 16u * lane)) = v;
                RL_FENCE();
                st_end += 1024u;
                if (st_end - st_lo > 4096u) st_lo = st_end - 4096u;
                if (lane == 0) RL_V64(lds->st_lo) = ((unsigned long long)st_end << 32) | st_lo;
                worked = true;
            }
        }
        // -- output: ring -> memory behind done_op --
        const uint32_t done = uni(RL_V32(lds->done_op));
        uint32_t fl = uni(RL_V32(lds->flushed));
        const bool req = stop || uni(RL_V32(lds->flush_req));
        const int32_t avail = (int32_t)(done - fl);
        RL_FENCE();
        if (avail > 0) {
            uint32_t n = 0;
            if (fl & 15u) { n = 16u - (fl & 15u); if (n > (uint32_t)avail) n = req ? (uint32_t)avail : 0u; if (lane < n) out[fl + lane] = ring[(fl + lane) & RL_MASK]; }
            else if (avail >= 1024) { n = 1024u; *(v4u_ua*)(out + fl + 16u * lane) = *(const lds_v4u*)(ring + ((fl + 16u * lane) & RL_MASK)); }
            else if (req) { n = avail < 64 ? (uint32_t)avail : 64u; if (lane < n) out[fl + lane] = ring[(fl + lane) & RL_MASK]; }
            if (n) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (in memory before `flushed` says so: a producer may read it from there)
                fl += n;
                if (lane == 0) RL_V32(lds->flushed) = fl;
                worked = true;
            }
        } else if (stop) return;
        if (!worked) { __builtin_amdgcn_s_sleep(2); if (stop && ++idle > RL_SPIN_CAP) return; }
    }
}

// ---- a speculator wave: for every payload position of a group of 64, what a window that started there would be ----
// Lane c walks the tokens from byte 64 g + c on by the producers' own rules (a token counts if it has no length bytes beyond the first literal
// one and its sequence ends within 64 bytes of the START): E = where the first token that does not fit stands, T = the output of those that do.
// The walk is per lane (two ds_bpermute per step: the bytes' parsed fields sit in the lane of the byte), all 64 starts at once; most of
// them are not tokens and nobody will ask - the one the chain lands on is among them, and its window's length is then one LDS read away.
template <int W, int S>
__device__ __forceinline__ void relay_speculator(const uint32_t sidx, RelayLds<W>* lds)
{
    const uint32_t lane = lane_id();
    lds_u8* const stage = (lds_u8*)lds->stage;
    lds_u32* const et = (lds_u32*)lds->et;
    auto info_at = [&](uint32_t q) -> uint32_t {                             // the byte at q read as a token: length of its sequence (clamped) | fits the lanes << 7 | output << 8
        const uint32_t a = q & ~3u;
        const uint32_t lo = *(const lds_u32*)(stage + (a & 4095u)), hi = *(const lds_u32*)(stage + ((a + 4u) & 4095u));
        const uint32_t d = (uint32_t)((((uint64_t)hi << 32) | lo) >> ((q & 3u) * 8u));
        const uint32_t t = d & 0xFFu, litn = t >> 4, ml = t & 15u, e1 = (d >> 8) & 0xFFu;
        const uint32_t hdr = litn == 15u ? 2u : 1u, lit = litn == 15u ? 15u + e1 : litn;
        const uint32_t len = hdr + lit + 2u + (ml == 15u ? 1u : 0u);
        const uint32_t ext = stage[(q + hdr + lit + 2u) & 4095u];               // (the match-length byte, if the token has one; read for nothing otherwise)
        const bool fine = (ml != 15u || ext <= RL_EXT_MAX) && !(litn == 15u && e1 == 255u) && len <= 64u;
        return (len > 127u ? 127u : len) | (fine ? 128u : 0u) | ((lit + ml + 4u + (ml == 15u ? ext : 0u)) << 8);
    };
    uint32_t g = sidx, idle = 0;
    DBP(unsigned long long z0 = clock64();)
    for (;;) {
        if (uni(RL_V32(lds->stop))) return;
        const uint32_t hint = uni(RL_V32(lds->pos_hint));
        const uint32_t gmin = hint >> 6;
        if (g < gmin) g = gmin + (sidx + S - gmin % S) % S;                  // (the reader jumped: the first group of mine at or behind it)
        const unsigned long long st = RL_V64(lds->st_lo);
        const uint32_t lo = uni((uint32_t)st), en = uni((uint32_t)(st >> 32));
        if (!(lo <= 64u * g && 64u * g + 208u <= en && 64u * g <= hint + 1024u)) {     // not staged yet, or far enough ahead
            __builtin_amdgcn_s_sleep(2);
            if (++idle > 4u * RL_SPIN_CAP) return;
            continue;
        }
        idle = 0;
        RL_FENCE();
        DBP(const unsigned long long z1 = clock64(); uint32_t n_it = 0;)
        const uint32_t ia = info_at(64u * g + lane), ib = info_at(64u * g + 64u + lane);
        auto look = [&](uint32_t x, uint32_t ta, uint32_t tb) -> uint32_t {     // entry x (0..127) of a table that lies in two registers
            const uint32_t xa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((x & 63u) << 2), (int)ta),
                           xb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((x & 63u) << 2), (int)tb);
            return (x & 64u) ? xb : xa;
        };
        // the same fields for two tokens in a row (0: not both for the lanes, or more than 64 bytes): the walk takes two at a time, then at most one
        auto two = [&](uint32_t x, uint32_t inf) -> uint32_t {
            const uint32_t nxp = x + (inf & 127u);
            const uint32_t ni = look(nxp & 127u, ia, ib);
            const uint32_t l2 = (inf & 127u) + (ni & 127u);
            return ((inf & 128u) && nxp < 128u && (ni & 128u) && l2 <= 64u) ? (l2 | 128u | (((inf >> 8) + (ni >> 8)) << 8)) : 0u;
        };
        const uint32_t ja = two(lane, ia), jb = two(64u + lane, ib);
        uint32_t cur = lane, rel = 0, tsum = 0;
        bool going = true;
        while (__ballot(going)) {
            const uint32_t inf = look(cur, ja, jb), len = inf & 127u;
            const bool ok = going && (inf & 128u) && rel + len <= 64u;
            if (ok) { tsum += inf >> 8; rel += len; cur += len; }
            going = ok && rel < 64u;
            DBP(n_it++;)
        }
        {
            const uint32_t inf = look(cur, ia, ib), len = inf & 127u;
            if (rel < 64u && (inf & 128u) && rel + len <= 64u) { tsum += inf >> 8; rel += len; }
        }
        const uint32_t q = 64u * g + lane;
        et[q & 2047u] = (((q >> 11) + 1u) << 19) | (tsum << 7) | rel;
        DBP(if (lane == 0 && blockIdx.x < 16) { const unsigned long long z2 = clock64(); atomicAdd(&g_dbprof[16], z1 - z0); atomicAdd(&g_dbprof[17], z2 - z1); atomicAdd(&g_dbprof[18], 1ull); atomicAdd(&g_dbprof[19], (unsigned long long)n_it); z0 = z2; })
        g += S;
    }
}

// ---- the finishing wave: the matches that had to wait, window by window, in stream order ----
// A producer copies what it can while the windows in front of its own are still in the works (literals; matches whose source lies below
// done_op) and leaves the rest of its window's matches here.  One wave takes them in order - the one chain of the block that is about
// bytes, not positions - and moves done_op along; nobody waits for a neighbour's window any more, and a window's step on this chain is
// its late matches (two or three), not a hand-over from wave to wave.
template <int W>
__device__ __forceinline__ void relay_finisher(RelayLds<W>* lds, BlockOut* __restrict__ entry)
{
    const uint32_t lane = lane_id();
    lds_u8* const ring = (lds_u8*)lds->ring;
    // A window's slot is read ahead - header, plan and list in one go, while the window before it waits for its bytes out of the ring: the LDS
    // performs a wave's reads in order, so what comes back behind a header with the right tag is what was written in front of it; a header
    // with the old tag means reading again.
    unsigned long long h = 0, pl = 0, tv = 0;
    auto read_slot = [&](uint32_t kk) {
        volatile lds_u64* const s_ = (volatile lds_u64*)lds->slot[kk % RL_Q];
        h = s_[0];
        pl = *(volatile lds_u64*)&lds->plan[kk % RL_Q][lane];
        tv = s_[1u + (lane < 22u ? lane : 0u)];
    };
    read_slot(0);
    for (uint32_t k = 0;; k++) {
        const uint32_t tagk = (k / RL_Q + 1u) & 0xFFFFu;
        uint32_t spins = 0;
        DBP(const unsigned long long f0 = clock64();)
        while ((uni((uint32_t)h) >> 16) != tagk) {
            if ((++spins & 31u) == 0u) {
                if (uni(RL_V32(lds->stop))) return;
                if (spins > 4u * RL_SPIN_CAP) { if (lane == 0) { entry->dst_size = (uint32_t)-1; RL_V32(lds->stop) = 1u; } return; }
            }
            read_slot(k);
        }
        RL_FENCE();
        const uint32_t count = uni((uint32_t)h) & 0xFFFFu, op_end = uni((uint32_t)(h >> 32));
        const unsigned long long pl_k = pl, tv_k = tv;
        DBP(const unsigned long long f1 = clock64();)
        {                                                                    // the planned ones: one read and one write for all of them
            const bool mine = (count & 0x8000u) && (uint32_t)pl_k != 0xFFFFFFFFu;
            uint8_t b = 0;
            if (mine) b = ring[(uint32_t)(pl_k >> 32) & RL_MASK];
            read_slot(k + 1);                                                // (on its way together with the ring's bytes)
            if (mine) ring[(uint32_t)pl_k & RL_MASK] = b;
        }
        if (const uint32_t nlist = count & 0x7FFFu) {                        // the listed ones, in order
            const uint32_t tlo = (uint32_t)tv_k, thi = (uint32_t)(tv_k >> 32);
            for (uint32_t j = 0; j < nlist; j++) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)tlo, (int)j), ok = (uint32_t)__builtin_amdgcn_readlane((int)thi, (int)j);
                const uint32_t dk = lo & 0xFFFFFFu, lk = lo >> 24;
                for (uint32_t c = 0; c < lk; c += WAVE) {                    // (up to 219 bytes: 64 at a time)
                    const uint32_t i = c + lane;
                    uint32_t idx = i;
                    if (ok < lk) idx = i % ok;                               // (overlapping: repeats its period - read from the period in front of the match)
                    if (i < lk) { const uint8_t b = ring[(dk - ok + idx) & RL_MASK]; ring[(dk + i) & RL_MASK] = b; }
                }
            }
        }
        RL_FENCE();
        if (lane == 0) { RL_V32(lds->done_op) = op_end; RL_V32(lds->fin_count) = k + 1u; }
        DBP(if (lane == 0 && blockIdx.x < 16) { const unsigned long long f2 = clock64(); atomicAdd(&g_dbprof[20], f1 - f0); atomicAdd(&g_dbprof[21], f2 - f1); atomicAdd(&g_dbprof[22], 1ull);
             if (count & 0x8000u) atomicAdd(&g_dbprof[23], 1ull); atomicAdd(&g_dbprof[24], (unsigned long long)(count & 0x7FFFu)); if (spins == 0) atomicAdd(&g_dbprof[25], 1ull); })
    }
}

// ---- the walker: the chain of positions ----
// pos += E(pos), op += T(pos), one read of the speculators' table per window, and the window goes on a list for the producers.  One wave does
// nothing else: handing the position from producer to producer (a write, the next wave's poll, its read of the table) was three LDS trips per
// window on the block's critical chain; this is one.  A window the lanes cannot take (E = 0, the block's end) goes on the list marked as a
// single sequence; the producer that gets it reports back where it left off.
template <int W>
__device__ __forceinline__ void relay_walker(uint32_t csize, uint32_t cap, RelayLds<W>* lds, BlockOut* __restrict__ entry)
{
    const uint32_t lane = lane_id();
    auto fail = [&]() { if (lane == 0) { entry->dst_size = (uint32_t)-1; RL_V32(lds->stop) = 1u; } };
    uint32_t pos = 0, op = 0, fin_seen = 0;
    for (uint32_t k = 0;; k++) {
        uint32_t spins = 0;
        while (k - fin_seen >= RL_QW) {                                      // (list entry k mod 16 is free once window k - 16 is finished)
            fin_seen = uni(RL_V32(lds->fin_count));
            if (k - fin_seen < RL_QW) break;
            if (uni(RL_V32(lds->stop))) return;
            if (++spins > RL_SPIN_CAP) { fail(); return; }
            __builtin_amdgcn_s_sleep(RL_SLEEP);
        }
        uint32_t etv = 0;
        if (pos <= csize && csize - pos >= 96u) {
            // (an entry of this lap of the table: the payload under it is staged)
            spins = 0;
            while ((etv = uni(*(volatile lds_u32*)((lds_u32*)lds->et + (pos & 2047u))), (etv >> 19) != (pos >> 11) + 1u)) {
                if ((++spins & 31u) == 0u) {
                    if (uni(RL_V32(lds->stop))) return;
                    if (spins > 4u * RL_SPIN_CAP) { fail(); return; }
                }
            }
        }
        // (E bytes of payload make T bytes; with T + 12 bytes of room none of the window's sequences can be the last one or run into the end-of-block rules)
        const uint32_t e = etv & 127u, t = (etv >> 7) & 4095u;
        const bool lanes = e != 0u && (uint64_t)op + t + 12u <= cap;
        const uint32_t tagk = (k / RL_QW) % 255u + 1u;
        if (lane == 0) {
            RL_V64(lds->wq[k % RL_QW][1]) = lanes ? (unsigned long long)(etv & 0x7FFFFu) : 0x80000000ull;
            RL_V64(lds->wq[k % RL_QW][0]) = ((unsigned long long)(op | (tagk << 24)) << 32) | (pos | (tagk << 24));
        }
        if (lanes) { pos += e; op += t; }
        else {
            const uint32_t mtag = k % 255u + 1u;
            unsigned long long m = 0;
            spins = 0;
            while ((m = RL_V64(lds->mail), uni((uint32_t)m >> 24) != mtag || uni((uint32_t)(m >> 32) >> 24) != mtag)) {
                if (uni(RL_V32(lds->stop))) return;
                if (++spins > 4u * RL_SPIN_CAP) { fail(); return; }
                __builtin_amdgcn_s_sleep(RL_SLEEP);
            }
            pos = uni((uint32_t)m) & 0xFFFFFFu; op = uni((uint32_t)(m >> 32)) & 0xFFFFFFu;
            if (lane == 0) RL_V64(lds->mail) = 0ull;                           // (taken: the tag comes round again every 255 windows)
        }
        if (lane == 0) RL_V32(lds->pos_hint) = pos;
    }
}

// ---- a producer wave ----
template <int W>
__device__ __forceinline__ void relay_producer(const uint32_t w, const uint8_t* __restrict__ in, uint32_t csize, uint8_t* out, uint32_t cap,
                                               RelayLds<W>* lds, BlockOut* __restrict__ entry)
{
    const uint32_t lane = lane_id();
    lds_u8* const ring = (lds_u8*)lds->ring;
    lds_u8* const stage = (lds_u8*)lds->stage;
    volatile lds_u32* const xp = (volatile lds_u32*)lds->expand[w];
    auto verdict = [&](int32_t got) {                                        // the block ends here
        if (lane == 0) { entry->dst_size = (uint32_t)got; RL_V32(lds->stop) = 1u; }
    };
    // polls: false = give up (stop seen, or stuck: the block is reported as failed)
#define RL_WAIT(cond)                                                                                      \
    { uint32_t spins_ = 0; bool ok_ = true;                                                                \
      while (!(cond)) {                                                                                    \
          if (uni(RL_V32(lds->stop))) { ok_ = false; break; }                                              \
          if (++spins_ > RL_SPIN_CAP) { verdict(-1); ok_ = false; break; }                                 \
          __builtin_amdgcn_s_sleep(RL_SLEEP);                                                              \
      }                                                                                                    \
      RL_FENCE();                                                                                          \
      if (!ok_) return; }
    auto staged = [&](uint32_t from, uint32_t upto) -> bool {                // payload [from, upto) is in `stage`
        const unsigned long long s = RL_V64(lds->st_lo);
        return uni((uint32_t)s) <= from && upto <= uni((uint32_t)(s >> 32));
    };
    auto lds32 = [&](uint32_t q) -> uint32_t {                               // 4 payload bytes from q on (any alignment)
        const uint32_t a = q & ~3u;
        const uint32_t lo = *(const lds_u32*)(stage + (a & 4095u)), hi = *(const lds_u32*)(stage + ((a + 4u) & 4095u));
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> ((q & 3u) * 8u));
    };
    uint32_t visit = 0;
    for (;;) {
        // ---- my next window, from the walker's list ----
        DBP(const unsigned long long y0 = clock64();)
        visit++;
        const uint32_t wk = (visit - 1u) * W + w, tag = (wk / RL_QW) % 255u + 1u;       // (wk: the window's number)
        uint32_t pos = 0, op = 0;
        {
            unsigned long long t = 0;
            RL_WAIT((t = RL_V64(lds->wq[wk % RL_QW][0]), uni((uint32_t)t >> 24) == tag && uni((uint32_t)(t >> 32) >> 24) == tag));
            pos = uni((uint32_t)t) & 0xFFFFFFu; op = uni((uint32_t)(t >> 32)) & 0xFFFFFFu;
        }
        DBP(const unsigned long long y1 = clock64();)
        const uint32_t etv = uni((uint32_t)RL_V64(lds->wq[wk % RL_QW][1]));
        const uint32_t spec_e = etv & 127u, spec_t = (etv >> 7) & 4095u;
        // ---- the lanes' path (decode.cuh has the why of every step) ----
        if (!(etv >> 31)) {
            const uint32_t d = lds32(pos + lane);
            DBP(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long y2 = clock64();)
            const uint32_t t = d & 0xFFu, litn = t >> 4, ml = t & 15u, e1 = (d >> 8) & 0xFFu;
            const uint32_t hdr = litn == 15u ? 2u : 1u, lit = litn == 15u ? 15u + e1 : litn;
            const uint32_t d2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + hdr + lit) << 2), (int)d);   // offset, match-length byte (of a token whose sequence ends inside the window)
            const uint32_t ext = (d2 >> 16) & 0xFFu, len = hdr + lit + 2u + (ml == 15u ? 1u : 0u);
            const bool easy = (ml != 15u || ext <= RL_EXT_MAX) && !(litn == 15u && e1 == 255u) && lane + len <= 64u;
            const uint32_t nx = easy ? lane + len : 255u;
            uint64_t mask = 0;
            uint32_t s = 0, sp = 0, n;
            do {
                n = (uint32_t)__builtin_amdgcn_readlane((int)nx, (int)s);
                asm("s_bitset1_b64 %0, %1" : "+s"(mask) : "s"(s));
                sp = s; s = n;
            } while (n < 64u);
            if (n > 64u) { mask &= ~(1ull << sp); s = sp; }
            if (!mask) { verdict(-1); return; }                              // (cannot be: the speculators walk by the same rules)
            {
                const bool is_tok = (mask >> lane) & 1ull;
                const uint32_t mlen = ml + 4u + (ml == 15u ? ext : 0u);
                const uint32_t tout = is_tok ? lit + mlen : 0u;
                const uint32_t incl = dpp_incl_scan_add(tout), ex = incl - tout;
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (s != spec_e || total != spec_t) { verdict(-1); return; }    // (cannot be: the speculators walk by the same rules)
                DBP(const unsigned long long y3 = clock64();)
                const uint32_t off = d2 & 0xFFFFu;
                const uint32_t mdst = op + ex + lit;                         // where the token's match goes
                if (__ballot(is_tok && (off == 0u || off > mdst))) { verdict(-1); return; }
                // the ring is 128 KiB: what this window overwrites must be in memory already (and is then 64 KiB behind every reader)
                RL_WAIT((int32_t)(uni(RL_V32(lds->flushed)) + 65536u - (op + total)) >= 0);
                // literals: byte lane l belongs to the nearest token at or below it
                const uint32_t g1 = dpp_incl_scan_max(is_tok ? lane + 1u : 0u);
                const uint32_t pg = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((g1 - 1u) << 2), (int)((ex << 8) | (lit << 1) | (hdr - 1u)));
                const uint32_t r = lane - g1 - (pg & 1u);
                if (lane >= g1 + (pg & 1u) && r < ((pg >> 1) & 127u)) ring[(op + (pg >> 8) + r) & RL_MASK] = (uint8_t)d;
                const uint32_t pa_mine = mdst | (mlen << 24);
                auto rounds = [&](bool elig) {                               // the eligible tokens' matches, 64 match bytes per round, a lane per byte
                    const uint32_t mc = elig ? mlen : 0u;                     // (eligible: at most 64 bytes)
                    const uint32_t inc = dpp_incl_scan_add(mc), exs = inc - mc;
                    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    uint32_t base = 0;
                    while (base < tot) {
                        const bool fits = elig && exs >= base && inc <= base + WAVE;
                        const uint64_t fm = __ballot(fits);
                        if (!fm) break;                                      // (never: a match here is at most 18 bytes)
                        const uint32_t nbase = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63 - (int)__builtin_clzll(fm));
                        xp[lane] = 0u;
                        if (fits) xp[exs - base] = lane + 1u;
                        const uint32_t k1 = dpp_incl_scan_max(xp[lane]);
                        const uint32_t kk = ((k1 ? k1 : 1u) - 1u) << 2;
                        const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)kk, (int)pa_mine),
                                       pb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)kk, (int)(off | ((exs - base) << 16)));
                        const uint32_t bi = lane - (pb >> 16), bd = pa & 0xFFFFFFu;
                        if (k1 && bi < (pa >> 24)) { const uint8_t b = ring[(bd - (pb & 0xFFFFu) + bi) & RL_MASK]; ring[(bd + bi) & RL_MASK] = b; }
                        base = nbase;
                    }
                };
                // Matches whose source is final already (below done_op) go now.  The first look comes when the finishing wave is at most RL_D windows
                // behind this one; the looks go on - each taking what has become final since - until that wave is RL_DPOST windows behind, and only
                // what is left then is left to it (it is the block's one chain of copies: every match it does not get shortens the block's time).
                RL_WAIT((int32_t)(uni(RL_V32(lds->fin_count)) + RL_D - wk) > 0);
                bool early = false;
                for (uint32_t looks = 0;; looks++) {
                    const uint32_t fc = uni(RL_V32(lds->fin_count)), snap = uni(RL_V32(lds->done_op));      // (done_op is written in front of fin_count: this done_op is at least that window's)
                    const bool ready = is_tok && !early && mlen <= WAVE && (int32_t)(snap - (mdst - off + mlen)) >= 0;
                    const uint64_t rm = __ballot(ready);
                    if (rm) { rounds(ready); early = early || ready; }
                    if ((int32_t)(fc + RL_DPOST - wk) > 0 || !__ballot(is_tok && !early)) break;
                    if (!rm) {
                        if (uni(RL_V32(lds->stop))) return;
                        if (looks > RL_SPIN_CAP) { verdict(-1); return; }
                        __builtin_amdgcn_s_sleep(RL_SLEEP);
                    }
                }
                DBP(const unsigned long long y4 = clock64();)
                // the others go to the finishing wave, in stream order (slot k mod 16: free once window k - 16 is finished)
                const bool latet = is_tok && !early;
                const uint64_t late = __ballot(latet);
                DBP(const unsigned long long y5 = clock64();)
                {
                    lds_u64* const sl = (lds_u64*)lds->slot[wk % RL_Q];
                    uint32_t count = (uint32_t)__builtin_popcountll(late);
                    if (count) {
                        // Those that read nothing another one of them writes (the source ends in front of the first one's place; the exact test, pair
                        // by pair, found 3 % more and cost more than it gave) go as a byte-per-lane plan, 64 bytes at most: one read and one write for
                        // the finishing wave, whatever their number.  The others - they are later in the stream than what they read, and the planned
                        // ones read nothing of theirs - follow as a list, in order.
                        const bool clash = latet && mdst - off + mlen > (uint32_t)__builtin_amdgcn_readlane((int)mdst, (int)__builtin_ctzll(late));
                        bool planned = latet && !clash;
                        const uint32_t mc = planned ? mlen : 0u;
                        const uint32_t inc = dpp_incl_scan_add(mc), exs = inc - mc;
                        if ((uint32_t)__builtin_amdgcn_readlane((int)inc, 63) > WAVE) planned = false;      // (too many bytes for one step: all of them by the list)
                        const uint64_t pm = __ballot(planned), lm = late & ~pm;
                        count = (uint32_t)__builtin_popcountll(lm);
                        if (pm) {
                            xp[lane] = 0u;
                            if (planned) xp[exs] = lane + 1u;
                            const uint32_t k1 = dpp_incl_scan_max(xp[lane]);
                            const uint32_t kk = ((k1 ? k1 : 1u) - 1u) << 2;
                            const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)kk, (int)pa_mine),
                                           pb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)kk, (int)(off | (exs << 16)));
                            const uint32_t bi = lane - (pb >> 16), bd = pa & 0xFFFFFFu;
                            const bool mine = k1 && bi < (pa >> 24);
                            ((lds_u64*)lds->plan[wk % RL_Q])[lane] = mine ? ((unsigned long long)(bd - (pb & 0xFFFFu) + bi) << 32) | (bd + bi) : 0xFFFFFFFFull;
                            count |= 0x8000u;
                        }
                        if (lm) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                            if (latet && !planned) sl[1u + rank] = ((unsigned long long)off << 32) | pa_mine;
                        }
                    }
                    RL_FENCE();
                    if (lane == 0) RL_V64(lds->slot[wk % RL_Q][0]) = ((unsigned long long)(op + total) << 32) | ((((wk / RL_Q) + 1u) & 0xFFFFu) << 16) | count;
                }
                DBP(const unsigned long long c_e = __builtin_popcountll(__ballot(early)), c_o = __builtin_popcountll(late);)
                DBP(if (lane == 0 && blockIdx.x < 16) { const unsigned long long y6 = clock64(); atomicAdd(&g_dbprof[0], y1 - y0); atomicAdd(&g_dbprof[1], y2 - y1); atomicAdd(&g_dbprof[2], y3 - y2);
                     atomicAdd(&g_dbprof[3], y4 - y3); atomicAdd(&g_dbprof[4], y5 - y4); atomicAdd(&g_dbprof[5], y6 - y5); atomicAdd(&g_dbprof[6], 1ull);
                     atomicAdd(&g_dbprof[7], c_e); atomicAdd(&g_dbprof[9], c_o); atomicAdd(&g_dbprof[26], (unsigned long long)(wk - RL_V32(lds->fin_count))); })
                continue;
            }
        }
        // ---- one sequence in stream order, behind everything in front of it ----
        RL_WAIT(uni(RL_V32(lds->done_op)) == op);
        RL_WAIT((int32_t)(uni(RL_V32(lds->flushed)) + 65536u - (op + 2u * WAVE)) >= 0);      // (the ring: see the lanes' path)
        auto fetch = [&](uint32_t qq, bool& gone) -> uint64_t {              // 8 payload bytes from qq on, wave-uniform
            if (!staged(qq, qq + 12u)) {
                if (lane == 0) RL_V32(lds->pos_hint) = qq;
                uint32_t spins = 0;
                while (!staged(qq, qq + 12u)) {
                    if (uni(RL_V32(lds->stop))) { gone = true; return 0; }
                    if (++spins > RL_SPIN_CAP) { verdict(-1); gone = true; return 0; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const uint32_t a = qq & ~3u;
            const uint32_t w0 = uni(*(const lds_u32*)(stage + (a & 4095u))), w1 = uni(*(const lds_u32*)(stage + ((a + 4u) & 4095u))),
                           w2 = uni(*(const lds_u32*)(stage + ((a + 8u) & 4095u)));
            const uint32_t sh = (qq & 3u) * 8u;
            uint64_t v = (((uint64_t)w1 << 32) | w0) >> sh;
            if (sh) v |= (uint64_t)w2 << (64u - sh);
            return v;
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
        // a copy longer than a wave goes to memory directly: the service wave has to be done with everything in front of it first
        auto memory_is_current = [&]() -> bool {
            if (lane == 0) RL_V32(lds->flush_req) = 1u;
            uint32_t spins = 0;
            while (uni(RL_V32(lds->flushed)) != op) {
                if (uni(RL_V32(lds->stop))) return false;
                if (++spins > RL_SPIN_CAP) { verdict(-1); return false; }
                __builtin_amdgcn_s_sleep(1);
            }
            return true;
        };
        auto mirror_and_publish = [&](uint32_t from, uint32_t nbytes) {      // out[from, from + nbytes) was written in memory: its last 64 KiB into the ring
            uint32_t f = from, c = nbytes;
            if (c > 65536u) { f += c - 65536u; c = 65536u; }
            for (uint32_t i = lane; i < c; i += WAVE) ring[(f + i) & RL_MASK] = out[f + i];
            RL_FENCE();
            if (lane == 0) { RL_V32(lds->flushed) = from + nbytes; RL_V32(lds->done_op) = from + nbytes; RL_V32(lds->flush_req) = 0u; }
        };
        DBP(if (lane == 0 && blockIdx.x < 16) atomicAdd(&g_dbprof[10], 1ull);)
        bool gone = false;
        uint64_t wv = fetch(pos, gone);
        if (gone) return;
        const uint32_t token = (uint32_t)wv & 0xFF;
        uint32_t lit = token >> 4, p = pos + 1;
        bool bad = false;
        if (lit == 15) {
            const uint64_t x = wv >> 8;                                      // 7 candidate length bytes, top byte 0 (never 0xFF)
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), k = f >> 3;
            lit = 15u + 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
            p = pos + 2 + k;
            if (k == 7) lit = 15u + ext_slow(pos + 1, p, bad);
        }
        if (bad || p > csize) { verdict(-1); return; }
        const uint32_t in_left = csize - p, out_left = cap - op;
        const bool is_last = (uint64_t)lit + 12 > out_left || (uint64_t)lit + 8 > in_left;
        if (is_last && (lit != in_left || lit > out_left)) { verdict(-1); return; }
        if (lit) {
            if (lit <= WAVE) {
                if (!staged(p, p + WAVE)) {
                    if (lane == 0) RL_V32(lds->pos_hint) = p;
                    RL_WAIT(staged(p, p + WAVE));
                }
                const uint8_t b = stage[(p + lane) & 4095u];
                if (lane < lit) ring[(op + lane) & RL_MASK] = b;
                RL_FENCE();
                if (lane == 0) RL_V32(lds->done_op) = op + lit;
            } else {
                if (!memory_is_current()) return;
                wave_copy_disjoint(out + op, in + p, lit);
                mirror_and_publish(op, lit);
            }
            op += lit;
        }
        if (is_last) { verdict((int32_t)op); return; }
        const uint32_t qo = p + lit;
        wv = fetch(qo, gone);
        if (gone) return;
        const uint32_t offset = (uint32_t)wv & 0xFFFF;
        uint32_t mlen = token & 15, npos = qo + 2;
        if (mlen == 15) {
            const uint64_t x = wv >> 16;                                     // 6 candidate length bytes
            const uint32_t f = (uint32_t)__builtin_ctzll(~x), k = f >> 3;
            mlen = 15u + 255u * k + (uint32_t)((x >> (f & 56u)) & 0xFF);
            npos = qo + 3 + k;
            if (k == 6) mlen = 15u + ext_slow(qo + 2, npos, bad);
            if (npos + 4 >= csize) bad = true;
        }
        mlen += 4;
        if (bad || offset == 0 || offset > op || (uint64_t)mlen + 5 > (uint64_t)(cap - op)) { verdict(-1); return; }   // (last 5 bytes must be literals)
        if (mlen <= WAVE) {                                                  // the usual short match: a byte per lane, out of the ring
            uint32_t idx = lane;
            if (offset < mlen) idx = lane % offset;                          // (an overlapping one repeats its period)
            if (lane < mlen) { const uint8_t b = ring[(op - offset + idx) & RL_MASK]; ring[(op + lane) & RL_MASK] = b; }
            RL_FENCE();
            if (lane == 0) RL_V32(lds->done_op) = op + mlen;
        } else {
            if (!memory_is_current()) return;
            wave_copy_match(out + op, offset, mlen);
            mirror_and_publish(op, mlen);
        }
        op += mlen;
        RL_FENCE();
        if (lane == 0) RL_V64(lds->mail) = ((unsigned long long)(op | ((wk % 255u + 1u) << 24)) << 32) | (npos | ((wk % 255u + 1u) << 24));      // (the walker goes on from here)
        if (lane == 0) RL_V64(lds->slot[wk % RL_Q][0]) = ((unsigned long long)op << 32) | ((((wk / RL_Q) + 1u) & 0xFFFFu) << 16);     // (nothing left to finish: the finishing wave moves on)
    }
#undef RL_WAIT
}

// workgroup-per-block decode of independent blocks: W producer waves, S speculator waves, the service wave, the finishing wave, the walker; 148 KiB of LDS (one workgroup to a CU)
template <int W, int S>
__global__ __launch_bounds__(64 * (W + S + 3)) void k_decode_blocks_relay(const uint8_t* __restrict__ frame, uint8_t* dst, BlockOut* __restrict__ table,
                                                                      const ResultRec* __restrict__ res, uint32_t n_max, uint64_t frame_cap,
                                                                      const uint32_t* __restrict__ only_if)
{
    static_assert(W >= 2 && S >= 1 && W + S + 3 <= 16, "1024 threads");
    __shared__ RelayLds<W> lds;
    if (res->status != ST_OK) return;
    if (only_if && *only_if == 0) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    const uint32_t w = uni(threadIdx.x >> 6);
    const BlockOut e = table[b];
    const uint32_t csz = e.word & 0x7FFFFFFFu;
    if (e.word >> 31) {                                                      // stored: a plain copy
        if (w != 0) return;
        int32_t got = -2;
        if (csz <= e.dst_size) { wave_copy_disjoint(dst + e.dst_off, frame + e.src_off, csz); got = (int32_t)csz; }
        if (lane_id() == 0) table[b].dst_size = (uint32_t)got;
        return;
    }
    if (csz == 0 || csz > (1u << 23) || e.dst_size > (1u << 23)) {           // (positions travel in 24 bits; a frame block is at most 4 MiB)
        if (threadIdx.x == 0) table[b].dst_size = (uint32_t)-1;
        return;
    }
    if (threadIdx.x < RL_QW) lds.wq[threadIdx.x][0] = 0ull;
    for (uint32_t i = threadIdx.x; i < 2048u; i += 64 * (W + S + 3)) lds.et[i] = 0u;
    if (threadIdx.x < RL_Q) lds.slot[threadIdx.x][0] = 0ull;                                  // (lap 0: no entry)
    if (threadIdx.x == 0) { lds.st_lo = 0; lds.st_end = 0; lds.done_op = 0; lds.flushed = 0; lds.pos_hint = 0; lds.stop = 0; lds.flush_req = 0; lds.fin_count = 0; lds.mail = 0ull; }
    __syncthreads();
    if (w == W + S + 2) relay_walker<W>(csz, e.dst_size, &lds, &table[b]);
    else if (w == W + S + 1) relay_finisher<W>(&lds, &table[b]);
    else if (w == W + S) relay_service<W>(frame + e.src_off, csz, frame_cap - e.src_off, dst + e.dst_off, &lds);
    else if (w >= W) relay_speculator<W, S>(w - W, &lds);
    else relay_producer<W>(w, frame + e.src_off, csz, dst + e.dst_off, e.dst_size, &lds, &table[b]);
}

}  // namespace lz4f
