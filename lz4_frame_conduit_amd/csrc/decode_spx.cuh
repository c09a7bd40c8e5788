// decode_spx.cuh -- foreign frames of BIG INDEPENDENT blocks (what `lz4 -c` and liblz4's LZ4F_compressFrame write by default with
// -B7: the frames the reference's decompress tests feed, /root/reference/test/Main.hs:33-36) onto the indexed kernels
// (SURVEY.md section 8a rows a3/a4).
//
// Such a frame brings no index, and the parse of a 4 MiB block is one dependent chain: 4 k sequences at 1280 cycles each on the
// fused decoder's scalar parser (2.3 ms per block, decode_fused.cuh), ~10 ms for a lane at memory latency (k_selfindex_walk).
// Here the chain is cut:
//   k_spx_index   a workgroup per block, a LANE per 32 KiB segment of the payload.  Lane u starts at a GUESS - it knows no token
//                 position in its segment - and follows the sequences from there to where lane u + 1 starts.  A chain that starts
//                 at a wrong byte is garbage, but a chain is a function of its position: lane 0 starts at byte 0 and is true, and
//                 lane u + 1 is true iff the (true) chain in front of it lands EXACTLY on its start.  One thread stitches: it
//                 follows the landings from lane to lane; where the true chain does not land on the next lane's start (a wrong
//                 guess) it walks on itself, from where the chain stands, until it meets a later lane's start.  Equality of
//                 positions is the whole proof, so a wrong guess can cost time, never change the result.
//                 Guessing well is what makes lanes meet.  Long literal runs (the payloads where a guessed chain would need ~20 KiB
//                 to fall into step: it hops 10-20 bytes at a time through 500 random bytes) announce themselves: a token with
//                 literal length >= 270 is a byte 0xF? followed by 0xFF - one position in 4096 of random bytes.  The waves scan the
//                 first KiB(s) of every segment for such pairs (64 lanes x 16 bytes of ONE segment per load), and a lane starts at
//                 the first pair of its segment whose sequence leads to another one (bench frame: 1-4 of 68 000 lanes guess wrong).
//                 Pairs but no token among them: the lane stays out and the lane in front walks on through its stretch.  No pair in
//                 8 KiB: short sequences, where a chain started anywhere falls into step within a few hundred bytes - or a literal
//                 run longer than that (the first 60 KiB of the bench's blocks, whose copies' sources lie in front of the block): a
//                 lane that then crawls (an eighth of the bytes per hop the density probe saw) gives up after 64 hops.
//                 Result per block: the sequence count, the output size, and a true (position, sequence number, output position)
//                 at the start of every stretch.
//   k_spx_parse   a lane per stretch: the sequences of that stretch -> descriptors, with parse_run()
//                 - the code k_parse_indexed runs per index entry, same rules, same direct-match marking.  Every stretch must end
//                 exactly where the next one starts, the last at the payload's end with the block's last sequence.
// Behind them k_resolve_direct and k_copy_indexed run as for a frame that came with the compressor's index.  Anything odd - a
// malformed payload, short blocks inside the frame, a stretch that does not end where it should - sets flags[0], and the generic
// decoder launched behind decodes the frame and gives the verdict.
#pragma once
#include "decode_indexed.cuh"

namespace lz4f {

#ifndef SPX_SEG_BYTES
#define SPX_SEG_BYTES 32768
#endif
constexpr uint32_t SPX_SEG = SPX_SEG_BYTES, SPX_SCAN = 8192, SPX_CAND = 6, SPX_MAXSEG = (4u << 20) / SPX_SEG;      // lanes per block <= 128
constexpr uint32_t SPX_LANE_HOPS = 2048;                        // sequences a lane follows before it gives its stretch up (sparse payloads: >= 24 bytes per sequence, 1400 per stretch)
constexpr uint32_t SPX_RESCUE = 1u << 16;                       // sequences the stitching thread may walk itself per block before it gives up
constexpr uint32_t SPX_NONE = 0xFFFFFFFFu;
constexpr uint32_t SPX_SUB = 8, SPX_SUB_STRIDE = 16;            // round 4: a lane also notes where it stands every 16 sequences (a stride that doubles when its eight places are full) - see k_spx_index
constexpr uint32_t SPX_MAXPT = SPX_MAXSEG * (SPX_SUB + 1);      // points per block: every stretch's start and what its lane noted on the way
struct SpxPoint { uint32_t pos, seq, out; };                    // a token position of the block's payload, the number of the sequence that starts there, its output position

// One sequence of the payload in[0, csize) at c.pos: lengths only (the walk that finds out WHERE sequences are; parse_run checks
// what they say).  Returns 0: went on to the next token; 1: that was the block's last sequence (c.pos = csize); 2: not a sequence
// this payload can hold.  One dependent 16-byte load per sequence (the word at the match offset holds the next token as well).
struct SpxCur { uint32_t pos, seq, out; uint64_t w, w_hi; };
__device__ __forceinline__ void spx_begin(const uint8_t* __restrict__ in, uint64_t readable, SpxCur& c) { pt_load16(in, c.pos, readable, c.w, c.w_hi); }
__device__ __forceinline__ int spx_step(const uint8_t* __restrict__ in, uint32_t csize, uint64_t readable, SpxCur& c)
{
    if (c.pos >= csize) return 2;
    const uint32_t token = (uint32_t)c.w & 0xFF;
    uint32_t lit = token >> 4, p = c.pos + 1;
    if (lit == 15) {
        const uint64_t x = c.w >> 8;
        const uint32_t f = (uint32_t)__builtin_ctzll(~x), kk = f >> 3;
        if (kk < 7) { lit += 255u * kk + (uint32_t)((x >> (f & 56u)) & 0xFF); p += kk + 1; }
        else { for (;;) { if (p >= csize || lit > (1u << 24)) return 2; const uint32_t v = in[p++]; lit += v; if (v != 255) break; } }
    }
    if (p > csize || lit >= (1u << 24)) return 2;
    const uint32_t in_left = csize - p;
    c.seq++;
    if (lit + 8 > in_left) { if (lit != in_left) return 2; c.out += lit; c.pos = csize; return 1; }
    const uint32_t q = p + lit;
    uint64_t w2, w2_hi;
    pt_load16(in, q, readable, w2, w2_hi);
    uint32_t mlen = token & 15, pn = q + 2;
    bool reload = false;
    if (mlen == 15) {
        const uint64_t x = w2 >> 16;
        const uint32_t f = (uint32_t)__builtin_ctzll(~x), kk = f >> 3;
        if (kk < 6) { mlen += 255u * kk + (uint32_t)((x >> (f & 56u)) & 0xFF); pn += kk + 1; }
        else { reload = true; for (;;) { if (pn >= csize || mlen > (1u << 24)) return 2; const uint32_t v = in[pn++]; mlen += v; if (v != 255) break; } }
    }
    c.out += lit + mlen + 4;
    c.pos = pn;
    if (c.out > (1u << 23)) return 2;
    if (reload) pt_load16(in, c.pos, readable, c.w, c.w_hi);
    else { const uint32_t sh = (pn - q) * 8u; c.w = sh >= 64 ? w2_hi : ((w2 >> sh) | (w2_hi << (64u - sh))); }
    return 0;
}

// lanes of a block: lane 0 starts at byte 0; lane u >= 1 exists while the stretch it looks for its start in lies inside the payload with room behind it
__device__ __forceinline__ uint32_t spx_lanes(uint32_t csize)
{
    if (csize <= SPX_SEG + SPX_SCAN + 2048u) return 1u;
    const uint32_t n = (csize - SPX_SCAN - 2048u - 1u) / SPX_SEG + 1u;   // largest u with u * SEG + SCAN + 2 KiB < csize, + 1
    return n < SPX_MAXSEG ? n : SPX_MAXSEG;
}

// Is `at` a likely token?  Its sequence must lead to the payload's end or to another token that announces a long literal run.
__device__ __forceinline__ bool spx_likely_token(const uint8_t* __restrict__ in, uint32_t csize, uint64_t readable, uint32_t at)
{
    SpxCur c{at, 0, 0, 0, 0};
    spx_begin(in, readable, c);
    const int r = spx_step(in, csize, readable, c);
    if (r == 1) return true;
    if (r == 0 && c.pos + 2 < csize) {
        const uint32_t t = (uint32_t)c.w & 0xFFFFu;
        return (t & 0xF0u) == 0xF0u && (t >> 8) == 0xFFu;
    }
    return false;
}

// ---- k_spx_index: a workgroup per block ----
// T (SPX_MAXPT + 1 points per block): T[k] = where run k starts (k = 0: byte 0) - a stretch's start, or what its lane noted on the way -, T[runs] = the payload's end; nr[b]: runs.
__global__ __launch_bounds__(SPX_MAXSEG) void k_spx_index(const uint8_t* __restrict__ frame, uint64_t frame_cap, const BlockOut* __restrict__ table,
                                                   const ResultRec* __restrict__ res, uint32_t n_max, uint32_t* __restrict__ cnt, uint32_t* __restrict__ osz,
                                                   SpxPoint* __restrict__ T, uint32_t* __restrict__ nr, uint32_t* __restrict__ flags,
                                                   const uint32_t* __restrict__ only_if)
{
    __shared__ uint32_t s_g[SPX_MAXSEG + 1];                    // where lane u starts: its guess (lane 0: byte 0); [lanes]: SPX_NONE
    __shared__ uint32_t s_cand[SPX_MAXSEG][SPX_CAND], s_ncand[SPX_MAXSEG];      // per lane: the first pairs of its segment
    __shared__ SpxPoint s_b[SPX_MAXSEG];                        // per lane: where its chain lands at or behind the NEXT lane's start (counts relative to its own start); pos SPX_NONE: nowhere
    __shared__ uint32_t s_end[SPX_MAXSEG];                      // per lane: 1 = s_b is the payload's end (pos == csize: the counts include the last sequence)
    // Round 4: the runs the copy kernel's first wave parses (k_copy_selffed) are as long as its chain of dependent loads, and a stretch of 32 KiB is 63 sequences
    // on the bench frame: a lane notes where it stands every SPX_SUB_STRIDE sequences (counts relative to its own start, like s_b), up to SPX_SUB places - when they
    // are full every other one goes and the stride doubles - and a true lane's notes become points of their own behind the stitch.
    __shared__ SpxPoint s_sub[SPX_MAXSEG][SPX_SUB];
    __shared__ uint32_t s_nsub[SPX_MAXSEG];
    __shared__ uint32_t s_tidx[SPX_MAXSEG], s_bseq[SPX_MAXSEG], s_bout[SPX_MAXSEG];      // behind the stitch, per lane: where its points go (SPX_NONE: not a true lane), the true counts at its start
    __shared__ uint32_t s_fail;
    if (res->status != ST_OK) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t b = blockIdx.x, u = threadIdx.x;
    if (b >= n) return;
    if (only_if && *only_if == 0) { if (u == 0) { cnt[b] = 0; osz[b] = 0; nr[b] = 0; atomicOr(flags, 1u); } return; }      // (dense payloads: not this path)
    const BlockOut e = table[b];
    const uint32_t csize = e.word & 0x7FFFFFFFu;
    if (e.word >> 31) { if (u == 0) { cnt[b] = 0; osz[b] = csize; nr[b] = 0; } return; }
    if (csize == 0 || e.src_off + csize > frame_cap) { if (u == 0) { atomicOr(flags, 1u); cnt[b] = 0; osz[b] = 0; nr[b] = 0; } return; }
    const uint8_t* in = frame + e.src_off;
    const uint64_t readable = frame_cap - e.src_off;
    const uint32_t nl = spx_lanes(csize);
    // ---- every lane's start: a guess (no overlap between lanes: lane u walks from its start to where lane u + 1 starts, and is
    // ---- followed by a true lane iff it lands exactly there) ----
#ifdef SPX_PROF
    const unsigned long long z0 = clock64();
#endif
    // ---- where could a lane start?  The "0xF? 0xFF" pairs in the first KiB(s) of every lane's segment, found by the WAVE: 64 lanes x 16
    // ---- bytes of one segment per load (a lane scanning its own segment 8 bytes at a time is 64 cache lines per wave instruction, and
    // ---- the slowest lane of 2000 waves took 0.35 ms over it); a segment is read on, a KiB at a time, until it has shown a pair.
    {
        const uint32_t lane = u & 63u, wbase = u & ~63u;
        for (uint32_t t = 0; t < 64; t++) {
            const uint32_t uo = wbase + t;                                   // (wave-uniform)
            if (uo == 0 || uo >= nl) continue;
            const uint32_t from = uo * SPX_SEG;
            uint32_t n = 0;
            for (uint32_t a = from; a < from + SPX_SCAN && a + 1024u + 24u < csize && n == 0; a += 1024u) {
                const uint8_t* q = in + a + lane * 16u;
                typedef uint64_t u64u __attribute__((aligned(1)));
                const uint64_t lo = *(const u64u*)q, hi = *(const u64u*)(q + 8);
                const uint32_t nb = (uint32_t)q[16];                       // (the byte behind my sixteen: a pair may straddle two lanes)
                // byte i is 0xF?, byte i + 1 is 0xFF (bit tricks that stay inside their byte)
                const uint64_t nlo = (lo >> 8) | (hi << 56), nhi = (hi >> 8) | ((uint64_t)nb << 56);
                uint64_t m0 = lo & (lo >> 2); m0 = m0 & (m0 >> 1) & 0x1010101010101010ull;
                uint64_t m1 = hi & (hi >> 2); m1 = m1 & (m1 >> 1) & 0x1010101010101010ull;
                uint64_t f0 = nlo & (nlo >> 4); f0 = f0 & (f0 >> 2); f0 = f0 & (f0 >> 1) & 0x0101010101010101ull;
                uint64_t f1 = nhi & (nhi >> 4); f1 = f1 & (f1 >> 2); f1 = f1 & (f1 >> 1) & 0x0101010101010101ull;
                const uint64_t c0 = (m0 >> 4) & f0, c1 = (m1 >> 4) & f1;      // bit 0 of byte i
                // sixteen bits per lane, bit i = byte i of my sixteen starts a pair
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) { bits |= (uint32_t)((c0 >> (8 * i)) & 1ull) << i; bits |= (uint32_t)((c1 >> (8 * i)) & 1ull) << (8 + i); }
                uint64_t have = __ballot(bits != 0);
                while (have && n < SPX_CAND) {                               // the first few, in position order
                    const uint32_t L = (uint32_t)__builtin_ctzll(have);
                    uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)bits, (int)L);
                    while (bl && n < SPX_CAND) { const uint32_t i = (uint32_t)__builtin_ctz(bl); bl &= bl - 1; if (lane == 0) s_cand[uo][n] = a + L * 16u + i; n++; }
                    have &= have - 1;
                }
            }
            if (lane == 0) s_ncand[uo] = n;
        }
    }
    __syncthreads();
    // ---- every lane's start: the first of its segment's pairs that looks like a token (see spx_likely_token).  Pairs and none of them
    // ---- a token: the lane stays out - a chain guessed into a long literal run hops ~10 bytes at a time (3000 hops through a stretch,
    // ---- the whole kernel waiting for that one lane), and the lane in front walks on through this stretch instead.  No pair at all
    // ---- in SPX_SCAN bytes: short sequences, where a chain started anywhere falls into step within a few hundred bytes.
    if (u < nl) {
        uint32_t g = 0;
        if (u) {
            const uint32_t n = s_ncand[u];
            g = n ? SPX_NONE : u * SPX_SEG;
            for (uint32_t k = 0; k < n; k++) { const uint32_t at = s_cand[u][k]; if (spx_likely_token(in, csize, readable, at)) { g = at; break; } }
        }
        s_g[u] = g;
    }
    if (u == nl) s_g[u] = SPX_NONE;
#ifdef SPX_PROF
    const unsigned long long z1 = clock64();
#endif
    __syncthreads();
#ifdef SPX_PROF
    const unsigned long long z2 = clock64();
#endif
    if (u < nl) {
        SpxPoint pb{SPX_NONE, 0, 0};
        uint32_t ended = 0, nsub = 0;
        if (s_g[u] != SPX_NONE) {
            uint32_t stop = SPX_NONE;                                        // the next lane that has a start (none: this lane walks to the end)
            for (uint32_t k = u + 1; k < nl; k++) if (s_g[k] != SPX_NONE && s_g[k] > s_g[u]) { stop = s_g[k]; break; }
            SpxCur c{s_g[u], 0, 0, 0, 0};
            spx_begin(in, readable, c);
            // (a lane that started without a pair to go by - none in SPX_SCAN bytes - may stand in a literal run longer than that, where its
            // chain crawls ~20 bytes per hop: if it makes less than an eighth of the way per hop that the probe saw sequences make in this
            // frame, it gives up after 64 hops and the stitching thread walks the stretch)
            // A lane that started at a pair may crawl as well: a wrong guess that looked like a token - a handful per 4 GiB - makes 900 hops
            // through its 32 KiB before it falls into step, and in most frames the whole kernel waited 0.6 ms for that one lane.  (The probe
            // cannot say what a hop should make where every block begins with a long literal run: it then has nothing to count.)  Such a
            // lane's premise was a token that announces a long literal run, in a stretch made of them; a chain through random bytes finds
            // one token in sixteen like that.  After 64 hops with fewer than half of them: it gives up, and the stitching thread walks
            // the stretch.
            const uint32_t slow = (u && s_g[u] == u * SPX_SEG && only_if) ? only_if[1] / 8u : 0u;
            const bool by_pair = u && s_g[u] != u * SPX_SEG;
            uint32_t longs = 0, sub_stride = SPX_SUB_STRIDE;
            for (uint32_t hops = 0;; hops++) {
                if (c.pos >= stop) { pb = SpxPoint{c.pos, c.seq, c.out}; break; }
                if (hops && hops % sub_stride == 0) {                        // (c.seq == hops: a token, its number and the output in front of it, all from my start)
                    if (nsub == SPX_SUB) { for (uint32_t i = 0; i < SPX_SUB / 2; i++) s_sub[u][i] = s_sub[u][2 * i + 1]; nsub = SPX_SUB / 2; sub_stride *= 2; }
                    if (hops % sub_stride == 0) s_sub[u][nsub++] = SpxPoint{c.pos, c.seq, c.out};
                }
                if ((hops & 63u) == 63u) {
                    if (slow && c.pos - s_g[u] < hops * slow) break;
                    if (by_pair && longs * 2u < hops) break;
                }
                longs += ((uint32_t)c.w & 0xF0u) == 0xF0u ? 1u : 0u;
#ifdef SPX_PROF
                if (hops > SPX_LANE_HOPS) atomicAdd(&flags[48], 1u);
#endif
                if (hops > SPX_LANE_HOPS) break;                             // (a chain that crawls is a wrong guess: the stitching thread walks this stretch)
                const int r = spx_step(in, csize, readable, c);
                if (r == 1) { pb = SpxPoint{c.pos, c.seq, c.out}; ended = 1; break; }      // the payload's end (for a guessed chain: what looks like it)
                if (r == 2) break;                                           // (a guessed chain may run into anything)
            }
        }
        s_b[u] = pb; s_end[u] = ended; s_nsub[u] = pb.pos != SPX_NONE ? nsub : 0u;
#ifdef SPX_PROF
        if (s_g[u] == SPX_NONE) atomicAdd(&flags[49], 1u);
        else if (u && s_g[u] == u * SPX_SEG) atomicAdd(&flags[50], 1u);
        if (pb.pos == SPX_NONE && s_g[u] != SPX_NONE) atomicAdd(&flags[51], 1u);
        atomicMax(&flags[52], pb.seq);
#endif
    }
#ifdef SPX_PROF
    const unsigned long long z3 = clock64();
#endif
    __syncthreads();
#ifdef SPX_PROF
    const unsigned long long z4 = clock64();
    if ((u & 63) == 0) { atomicMax(&flags[40], (uint32_t)(z1 - z0)); atomicMax(&flags[41], (uint32_t)(z3 - z2)); atomicAdd(&flags[44], (uint32_t)((z1 - z0) >> 8)); atomicAdd(&flags[45], (uint32_t)((z3 - z2) >> 8)); }
#endif
    // ---- the stitch: lane 0 starts at byte 0 and is true; lane k + 1 is true iff the true chain lands exactly on its start.  A lane
    // ---- that is not (a wrong guess) is skipped: the thread walks on from where the true chain stands until it meets a later lane's start.
    SpxPoint* Tb = T + (size_t)b * (SPX_MAXPT + 1);
    if (u < SPX_MAXSEG) s_tidx[u] = SPX_NONE;
    if (u == 0) s_fail = 0;
    __syncthreads();
    uint32_t tpos = 0, total_seq = 0, total_out = 0;
    if (u == 0) {
        uint32_t k = 0, rescue = 0;
        bool bad = false;
        SpxPoint cur{0, 0, 0};                                               // the true chain: where the stretch of lane k starts
        for (;;) {
            // lane k starts where the true chain stands (cur.pos == s_g[k]): its walk is the true chain's, and so are its notes
            s_tidx[k] = tpos; s_bseq[k] = cur.seq; s_bout[k] = cur.out;
            tpos += 1u + s_nsub[k];
            SpxPoint nx{SPX_NONE, 0, 0}; uint32_t ended = 0;
            if (s_b[k].pos != SPX_NONE) { nx = SpxPoint{s_b[k].pos, cur.seq + s_b[k].seq, cur.out + s_b[k].out}; ended = s_end[k]; }
            else { bad = true; break; }                                      // (a true chain that runs into something: malformed payload)
            uint32_t kn = k + 1;
            while (!ended && !(kn < nl && s_g[kn] == nx.pos)) {
                // the true chain did not land on the next lane's start (or there is none): walk on from where it stands, to the start of
                // the first lane it meets (or the end); that lane's own walk is the true chain again
                if (kn < nl && (s_g[kn] == SPX_NONE || s_g[kn] < nx.pos)) { kn++; continue; }      // (a lane without a start, or one whose start lies behind us already)
                const uint32_t stop = kn < nl ? s_g[kn] : SPX_NONE;
                atomicAdd(&flags[2], 1u);                                    // (how often: a developer's number, LZ4F_MI355X_PROF prints it)
                SpxCur c{nx.pos, nx.seq, nx.out, 0, 0};
                spx_begin(in, readable, c);
                for (;;) {
                    if (c.pos >= stop) break;
                    if (++rescue > SPX_RESCUE) { bad = true; break; }
                    const int r = spx_step(in, csize, readable, c);
                    if (r == 1) { ended = 1; break; }
                    if (r == 2) { bad = true; break; }
                }
                if (bad) break;
                nx = SpxPoint{c.pos, c.seq, c.out};
            }
            if (bad) break;
            if (ended) {
                if (nx.pos != csize) { bad = true; break; }
                total_seq = nx.seq; total_out = nx.out; break;
            }
            k = kn; cur = nx;
        }
#ifdef SPX_PROF
        { const unsigned long long z5 = clock64(); atomicMax(&flags[42], (uint32_t)(z5 - z4)); atomicMax(&flags[43], (uint32_t)(z5 - z0)); atomicAdd(&flags[46], (uint32_t)((z5 - z4) >> 8)); }
#endif
        if (bad || total_seq == 0) { atomicOr(flags, 1u); cnt[b] = 0; osz[b] = 0; nr[b] = 0; s_fail = 1; }
        else {
            Tb[tpos] = SpxPoint{csize, total_seq, total_out};                // the end, as the last run's stop
            cnt[b] = total_seq; osz[b] = total_out; nr[b] = tpos;
        }
    }
    __syncthreads();
    // ---- every true lane writes its points: its start, and what it noted on the way, in true counts ----
    if (s_fail || u >= nl || s_tidx[u] == SPX_NONE) return;
    {
        const uint32_t at = s_tidx[u], bs = s_bseq[u], bo = s_bout[u], ns = s_nsub[u];
        Tb[at] = SpxPoint{s_g[u], bs, bo};
        for (uint32_t i = 0; i < ns; i++) { const SpxPoint q = s_sub[u][i]; Tb[at + 1 + i] = SpxPoint{q.pos, bs + q.seq, bo + q.out}; }
    }
}

// ---- k_spx_parse: a lane per stretch ----
__global__ __launch_bounds__(128) void k_spx_parse(const uint8_t* __restrict__ frame, uint64_t frame_cap, const BlockOut* __restrict__ table,
                                                  const ResultRec* __restrict__ res, uint32_t n_max, const void* __restrict__ ix,
                                                  const SpxPoint* __restrict__ T, const uint32_t* __restrict__ nr, SeqDesc* __restrict__ desc,
                                                  uint32_t* __restrict__ flags)
{
    if (res->status != ST_OK || *flags) return;
    const uint32_t n = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t b = blockIdx.x;
    if (b >= n) return;
    const uint32_t stretches = nr[b];
    const uint64_t desc_cap = flags[9];
    const IxBlock blk = ix_blocks(ix)[b];
    const BlockOut e = table[b];
    const uint32_t csize = e.word & 0x7FFFFFFFu;
    if (stretches && ((e.word >> 31) || e.src_off + csize > frame_cap || (uint64_t)blk.seq_base + blk.nseq > desc_cap)) { if (threadIdx.x == 0) atomicOr(flags, 1u); return; }
    const SpxPoint* Tb = T + (size_t)b * (SPX_MAXPT + 1);
    for (uint32_t k = threadIdx.x; k < stretches; k += blockDim.x) {
        const SpxPoint from = Tb[k], to = Tb[k + 1];
        const bool is_tail = k + 1 == stretches;
        bool bad = to.seq <= from.seq || to.seq > blk.nseq || (is_tail && (to.seq != blk.nseq || to.pos != csize)) || (k == 0 && (from.pos | from.seq | from.out) != 0);
        uint32_t pos = from.pos;
        if (!bad) bad = parse_run(frame + e.src_off, csize, frame_cap - e.src_off, pos, from.out, to.seq - from.seq, is_tail, desc + blk.seq_base + from.seq, 0u, 0ull);
        if (!bad && pos != to.pos) bad = true;                                   // must end exactly where the next stretch starts
        if (bad) atomicOr(flags, 1u);
    }
}


// ---- round 4: the stretches as the runs of the self-feeding copy kernel (decode_indexed.cuh: k_copy_selffed) - k_spx_parse and
// ---- k_resolve_direct are then not launched: the workgroup's first wave parses a round of stretches and resolves slot by slot
struct FzRunsSpx {
    const SpxPoint* T; uint32_t n, nseq_blk, csize;
    __device__ __forceinline__ uint32_t count() const { return n; }
    __device__ __forceinline__ FzRun get(uint32_t u) const
    {
        const SpxPoint from = T[u], to = T[u + 1];
        FzRun r{from.pos, from.out, from.seq, to.seq - from.seq, to.pos, u + 1 == n, false};
        r.bad = to.seq <= from.seq || to.seq > nseq_blk || (r.tail && (to.seq != nseq_blk || to.pos != csize)) || (u == 0 && (from.pos | from.seq | from.out) != 0) || from.pos >= csize;
        return r;
    }
};
struct FzSrcSpx {
    const SpxPoint* T; const uint32_t* nr;
    typedef FzRunsSpx Runs;
    __device__ __forceinline__ bool make(uint32_t b, const IxBlock& blk, uint32_t csize, uint32_t, FzRunsSpx& r) const
    {
        const uint32_t stretches = nr[b];
        if (stretches == 0 || stretches > SPX_MAXPT) return false;
        r = FzRunsSpx{T + (size_t)b * (SPX_MAXPT + 1), stretches, blk.nseq, csize};
        return true;
    }
};

}  // namespace lz4f
