// decode_2k.cuh -- two-kernel LZ4 block decode for large blocks.
//
// At 4 MiB blocks a 4 GiB stream has only 1024 independent blocks (4 per CU).  The sequence PARSE of a
// block is one serial dependent chain; the COPIES are where the bytes are.  So they are separate kernels:
//
//   k_parse_blocks   one WAVE per block, everything wave-uniform (SALU).  The payload streams through two
//                    16 KiB LDS stages per wave, filled by direct-to-LDS loads (global_load_lds_dwordx4,
//                    1 KiB per instruction) one stage AHEAD of the parser, so the chain
//                    token -> lengths -> next token never waits on HBM: one 8-byte LDS read per sequence.
//                    Output: one 16-byte descriptor per sequence {lit_src, lit_len, dst, match_len, offset},
//                    gathered 64 at a time in VGPR lanes (v_cndmask on lane == slot) and stored as one coalesced 1 KiB write.
//   k_copy_blocks    one WORKGROUP (8 waves) per block walks the descriptors in batches of 512:
//                    literals (no dependencies at all) are copied HBM -> HBM 16 B per lane by all waves; a
//                    match is "safe" when its source does not overlap the destination of an earlier match of
//                    the same batch (one lane per sequence, binary search in LDS) and all safe matches are
//                    copied in parallel; the rest are replayed in stream order by one wave.
// Both kernels enforce the accept/reject rules of wave_decode_block (decode.cuh) / the oracle.
#pragma once
#include "common.cuh"
#include "decode.cuh"

namespace lz4f {

constexpr uint32_t PK_STAGE = 16384;            // payload bytes per LDS stage
constexpr uint32_t PK_OVER = 64;                // stages overlap: an 8-byte header read never straddles a stage end
constexpr int      PK_WAVES = 4;                // parser waves (= blocks) per workgroup: 4 x 2 x 16.06 KiB = 128.5 KiB LDS
constexpr int      CK_WAVES = 8;                // copy kernel: waves per workgroup
constexpr int      CK_NB = 512;                 // descriptors per batch (one per thread)

// descriptor: x = lit_src | off[7:0] << 24, y = lit_len | off[15:8] << 24, z = dst, w = match_len (0: last sequence)
// (positions and lengths are < 2^23 because a block holds at most 4 MiB)
struct SeqDesc { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t max_seq_per_block(uint32_t block_size) { return block_size / 4 + 2; }


// issue the direct-to-LDS loads of payload bytes [s*STAGE, min(csize, (s+1)*STAGE + OVER)) into `slot`
__device__ __forceinline__ void pk_stage_issue(uint8_t* slot, const uint8_t* __restrict__ in, uint32_t csize, uint32_t s)
{
    const uint32_t lane = lane_id();
    const uint32_t base = s * PK_STAGE;
    if (base >= csize) return;
    const uint32_t end = (base + PK_STAGE + PK_OVER < csize) ? base + PK_STAGE + PK_OVER : csize;
    const uint32_t span = end - base;
    for (uint32_t piece = 0; piece < span; piece += 1024) {
        const uint32_t o = piece + lane * 16;
        if (o + 16 <= span) __builtin_amdgcn_global_load_lds((gptr_t)(in + base + o), (lptr_t)(slot + piece), 16, 0, 0);
    }
    const uint32_t tail0 = span & ~15u;                          // ragged tail (< 16 B): bytewise, never past the payload
    if (lane < span - tail0) slot[tail0 + lane] = in[base + tail0 + lane];
}

// 8 payload bytes at position q from the stage slots (all operands wave-uniform).
// The three dword reads are inline asm on purpose: with ordinary LDS loads hipcc inserts `s_waitcnt vmcnt(0)` in front
// of every read (a direct-to-LDS load may be pending, and it cannot know that it targets the OTHER slot), which would
// serialise the parser behind its own prefetch.  Ordering is kept by hand: k_parse_blocks waits vmcnt(0) exactly when
// it switches to a freshly loaded stage.
__device__ __forceinline__ uint64_t pk_fetch8(const uint8_t* stages /* [2][STAGE+OVER] */, uint32_t q)
{
    const uint32_t s = q / PK_STAGE;
    const uint32_t o = q - s * PK_STAGE;
    const uint32_t addr = (uint32_t)(uintptr_t)(lptr_t)(stages + (s & 1) * (PK_STAGE + PK_OVER)) + (o & ~3u);
    uint64_t w01; uint32_t w2;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w01), "=&v"(w2) : "v"(addr) : "memory");
    const uint64_t lo = uni64(w01);
    const uint32_t hi = uni(w2);
    const uint32_t shv = (o & 3u) * 8;
    // funnel shift of the 96-bit value {hi, lo} by shv in {0, 8, 16, 24}
    return (lo >> shv) | (((uint64_t)hi << 32) << (32 - shv));
}

// ------------------------------- kernel 1: parse -------------------------------------------------
// seq_count[b] = number of descriptors of block b, or 0xFFFFFFFF when the block is malformed;
// out_size[b]  = decoded size implied by the sequences.
__global__ __launch_bounds__(64 * PK_WAVES) void k_parse_blocks(const uint8_t* __restrict__ frame, const BlockOut* __restrict__ table,
                                                                const ResultRec* __restrict__ res, uint32_t n_max, uint32_t block_size, uint32_t linked,
                                                                SeqDesc* __restrict__ desc, uint32_t* __restrict__ seq_count, uint32_t* __restrict__ out_size)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[PK_WAVES][2][PK_STAGE + PK_OVER];
    if (res->status != ST_OK) return;
    const uint32_t n_blocks = res->n_blocks < n_max ? res->n_blocks : n_max;
    const uint32_t wave = uni(threadIdx.x >> 6), lane = lane_id();
    const uint32_t blk = uni(blockIdx.x * PK_WAVES + wave);
    if (blk >= n_blocks) return;
    const BlockOut e = table[blk];
    if (e.word >> 31) { if (lane == 0) { seq_count[blk] = 0; out_size[blk] = e.word & 0x7FFFFFFFu; } return; }   // stored block: nothing to parse
    const uint32_t csize = e.word & 0x7FFFFFFFu;
    const uint32_t cap = linked ? block_size : e.dst_size;       // linked frames: exact room is checked by the copy kernel
    const uint8_t* __restrict__ in = frame + e.src_off;
    uint8_t* stages = &s_stage[wave][0][0];
    SeqDesc* dout = desc + (uint64_t)blk * max_seq_per_block(block_size);
    const uint32_t dcap = max_seq_per_block(block_size);

    uint32_t nseq = 0, status = 0;
    uint32_t q = 0, op = 0;                      // payload cursor / output cursor
    uint64_t w = 0; uint32_t avail = 0;          // w holds the next `avail` payload bytes at q
    int32_t  cur = -1;                           // stage the parser is reading (resident)
    bool     next_issued = false;                // stage cur+1 has been requested into the other slot
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;     // 64 descriptors gathered across the lanes

    if (csize == 0) status = 1;
    else { pk_stage_issue(stages, in, csize, 0); next_issued = true; }
    // make the stage that holds position qq readable.  Common case: still the current stage.  Next stage: wait for
    // the loads issued one stage ago, then immediately request the one after it over the slot that just died.
    auto need = [&](uint32_t qq) {
        const int32_t s = (int32_t)(qq / PK_STAGE);
        if (s == cur) return;
        if (!(s == cur + 1 && next_issued)) {                                    // a jump (long literal run): load it now
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // never refill a slot that is still being filled
            pk_stage_issue(stages + (uint32_t)(s & 1) * (PK_STAGE + PK_OVER), in, csize, (uint32_t)s);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur = s;
        pk_stage_issue(stages + (uint32_t)((s + 1) & 1) * (PK_STAGE + PK_OVER), in, csize, (uint32_t)s + 1);
        next_issued = true;
    };
    auto refill = [&]() { need(q); w = pk_fetch8(stages, q); avail = 8; };
    // LZ4 length bytes (255, 255, ..., <255) decoded without a per-byte loop: the number of leading 0xFF bytes of
    // the window is ctz(~w) / 8.  Returns the value added by the length bytes; q/w/avail advance past them.
    auto len_bytes = [&]() -> uint32_t {
        uint32_t add = 0;
        for (;;) {
            if (avail == 0) refill();
            const uint64_t inv = ~w;
            const uint32_t k = inv ? (uint32_t)(__builtin_ctzll(inv) >> 3) : 8u;      // leading 0xFF bytes in the window
            if (k < avail) {                                                          // terminator is in the window
                add += 255u * k + (uint32_t)((w >> (8 * k)) & 0xFF);
                const uint32_t used = k + 1;
                w = used < 8 ? (w >> (8 * used)) : 0; avail -= used; q += used;
                return add;
            }
            add += 255u * avail; q += avail; avail = 0;                               // all 0xFF so far: keep going
            if (add > 0x7FFF0000u || q >= csize) { status = 1; return add; }
        }
    };

    while (status == 0) {
        // ---- token + literal length ----
        if (q >= csize) { status = 1; break; }
        if (avail == 0) refill();
        const uint32_t token = (uint32_t)w & 0xFF; w >>= 8; avail--; q++;
        uint32_t lit = token >> 4;
        if (lit == 15) { lit += len_bytes(); if (status) break; }
        if (q > csize) { status = 1; break; }
        const uint32_t in_left = csize - q, out_left = cap - op;
        const uint32_t lit_src = q;
        uint32_t mlen = 0, off = 0;
        const bool is_last = (uint64_t)lit + 12 > out_left || (uint64_t)lit + 8 > in_left;
        if (is_last) {
            if (lit != in_left || lit > out_left) { status = 1; break; }
        } else {
            // ---- match header (>= 8 payload bytes remain after the literals) ----
            if (lit) { q += lit; if (lit >= avail) avail = 0; else { w >>= 8 * lit; avail -= lit; } }
            if (avail < 2) refill();
            off = (uint32_t)w & 0xFFFF; w >>= 16; avail -= 2; q += 2;
            if (off == 0) { status = 1; break; }
            if (off > op + lit + (linked ? 65535u : 0u)) { status = 1; break; }     // exact history check for linked frames: copy kernel
            mlen = token & 15;
            if (mlen == 15) {
                mlen += len_bytes(); if (status) break;
                if (q + 4 >= csize) { status = 1; break; }                          // length bytes may not reach the last 5 payload bytes
            }
            mlen += 4;
            if ((uint64_t)mlen + 5 > (uint64_t)(cap - (op + lit))) { status = 1; break; }
        }
        // ---- commit the descriptor into lane (nseq & 63) ----
        if (nseq >= dcap) { status = 1; break; }
        const uint32_t slot = nseq & 63;
        const bool mine = lane == slot;                  // select, not branch: the loop stays scalar
        d0 = mine ? (lit_src | ((off & 0xFFu) << 24)) : d0;
        d1 = mine ? (lit | ((off >> 8) << 24)) : d1;
        d2 = mine ? op : d2;
        d3 = mine ? mlen : d3;
        nseq++;
        op += lit + mlen;
        if (slot == 63) {                         // 64 gathered: one coalesced 1 KiB store, no lane predicate (keeps the loop scalar)
            SeqDesc d; d.x = d0; d.y = d1; d.z = d2; d.w = d3; dout[nseq - 64 + lane] = d;
        }
        if (is_last) break;
    }
    if (status == 0 && (nseq & 63) && lane < (nseq & 63)) { SeqDesc d; d.x = d0; d.y = d1; d.z = d2; d.w = d3; dout[(nseq & ~63u) + lane] = d; }
    if (lane == 0) { seq_count[blk] = status ? 0xFFFFFFFFu : nseq; out_size[blk] = op; }
}

// ------------------------------- kernel 2: copy --------------------------------------------------
constexpr uint32_t CK_MAX_LEVEL = 12;           // dependency chains deeper than this are replayed in stream order
constexpr uint32_t CK_SERIAL = 255;
struct alignas(16) CkShared {
    uint32_t lit_src[CK_NB], lit_len[CK_NB], dst[CK_NB], mlen[CK_NB], moff[CK_NB];
    uint32_t dep[CK_NB];                        // jlo | jhi << 16: earlier matches of the batch whose destination overlaps my source
    uint32_t level[CK_NB];                      // 0: no in-batch dependency; L: longest dependency chain; CK_SERIAL: replay in order
    uint32_t bad, any_long, changed, max_level, n_serial, pad[3];
};

// Replays the descriptors of one block with the whole workgroup.  `hist`: valid bytes in front of `out`.
// Returns false when a descriptor violates a bound that only this kernel can check (linked-frame history).
__device__ __forceinline__ bool wg_copy_block(CkShared& sh, const uint8_t* __restrict__ in, uint8_t* out, const SeqDesc* __restrict__ dsc, uint32_t nseq,
                                              uint64_t hist, uint32_t room, uint32_t dbg)
{
    const uint32_t tid = threadIdx.x, wave = uni(tid >> 6);
    bool ok = true;
    for (uint32_t b0 = 0; b0 < nseq; b0 += CK_NB) {
        const uint32_t n = uni((nseq - b0 < (uint32_t)CK_NB) ? nseq - b0 : (uint32_t)CK_NB);
        __syncthreads();                                                // previous batch is completely done with LDS
        if (tid == 0) { sh.bad = 0; sh.any_long = 0; sh.changed = 0; sh.max_level = 0; sh.n_serial = 0; }
        if (tid < n) {
            const SeqDesc d = dsc[b0 + tid];
            sh.lit_src[tid] = d.x & 0xFFFFFFu; sh.lit_len[tid] = d.y & 0xFFFFFFu; sh.dst[tid] = d.z; sh.mlen[tid] = d.w;
            sh.moff[tid] = (d.x >> 24) | ((d.y >> 24) << 8);
            sh.level[tid] = 0;
        }
        __syncthreads();
        // ---- dependency test (one lane per sequence): which earlier matches of this batch write bytes my match reads?
        // Match destinations D_j = [dm_j, dm_j + mlen_j) are disjoint and increasing in j, so the overlapping ones form a
        // contiguous range [jlo, jhi] found by two binary searches.
        bool has_dep = false; uint32_t jlo = 0, jhi = 0;
        if (tid < n && !(dbg & 4)) {
            const uint32_t k = tid;
            const uint32_t ml = sh.mlen[k];
            const uint32_t ll = sh.lit_len[k], dd = sh.dst[k];
            if ((uint64_t)dd + ll + ml > room) atomicOr(&sh.bad, 1u);
            if (ll > 65536) atomicOr(&sh.any_long, 1u);
            if (ml) {
                const uint32_t dm = dd + ll;
                const uint32_t off = sh.moff[k];
                if ((uint64_t)off > (uint64_t)dm + hist) atomicOr(&sh.bad, 1u);
                const int64_t s_start = (int64_t)dm - off;
                const int64_t s_end = (off < ml) ? (int64_t)dm : s_start + ml;
                uint32_t lo = 0, hi = k;                                // #(j < k with dm_j < s_end)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((int64_t)(sh.dst[mid] + sh.lit_len[mid]) < s_end) lo = mid + 1; else hi = mid;
                }
                const uint32_t cnt_hi = lo;
                lo = 0; hi = k;                                         // #(j < k with dm_j + mlen_j <= s_start)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((int64_t)(sh.dst[mid] + sh.lit_len[mid]) + sh.mlen[mid] <= s_start) lo = mid + 1; else hi = mid;
                }
                jlo = lo;
                if (cnt_hi > jlo) { has_dep = true; jhi = cnt_hi - 1; sh.level[k] = 1; }
            }
        }
        __syncthreads();
        if (uni(sh.bad)) { ok = false; break; }
        // ---- levels by relaxation: level[k] = 1 + max(level[jlo..jhi]); converges in (longest chain) rounds ----
        for (uint32_t it = 0; it < CK_MAX_LEVEL + 1; it++) {
            if (has_dep) {
                uint32_t m = 0;
                if (jhi - jlo >= 8) m = CK_SERIAL;                       // very wide source: do not bother, replay in order
                else for (uint32_t j = jlo; j <= jhi; j++) { const uint32_t l = sh.mlen[j] ? sh.level[j] : 0; m = l > m ? l : m; }
                uint32_t nl = m >= CK_MAX_LEVEL ? CK_SERIAL : m + 1;
                if (nl != sh.level[tid]) { sh.level[tid] = nl; sh.changed = 1; }
            }
            __syncthreads();
            const uint32_t ch = uni(sh.changed);
            __syncthreads();
            if (!ch) break;
            if (tid == 0) sh.changed = 0;
            __syncthreads();
        }
        if (has_dep) { const uint32_t l = sh.level[tid]; if (l == CK_SERIAL) atomicAdd(&sh.n_serial, 1u); else atomicMax(&sh.max_level, l); }
        // ---- literals: no dependencies.  Each wave takes every 8th sequence, 4 at a time: 4 loads in flight, then 4 stores.
        for (uint32_t i0 = wave; i0 < n && !(dbg & 1); i0 += 4 * CK_WAVES) {
            uint32_t len[4]; const uint8_t* sp[4]; uint8_t* dp[4]; Piece pc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k = i0 + u * CK_WAVES;
                const uint32_t kk = k < n ? k : 0;
                len[u] = k < n ? uni(sh.lit_len[kk]) : 0;
                sp[u] = in + uni(sh.lit_src[kk]); dp[u] = out + uni(sh.dst[kk]);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (len[u] && len[u] <= 1024) piece_load(pc[u], sp[u], len[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) if (len[u] && len[u] <= 1024) piece_store(pc[u], dp[u], len[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) if (len[u] > 1024 && len[u] <= 65536) wave_copy_disjoint(dp[u], sp[u], len[u]);
        }
        if (uni(sh.any_long)) {                                         // very long runs: sliced over all waves
            for (uint32_t k = 0; k < n; k++) {
                const uint32_t len = uni(sh.lit_len[k]);
                if (len <= 65536) continue;
                const uint32_t per = (((len + CK_WAVES - 1) / CK_WAVES) + 15) & ~15u;
                const uint32_t a = wave * per;
                if (a < len) wave_copy_disjoint(out + uni(sh.dst[k]) + a, in + uni(sh.lit_src[k]) + a, (len - a < per) ? len - a : per);
            }
        }
        __syncthreads();                                                // literals of this batch are in memory
        // ---- matches level by level: everything a level-L match reads was written by literals or by levels < L ----
        const uint32_t max_level = uni(sh.max_level), n_serial = uni(sh.n_serial);
        for (uint32_t L = 0; L <= max_level && !(dbg & 2); L++) {
            if (L) __syncthreads();
            for (uint32_t i0 = wave; i0 < n; i0 += 4 * CK_WAVES) {
                uint32_t ml[4], mo[4]; uint8_t* dp[4]; Piece pc[4]; bool fast[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t k = i0 + u * CK_WAVES;
                    const uint32_t kk = k < n ? k : 0;
                    ml[u] = (k < n && uni(sh.level[kk]) == L) ? uni(sh.mlen[kk]) : 0;
                    mo[u] = uni(sh.moff[kk]);
                    dp[u] = out + uni(sh.dst[kk]) + uni(sh.lit_len[kk]);
                    fast[u] = ml[u] && ml[u] <= 1024 && mo[u] >= ml[u];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) if (fast[u]) piece_load(pc[u], dp[u] - mo[u], ml[u]);
#pragma unroll
                for (int u = 0; u < 4; u++) if (fast[u]) piece_store(pc[u], dp[u], ml[u]);
#pragma unroll
                for (int u = 0; u < 4; u++) if (ml[u] && !fast[u]) wave_copy_match(dp[u], mo[u], ml[u]);
            }
        }
        if (n_serial) {                                                 // chains deeper than CK_MAX_LEVEL: stream order, one wave
            __syncthreads();
            if (wave == 0) {
                for (uint32_t k = 0; k < n; k++)
                    if (uni(sh.level[k]) == CK_SERIAL) wave_copy_match(out + uni(sh.dst[k]) + uni(sh.lit_len[k]), uni(sh.moff[k]), uni(sh.mlen[k]));
            }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(64 * CK_WAVES) void k_copy_blocks(const uint8_t* __restrict__ frame, uint8_t* dst, uint64_t dst_cap,
                                                               BlockOut* __restrict__ table, const ResultRec* __restrict__ res,
                                                               uint32_t n_max, uint32_t linked, uint32_t block_size, uint64_t hist0,
                                                               const SeqDesc* __restrict__ desc, const uint32_t* __restrict__ seq_count,
                                                               const uint32_t* __restrict__ out_size, uint32_t dbg)
{
    __shared__ CkShared sh;
    if (res->status != ST_OK) return;
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
                const uint32_t per = (((csz + CK_WAVES - 1) / CK_WAVES) + 15) & ~15u;
                const uint32_t a = (tid >> 6) * per;
                if (a < csz) wave_copy_disjoint(dst + at + a, frame + e.src_off + a, (csz - a < per) ? csz - a : per);
                got = (int32_t)csz;
            }
        } else {
            const uint32_t ns = seq_count[b];
            if (ns == 0xFFFFFFFFu || out_size[b] > room) got = -1;
            else {
                const bool ok = wg_copy_block(sh, frame + e.src_off, dst + at, desc + (uint64_t)b * max_seq_per_block(block_size), ns,
                                              linked ? out + hist0 : 0, room, dbg);
                got = ok ? (int32_t)out_size[b] : -1;
            }
        }
        if (tid == 0) { table[b].dst_off = at; table[b].dst_size = (uint32_t)got; }
        if (got < 0) {
            if (linked) for (uint32_t k = b + 1 + tid; k < n; k += 64 * CK_WAVES) table[k].dst_size = 0;
            break;
        }
        out += (uint32_t)got;
        __syncthreads();
    }
}

}  // namespace lz4f
