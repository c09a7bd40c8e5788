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
