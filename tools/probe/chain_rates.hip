// Probe: how fast can ONE wave run a dependent chain on gfx950?  (XXH32's accumulator: v = rotl(v + m, 13) * P1; the LZ4 parser's
// "next token" chain.)  One wave per workgroup, one workgroup per CU, so nothing else competes for the issue slots.
//   0  VALU chain, operand from v_readlane (what hipcc makes of the C++)
//   1  SALU chain (inline asm: s_add, s_lshl, s_lshr, s_or, s_mul_i32), operand from v_readlane
//   2  VALU chain in lanes 0..3, operand read from LDS (ds_read_b32, immediate offsets)
//   3  four independent SALU chains interleaved (one wave doing all four accumulators)
//   4  SALU dependent adds only (s_add_u32 x 8 per step): single-wave dependent SALU issue interval
//   5  v_readlane -> s_add (lane index from the chain): the parser's "next = next_of[lane]" hop
//   6  ds_read_b32 uniform address dependent chain (LDS pointer chase)
//   7  ds_bpermute dependent chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define P1 0x9E3779B1u
template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* out, const uint32_t* __restrict__ in, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) lds[i] = (KIND == 6) ? ((i * 37u + 11u) & 4095u) * 4u : i * 2654435761u + seed;
    __syncthreads();
    uint32_t m = in[lane] * 0x85EBCA77u;
    uint32_t v = seed, v2 = seed + 1, v3 = seed + 2, v4 = seed + 3;
    const int N = 256;                                    // steps of 64
    const unsigned long long t0 = clock64();
    if (KIND == 0) {
        for (int it = 0; it < N; it++) {
#pragma unroll
            for (int i = 0; i < 64; i++) { const uint32_t x = v + (uint32_t)__builtin_amdgcn_readlane(m, i); v = ((x << 13) | (x >> 19)) * P1; }
            m += 0x1234567u;
        }
    }
    if (KIND == 1) {
        uint32_t s = __builtin_amdgcn_readfirstlane(seed);
        for (int it = 0; it < N; it++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                const uint32_t mi = (uint32_t)__builtin_amdgcn_readlane(m, i);
                uint32_t t1, t2;
                asm volatile("s_add_u32 %0, %0, %3\n\ts_lshl_b32 %1, %0, 13\n\ts_lshr_b32 %2, %0, 19\n\ts_or_b32 %0, %1, %2\n\ts_mul_i32 %0, %0, %4"
                             : "+s"(s), "=&s"(t1), "=&s"(t2) : "s"(mi), "s"(P1) : "scc");
            }
            m += 0x1234567u;
        }
        v = s;
    }
    if (KIND == 2) {
        const uint32_t base = (lane & 3u) * 4u;
        for (int it = 0; it < N; it++) {
#pragma unroll
            for (int i = 0; i < 64; i++) { const uint32_t x = v + *(const uint32_t*)((const uint8_t*)lds + base + i * 16 + (it & 15) * 1024); v = ((x << 13) | (x >> 19)) * P1; }
        }
    }
    if (KIND == 3) {
        uint32_t s1 = __builtin_amdgcn_readfirstlane(seed), s2 = s1 + 1, s3 = s1 + 2, s4 = s1 + 3;
        for (int it = 0; it < N; it++) {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                const uint32_t mi = (uint32_t)__builtin_amdgcn_readlane(m, i);
                uint32_t t1, t2;
#define STEP(S) asm volatile("s_add_u32 %0, %0, %3\n\ts_lshl_b32 %1, %0, 13\n\ts_lshr_b32 %2, %0, 19\n\ts_or_b32 %0, %1, %2\n\ts_mul_i32 %0, %0, %4" : "+s"(S), "=&s"(t1), "=&s"(t2) : "s"(mi), "s"(P1) : "scc");
                STEP(s1) STEP(s2) STEP(s3) STEP(s4)
            }
            m += 0x1234567u;
        }
        v = s1 ^ s2 ^ s3 ^ s4;
    }
    if (KIND == 4) {
        uint32_t s = __builtin_amdgcn_readfirstlane(seed);
        for (int it = 0; it < N * 8; it++)
            asm volatile("s_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1" : "+s"(s) : "s"(seed) : "scc");
        v = s;
    }
    if (KIND == 5) {
        uint32_t nxt = (lane * 37u + 11u) & 63u;           // a permutation walk
        uint32_t s = 0;
        for (int it = 0; it < N * 64; it++) s = (uint32_t)__builtin_amdgcn_readlane(nxt, s);
        v = s;
    }
    if (KIND == 6) {
        uint32_t a = 0;
        for (int it = 0; it < N * 64; it++) a = __builtin_amdgcn_readfirstlane(*(const uint32_t*)((const uint8_t*)lds + a));
        v = a;
    }
    if (KIND == 7) {
        uint32_t a = lane;
        for (int it = 0; it < N * 64; it++) a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a * 4u), (int)((a * 37u + 11u) & 63u));
        v = a;
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = v ^ v2 ^ v3 ^ v4; }
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 4096 * 16);
    uint32_t* in; hipMalloc(&in, 4096); hipMemset(in, 0x5A, 4096);
    const char* names[8] = {"VALU chain + readlane", "SALU chain + readlane", "VALU chain, 4 lanes, LDS operand", "4 SALU chains interleaved", "SALU dependent s_add x8", "readlane hop chain", "LDS pointer chase (uniform)", "ds_bpermute chain"};
    const double per[8] = {256.0 * 64, 256.0 * 64, 256.0 * 64, 256.0 * 64, 256.0 * 8 * 8, 256.0 * 64, 256.0 * 64, 256.0 * 64};
    for (int kind = 0; kind < 8; kind++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (kind) {
#define L(K) case K: hipLaunchKernelGGL(k<K>, dim3(256), dim3(64), 0, 0, d, in, 7u); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
            }
            hipDeviceSynchronize();
        }
        unsigned long long h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; i++) avg += h[2 * i]; avg /= 256;
        printf("%-36s %10.0f cycles -> %.1f cycles per step\n", names[kind], avg, avg / per[kind]); fflush(stdout);
    }
    return 0;
}
