// Probe: LDS throughput of the access shapes pass E1 uses (gfx950): cycles per wave instruction with 16 waves of one workgroup issuing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[131072];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 131072 / 4; i += 1024) ((uint32_t*)buf)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = tid * 2654435761u + seed, acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < 2000; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t a = (x >> 8) & 0x1FFFF;
        if (KIND == 0) acc += *(const uint32_t*)(buf + (a & ~3u));                       // ds_read_b32 random aligned
        if (KIND == 1) { struct __attribute__((aligned(4))) w2 { uint32_t a, b; }; const w2 w = *(const w2*)(buf + ((a & 0x1FFF8) & ~3u)); acc += w.a ^ w.b; }   // ds_read2_b32 random
        if (KIND == 2) acc += *(const uint16_t*)(buf + (a & ~1u));                       // ds_read_u16 random
        if (KIND == 3) acc += buf[a];                                                    // ds_read_u8 random
        if (KIND == 4) *(uint32_t*)(buf + (a & ~3u)) = x;                                // ds_write_b32 random
        if (KIND == 5) *(uint16_t*)(buf + (a & ~1u)) = (uint16_t)x;                      // ds_write_b16 random
        if (KIND == 6) buf[a] = (uint8_t)x;                                              // ds_write_b8 random
        if (KIND == 7) { typedef uint32_t u32_ua __attribute__((aligned(1))); acc += *(const u32_ua*)(buf + a); }   // unaligned ds_read_b32 random
        if (KIND == 8) { struct __attribute__((aligned(4))) w2 { uint32_t a, b; }; const uint32_t b = (it * 64 + (tid & 63)) & 0x1FFF8; const w2 w = *(const w2*)(buf + (b & ~3u)); acc += w.a ^ w.b; }   // ds_read2_b32 consecutive bytes (stride-1 probe)
    }
    const unsigned long long t1 = clock64();
    if (acc == 0x12345) out[63] = acc;
    if ((tid & 63) == 0) out[tid >> 6] = t1 - t0;
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 1024);
    const char* names[9] = {"ds_read_b32 random", "ds_read2_b32 random", "ds_read_u16 random", "ds_read_u8 random", "ds_write_b32 random", "ds_write_b16 random", "ds_write_b8 random", "ds_read_b32 unaligned random", "ds_read2_b32 stride-1"};
    for (int kind = 0; kind < 9; kind++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (kind) {
#define L(K) case K: hipLaunchKernelGGL(k<K>, dim3(256), dim3(1024), 0, 0, d, 7u); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8)
            }
            hipDeviceSynchronize();
        }
        unsigned long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 16; i++) avg += h[i]; avg /= 16;
        printf("%-30s %8.0f cycles per wave for 2000 instr -> %.1f cycles/instr/wave, %.2f cycles per wave-instr at the CU (16 waves)\n", names[kind], avg, avg / 2000, avg / 2000 / 16);
    }
    return 0;
}
