// Probe: do ds_read_b128 / ds_write_b128 / b64 / b32 work at byte-unaligned LDS addresses on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__global__ void k(uint8_t* out, int shift_r, int shift_w)
{
    __shared__ __attribute__((aligned(16))) uint8_t a[4096];
    __shared__ __attribute__((aligned(16))) uint8_t b[4096];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) { a[i] = (uint8_t)(i * 7 + 3); b[i] = 0; }
    __syncthreads();
    const uint32_t ra = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(a + shift_r + lane * 16);
    const uint32_t wa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(b + shift_w + lane * 16);
    v4u v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ra) : "memory");
    asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(wa), "v"(v) : "memory");
    __syncthreads();
    for (uint32_t i = lane; i < 4096; i += 64) out[i] = b[i];
}
int main()
{
    uint8_t* d; hipMalloc(&d, 4096);
    int bad_total = 0;
    for (int sr = 0; sr < 16; sr++) for (int sw = 0; sw < 16; sw += 5) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sr, sw);
        uint8_t h[4096]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; i++) { uint8_t want = (uint8_t)((i + sr) * 7 + 3); if (h[sw + i] != want) bad++; }
        if (bad) { printf("shift_r %d shift_w %d: %d bad bytes\n", sr, sw, bad); bad_total += bad; }
    }
    printf("unaligned ds_read_b128/ds_write_b128: %s\n", bad_total ? "BROKEN" : "OK");
    return 0;
}
