// lds_align_probe -- what does ds_read_b32 return for a byte address that is not a multiple of 4 on gfx950 under ROCm?
// (If the low address bits were ignored, k_synth could index its carrier table with (int)(2044 p) and drop the
// shift-add of every sample.)   hipcc --offload-arch=gfx950 -O2 -o tools/lds_align_probe tools/lds_align_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out)
{
    __shared__ uint32_t t[64];
    t[threadIdx.x] = 0x03020100u + 0x04040404u * threadIdx.x;  // byte i of the array holds i
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)t;
    const uint32_t a = base + 16u + threadIdx.x;  // byte address 16 + lane
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    out[threadIdx.x] = v;
}
int main()
{
    uint32_t *d, h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("byte address 16+%d -> %08x\n", i, h[i]);
    return 0;
}
