// wrcal -- calibration of rocprofv3's WRITE_SIZE on gfx950 for the store patterns of k_synth
// (MI355X_MICROARCH.md: "WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own access
// pattern").  Each kernel writes exactly 1 GiB:
//   wr_coalesced : lane i writes 16 B at base + i*16 (1 KiB per wave instruction)
//   wr_lane64    : every lane owns a 4 KiB region and writes it 64 B at a time (4 x 16 B back to back)
//   wr_lane16    : same, 16 B at a time with ~R steps of ALU work in between (the round-1 k_synth pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void wr_coalesced(uint4 *out, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

template <int PIECES>
__global__ void wr_lane(uint4 *out, int spin)
{
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 *p = out + lane * 256;  // 4 KiB per lane
    uint32_t x = (uint32_t)lane;
    for (int k = 0; k < 256; k += PIECES) {
        for (int s = 0; s < spin; ++s) x = x * 1664525u + 1013904223u;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) p[k + j] = make_uint4(x, (uint32_t)k, (uint32_t)j, 3);
    }
}

int main()
{
    const size_t bytes = 1ull << 30;
    uint4 *d;
    hipMalloc(&d, bytes);
    const int lanes = (int)(bytes / 4096);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(wr_coalesced, dim3(2048), dim3(256), 0, 0, d, bytes / 16);
        hipLaunchKernelGGL(wr_lane<4>, dim3(lanes / 256), dim3(256), 0, 0, d, 4000);
        hipLaunchKernelGGL(wr_lane<1>, dim3(lanes / 256), dim3(256), 0, 0, d, 1000);
        hipLaunchKernelGGL(wr_lane<8>, dim3(lanes / 256), dim3(256), 0, 0, d, 8000);
    }
    hipDeviceSynchronize();
    printf("wrote 1 GiB per kernel\n");
    return 0;
}
