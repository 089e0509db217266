// step_ubench -- issue ceiling of k_synth's fast step on gfx950: the 10 VALU instructions of one channel-sample
// (cvt, shift-add, bfe_i32, FP64 mul, cvt, shift-add, pk_mad, 2 FP64 add, fract), four independent channels
// interleaved, in a loop without any of the kernel's other work.  Variants: with / without the LDS read of the carrier
// table (random addresses like the real one), 1 / 2 / 3 waves per SIMD.  Prints cycles per VALU instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/step_ubench tools/step_ubench.hip && tools/step_ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef short s2 __attribute__((ext_vector_type(2)));

template <int LDS>
__global__ __launch_bounds__(256) void k(int *out, double seed, int iters, uint32_t wseed)
{
    __shared__ int lut[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lut[i] = i * 2654435761u;
    __syncthreads();
    double y[4], p[4];
    uint32_t W[4];
    int m[4];
    const double cs = 0.7869 + seed * 1e-9, ds = 1.1e-3 + seed * 1e-9, k511 = 511.0;
    for (int j = 0; j < 4; ++j) {
        y[j] = 100.0 + j * 7.3 + threadIdx.x * 0.37;
        p[j] = 0.01 * j + threadIdx.x * 1e-3;
        W[j] = wseed * (j + 1) + threadIdx.x;
        m[j] = -2 * (int)y[j];
    }
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) int *)lut + 4096;
    int acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ic = (int)y[j];
                int off;
                asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(off) : "v"(ic), "v"(m[j]));
                const int v = __builtin_amdgcn_sbfe((int)W[j], (uint32_t)off, 2);
                const int kk = (int)(k511 * p[j]);
                uint32_t a;
                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(kk), "s"(base));
                int t;
                if (LDS) t = *(const __attribute__((address_space(3))) int *)(uintptr_t)a;
                else t = (int)a;
                const s2 t2 = __builtin_bit_cast(s2, t);
                const s2 v2 = {(short)v, (short)v};
                acc = __builtin_bit_cast(int, (s2)(t2 * v2 + __builtin_bit_cast(s2, acc)));
                y[j] = y[j] + cs;
                p[j] = __builtin_amdgcn_fract(p[j] + ds);
            }
            if (u & 1) asm volatile("" : "+v"(acc), "+v"(y[0]), "+v"(p[0]), "+v"(y[1]), "+v"(p[1]), "+v"(y[2]), "+v"(p[2]), "+v"(y[3]), "+v"(p[3]));
        }
        // keep y in the table's range
#pragma unroll
        for (int j = 0; j < 4; ++j) { y[j] = y[j] - 6.0; m[j] = -2 * (int)y[j]; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + (int)y[0] + (int)(p[1] * 1000);
}

template <int LDS>
static void run(int *d, int waves)
{
    const int iters = 2048, blocks = 256 * waves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<LDS>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 16, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<LDS>, dim3(blocks), dim3(256), 0, 0, d, 1.0, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double valu = (double)iters * (8 * 4 * 10 + 4 * 4) * waves;  // wave-instructions per SIMD
    printf("LDS %d, %d waves/SIMD: %.3f ms, %.2f ns per VALU instruction and SIMD (= %.2f cycles at 2.3 GHz)\n", LDS, waves, ms,
           ms * 1e6 / valu, ms * 1e6 / valu * 2.3);
}

int main()
{
    int *d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(int));
    for (int w = 1; w <= 4; ++w) run<0>(d, w);
    for (int w = 1; w <= 4; ++w) run<1>(d, w);
    return 0;
}
