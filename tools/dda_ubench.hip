// dda_ubench -- what a carrier index taken from a fixed-point DDA instead of the exact FP64 recurrence would buy
// ("the next lever").  Two sample steps of one channel of a resampled group, four channels interleaved,
// 16 samples per group, no other work:
//   A  today's chan_step_rw: v_bfe_i32 (chip), v_mul_f64 511 p, v_cvt_i32_f64, v_lshl_add_u32, ds_read_b32, v_pk_mad_u16,
//      v_add_f64, v_fract_f64                                                        = 7 VALU + 1 LDS
//   B  DDA: t = 2^20 + 511 p as a double on the 2^-32 grid, t += c (exact: one v_add_f64 IS a 52-bit integer add),
//      address = v_lshl_add_u32(hi(t), 2, base') -- floor(t - 2^20) sits in the low bits of the high word --, ds_read_b32,
//      v_bfe_i32, v_pk_mad_u16, and one v_min3_u32 per TWO samples over the low words (the fraction of 511 p: a sample
//      whose fraction is within the error bound of 0 makes the group ambiguous -> exact replay, rare)  = 4.5 VALU + 1 LDS
//   C  B with conflict-free gathers (every lane of a wave reads its own bank: the table index is replaced by the lane id
//      in the low five bits)   D  B without the LDS read: the VALU floor of the step
// Prints ns per channel-sample and SIMD at 1..3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/dda_ubench tools/dda_ubench.hip && tools/dda_ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void acc_mad(int &acc, int t, int v)
{
    const s2 t2 = __builtin_bit_cast(s2, t);
    const s2 v2 = {(short)v, (short)v};
    acc = __builtin_bit_cast(int, (s2)(t2 * v2 + __builtin_bit_cast(s2, acc)));
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k(int *out, double seed, int iters, uint32_t wseed)
{
    __shared__ int lut[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lut[i] = i * 2654435761u;
    __syncthreads();
    double p[4], t[4];
    uint32_t X[4];
    const double ds = 1.1e-3 + seed * 1e-9, k511 = 511.0;
    const double c = 511.0 * ds;  // (the real thing rounds this to the 2^-32 grid and corrects the residue per group)
    for (int j = 0; j < 4; ++j) {
        p[j] = 0.01 * j + threadIdx.x * 1e-3;
        t[j] = 1048576.0 + 511.0 * p[j];
        X[j] = wseed * (j + 1) + threadIdx.x;
    }
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) int *)lut;
    const uint32_t based = base - (0x41300000u << 2);
    int o[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) o[u] = 0;
    uint32_t amb = ~0u;
    for (int i = 0; i < iters; ++i) {
        uint32_t lo_prev[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int acc = o[u];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = __builtin_amdgcn_sbfe((int)X[j], (uint32_t)(2 * u), 2);
                uint32_t a;
                if (MODE == 0) {
                    const int kk = (int)(k511 * p[j]);
                    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(kk), "s"(base));
                } else {
                    const uint32_t hi = (uint32_t)(__builtin_bit_cast(uint64_t, t[j]) >> 32);
                    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(hi), "s"(based));
                    if (MODE == 2) a = (a & ~0x7cu) | ((threadIdx.x & 31u) << 2);  // (one extra v_bfi per sample)
                }
                const int tt = MODE == 3 ? (int)a : *(const __attribute__((address_space(3))) int *)(uintptr_t)a;
                acc_mad(acc, tt, v);
                if (MODE == 0) {
                    p[j] = __builtin_amdgcn_fract(p[j] + ds);
                } else {
                    const uint32_t lo = (uint32_t)__builtin_bit_cast(uint64_t, t[j]);
                    if (u & 1) {
                        uint32_t m;
                        asm("v_min3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(amb), "v"(lo), "v"(lo_prev[j]));
                        amb = m;
                    } else {
                        lo_prev[j] = lo;
                    }
                    t[j] = t[j] + c;
                }
            }
            if (u & 1) {
                if (MODE == 0) asm volatile("" : "+v"(acc), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));
                else asm volatile("" : "+v"(acc), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
            }
            o[u] = acc;
        }
        // group end: keep the DDA inside the table (what the renormalisation at a group start does)
        if (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = t[j] >= 1048576.0 + 511.0 ? t[j] - 511.0 : t[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) X[j] = X[j] * 1664525u + 1013904223u;
    }
    int s = (int)amb;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += o[u];
    out[blockIdx.x * 256 + threadIdx.x] = s + (int)(p[1] * 1000) + (int)t[2];
}

template <int MODE>
static void run(int *d, int waves)
{
    const int iters = 1024, blocks = 256 * waves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 16, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double chs = (double)iters * 16 * 4 * waves;  // channel-samples per SIMD (wave-level)
    static const char *names[4] = {"A exact", "B dda  ", "C dda, conflict-free", "D dda, no LDS"};
    printf("%s, %d waves/SIMD: %.3f ms, %.2f ns per channel-sample and SIMD\n", names[MODE], waves, ms,
           ms * 1e6 / chs);
}

int main()
{
    int *d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(int));
    for (int w = 1; w <= 3; ++w) run<0>(d, w);
    for (int w = 1; w <= 3; ++w) run<1>(d, w);
    for (int w = 1; w <= 3; ++w) run<2>(d, w);
    for (int w = 1; w <= 3; ++w) run<3>(d, w);
    return 0;
}
