// mfma_add_probe -- can the FP64 matrix pipe of gfx950 do k_synth's NCO additions, bit for bit, BESIDE the VALU?
//
// k_synth is VALU-issue bound; 3 of its 11 per-channel-sample VALU instructions are FP64 additions of a
// WAVE-UNIFORM step to per-lane state (y += cs2, p + |d|).  v_mfma_f64_16x16x4_f64 computes D = C + A x B on a
// 16x16 tile = 4 doubles per lane; with A[row][k] = (k == row / 4) and B[k][col] = step_k every lane's register r
// receives  C_r + 1.0 * step_r + three exact zeros, i.e. ONE correctly rounded addition per register -- the same
// bits as v_add_f64 -- for four channels at once.  This probe checks (1) that claim on random and adversarial
// operands (ties, binade crossings, tiny phases, -0.0), (2) what an MFMA costs when it is issued between VALU
// instructions that do not depend on it.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_add_probe tools/mfma_add_probe.hip && tools/mfma_add_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double lane_a()
{
    const int l = threadIdx.x & 63;
    return ((l >> 4) == ((l & 15) >> 2)) ? 1.0 : 0.0;  // A[row = l & 15][k = l >> 4] = (k == row / 4)
}

// out[4 * t + r] = x[4 * t + r] (+) step[(wave of t)][r], once through the matrix pipe and once through v_add_f64
__global__ void k_exact(const double *x, const double *steps, double *out_mfma, double *out_valu)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = t >> 6, l = threadIdx.x & 63;
    const double a = lane_a();
    const double b = steps[wave * 4 + (l >> 4)];  // B[k = l >> 4][col] = step_k
    v4f64 c;
    c.x = x[4 * t + 0]; c.y = x[4 * t + 1]; c.z = x[4 * t + 2]; c.w = x[4 * t + 3];
    const v4f64 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    out_mfma[4 * t + 0] = d.x; out_mfma[4 * t + 1] = d.y; out_mfma[4 * t + 2] = d.z; out_mfma[4 * t + 3] = d.w;
    for (int r = 0; r < 4; ++r) {
        double s = steps[wave * 4 + r], v = x[4 * t + r], o;
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(o) : "v"(v), "v"(s));
        out_valu[4 * t + r] = o;
    }
}

// MODE 0: NV VALU instructions per iteration on 4 independent chains; 1: one dependent MFMA per iteration only;
// 2: both (the MFMA result is not read by the VALU instructions of the same iteration)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_mix(double *out, double seed, int iters)
{
    const double a = lane_a();
    const double b = seed * 1e-3 + (threadIdx.x >> 4 & 3) * 1e-4;
    v4f64 acc = {seed, seed + 1, seed + 2, seed + 3};
    double y0 = seed + threadIdx.x * 1e-6, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    const double c = seed * 1e-3;
    for (int i = 0; i < iters; ++i) {
        if (MODE != 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        if (MODE != 1) {
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(y0) : "v"(c));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(y1) : "v"(c));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(y2) : "v"(c));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(y3) : "v"(c));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i0) : "v"(y0));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i1) : "v"(y1));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i2) : "v"(y2));
                asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i3) : "v"(y3));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + y0 + y1 + y2 + y3 + i0 + i1 + i2 + i3;
}

template <int MODE, int NV>
static double run_mix(const char *name, double *d, int waves_per_simd)
{
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd;  // blocks of 4 waves -> waves_per_simd per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<MODE, NV>), dim3(blocks), dim3(256), 0, 0, d, 1.5, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<MODE, NV>), dim3(blocks), dim3(256), 0, 0, d, 1.5, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_iter = ms * 1e6 / ((double)iters * waves_per_simd);  // per wave-iteration per SIMD
    printf("%-44s %d waves/SIMD  %8.3f ms  %7.2f ns per wave-iteration per SIMD\n", name, waves_per_simd, ms, ns_iter);
    return ns_iter;
}

static uint64_t bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
static double from_bits(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

int main()
{
    // ---- 1. exactness
    const int waves = 4096, n = waves * 64 * 4;
    std::vector<double> x(n), st(waves * 4);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int w = 0; w < waves; ++w) {
        const int kind = w % 8;
        for (int r = 0; r < 4; ++r) {
            double s;
            if (kind < 2) s = 2.0 * 1.023e6 / 2.6e6 * (1.0 + 1e-6 * (U(rng) - 0.5));          // code step, half chips
            else if (kind < 4) s = 3500.0 / 2.6e6 * U(rng);                                     // carrier step
            else if (kind == 4) s = ldexp(1.0 + ldexp((double)(rng() >> 12), -52), -(int)(rng() % 40));  // random binade
            else if (kind == 5) s = ldexp((double)(rng() % 4096 + 1), -53);                      // tie-prone: few bits at 2^-53
            else if (kind == 6) s = ldexp((double)(rng() % 1024 + 1), -60 + (int)(rng() % 16));
            else s = U(rng) * 1e-5;
            st[w * 4 + r] = s;
        }
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                double v;
                const uint64_t z = rng();
                if (kind < 2) v = 8184.0 * U(rng);
                else if (kind < 4) v = U(rng);
                else if (kind == 4) v = ldexp(1.0 + ldexp((double)(z >> 12), -52), (int)(z % 14) - 13 + 12 * (int)(z >> 8 & 1));
                else if (kind == 5) v = (z & 1) ? from_bits(bits(1.0) - 1 - (z >> 40)) : ldexp((double)(z >> 12), -52);  // just below 1 / on the 2^-52 grid
                else if (kind == 6) v = from_bits(bits(ldexp(1.0, -(int)(z % 12))) - (z >> 50));  // just below a binade boundary
                else v = (z & 7) == 0 ? -0.0 : ldexp((double)(z >> 20), -52 - (int)(z % 11));      // tiny phases, -0.0
                x[((size_t)w * 64 + l) * 4 + r] = v;
            }
    }
    double *dx, *ds, *dm, *dv;
    hipMalloc(&dx, n * 8); hipMalloc(&ds, waves * 4 * 8); hipMalloc(&dm, n * 8); hipMalloc(&dv, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(ds, st.data(), waves * 4 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_exact, dim3(waves / 4), dim3(256), 0, 0, dx, ds, dm, dv);
    std::vector<double> om(n), ov(n);
    hipMemcpy(om.data(), dm, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ov.data(), dv, n * 8, hipMemcpyDeviceToHost);
    long bad_valu = 0, bad_host = 0, bad_zero = 0;
    for (int i = 0; i < n; ++i) {
        const int w = i / 256, r = i % 4;
        const volatile double h = x[i] + st[w * 4 + r];
        if (bits(om[i]) != bits(ov[i])) {
            if (om[i] == ov[i]) ++bad_zero;  // signed-zero difference only
            else if (++bad_valu <= 5) printf("  MISMATCH x=%a step=%a mfma=%a valu=%a\n", x[i], st[w * 4 + r], om[i], ov[i]);
        }
        if (bits(ov[i]) != bits((double)h)) ++bad_host;
    }
    printf("exactness: %d additions, mfma != v_add_f64: %ld (signed-zero-only differences: %ld), v_add_f64 != host: %ld\n",
           n, bad_valu, bad_zero, bad_host);

    // ---- 2. cost beside VALU work
    double *d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(double));
    for (int wps = 1; wps <= 3; wps += 2) {
        run_mix<0, 32>("32 VALU (16 add_f64 + 16 cvt_i32_f64)", d, wps);
        run_mix<1, 32>("1 dependent MFMA f64 16x16x4", d, wps);
        run_mix<2, 32>("32 VALU + 1 MFMA", d, wps);
        run_mix<0, 64>("64 VALU", d, wps);
        run_mix<2, 64>("64 VALU + 1 MFMA", d, wps);
    }
    return bad_valu != 0;
}
