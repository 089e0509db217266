// ubench -- issue cost of the VALU instructions on k_synth's critical path (gfx950), relative to v_add_u32.
// 8 independent chains per lane, 4 waves per SIMD, every SIMD busy; prints ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed, int iters)
{
    double a[CHAINS], b2[CHAINS];
    int ia[CHAINS], ib[CHAINS];
    __shared__ int lds[4096];
    lds[threadIdx.x] = threadIdx.x;
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) {
        a[j] = seed + j * 0.125 + threadIdx.x * 1e-6;
        ia[j] = threadIdx.x + j;
        ib[j] = j;
        b2[j] = 0.0;
    }
    const double c = seed * 1e-3;
    const double cs = __builtin_bit_cast(double, ((uint64_t)__builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(uint64_t, c) >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint64_t, c)));
    const int si = __builtin_amdgcn_readfirstlane(iters | 3);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 3) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[j]) : "v"(a[j]));
            if (OP == 4) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[j]) : "v"(ia[j]));
            if (OP == 5) asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(a[j]), "v"(c) : "vcc");
            if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[j]) : "v"(i) : "vcc");
            if (OP == 7) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(ia[j]) : "v"(i));
            if (OP == 8) asm volatile("v_trunc_f64 %0, %1" : "=v"(a[j]) : "v"(a[j]));
            if (OP == 9) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 10) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 11) asm volatile("v_fract_f64 %0, %1" : "=v"(a[j]) : "v"(a[j]));
            if (OP == 12) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(ia[j]) : "v"(a[j]));
            if (OP == 13) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a[j]) : "v"(i));
            if (OP == 14) asm volatile("v_cmp_le_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[j]) : "v"(c), "v"(ia[j]), "v"(i) : "vcc");
            if (OP == 15) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[j]) : "s"(cs));
            if (OP == 16) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 17) asm volatile("v_bfe_u32 %0, %0, %1, 2" : "+v"(ia[j]) : "v"(i));
            if (OP == 18) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(ia[j]) : "s"(si));
            if (OP == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 20) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 21) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ia[j]) : "v"(i), "s"(si));
            if (OP == 22) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(ia[j]));
            if (OP == 23) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ia[j]) : "s"(si), "v"(i));
            if (OP == 24) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 25) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(ia[j]) : "s"(si), "v"(i));
            if (OP == 26) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(ia[j]) : "s"(si), "v"(i));
            if (OP == 27) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(ia[j]), "+v"(ib[j]) : "v"(i) : "vcc");
            if (OP == 28) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ia[j]) : "s"(si));
            if (OP == 29) asm volatile("v_bfe_u32 %0, %0, 4, 2" : "+v"(ia[j]));
            if (OP == 30) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "s"(cs));
            if (OP == 31) asm volatile("v_add_f64 %0, %1, %2" : "=v"(b2[j]) : "v"(a[j]), "s"(cs));
            if (OP == 32) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 33) asm volatile("v_and_b32 %0, 0x3000, %0" : "+v"(ia[j]));
            if (OP == 34) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 35) asm volatile("v_mov_b32 %0, %1" : "=v"(ia[j]) : "v"(i));
            if (OP == 36) asm volatile("v_add_u32 %0, %1, %0" : "+v"(ia[j]) : "s"(si));
            if (OP == 37) asm volatile("v_cvt_i32_f64 %0, %1\n v_lshlrev_b32 %0, 1, %0" : "=v"(ia[j]) : "v"(a[j]));
            if (OP == 38) asm volatile("v_fract_f64 %0, %1" : "=v"(b2[j]) : "v"(a[j]));
            if (OP == 39) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[j]) : "v"(ia[j]));
            if (OP == 40) asm volatile("ds_read_b32 %0, %1" : "=v"(ib[j]) : "v"(ia[j]));
            if (OP == 41) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 42) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[j]) : "v"(i));
            if (OP == 43) asm volatile("v_add_f64 %0, %0, %1 \n v_fract_f64 %0, %0" : "+v"(a[j]) : "s"(cs));
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) s += a[j] + ia[j] + ib[j] + b2[j];
    if (OP == 40) asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, double *d)
{
    const int blocks = 256 * 4;  // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.5, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)ITER * CHAINS * 4;  // 4 waves per SIMD
    const double ns = ms * 1e6 / wave_instr_per_simd;
    printf("%-16s %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", name, ms, ns);
    return ns;
}

int main()
{
    double *d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(double));
    const double ref = run<0>("v_add_u32", d);
    run<0>("v_add_u32", d);
    const char *names[] = {"v_add_u32", "v_add_f64", "v_mul_f64", "v_cvt_i32_f64", "v_cvt_f64_i32", "v_cmp_le_f64",
                           "v_cndmask_b32", "v_lshrrev_b32", "v_trunc_f64", "v_fma_f64", "v_xad_u32", "v_fract_f64",
                           "v_cvt_u32_f64", "v_lshrrev_b64", "cmp_f64+cndmask"};
    double r[15];
    r[1] = run<1>(names[1], d); r[2] = run<2>(names[2], d); r[3] = run<3>(names[3], d); r[4] = run<4>(names[4], d);
    r[5] = run<5>(names[5], d); r[6] = run<6>(names[6], d); r[7] = run<7>(names[7], d); r[8] = run<8>(names[8], d);
    r[9] = run<9>(names[9], d); r[10] = run<10>(names[10], d); r[11] = run<11>(names[11], d);
    r[12] = run<12>(names[12], d); r[13] = run<13>(names[13], d); r[14] = run<14>(names[14], d);
    const double r15 = run<15>("v_add_f64 sgpr", d), r16 = run<16>("v_lshl_add_u32", d), r17 = run<17>("v_bfe_u32", d);
    const double r18 = run<18>("v_mul_i32_i24 s", d), r19 = run<19>("v_mul_lo_u32", d), r20 = run<20>("v_add3_u32", d);
    const double r21 = run<21>("v_perm_b32", d);
    struct { const char *n; double v; } more[] = {
        {"v_lshlrev_b32 imm", run<22>("v_lshlrev_b32 imm", d)}, {"v_and_or_b32", run<23>("v_and_or_b32", d)},
        {"v_alignbit_b32", run<24>("v_alignbit_b32", d)}, {"v_mad_u32_u24", run<25>("v_mad_u32_u24", d)},
        {"v_bfi_b32", run<26>("v_bfi_b32", d)}, {"add_co+addc (2)", run<27>("add_co+addc", d)},
        {"v_lshl_add_u32 sgpr", run<28>("v_lshl_add_u32 sgpr", d)}, {"v_bfe_u32 imm", run<29>("v_bfe_u32 imm", d)},
        {"v_mul_f64 sgpr", run<30>("v_mul_f64 sgpr", d)}, {"v_add_f64 sgpr, dst!=src", run<31>("v_add_f64 sgpr dst!=src", d)},
        {"v_mul_u32_u24", run<32>("v_mul_u32_u24", d)}, {"v_and_b32 literal", run<33>("v_and_b32 literal", d)},
        {"v_lshl_or_b32", run<34>("v_lshl_or_b32", d)}, {"v_mov_b32", run<35>("v_mov_b32", d)},
        {"v_add_u32 sgpr", run<36>("v_add_u32 sgpr", d)}, {"cvt_i32_f64+lshlrev (2)", run<37>("cvt+lshlrev", d)},
        {"v_fract_f64 dst!=src", run<38>("v_fract_f64 dst!=src", d)}, {"v_ldexp_f64", run<39>("v_ldexp_f64", d)},
        {"ds_read_b32", run<40>("ds_read_b32", d)}, {"v_sub_u32", run<41>("v_sub_u32", d)}, {"v_xor_b32", run<42>("v_xor_b32", d)},
        {"add_f64 sgpr + fract (2)", run<43>("add_f64+fract", d)}};
    printf("\nrelative to v_add_u32 (= 1 issue slot):\n");
    printf("  v_add_f64 sgpr   %.2f\n  v_lshl_add_u32   %.2f\n  v_bfe_u32        %.2f\n  v_mul_i32_i24 s  %.2f\n  v_mul_lo_u32     %.2f\n  v_add3_u32       %.2f\n  v_perm_b32       %.2f\n",
           r15 / ref, r16 / ref, r17 / ref, r18 / ref, r19 / ref, r20 / ref, r21 / ref);
    for (int i = 1; i < 15; ++i) printf("  %-16s %.2f\n", names[i], r[i] / ref);
    for (auto &m : more) printf("  %-26s %.2f\n", m.n, m.v / ref);
    return 0;
}
