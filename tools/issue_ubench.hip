// issue_ubench -- ABSOLUTE issue cost (shader cycles per wave-instruction per SIMD) of the VALU / LDS instructions that are on
// k_synth_g's sample step or are candidates to replace them (MI355X_MICROARCH.md lists v_fma_f32 at 2 cycles
// per wave64; tools/ubench.hip priced everything RELATIVE to v_add_u32 and called that one slot of 4 cycles).
// The timed loop is ONE asm statement (the compiler puts an s_nop between separate asm statements that write VGPRs: the first
// version of this probe measured those): 8 independent chains x 8 repetitions per trip, W waves per SIMD (1, 2, 4, 8), every
// SIMD of the chip busy.  Clock: s_memtime around the loop (averaged over the blocks) and HIP events around the launch.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define TRIPS 512
#define REPS 8

// \r: register index of a 32-bit chain (v10..v17) or the first of a 64-bit chain (v20, v22, .. v34); constants: v40 (int),
// v41 (float), v[42:43] (f64 / packed f32), v44 an LDS address
#define OPS(X)                                                                            \
    X(0, "v_add_u32", 1, 0, "v_add_u32 v\\r, v\\r, v40")                                   \
    X(1, "v_fma_f32", 1, 0, "v_fma_f32 v\\r, v\\r, v41, v41")                              \
    X(2, "v_add_f32", 1, 0, "v_add_f32 v\\r, v\\r, v41")                                   \
    X(3, "v_mul_f32", 1, 0, "v_mul_f32 v\\r, v\\r, v41")                                   \
    X(4, "v_pk_fma_f32", 1, 1, "v_pk_fma_f32 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43], v[42:43]") \
    X(5, "v_pk_add_f32", 1, 1, "v_pk_add_f32 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]")         \
    X(6, "v_add_f64", 1, 1, "v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]")               \
    X(7, "v_fma_f64", 1, 1, "v_fma_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43], v[42:43]")     \
    X(8, "v_pk_mad_u16", 1, 0, "v_pk_mad_u16 v\\r, v40, v\\r, v\\r op_sel_hi:[1,0,1]")      \
    X(9, "v_pk_add_u16", 1, 0, "v_pk_add_u16 v\\r, v\\r, v40")                             \
    X(10, "v_bfe_i32", 1, 0, "v_bfe_i32 v\\r, v\\r, 4, 2")                                 \
    X(11, "v_lshl_add_u32", 1, 0, "v_lshl_add_u32 v\\r, v\\r, 2, v40")                     \
    X(12, "v_min3_u32", 1, 0, "v_min3_u32 v\\r, v\\r, v40, v41")                           \
    X(13, "v_min_u32", 1, 0, "v_min_u32 v\\r, v\\r, v40")                                  \
    X(14, "v_and_b32", 1, 0, "v_and_b32 v\\r, v\\r, v40")                                  \
    X(15, "v_lshlrev_b32", 1, 0, "v_lshlrev_b32 v\\r, 1, v\\r")                            \
    X(16, "v_lshrrev_b32", 1, 0, "v_lshrrev_b32 v\\r, 1, v\\r")                            \
    X(17, "v_mad_i32_i24", 1, 0, "v_mad_i32_i24 v\\r, v\\r, v40, v40")                     \
    X(18, "v_mul_i32_i24", 1, 0, "v_mul_i32_i24 v\\r, v\\r, v40")                          \
    X(19, "v_dot2_i32_i16", 1, 0, "v_dot2_i32_i16 v\\r, v\\r, v40, v\\r")                  \
    X(20, "v_dot4_i32_i8", 1, 0, "v_dot4_i32_i8 v\\r, v\\r, v40, v\\r")                    \
    X(21, "v_perm_b32", 1, 0, "v_perm_b32 v\\r, v\\r, v40, v41")                           \
    X(22, "v_bfi_b32", 1, 0, "v_bfi_b32 v\\r, v40, v\\r, v41")                             \
    X(23, "v_alignbit_b32", 1, 0, "v_alignbit_b32 v\\r, v\\r, v40, v41")                   \
    X(24, "v_mov_b32", 1, 0, "v_mov_b32 v\\r, v40")                                        \
    X(25, "v_cndmask_b32 (vcc)", 1, 0, "v_cndmask_b32 v\\r, v\\r, v40, vcc")               \
    X(26, "v_cvt_i32_f64", 1, 1, "v_cvt_i32_f64 v\\r, v[42:43]")                           \
    X(27, "v_fract_f64", 1, 1, "v_fract_f64 v[\\r:\\r+1], v[\\r:\\r+1]")                    \
    X(28, "v_xor_b32", 1, 0, "v_xor_b32 v\\r, v\\r, v40")                                  \
    X(29, "v_sub_u32", 1, 0, "v_sub_u32 v\\r, v\\r, v40")                                  \
    X(30, "v_max_f32", 1, 0, "v_max_f32 v\\r, v\\r, v41")                                  \
    X(31, "v_add_u32 sdwa byte1", 1, 0, "v_add_u32_sdwa v\\r, v\\r, v40 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") \
    X(32, "v_add_f32 dpp row_shr:1", 1, 0, "v_add_f32_dpp v\\r, v\\r, v41 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(33, "v_pk_fma_f16", 1, 0, "v_pk_fma_f16 v\\r, v\\r, v40, v\\r")                      \
    X(34, "v_mad_u32_u24", 1, 0, "v_mad_u32_u24 v\\r, v\\r, v40, v40")                     \
    X(35, "v_add3_u32", 1, 0, "v_add3_u32 v\\r, v\\r, v40, v40")                           \
    X(36, "v_or_b32", 1, 0, "v_or_b32 v\\r, v\\r, v40")                                    \
    X(37, "v_bfe_u32", 1, 0, "v_bfe_u32 v\\r, v\\r, 4, 9")                                 \
    X(38, "v_and_or_b32", 1, 0, "v_and_or_b32 v\\r, v\\r, v40, v41")                       \
    X(39, "v_lshl_or_b32", 1, 0, "v_lshl_or_b32 v\\r, v\\r, 2, v40")                       \
    X(40, "v_fma_f32 + v_add_u32 (2)", 2, 1, "v_fma_f32 v\\r, v\\r, v41, v41\n v_add_u32 v[\\r+1], v[\\r+1], v40") \
    X(41, "v_add_f64 + v_and_b32 (2)", 2, 1, "v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]\n v_and_b32 v[\\r-10], v[\\r-10], v40") \
    X(42, "v_pk_mad_u16 + v_add_u32 (2)", 2, 1, "v_pk_mad_u16 v\\r, v40, v\\r, v\\r op_sel_hi:[1,0,1]\n v_add_u32 v[\\r+1], v[\\r+1], v40") \
    X(43, "sample step: bfe, lshl_add, pk_mad, add_f64 (4)", 4, 1, "v_bfe_i32 v[\\r-10], v40, 4, 2\n v_lshl_add_u32 v[\\r-9], v[\\r+1], 2, v40\n v_pk_mad_u16 v[\\r-10], v40, v[\\r-10], v[\\r-10] op_sel_hi:[1,0,1]\n v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]") \
    X(44, "ds_read_b32 (same address)", 1, 0, "ds_read_b32 v\\r, v44")                     \
    X(45, "ds_read_b32 + 4 x v_add_f64 (5)", 5, 1, "ds_read_b32 v[\\r-10], v44\n v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]\n v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]\n v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]\n v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]") \
    X(46, "ds_read_b32 + 4 x v_add_u32 (5)", 5, 1, "ds_read_b32 v[\\r-10], v44\n v_add_u32 v\\r, v\\r, v40\n v_add_u32 v[\\r+1], v[\\r+1], v40\n v_add_u32 v\\r, v\\r, v40\n v_add_u32 v[\\r+1], v[\\r+1], v40") \
    X(47, "s_nop 0", 1, 0, "s_nop 0")                                                     \
    X(48, "v_add_u32 + s_nop 0 (2)", 2, 0, "v_add_u32 v\\r, v\\r, v40\n s_nop 0")   \
    X(49, "v_fma_f32 distinct regs", 1, 1, "v_fma_f32 v\\r, v\\r, v[\\r+1], v41")             \
    X(50, "N,P,P: pk_mad, add_u32, and_b32 (3)", 3, 1, "v_pk_mad_u16 v\\r, v40, v\\r, v\\r op_sel_hi:[1,0,1]\n v_add_u32 v[\\r+1], v[\\r+1], v40\n v_and_b32 v[\\r-10], v[\\r-10], v40") \
    X(51, "N,N,P,P: pk_mad, bfe, add_u32, and_b32 (4)", 4, 1, "v_pk_mad_u16 v\\r, v40, v\\r, v\\r op_sel_hi:[1,0,1]\n v_bfe_i32 v[\\r-9], v[\\r-9], 4, 2\n v_add_u32 v[\\r+1], v[\\r+1], v40\n v_and_b32 v[\\r-10], v[\\r-10], v40") \
    X(52, "N,P,P,P,P: add_f64 + 4 simple (5)", 5, 1, "v_add_f64 v[\\r:\\r+1], v[\\r:\\r+1], v[42:43]\n v_add_u32 v[\\r-10], v[\\r-10], v40\n v_and_b32 v[\\r-9], v[\\r-9], v40\n v_lshrrev_b32 v[\\r-10], 1, v[\\r-10]\n v_xor_b32 v[\\r-9], v[\\r-9], v40") \
    X(53, "v_add_co_u32 (vcc)", 1, 0, "v_add_co_u32 v\\r, vcc, v\\r, v40")                   \
    X(54, "v_addc_co_u32 (vcc)", 1, 0, "v_addc_co_u32 v\\r, vcc, v\\r, v40, vcc")            \
    X(55, "add_co + addc pair (2)", 2, 1, "v_add_co_u32 v\\r, vcc, v\\r, v40\n v_addc_co_u32 v[\\r+1], vcc, v[\\r+1], v40, vcc") \
    X(56, "v_lshlrev_b32 by vgpr", 1, 0, "v_lshlrev_b32 v\\r, v40, v\\r")                    \
    X(57, "v_lshrrev_b32 by vgpr", 1, 0, "v_lshrrev_b32 v\\r, v40, v\\r")                    \
    X(58, "v_ashrrev_i32", 1, 0, "v_ashrrev_i32 v\\r, 1, v\\r")                              \
    X(59, "v_cvt_f32_i32", 1, 0, "v_cvt_f32_i32 v\\r, v\\r")                                 \
    X(60, "v_cvt_f32_ubyte0", 1, 0, "v_cvt_f32_ubyte0 v\\r, v\\r")                           \
    X(61, "v_mac_f32 / v_fmac_f32", 1, 0, "v_fmac_f32 v\\r, v40, v41")                       \
    X(62, "v_fmac_f32 + v_add_u32 (2)", 2, 1, "v_fmac_f32 v\\r, v40, v41\n v_add_u32 v[\\r+1], v[\\r+1], v40") \
    X(63, "v_fmac_f32 x2 + add_u32 + lshrrev (4)", 4, 1, "v_fmac_f32 v\\r, v40, v41\n v_add_u32 v[\\r+1], v[\\r+1], v40\n v_fmac_f32 v[\\r-10], v40, v41\n v_lshrrev_b32 v[\\r-9], 1, v[\\r-9]") \
    X(64, "v_mul_u32_u24", 1, 0, "v_mul_u32_u24 v\\r, v\\r, v40")                            \
    X(65, "v_subrev_u32", 1, 0, "v_subrev_u32 v\\r, v40, v\\r")                              \
    X(66, "v_sub_f32", 1, 0, "v_sub_f32 v\\r, v\\r, v41")                                    \
    X(67, "v_mul_f32 x v_add_f32 alternating (2)", 2, 1, "v_mul_f32 v\\r, v\\r, v41\n v_add_f32 v[\\r+1], v[\\r+1], v41") \
    X(68, "v_pk_mad_u16 x2 + 4 simple (6)", 6, 1, "v_pk_mad_u16 v\\r, v40, v\\r, v\\r op_sel_hi:[1,0,1]\n v_pk_mad_u16 v[\\r+1], v40, v[\\r+1], v[\\r+1] op_sel_hi:[1,0,1]\n v_add_u32 v[\\r-10], v[\\r-10], v40\n v_and_b32 v[\\r-9], v[\\r-9], v40\n v_lshrrev_b32 v[\\r-10], 1, v[\\r-10]\n v_xor_b32 v[\\r-9], v[\\r-9], v40") \
    X(69, "ds_read_b32 + 8 x v_add_u32 (9)", 9, 1, "ds_read_b32 v[\\r-10], v44\n v_add_u32 v\\r, v\\r, v40\n v_add_u32 v[\\r+1], v[\\r+1], v40\n v_add_u32 v\\r, v\\r, v40\n v_add_u32 v[\\r+1], v[\\r+1], v40\n v_add_u32 v\\r, v\\r, v40\n v_add_u32 v[\\r+1], v[\\r+1], v40\n v_add_u32 v\\r, v\\r, v40\n v_add_u32 v[\\r+1], v[\\r+1], v40")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cyc, float seed, int trips)
{
    __shared__ int lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    int iv = (int)(seed * 3.0f) | 1;
    float fv = seed * 1e-3f;
    double dv = (double)fv;
    uint32_t res = 0;
#define X(id, name, ninstr, wide, txt)                                                                                   \
    if constexpr (OP == id)                                                                                              \
        asm volatile(                                                                                                    \
            "v_mov_b32 v40, %3\n v_mov_b32 v41, %4\n v_mov_b32 v42, %5\n v_mov_b32 v43, %6\n v_mov_b32 v44, %7\n"        \
            ".irp q,10,11,12,13,14,15,16,17,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35\n v_mov_b32 v\\q, %3\n .endr\n" \
            "s_mov_b32 s20, %8\n s_waitcnt lgkmcnt(0)\n s_memtime %1\n s_waitcnt lgkmcnt(0)\n"                           \
            "1:\n .rept " "8" "\n"                                                                                       \
            ".if " #wide "\n .irp r,20,22,24,26,28,30,32,34\n " txt "\n .endr\n"                                         \
            ".else\n .irp r,10,11,12,13,14,15,16,17\n " txt "\n .endr\n .endif\n"                                        \
            ".endr\n s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                  \
            "s_waitcnt lgkmcnt(0)\n s_memtime %2\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, v10, v20\n"                      \
            : "=v"(res), "=s"(t0), "=s"(t1)                                                                              \
            : "v"(iv), "v"(fv), "v"((uint32_t)__builtin_bit_cast(uint64_t, dv)), "v"((uint32_t)(__builtin_bit_cast(uint64_t, dv) >> 32)), \
              "v"((uint32_t)(threadIdx.x & 63) * 4u), "s"(trips)                                                         \
            : "memory", "vcc", "scc", "s20", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", \
              "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v40", "v41", "v42", "v43", "v44");
    OPS(X)
#undef X
    out[blockIdx.x * blockDim.x + threadIdx.x] = res + lds[(threadIdx.x + 1) & 1023];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double g_mhz = 2400.0;

template <int OP>
void run(const char *name, int ninstr, uint32_t *d, unsigned long long *dc)
{
    double res_evt[4], res_mt[4];
    const int Ws[4] = {1, 2, 4, 8};
    for (int wi = 0; wi < 4; ++wi) {
        const int W = Ws[wi];
        const int blocks = 256 * W;  // W blocks of 4 waves per CU = W waves per SIMD
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.5f, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.5f, TRIPS);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long hc[4096];
        (void)hipMemcpy(hc, dc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mt = 0;
        for (int b = 0; b < blocks; ++b) mt += (double)hc[b];
        mt /= blocks;
        const double wi_per_simd = (double)TRIPS * REPS * 8 * ninstr * W;
        res_evt[wi] = ms * 1e-3 * g_mhz * 1e6 / wi_per_simd;
        res_mt[wi] = mt / wi_per_simd;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    printf("%-50s memtime %6.2f %6.2f %6.2f %6.2f   evt %6.2f %6.2f %6.2f %6.2f\n", name, res_mt[0], res_mt[1], res_mt[2], res_mt[3],
           res_evt[0], res_evt[1], res_evt[2], res_evt[3]);
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    g_mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, clockRate %.0f MHz; cycles per wave-instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD\n"
           "(memtime: s_memtime ticks around the loop / instructions issued on the SIMD; evt: wall time x clockRate)\n",
           prop.gcnArchName, prop.multiProcessorCount, g_mhz);
    uint32_t *d;
    unsigned long long *dc;
    (void)hipMalloc(&d, 4096 * 256 * sizeof(uint32_t));
    (void)hipMalloc(&dc, 4096 * sizeof(unsigned long long));
#define X(id, name, ninstr, wide, txt) run<id>(name, ninstr, d, dc);
    OPS(X)
#undef X
    return 0;
}
