// synth_common.h -- small device helpers and table geometry shared by the synthesis kernels
// (synth_kernels.hip: k_synth, one chunk per lane; synth_group.hip: k_synth_g, one 16-sample group per lane).
#ifndef GAL_SYNTH_COMMON_H_
#define GAL_SYNTH_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nco_walk.h"

#define SYN_BLOCK 256
#define SYN_GROUP 16
#define GAL_ACT_ROW 32  // bytes per epoch in the active-position lists (<= 24 entries used, zero-padded; k_synth reads the first 16)
#define STR_WORDS 512
#define STR_PITCH 513  // LDS words per channel: one pad word (= word 0) so that "the next word" never wraps
#define RW_BINS 128        // bins of the group-start fraction (k_synth<.., RW = 1>)
#define RW_BIN_PITCH 130   // 129 entries used: a fraction that rounds to 1.0f lands in the (undecidable) entry 128
#define CB_BINS 64         // the same for the CBOC mode, which keeps TWO bin tables per channel (chip holds, BOC(6,1) parity)
#define CB_BIN_PITCH 66
#define RW_EDGE 9.5367431640625e-07f    // 2^-20: a threshold this close outside a bin is registered in the bin as well
#define RW_DELTA 2.384185791015625e-07f  // 2^-22: a fraction this close to its threshold is not decided by the table

namespace galdev {
using namespace galnco;

__device__ __forceinline__ double uniform_f64(double v)
{
    const uint64_t u = d2u(v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return u2d(((uint64_t)hi << 32) | lo);
}

// sg * 0x55555555: 0, 0x5555.., 0xAAAA.., 0xFFFF.. = the XOR mask of the sign pair on all 16 half chips
// (a 24-bit multiply and a shift-or: v_mul_lo_u32 is a quarter-rate instruction)
__device__ __forceinline__ uint32_t gal_sign_mask(const uint32_t sg)
{
    const uint32_t m = __umul24(sg, 0x555555u);
    uint32_t d;  // (asm: the combiner otherwise folds the shift into a second, 32-bit multiply)
    asm("v_lshl_or_b32 %0, %1, 16, %1" : "=v"(d) : "v"(m));
    return d;
}
#define GAL_SIGN_MASK(sg) gal_sign_mask(sg)
// the same mask for the CURRENT symbol of a channel, out of byte 2 of its packed state (sym_state)
#define GAL_SIGN_MASK_ST(st) __builtin_amdgcn_perm((st), (st), 0x02020202u)

// (m & a) | (~m & b) as the one instruction it is (the compiler expands the expression to not / and / and / or)
__device__ __forceinline__ uint32_t gal_bfi(const uint32_t m, const uint32_t a, const uint32_t b)
{
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
    return d;
}

// (non-zero, negative-if-non-zero) bit pairs -> two's-complement 2-bit fields 00 / 01 / 11 = 0 / +1 / -1: the upper
// bit survives only where the lower one is set
__device__ __forceinline__ uint32_t window_signed(uint32_t w)
{
    return w & (((w & 0x55555555u) << 1) | 0x55555555u);
}

// The spread of a window with up to four holds (masks M_d = ~0 << 2 u_d, nested: u_1 < u_2 < u_3 < u_4, unused ones 0): the
// fields from u_d on read the window shifted by d fields.  Sequentially that is x <- bfi(M_d, x << 2, x) four times, eight
// DEPENDENT instructions; the same result as a tree -- four independent shifts of the original window, then
// bfi(M_2, bfi(M_4, x << 8, bfi(M_3, x << 6, x << 4)), bfi(M_1, x << 2, x)) -- has the same eight instructions at half the depth
// (1.216 -> 1.211 ms per pipelined step in four same-box alternations).
__device__ __forceinline__ uint32_t rw_spread(const uint32_t x, const uint4 M)
{
    const uint32_t lo = gal_bfi(M.x, x << 2, x);         // fields below u_2
    const uint32_t mid = gal_bfi(M.z, x << 6, x << 4);   // fields u_2 .. u_4
    const uint32_t hi = gal_bfi(M.w, x << 8, mid);       // fields from u_2 on
    return gal_bfi(M.y, hi, lo);
}

typedef short gal_s2 __attribute__((ext_vector_type(2)));

// acc.(I,Q) += entry.(I,Q) * v  as ONE v_pk_mad_u16 ... op_sel_hi:[1,0,1] (the compiler folds the splat of v into the
// operand select).  Written with vector types, not asm: the scheduler must see the LDS latency of `entry`; the
// empty asm keeps the four channels of a part one accumulate chain (re-associated into a tree it costs 5
// instructions instead of 4).
__device__ __forceinline__ void gal_acc(int &acc, const int entry, const int v)
{
    const gal_s2 t2 = __builtin_bit_cast(gal_s2, entry);
    const gal_s2 v2 = {(short)v, (short)v};
    const gal_s2 a2 = t2 * v2 + __builtin_bit_cast(gal_s2, acc);
    acc = __builtin_bit_cast(int, a2);
#ifdef GAL_ACC_CHAIN
    asm("" : "+v"(acc));
#endif
}

}  // namespace galdev
#endif  // GAL_SYNTH_COMMON_H_
