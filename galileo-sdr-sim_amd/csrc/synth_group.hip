// synth_group.hip -- k_synth_g: the hot kernel of the reference geometry (BOC(1,1), 0.74 <= 2 f_code / fs < 1, i.e. 2.6 MS/s),
// ONE 16-SAMPLE GROUP PER LANE, and k_repair_g, which replays the few groups it could not decide with the reference's exact
// operation sequence.  Replaces the per-sample loop of the reference, src/galileo-sdr.cpp:481-539.
//
// k_synth (synth_kernels.hip) lets a lane replay 1040 consecutive samples because the two NCO recurrences (:528-532) are
// chains of ROUNDED additions: the exact phase of sample n is only available by stepping.  But the loop does not use the
// phases, it uses two INTEGERS derived from them -- the half-chip index (int)(2 x) (:512) and the table index
// (int)(511 p) (:509-510) -- and those depend on the rounding history only where the real-valued phase lies within the
// accumulated rounding drift of an integer boundary.  Everywhere else any approximation with a known error bound has the
// same floor.  The walkers leave an EXACT checkpoint of both chains every R = 1024 samples (k_walk_code / k_walk_carr); from
// it every 16-sample group of the chunk gets its start state in closed form,
//     y_g = fma(16 g, s, y_c)                      |y_g - sequential y| <= 1041 ulp(8192)/2 = 2^-31
//     p_g = frac(fma(16 g, |d|, p_c))              |p_g - sequential p| <= 1040 x 2^-53
// and the group is synthesised from those: the chips through the resampled-window pattern look-up of k_synth<.., RW = 1>
// (rw_phase_a/b/c: a group whose fraction lies within 2^-22 of a pattern threshold is UNDECIDED), the carrier index through
// the fixed-point DDA t = 2^20 + 512 + 2^-26 + 511 p advanced by t += 511 |d| (a double in [2^20, 2^21) is a fixed-point
// number with 32 fraction bits: the high word is the table address, the low word the fraction of 511 p; a sample whose
// fraction word is below 2^7, i.e. 511 p within 2^-26 of an integer, is UNDECIDED; the error of t against the sequential
// phase is below 18 x 2^-33 + 2^-34 < 2^-28).  A lane that met an undecided sample appends its group to a list; k_repair_g
// walks each listed group's exact start state out of the same checkpoint (nco_walk.h: code_walk / carr_walk_track) and
// replays its 16 samples x all channels with the reference's statements.  About 2000 of the 19.5 million groups of a 120 s
// batch are listed (1.7e-4 of them for a chip threshold, 6e-6 for a carrier index).
//
// What the layout buys over one chunk per lane: no per-lane state survives a group, so nothing is loop-carried but the
// accumulators (k_synth: 60 VGPRs of channel state; this kernel runs four waves per SIMD instead of three); a wave's 64
// lanes write 4 KB of CONTIGUOUS output per iteration (k_synth: 64 bursts of 64 B in 64 different places); the code wrap is
// a matter of the window's contents, not of the sample loop (k_synth: slow groups); the chunk's checkpoints are wave-uniform
// scalar loads (k_synth: 36 gathers per lane); neighbouring lanes read neighbouring carrier-table entries (4.7 instead of
// 6.7 LDS cycles per gather at +-3.5 kHz).
// The carrier checkpoints, which k_synth's exact replay verifies on its way, are verified by k_verify_carr (synth_kernels.hip)
// on the walker stream, beside this kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "nco_walk.h"
#include "synth_dev.h"
#include "synth_common.h"

using namespace galnco;
using namespace galdev;

// Compiled as three translation units, side by side under `make -j` (Makefile: -DGAL_GTU=0..2), because its instantiations take five
// minutes in one go: 0 = the instances for up to 12 channels (BOC(1,1) and CBOC), k_repair_g and the launchers; 1 = the wide
// instances (13-24 channels); 2 = the bisection instances (crowded pattern thresholds).  Without GAL_GTU everything is one unit.
#if !defined(GAL_GTU)
#define GAL_GTU_MAIN 1
#define GAL_GTU_WIDE 1
#define GAL_GTU_SEARCH 1
#else
#define GAL_GTU_MAIN (GAL_GTU == 0)
#define GAL_GTU_WIDE (GAL_GTU == 1)
#define GAL_GTU_SEARCH (GAL_GTU == 2)
#endif

#ifndef SG_THREADS
#define SG_THREADS 1024  // largest block the kernel is compiled for (its waves' records are sized for 16); the launch takes 512
                         // threads: 8 waves share one epoch's tables (61 KB of LDS), two blocks per CU, four waves per SIMD
#endif
#ifndef SG_WAVES_PER_EU
#define SG_WAVES_PER_EU 4
#endif
#define SG_CHUNK 1024   // samples per wave iteration: 64 lanes x 16
#define SG_SYMS 64      // symbol sign pairs per channel and epoch (host gate: an epoch spans fewer symbols)
#define SG_STR_PITCH 580  // LDS words per stream row: the 8184 half chips of the period, then its first 1088 once more -- a chunk
                          // starts below the code wrap and advances by < 1008 half chips, so no window index ever wraps
#define SG_MPOS 17      // sign-mask table: positions of the code wrap relative to a window (0: behind it .. 16: in front of it)
#define SG_LUT_N 1152   // entries per carrier table: 512 (mirrored phase still negative) + 511 + 129 (behind a wrap inside a group)
#define SG_AMB 128u
#define SG_BIAS (1049088.0 + 1.4901161193847656250e-08)  // 2^20 + 512 + 2^-26
#define SG_MAXCH 12     // channel positions of the instances for up to 12 channels (two blocks of 512 threads per CU) ...
#define SG_MAXCH_WIDE 24  // ... and of the wide instances, 13 .. 24 positions in ONE launch (round 6: 24 stream rows = 56 KB of 133 KB
                          // of LDS, one block of 1024 threads per CU -- the same four waves per SIMD; rounds 3-5 ran such a batch as two
                          // launches of <= 12, the second one adding onto the first's samples: a read-modify-write of the whole output)

// What a group start needs of its channel and chunk, wave-uniform: written once per wave iteration by a LOADER lane (lane j < NCH
// fetches channel j's checkpoint of the NEXT chunk while the wave works on the current one) into the wave's own LDS slots, read
// back by every lane as one-address reads.  No scalar registers, nothing loop-invariant for the compiler to hoist into them.
struct SgRec {
    double yc;   // code phase of the chunk's first sample (pre-check, :491), in half chips: 2 x
    double pm;   // carrier phase there, MIRRORED: times the sign of the epoch's step (k_synth: ChanState::p)
    double s;    // code step in half chips per sample, 2 f_code delt
    double dabs; // |f_carr delt|
};
typedef uint32_t sg_u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) sg_u4 *sg_lds_u4;
typedef const __attribute__((address_space(3))) int *sg_lds_int;
typedef const __attribute__((address_space(3))) uint32_t *sg_lds_u32;

// all 16 fields = field d of w (a two's-complement 2-bit value 00 / 01 / 11)
__device__ __forceinline__ uint32_t sg_rep(const uint32_t w, const int d)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)w, 2 * d, 1);      // 0 / ~0: the field's low bit
    const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)w, 2 * d + 1, 1);  // ... its high bit
    return gal_bfi(0x55555555u, lo, hi);
}

__device__ __forceinline__ double sg_f64(const uint32_t lo, const uint32_t hi) { return u2d(((uint64_t)hi << 32) | lo); }

// One part: CNT channel positions J0 .. J0+CNT-1 of one 16-sample group, in phases, each run for all positions of the part
// before the next one starts, so that the part waits ONCE for each round of LDS reads.  o[]: the group's int16 pairs (I, Q) as
// they go to memory; amb: the smallest fraction word the lane has seen; undec: lanes with an undecided chip pattern.
// MODE: the form of the resampled window (k_synth's RW): 1 = the window advances every sample except at <= 4 HOLDS (code step 0.74 ..
// 1 half chips per sample: the reference's 2.6 MS/s), 2 = it advances at <= 2 samples of the group (code step <= 0.133: 15.4 MS/s and
// above), 3 = at <= 4 samples (<= 0.266: 7.7 .. 15.4 MS/s), 4 = ANY pattern of holds (round 5: what lies between, 2.77 .. 7.7 MS/s) --
// sample u reads field u - h(u) of the window, h(u) = the holds up to u, through a four-stage shift network (shifts of 8, 4, 2, 1
// fields where the bits of h say so); the masks of the stages are made with the patterns (build below)
// SIG: 0 = BOC(1,1) as the reference generates it; 1 = the opt-in CBOC(6,1,1/11) mode (GAL_CFG_CBOC; hold form only): a second
// pattern look-up per group -- the PARITY of the BOC(6,1) half period of every sample, k_synth's rw_phase_a6 / b6 -- two chip words
// (the (B - C) and the (B + C) factor of every sample), 8-byte table entries (TA[k], TB[k]) and two multiply-adds per sample
// SEARCH (round 6): the pattern of a group is found by a four-step bisection over the channel's 15 sorted thresholds instead of through
// the bin table, which holds ONE threshold per bin -- sample rates at which 2 f_code / fs is close to a fraction with a small
// denominator (4.092, 8.184, 16.368 MS/s: two, four, eight samples per half chip; 2.5, 2.728 ...) have thresholds that coincide or lie
// 1e-6 apart (the Doppler's share of the step) and used to send the whole batch to the exact-replay kernel.  The fraction is known to
// 2^-31, the undecided band stays 2^-22 around every threshold: a few groups in 10^5 more are listed.
template <int J0, int CNT, int MODE, int SIG, int BINS, int BPITCH, bool SEARCH>
__device__ __forceinline__ void sg_part(int (&o)[16], uint32_t &amb, uint64_t &undec, const double g16, const SgRec *rec, const uint32_t *sya,
                                        const double *s_c511, const uint32_t *s_lutd,
                                        const uint32_t *s_str, const uint2 *s_bin, const uint4 *s_pat, const uint2 *s_bin6,
                                        const uint32_t *s_pat6, const float *s_thr, const int nact)
{
    uint32_t X[CNT];
    [[maybe_unused]] uint32_t XB[CNT];
    double t[CNT];
    // the DDA's step 511 |d| and the table base of the part's positions: wave-uniform and constant over the epoch, but read
    // from LDS for every group -- kept in registers for all twelve positions they would cost 36 of them
    double c511[CNT];
    uint32_t lutd[CNT];
#pragma unroll
    for (int q = 0; q < CNT; ++q) {
        c511[q] = ((const volatile __attribute__((address_space(3))) double *)(uintptr_t)s_c511)[J0 + q];
        lutd[q] = ((const volatile __attribute__((address_space(3))) uint32_t *)(uintptr_t)s_lutd)[J0 + q];
    }
    {
        // ---- phase A: the wave's records; approximate pre-check code phase at the group start; bin entry, stream words and
        // sign masks in flight
        int ic0[CNT];
        float f[CNT];
        uint2 be[CNT];
        uint32_t lo[CNT], hi[CNT], mask[CNT];
        double praw[CNT];
        [[maybe_unused]] float f6[CNT];
        [[maybe_unused]] uint2 be6[CNT];
        [[maybe_unused]] int i12[CNT];
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            const int j = J0 + q;
            const sg_u4 r0 = ((sg_lds_u4)(uintptr_t)(rec + j))[0], r1 = ((sg_lds_u4)(uintptr_t)(rec + j))[1];
            const uint32_t sa0 = *(sg_lds_u32)(uintptr_t)(sya + j);
            const double A = __builtin_fma(g16, sg_f64(r1.x, r1.y), sg_f64(r0.x, r0.y));
            ic0[q] = (int)A;  // < 8184 + 1008: the row continues behind the period's end (:491-507 is a matter of the signs)
            f[q] = (float)__builtin_amdgcn_fract(A);
            if constexpr (!SEARCH) {
                const int bi = (int)(f[q] * (float)BINS);
                be[q] = s_bin[j * BPITCH + bi];
            }
            if constexpr (SIG == 1) {  // the BOC(6,1) half period of sample u is (int)(6 y_u) = i12 + floor(f6 + u 6 s)
                const double y6 = 6.0 * A;
                i12[q] = (int)y6;
                f6[q] = (float)__builtin_amdgcn_fract(y6);
                be6[q] = s_bin6[j * CB_BIN_PITCH + (int)(f6[q] * (float)CB_BINS)];
            }
            const uint32_t *wp = s_str + j * SG_STR_PITCH + (ic0[q] >> 4);
            lo[q] = wp[0];
            hi[q] = wp[1];
            // XOR mask of the data / secondary-code signs (:517-518) on the window's 16 half chips: the chunk's first symbol
            // up to the code wrap, its successor behind it -- one table row per pair of sign pairs, one entry per position of
            // the wrap relative to the window
            int pos;
            asm("v_med3_i32 %0, %1, 0, 16" : "=v"(pos) : "v"(ic0[q] - (8184 - 16)));
            mask[q] = *(sg_lds_u32)(uintptr_t)(sa0 + ((uint32_t)pos << 2));
            praw[q] = __builtin_fma(g16, sg_f64(r1.z, r1.w), sg_f64(r0.z, r0.w));
        }
        // ---- phase B: threshold compare, pattern masks in flight; the carrier's DDA word meanwhile
        uint4 M[CNT];
        [[maybe_unused]] uint32_t p6[CNT];
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            const int j = J0 + q;
            if constexpr (SIG == 1) {
                const float thr6 = __uint_as_float(be6[q].x);
                const uint32_t po6 = be6[q].y + (f6[q] >= thr6 ? 4u : 0u);  // be6.y = 4 x (thresholds below the bin)
                // (pattern: parity relative to sample 0's half period, whose own parity is added here)
                p6[q] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_pat6 + j * 16) + po6) ^
                        (((uint32_t)i12[q] & 1u) ? 0xAAAAAAAAu : 0u);
                undec |= __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(f6[q] - thr6) >= RW_DELTA));
            }
            if constexpr (SEARCH) {
                // id = the number of thresholds <= f (entry 15 of the sorted row is the sentinel 2.0); undecided within 2^-22 of the
                // neighbours on either side, and of 0 and 1 (sample 0's own half chip hangs on the rounding history there)
                const float *th = s_thr + j * 16;
                int id = th[7] <= f[q] ? 8 : 0;
                id += th[id + 3] <= f[q] ? 4 : 0;
                id += th[id + 1] <= f[q] ? 2 : 0;
                id += th[id] <= f[q] ? 1 : 0;
                const float up = th[id], lo = th[id > 0 ? id - 1 : 0];
                M[q] = s_pat[j * 16 + id];
                const bool near = !(up - f[q] >= RW_DELTA) | ((id > 0) & !(f[q] - lo >= RW_DELTA)) | !(f[q] >= RW_DELTA) | !(f[q] <= 1.0f - RW_DELTA);
                undec |= __builtin_amdgcn_ballot_w64(near);
            } else {
            const float thr = __uint_as_float(be[q].x);
            const uint32_t po = be[q].y + (f[q] >= thr ? 16u : 0u);
            M[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(s_pat + j * 16) + po);
            // undecided: the fraction too close to its bin's threshold (a NaN threshold = a bin near two of them); 0 and 1 are
            // thresholds of the first and the last bin, because sample 0's own half chip hangs on the rounding history there
            undec |= __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(f[q] - thr) >= RW_DELTA));
            }
            const double pg = praw[q] - __builtin_trunc(praw[q]);  // (a mirrored phase that is still negative stays negative)
            t[q] = __builtin_fma(511.0, pg, SG_BIAS);
        }
        // ---- phase C: window (signs applied), spread
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            const uint32_t W = __builtin_amdgcn_alignbit(hi[q], lo[q], (uint32_t)ic0[q] << 1) ^ mask[q];
            if constexpr (SIG == 1) {
                // fields (B != C, sign of C x secondary) per SAMPLE; the half chip of sample u is ic0 + u - (holds before u): its
                // parity comes out of the hold masks.  X: the (B - C) factor of every sample as a signed 2-bit field; XB: the (B + C)
                // factor, whose sign is bit 1 ^ 1 ^ parity(half chip) ^ parity(half period) (k_synth: rw_phase_c1_cboc / _d_cboc)
                // (the spread of the RAW window, by the batch's form; the parity of sample u's half chip relative to ic0's: forms 1
                // and 4 hold -- half chip ic0 + u - h(u), parity(h) out of the masks (form 1: nested, their XOR; form 4: stage 0 of
                // the network sits at the output position) and 1 ^ parity(u) = 0x2222.. --, forms 2 and 3 advance -- half chip
                // ic0 + a(u), parity(a) = the XOR of the nested masks, and the 1 is 0xAAAA..)
                uint32_t x, hw;
                if constexpr (MODE == 1) {
                    x = rw_spread(W, M[q]);
                    hw = ((M[q].x ^ M[q].y ^ M[q].z ^ M[q].w) & 0xAAAAAAAAu) ^ 0x22222222u;
                } else if constexpr (MODE == 4) {
                    x = gal_bfi(M[q].w, W << 16, W);
                    x = gal_bfi(M[q].z, x << 8, x);
                    x = gal_bfi(M[q].y, x << 4, x);
                    x = gal_bfi(M[q].x, x << 2, x);
                    hw = (M[q].x & 0xAAAAAAAAu) ^ 0x22222222u;
                } else {
                    x = gal_bfi(M[q].x, sg_rep(W, 1), sg_rep(W, 0));
                    x = gal_bfi(M[q].y, sg_rep(W, 2), x);
                    if constexpr (MODE == 3) {
                        x = gal_bfi(M[q].z, sg_rep(W, 3), x);
                        x = gal_bfi(M[q].w, sg_rep(W, 4), x);
                    }
                    hw = ((M[q].x ^ M[q].y ^ M[q].z ^ M[q].w) & 0xAAAAAAAAu) ^ 0xAAAAAAAAu;
                }
                hw ^= ((uint32_t)ic0[q] & 1u) ? 0xAAAAAAAAu : 0u;
                X[q] = window_signed(x);
                const uint32_t sb = x ^ hw ^ p6[q];
                const uint32_t lo1 = ~x & 0x55555555u;  // B == C: this term is the one that is non-zero
                XB[q] = (J0 + q) < nact ? (lo1 | (sb & (lo1 << 1))) : 0u;  // (an idle position's all-zero row has B == C everywhere)
            } else if constexpr (MODE == 1) {
                X[q] = rw_spread(window_signed(W), M[q]);
            } else if constexpr (MODE == 4) {
                uint32_t x = window_signed(W);
                x = gal_bfi(M[q].w, x << 16, x);
                x = gal_bfi(M[q].z, x << 8, x);
                x = gal_bfi(M[q].y, x << 4, x);
                X[q] = gal_bfi(M[q].x, x << 2, x);
            } else {  // field 0 everywhere, field 1 from the first advance on, field 2 from the second, ...
                const uint32_t w = window_signed(W);
                uint32_t x = gal_bfi(M[q].x, sg_rep(w, 1), sg_rep(w, 0));
                x = gal_bfi(M[q].y, sg_rep(w, 2), x);
                if constexpr (MODE == 3) {
                    x = gal_bfi(M[q].z, sg_rep(w, 3), x);
                    x = gal_bfi(M[q].w, sg_rep(w, 4), x);
                }
                X[q] = x;
            }
        }
    }
    // ---- the 16 samples: chip value from field u of X, table address from the high word of t.  The table reads are issued
    // one sample ahead of the multiply-adds that consume them and waited for with explicit counts: written as volatile asm,
    // because the compiler's own schedule issues two reads, waits, uses them, issues two more -- two exposed LDS latencies per
    // sample.  (Its own wait counts stay valid: more reads in flight than it knows of only make them conservative.  No scalar-
    // memory or FLAT load is in flight here, so the counter counts LDS reads, which return in order.  Two samples ahead: 0.6 %,
    // inside the noise.)
    __builtin_amdgcn_sched_barrier(0);
    typedef int sg_i2 __attribute__((ext_vector_type(2)));
    typename std::conditional<SIG == 1, sg_i2, int>::type e[16][CNT];
    uint32_t lw[16][CNT];
#define SG_ISSUE(u)                                                                                                       \
    _Pragma("unroll") for (int q = 0; q < CNT; ++q) {                                                                     \
        if constexpr (SIG == 1) {                                                                                         \
            uint32_t a_;                                                                                                  \
            asm volatile("v_lshl_add_u32 %1, %2, 3, %3\n\tds_read_b64 %0, %1"                                             \
                         : "=&v"(e[u][q]), "=&v"(a_) : "v"((uint32_t)(d2u(t[q]) >> 32)), "v"(lutd[q]) : "memory");        \
        } else {                                                                                                          \
            asm volatile("v_lshl_add_u32 %0, %1, 2, %2\n\tds_read_b32 %0, %0"                                             \
                         : "=&v"(e[u][q]) : "v"((uint32_t)(d2u(t[q]) >> 32)), "v"(lutd[q]) : "memory");                   \
        }                                                                                                                 \
        lw[u][q] = (uint32_t)d2u(t[q]);                                                                                   \
        t[q] = t[q] + c511[q];                                                                                            \
    }
// the reads of sample u have landed once no more than the CNT issued behind them are outstanding (none behind the last sample's);
// the operands tie the multiply-adds of this sample (e), the running minimum of the fraction words (amb) and the multiply-adds of
// the PREVIOUS sample (its finished accumulator) to the wait: without the last the compiler may sink a sample's multiply-adds
// below later reads and park the loaded entries in scratch meanwhile (the CBOC form did: 290 spills)
#define SG_WAIT(TXT, u)                                                                                                               \
    if constexpr (CNT == 1) asm volatile(TXT : "+v"(e[u][0]), "+v"(amb), "+v"(o[(u) ? (u) - 1 : 0]) :: "memory");                     \
    if constexpr (CNT == 2) asm volatile(TXT : "+v"(e[u][0]), "+v"(e[u][1]), "+v"(amb), "+v"(o[(u) ? (u) - 1 : 0]) :: "memory");      \
    if constexpr (CNT == 3) asm volatile(TXT : "+v"(e[u][0]), "+v"(e[u][1]), "+v"(e[u][2]), "+v"(amb), "+v"(o[(u) ? (u) - 1 : 0]) :: "memory"); \
    if constexpr (CNT == 4) asm volatile(TXT : "+v"(e[u][0]), "+v"(e[u][1]), "+v"(e[u][2]), "+v"(e[u][3]), "+v"(amb), "+v"(o[(u) ? (u) - 1 : 0]) :: "memory");
    SG_ISSUE(0)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u < 15) {
            SG_ISSUE(u + 1)
            if constexpr (CNT == 1) { SG_WAIT("s_waitcnt lgkmcnt(1)", u) }
            if constexpr (CNT == 2) { SG_WAIT("s_waitcnt lgkmcnt(2)", u) }
            if constexpr (CNT == 3) { SG_WAIT("s_waitcnt lgkmcnt(3)", u) }
            if constexpr (CNT == 4) { SG_WAIT("s_waitcnt lgkmcnt(4)", u) }
        } else {
            SG_WAIT("s_waitcnt lgkmcnt(0)", u)
        }
        int acc = o[u];
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            if constexpr (SIG == 1) {  // exactly one of the two factors is non-zero
                gal_acc(acc, e[u][q].x, __builtin_amdgcn_sbfe((int)X[q], (uint32_t)(2 * u), 2));
                gal_acc(acc, e[u][q].y, __builtin_amdgcn_sbfe((int)XB[q], (uint32_t)(2 * u), 2));
            } else {
                gal_acc(acc, e[u][q], __builtin_amdgcn_sbfe((int)X[q], (uint32_t)(2 * u), 2));
            }
        }
#pragma unroll
        for (int q = 0; q < CNT; ++q) amb = amb < lw[u][q] ? amb : lw[u][q];
        o[u] = acc;
    }
#undef SG_WAIT
#undef SG_ISSUE
    __builtin_amdgcn_sched_barrier(0);
}

// ACC: add onto samples already in `iq` (second and later channel groups when more than 12 channels are active)
template <int NCH, bool ACC, int MODE, int SIG = 0, bool SEARCH = false>
__global__ __launch_bounds__(SG_THREADS) __attribute__((amdgpu_waves_per_eu(SG_WAVES_PER_EU)))
void k_synth_g(const DevPlan *__restrict__ Pd, SynGeom G, const uint8_t *__restrict__ act_all, const int *__restrict__ nact_all,
               uint32_t *__restrict__ iq, uint32_t *__restrict__ flist, const int flist_cap)
{
    static_assert(NCH >= 1 && NCH <= SG_MAXCH_WIDE, "1..24 channel positions per launch");
    static_assert(SIG == 0 || NCH <= SG_MAXCH, "the CBOC mode is built for up to 12 positions per launch");
    static_assert(!SEARCH || (SIG == 0 && NCH <= SG_MAXCH), "the threshold search is built for BOC(1,1), up to 12 positions per launch");
    constexpr int MC = NCH <= SG_MAXCH ? SG_MAXCH : SG_MAXCH_WIDE;  // positions the per-wave records are laid out for
    // CBOC keeps two bin tables per channel (chip holds, half-period parity) of 64 bins each, as in k_synth
    constexpr int BINS = SIG ? CB_BINS : RW_BINS, BPITCH = SIG ? CB_BIN_PITCH : RW_BIN_PITCH;
    __shared__ uint32_t s_str[NCH * SG_STR_PITCH];
    __shared__ uint32_t s_mtab[16 * SG_MPOS];  // [sign pair of the first symbol * 4 + of its successor][position of the wrap]
    __shared__ __attribute__((aligned(8))) int s_lut[2 * SG_LUT_N * (SIG ? 2 : 1)];  // CBOC: 8-byte entries (TA[k], TB[k])
    __shared__ uint2 s_bin[NCH * BPITCH];
    __shared__ uint4 s_pat[NCH * 16];
    __shared__ float s_thr[NCH * 16];
    __shared__ uint2 s_bin6[SIG ? NCH * CB_BIN_PITCH : 1];
    __shared__ uint32_t s_pat6[SIG ? NCH * 16 : 1];
    __shared__ float s_thr6[SIG ? NCH * 16 : 1];
    __shared__ uint32_t s_sym[NCH * SG_SYMS];
    __shared__ __attribute__((aligned(16))) SgRec s_rec[(SG_THREADS / 64) * 2 * MC];  // [wave][buffer][position]
    __shared__ double s_c511[MC];
    __shared__ uint32_t s_lutd[MC];
    __shared__ uint32_t s_sya[(SG_THREADS / 64) * 2 * MC];  // ... LDS byte address of the sign mask of the chunk's first symbol
    __shared__ int s_rwbad;

    // (s_setprio 1 / 2 here, to keep the verification kernel that runs beside this one out of its issue slots: the kernel ALONE gets
    // slower, 0.895 -> 0.960 ms, profiles/r05d_prio_ab.log)
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    // ---- the block's share of the launch: a CONTIGUOUS run of its chunks in (epoch, chunk) order.  ne x bpe blocks, block boundaries
    // on epoch boundaries: a block's run lies inside ONE epoch -- one set of tables (sg_grid).  The balanced layouts that ignore the
    // epoch boundaries (a range that crosses one is worked off segment by segment, the block rebuilds its per-epoch tables in
    // between; consecutive ranges on the same XCD), measured slower, are in variant builds only: -DSG_BALANCED_RANGES,
    // build_variant_g.sh (a tool of rounds 3-5: git history)
#ifdef SG_BALANCED_RANGES
    const int nb = (int)gridDim.x;
    const int xb = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const long long U = (long long)G.ne * G.nchunks;
    long long uq = U * xb / nb;
    const long long u_end = U * (xb + 1) / nb;
#else
    const int bpe = G.blocks_per_epoch;
    const int er = (int)blockIdx.x / bpe, tg = (int)blockIdx.x - er * bpe;  // epoch relative to the executed range, part of it
    const int c_begin = G.nchunks * tg / bpe, c_end = G.nchunks * (tg + 1) / bpe;
#endif
    const int *const p_lut = Pd->lut;
    const int *const p_prn = Pd->prn;
    const int *const p_ib0 = Pd->ib0;
    const uint32_t *const p_str = Pd->str;
    const double *const p_cpx = Pd->cp_x, *const p_cpp = Pd->cp_p;
    const uint32_t *const p_cpi = Pd->cp_ib;
    const double *const p_cstep = Pd->cstep, *const p_dstep = Pd->dstep;
    const uint32_t *const p_pcur = Pd->page_cur, *const p_pnext = Pd->page_next;
    const uint32_t cs25 = Pd->cs25;
    int *const p_ctr = Pd->ctr;
    // ---- tables that do not depend on the epoch
    for (int i = tid; i < 16 * SG_MPOS; i += nthr) {
        const int pair = i / SG_MPOS, pos = i - pair * SG_MPOS;
        const uint32_t mc = (uint32_t)(pair >> 2) * 0x55555555u, mn = (uint32_t)(pair & 3) * 0x55555555u;
        // pos = clamp(first half chip of the window - (8184 - 16), 0, 16): the fields from 16 - pos on lie behind the wrap
        s_mtab[i] = pos == 0 ? mc : pos == 16 ? mn : (mc ^ ((mc ^ mn) & (~0u << (2 * (16 - pos)))));
    }
    for (int i = tid; i < 2 * SG_LUT_N * (SIG ? 2 : 1); i += nthr) {
        if constexpr (SIG == 1) {  // int i = (table (plain / conjugate) x SG_LUT_N + entry) x 2 + (0: TA, 1: TB); Pd->lut = [TA 512][TB 512]
            const int ie = i >> 1, tab = ie >= SG_LUT_N, ii = ie - tab * SG_LUT_N;
            int k = ii < 512 ? ii - 511 : ii - 512;
            k = k >= 511 ? k - 511 : k;
            k = k >= 511 ? k - 511 : k;
            s_lut[i] = p_lut[((i & 1) << 9) + ((tab ? -k : k) & 511)];
            continue;
        }
        // table (plain / conjugate) x SG_LUT_N + entry i = floor(511 p + 512): i >= 512: LUT[(i - 512) mod 511] (the phase wraps
        // at 1, so 511 p wraps at 511); i < 512 (mirrored phase still negative after a Doppler sign change): (int) truncates
        // towards zero (:509), so entry i holds k = i - 511
        const int tab = i >= SG_LUT_N, ii = i - tab * SG_LUT_N;
        int k = ii < 512 ? ii - 511 : ii - 512;
        k = k >= 511 ? k - 511 : k;
        k = k >= 511 ? k - 511 : k;
        s_lut[i] = p_lut[(tab ? -k : k) & 511];
    }
#ifdef SG_BALANCED_RANGES
    int row_prn[NCH];  // PRN whose stream row position j holds (0: none yet, -1: the all-zero row of an idle position)
#pragma unroll
    for (int j = 0; j < NCH; ++j) row_prn[j] = 0;
#endif
#ifdef SG_BALANCED_RANGES
    bool first_seg = true;
    while (uq < u_end) {
    const int er = (int)(uq / G.nchunks);  // epoch relative to the executed range
    const int c_begin = (int)(uq - (long long)er * G.nchunks);
    const int c_end = (long long)(G.nchunks - c_begin) < u_end - uq ? G.nchunks : c_begin + (int)(u_end - uq);
    uq += c_end - c_begin;
    if (!first_seg) __syncthreads();  // every wave is through with the tables of the segment before
    first_seg = false;
#else
    {
#endif
    const int e = G.e0 + er;

    // ---- phase 0 (scalar): the epoch's active list, slot indices
    const int nact = __builtin_amdgcn_readfirstlane(nact_all[e]);
    const uint4 aw = *reinterpret_cast<const uint4 *>(act_all + (size_t)e * GAL_ACT_ROW);
    const uint4 aw2 = MC > 16 ? *reinterpret_cast<const uint4 *>(act_all + (size_t)e * GAL_ACT_ROW + 16) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t awv[8] = {aw.x, aw.y, aw.z, aw.w, aw2.x, aw2.y, aw2.z, aw2.w};
    int ixs[MC];
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        // slot index e * S + act[j] of position j (idle positions alias slot act[0]: loads stay in bounds, results unused)
        ixs[j] = __builtin_amdgcn_readfirstlane(e * G.S + (int)((awv[j >> 2] >> (8 * (j & 3))) & 0xffu));
    }

    // ---- phase 1: the epoch's tables in LDS
    if (tid == 0) s_rwbad = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        int prn = __builtin_amdgcn_readfirstlane(p_prn[ixs[j]]);
        prn = prn < 1 ? 1 : prn;  // idle position: any valid row, zeroed below
        const bool on = j < nact;
#ifdef SG_BALANCED_RANGES
        const int want = on ? prn : -1;
        if (want == row_prn[j]) continue;  // the row of the segment before (same satellite in this position) stays
        row_prn[j] = want;
#endif
        const uint32_t *src = p_str + (size_t)(prn - 1) * STR_WORDS;
        // the row continues behind half chip 8183 (the middle of word 511) with the start of the period, so that a window at or
        // across the code wrap is one contiguous read: a plain copy of words 0 .. 510, then the 69 spliced ones
        for (int t = tid; t < STR_WORDS - 1; t += nthr) s_str[j * SG_STR_PITCH + t] = on ? src[t] : 0u;
        if (tid < SG_STR_PITCH - (STR_WORDS - 1)) {
            const int t = tid;  // row word 511 + t
            const uint32_t w = t == 0 ? ((src[STR_WORDS - 1] & 0xffffu) | (src[0] << 16)) : ((src[t - 1] >> 16) | (src[t] << 16));
            s_str[j * SG_STR_PITCH + STR_WORDS - 1 + t] = on ? w : 0u;
        }
    }
    // symbol sign pairs: entry k = sg = (data ^ secondary) | secondary << 1 of the k-th symbol after the one in force at the
    // epoch start (the XOR mask of a symbol on its half chips is sg x 0x55555555); data symbol = page bit, secondary =
    // CS25[ibit % 25] (:517-518); the page changes when the symbol counter wraps inside the epoch (:497-506)
    for (int i = tid; i < NCH * SG_SYMS; i += nthr) {
        const int j = i / SG_SYMS, k = i - j * SG_SYMS;
        int ix = 0;
#pragma unroll
        for (int q = 0; q < NCH; ++q) ix = j == q ? ixs[q] : ix;
        int ibit = p_ib0[ix] + k;
        const bool nx = ibit >= GAL_N_SYM_PAGE;
        ibit = nx ? ibit - GAL_N_SYM_PAGE : ibit;
        const uint32_t *pg = (nx ? p_pnext : p_pcur) + (size_t)ix * GAL_PAGE_WORDS;
        const uint32_t dbit = (pg[ibit >> 5] >> (ibit & 31)) & 1u;
        const uint32_t sbit = (cs25 >> (ibit % 25)) & 1u;
        const uint32_t sg = (dbit ^ sbit) | (sbit << 1);
        s_sym[i] = j < nact ? sg : 0u;
    }
    double rw_s[NCH];  // the channels' code steps in half chips (wave-uniform)
#pragma unroll
    for (int j = 0; j < NCH; ++j) rw_s[j] = j < nact ? uniform_f64(2.0 * p_cstep[ixs[j]]) : 0.0;
    auto rw_step_of = [&](const int jj) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < NCH; ++j) v = jj == j ? rw_s[j] : v;
        return v;
    };
    // the DDA's step 511 |d| and the table base (plain / conjugate table minus the exponent field of [2^20, 2^21) << 2, so that
    // the address is ONE shift-add of t's high word) of every position
    if (tid < MC) {
        int ix = 0;
#pragma unroll
        for (int q = 0; q < NCH; ++q) ix = tid == q ? ixs[q] : ix;
        const double d = (tid < NCH && tid < nact) ? p_dstep[ix] : 0.0;
        s_c511[tid] = 511.0 * __builtin_fabs(d);
        s_lutd[tid] = SIG ? (uint32_t)(uintptr_t)(sg_lds_int)s_lut + (((uint32_t)(d2u(d) >> 32) >> 31) ? SG_LUT_N * 8u : 0u) - (0x41300000u << 3)
                          : (uint32_t)(uintptr_t)(sg_lds_int)s_lut + (((uint32_t)(d2u(d) >> 32) >> 31) ? SG_LUT_N * 4u : 0u) - (0x41300000u << 2);
    }
    // ---- hold patterns (k_synth<.., RW = 1>, rw_phase_a): step A, the 15 thresholds T_u = 1 - frac(u s) of each channel, sorted
    for (int t = tid; t < NCH * 16; t += nthr) {
        const int j = t >> 4, u = (t & 15) + 1;
        const double s = rw_step_of(j);
        if (u <= 15) {
            const double us = (double)u * s;
            const double T = 1.0 - (us - __builtin_floor(us));
            int rank = 0;
            for (int v = 1; v <= 15; ++v) {
                const double vs = (double)v * s;
                const double Tv = 1.0 - (vs - __builtin_floor(vs));
                rank += (Tv < T) || (Tv == T && v < u);
            }
            s_thr[j * 16 + rank] = (float)T;
        } else {
            s_thr[j * 16 + 15] = 2.0f;
        }
        if constexpr (SIG == 1) {  // the same ranking for the BOC(6,1) half periods: step 6 s per sample
            const double s6 = 6.0 * s;
            if (u <= 15) {
                const double us = (double)u * s6;
                const double T = 1.0 - (us - __builtin_floor(us));
                int rank = 0;
                for (int v = 1; v <= 15; ++v) {
                    const double vs = (double)v * s6;
                    const double Tv = 1.0 - (vs - __builtin_floor(vs));
                    rank += (Tv < T) || (Tv == T && v < u);
                }
                s_thr6[j * 16 + rank] = (float)T;
            } else {
                s_thr6[j * 16 + 15] = 2.0f;
            }
        }
    }
    __syncthreads();
    // ---- step B: the bin tables (NCH x (BINS + 1) entries) and the 16 patterns of each channel
    auto build_bins = [&](uint2 *bins, const float *thrs, const int nbins, const int pitch, const uint32_t unit) {
        for (int t = tid; t < NCH * (nbins + 1); t += nthr) {
            const int j = t / (nbins + 1), b = t - j * (nbins + 1);
            const float lo = (float)b * (1.0f / (float)nbins) - RW_EDGE, hi = (float)(b + 1) * (1.0f / (float)nbins) + RW_EDGE;
            // the thresholds are sorted (entry 15: the sentinel 2.0): the number below the bin by bisection, then the (at most two
            // that matter) inside it
            const float *th = thrs + j * 16;
            int idb = th[7] < lo ? 8 : 0;
            idb += th[idb + 3] < lo ? 4 : 0;
            idb += th[idb + 1] < lo ? 2 : 0;
            idb += th[idb] < lo ? 1 : 0;
            const float t0 = th[idb], t1 = idb < 15 ? th[idb + 1] : 2.0f;
            int cnt = (t0 < hi) + (t1 < hi);
            float thr = t0 < hi ? t0 : 4.0f;  // 4: "no threshold near this bin" -- never reached, never close
            // 0 and 1 are thresholds too -- of sample 0's own half chip (half period), which the approximate phase decides only away
            // from them (k_synth knows its group-start phase exactly): bin 0 compares with 0 (never below it: the pattern offset
            // makes up for the unconditional "f >= threshold"), the last bin with 1; a pattern threshold beside them leaves the bin
            // undecidable
            uint32_t off = (uint32_t)idb * unit;
            if (b == 0) { thr = cnt ? __builtin_nanf("") : 0.0f; off -= cnt ? 0u : unit; cnt = 0; }
            if (b == nbins - 1) { thr = cnt ? __builtin_nanf("") : 1.0f; cnt = 0; }
            if (cnt >= 2 || b == nbins) thr = __builtin_nanf("");  // undecidable here
            if (j >= nact) { thr = 4.0f; off = 0u; }
            bins[j * pitch + b] = make_uint2(__float_as_uint(thr), off);
        }
    };
    if constexpr (!SEARCH) build_bins(s_bin, s_thr, BINS, BPITCH, 16u);
    if constexpr (SIG == 1) {
        build_bins(s_bin6, s_thr6, CB_BINS, CB_BIN_PITCH, 4u);
        for (int t = tid; t < NCH * 16; t += nthr) {
            // pattern `id` (id thresholds <= f6): bit 2u+1 = parity of floor(f6 + u 6s), the number of half periods sample u lies
            // beyond sample 0's
            const int j = t >> 4, id = t & 15;
            const double s6 = 6.0 * rw_step_of(j);
            const double Tlo = id ? (double)s_thr6[j * 16 + id - 1] : 0.0;
            double Thi = (double)s_thr6[j * 16 + id];
            Thi = Thi > 1.0 ? 1.0 : Thi;
            const double f = 0.5 * (Tlo + Thi);
            uint32_t w = 0u;
            for (int u = 1; u <= 15; ++u)
                if ((long long)__builtin_floor(f + (double)u * s6) & 1LL) w |= 2u << (2 * u);
            s_pat6[t] = j < nact ? w : 0u;
        }
    }
    for (int t = tid; t < NCH * 16; t += nthr) {
        const int j = t >> 4, id = t & 15;  // id = number of thresholds <= f
        const double s = rw_step_of(j);
        const double Tlo = id ? (double)s_thr[j * 16 + id - 1] : 0.0;
        double Thi = (double)s_thr[j * 16 + id];
        Thi = Thi > 1.0 ? 1.0 : Thi;
        const double f = 0.5 * (Tlo + Thi);
        uint32_t m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
        if constexpr (MODE == 4) {
            // The shift network of a pattern: sample u reads field u - h(u) of the window, h(u) = u - floor(f + u s) (f < 1) the holds
            // up to u.  Read from the output back to the window, sample u's value comes from position p0 = u, p1 = p0 - (h & 1),
            // p2 = p1 - (h & 2), p3 = p2 - (h & 4), p3 - (h & 8) = u - h(u): stage k (applied to the window in the order 3, 2, 1, 0)
            // shifts the field AT p_k by 2^k fields iff bit k of h is set.  Two samples whose paths meet read the same field of
            // the window (if p_k(u) = p_k(u'), u < u', but the fields differ by d > 0, then floor(h(u') / 2^k) - floor(h(u) / 2^k)
            // = -d although h(u') >= h(u)), so the masks never ask two things of one position.
            for (int u = 0; u <= 15; ++u) {
                const int hld = u - (int)__builtin_floor(f + (double)u * s);
                int pp = u;
                m0 |= (hld & 1) ? 3u << (2 * pp) : 0u;
                pp -= hld & 1;
                m1 |= (hld & 2) ? 3u << (2 * pp) : 0u;
                pp -= hld & 2;
                m2 |= (hld & 4) ? 3u << (2 * pp) : 0u;
                pp -= hld & 4;
                m3 |= (hld & 8) ? 3u << (2 * pp) : 0u;
            }
            if (j >= nact) m0 = m1 = m2 = m3 = 0u;
        } else {
        int d = 0;
        double gp = 0.0;  // floor(f), f < 1
        for (int u = 1; u <= 15; ++u) {
            const double g = __builtin_floor(f + (double)u * s);
            if ((MODE >= 2) ? (g != gp) : (g == gp)) {  // MODE 1: sample u HOLDS the half chip of sample u - 1; 2, 3: ADVANCES
                const uint32_t m = ~0u << (2 * u);
                m0 = d == 0 ? m : m0; m1 = d == 1 ? m : m1; m2 = d == 2 ? m : m2; m3 = d == 3 ? m : m3;
                ++d;
            }
            gp = g;
        }
        if (j >= nact) m0 = m1 = m2 = m3 = 0u;
        else if (d > (MODE == 2 ? 2 : 4)) s_rwbad = 1;  // more holds / advances than the masks carry (the host's gate excludes it):
                                                        // every group is listed
        }
        s_pat[t] = make_uint4(m0, m1, m2, m3);
    }
    __syncthreads();
    const bool rw_off = __builtin_amdgcn_readfirstlane(s_rwbad) != 0;

    // ---- the segment's chunks: wave w of the block takes chunks c_begin + w, + nw, ...
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int nw = nthr >> 6;
    const double g16 = (double)(16 * lane);
    // loader lanes: lane j < NCH fetches position j's checkpoint (idle positions alias slot act[0], see ixs)
    const int jl = lane < NCH ? lane : NCH - 1;
    int ixl = 0;
#pragma unroll
    for (int q = 0; q < NCH; ++q) ixl = jl == q ? ixs[q] : ixl;
    const bool onl = jl < nact;
    const size_t cpl = (size_t)ixl * G.CP1;
    const double dl = onl ? p_dstep[ixl] : 0.0;
    const double sl = onl ? 2.0 * p_cstep[ixl] : 0.0, dabsl = __builtin_fabs(dl);
    const uint32_t dsgnl = (uint32_t)(d2u(dl) >> 32) & 0x80000000u;
    // the position's symbol sign pairs, indexed by the symbol counter (+ 500 once the page has flipped) minus its value at the
    // epoch start (idle: an all-zero row)
    const uint32_t *const syml = s_sym + jl * SG_SYMS - (onl ? p_ib0[ixl] : 0);
    const uint32_t mtab0 = (uint32_t)(uintptr_t)(sg_lds_u32)s_mtab;
    SgRec *const recw = s_rec + wv * 2 * MC;
    uint32_t *const syaw = s_sya + wv * 2 * MC;
    const int cstep_w = nw;
    int c = __builtin_amdgcn_readfirstlane(c_begin + wv);
    // checkpoint of chunk cc: x (pre-check code phase, chips), p (carrier phase), ibit | flipped << 16
    double lx = 0.0, lp = 0.0;
    uint32_t lib = 0u;
    // (global-address-space loads: a FLAT load counts on the LDS counter as well, and the sample loop's explicit lgkmcnt waits
    // would then wait for this prefetch instead of running a sample ahead)
    typedef const __attribute__((address_space(1))) double *sg_gbl_f64;
    typedef const __attribute__((address_space(1))) uint32_t *sg_gbl_u32;
    const sg_gbl_f64 g_cpx = (sg_gbl_f64)(uintptr_t)p_cpx + cpl, g_cpp = (sg_gbl_f64)(uintptr_t)p_cpp + cpl;
    const sg_gbl_u32 g_cpi = (sg_gbl_u32)(uintptr_t)p_cpi + cpl;
    auto fetch = [&](const int cc) {
        const int ce = cc < c_end ? cc : c_end - 1;
        lx = g_cpx[ce];
        lp = g_cpp[ce];
        lib = g_cpi[ce];
    };
    auto stage = [&](const int buf) {
        SgRec r;
        // a wrap that is pending at the chunk's first sample (:491) is taken here: every group of the chunk lies behind it
        const bool pend = lx >= 4092.0;
        const double x = pend ? lx - 4092.0 : lx;
        // (idle positions: phases far from every boundary the kernel looks at -- their zero checkpoints would sit ON one, 511 p = 0,
        // and list every group of the epoch)
        r.yc = onl ? x + x : 0.5;
        r.pm = onl ? u2d(d2u(lp) ^ ((uint64_t)dsgnl << 32)) : 0.25;
        if (onl && dabsl == 0.0) {
            // a carrier that stands still (step exactly 0: `p += 0; p -= (long)p` leaves every phase of the epoch at the checkpoint's):
            // the table index of all its samples is the reference's own expression on that phase, known EXACTLY here, so the group
            // gets a phase in the middle of that entry -- no sample of it is ever undecided, even with the phase ON an index boundary
            // (rounds 2-4 kept such batches off this kernel for that).  Entry = k + 512 for k >= 0, k + 511 below (SG_LUT_N layout).
            // (on the mirrored phase, like every other: a step of -0.0 selects the conjugate table, whose entry for -k is LUT[k])
            const int k = (int)(511.0 * r.pm);  // :509, before the & 511
            r.pm = ((double)(k >= 0 ? k : k - 1) + 0.5) * (1.0 / 511.0);
        }
        r.s = sl;
        r.dabs = dabsl;
        if (lane < NCH) {
            recw[buf * MC + lane] = r;
            // (idle positions: steps zero, stream row zero, sign pairs zero -- no contribution)
            const uint32_t ks = onl ? (lib & 0xffffu) + 500u * (lib >> 16) + (pend ? 1u : 0u) : 0u;
            const uint32_t pair = syml[ks] * 4u + syml[ks + 1];
            syaw[buf * MC + lane] = mtab0 + pair * (SG_MPOS * 4u);
        }
    };
    if (c < c_end) fetch(c);
    int buf = 0;
    for (; c < c_end; c += cstep_w, buf ^= 1) {
        stage(buf);
        fetch(c + cstep_w);  // in flight while this chunk is synthesised
        const int n0 = c * SG_CHUNK;
        const int nsteps = G.N - n0 - 16 * lane;  // samples of this lane's group inside the epoch
        if (nsteps > 0) {
        uint32_t *out = iq + (size_t)er * G.N + n0 + 16 * lane;  // iq holds the executed range only
        const bool vec_ok = ((((size_t)er * G.N + n0) & 3) == 0) && nsteps >= 16;
        int o[16];
        if (ACC) {
            if (vec_ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(out + 4 * q);
                    o[4 * q] = (int)v.x; o[4 * q + 1] = (int)v.y; o[4 * q + 2] = (int)v.z; o[4 * q + 3] = (int)v.w;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) o[u] = u < nsteps ? (int)out[u] : 0;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) o[u] = 0;
        }
        uint32_t amb = ~0u;
        uint64_t undec = rw_off ? ~0ull : 0ull;
        const SgRec *rec = recw + buf * MC;
        const uint32_t *sya = syaw + buf * MC;
        // balanced parts: 5, 6, 7, 9, 10, 11 positions are cut 3+2, 3+3, 4+3, 3+3+3, 4+3+3, 4+4+3
#define SG_PART(J0, CNT) sg_part<J0, CNT, MODE, SIG, BINS, BPITCH, SEARCH>(o, amb, undec, g16, rec, sya, s_c511, s_lutd, s_str, s_bin, s_pat, s_bin6, s_pat6, s_thr, nact); \
                         __builtin_amdgcn_sched_barrier(0);
        // (CBOC: parts of at most three -- a part's group start keeps 17 values per position alive, 13 in the BOC(1,1) form)
        if constexpr (SIG == 1 && NCH <= 3) { SG_PART(0, NCH) }
        else if constexpr (SIG == 1 && NCH == 4) { SG_PART(0, 2) SG_PART(2, 2) }
        else if constexpr (SIG == 1 && NCH == 5) { SG_PART(0, 3) SG_PART(3, 2) }
        else if constexpr (SIG == 1 && NCH == 6) { SG_PART(0, 3) SG_PART(3, 3) }
        else if constexpr (SIG == 1 && NCH == 7) { SG_PART(0, 3) SG_PART(3, 2) SG_PART(5, 2) }
        else if constexpr (SIG == 1 && NCH == 8) { SG_PART(0, 3) SG_PART(3, 3) SG_PART(6, 2) }
        else if constexpr (SIG == 1 && NCH == 9) { SG_PART(0, 3) SG_PART(3, 3) SG_PART(6, 3) }
        else if constexpr (SIG == 1 && NCH == 10) { SG_PART(0, 3) SG_PART(3, 3) SG_PART(6, 2) SG_PART(8, 2) }
        else if constexpr (SIG == 1 && NCH == 11) { SG_PART(0, 3) SG_PART(3, 3) SG_PART(6, 3) SG_PART(9, 2) }
        else if constexpr (SIG == 1 && NCH == 12) { SG_PART(0, 3) SG_PART(3, 3) SG_PART(6, 3) SG_PART(9, 3) }
        else if constexpr (NCH <= 4) { SG_PART(0, NCH) }
        else if constexpr (NCH == 5) { SG_PART(0, 3) SG_PART(3, 2) }
        else if constexpr (NCH == 6) { SG_PART(0, 3) SG_PART(3, 3) }
        else if constexpr (NCH == 7) { SG_PART(0, 4) SG_PART(4, 3) }
        else if constexpr (NCH == 8) { SG_PART(0, 4) SG_PART(4, 4) }
        else if constexpr (NCH == 9) { SG_PART(0, 3) SG_PART(3, 3) SG_PART(6, 3) }
        else if constexpr (NCH == 10) { SG_PART(0, 4) SG_PART(4, 3) SG_PART(7, 3) }
        else if constexpr (NCH == 11) { SG_PART(0, 4) SG_PART(4, 4) SG_PART(8, 3) }
        else if constexpr (NCH == 12) { SG_PART(0, 4) SG_PART(4, 4) SG_PART(8, 4) }
        else {
            // the wide instances: parts of four, the last 5 / 6 / 7 positions cut 3+2 / 3+3 / 4+3 as above
            SG_PART(0, 4) SG_PART(4, 4)
            constexpr int R8 = NCH - 8;  // 5 .. 16 positions left
            if constexpr (R8 == 5) { SG_PART(8, 3) SG_PART(11, 2) }
            else if constexpr (R8 == 6) { SG_PART(8, 3) SG_PART(11, 3) }
            else if constexpr (R8 == 7) { SG_PART(8, 4) SG_PART(12, 3) }
            else {
                SG_PART(8, 4)
                constexpr int R12 = NCH - 12;  // 4 .. 12
                if constexpr (R12 <= 4) { SG_PART(12, R12) }
                else if constexpr (R12 == 5) { SG_PART(12, 3) SG_PART(15, 2) }
                else if constexpr (R12 == 6) { SG_PART(12, 3) SG_PART(15, 3) }
                else if constexpr (R12 == 7) { SG_PART(12, 4) SG_PART(16, 3) }
                else {
                    SG_PART(12, 4)
                    constexpr int R16 = NCH - 16;  // 4 .. 8
                    if constexpr (R16 <= 4) { SG_PART(16, R16) }
                    else if constexpr (R16 == 5) { SG_PART(16, 3) SG_PART(19, 2) }
                    else if constexpr (R16 == 6) { SG_PART(16, 3) SG_PART(19, 3) }
                    else if constexpr (R16 == 7) { SG_PART(16, 4) SG_PART(20, 3) }
                    else { SG_PART(16, 4) SG_PART(20, 4) }
                }
            }
        }
#undef SG_PART
        if (vec_ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4 *>(out + 4 * q) = make_uint4((uint32_t)o[4 * q], (uint32_t)o[4 * q + 1], (uint32_t)o[4 * q + 2], (uint32_t)o[4 * q + 3]);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < nsteps) out[u] = (uint32_t)o[u];
        }
        if (((undec >> lane) & 1ull) | (amb < SG_AMB)) {  // undecided: k_repair_g replays the group exactly
            const int slot = atomicAdd(p_ctr + CTR_GFLAGS, 1);
            if (slot < flist_cap) flist[slot] = ((uint32_t)er * (uint32_t)G.nchunks + (uint32_t)c) * 64u + (uint32_t)lane;
        }
        }
    }
    }  // segments
}

#if GAL_GTU_MAIN
// ------------------------------------------------------------------------------------------------
// k_repair_g: the listed groups once more, with the reference's statements (:489-533) from their EXACT start state -- the
// chunk's checkpoint advanced by 16 g samples in closed form (nco_walk.h) -- for ALL active channels of the epoch (every
// channel group of a > 12-channel batch), written over what the synthesis launches left.  A row of 16 lanes takes one
// group: lane r the slots r, r + 16, ..; the row then sums its 16 x (I, Q) and lane u stores sample u (:536-537).
__global__ __launch_bounds__(256) void k_repair_g(DevPlan P, SynGeom G, uint32_t *__restrict__ iq, const uint32_t *__restrict__ flist,
                                                  const int flist_cap)
{
    __shared__ int s_lut[1024];  // int16 pairs (2 cos, 2 sin): the one table the 16 samples gather from; CBOC: [TA 512][TB 512]
    const int n_raw = P.ctr[CTR_GFLAGS];
    const int n = n_raw < flist_cap ? n_raw : flist_cap;
    if (n_raw > flist_cap && blockIdx.x == 0 && threadIdx.x == 0) P.ctr[CTR_GOVER] = 1;  // gal_synth_finish repeats the batch exactly
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = threadIdx.x & 15;
    const int nrows = (gridDim.x * blockDim.x) >> 4;
    if (((int)blockIdx.x * (int)blockDim.x >> 4) >= n) return;  // no row of this block has a group
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_lut[i] = P.lut[i];  // (BOC(1,1): the second half repeats the first)
    __syncthreads();
    for (int i = gt >> 4; i < n; i += nrows) {
        const uint32_t ent = flist[i];
        const int g = (int)(ent & 63u);
        const uint32_t ec = ent >> 6;
        const int er = (int)(ec / (uint32_t)G.nchunks), c = (int)(ec - (uint32_t)er * (uint32_t)G.nchunks);
        const int e = G.e0 + er;
        const int n0 = c * G.R + 16 * g;
        int cnt = G.N - n0;
        cnt = cnt > 16 ? 16 : cnt;
        int aI[16], aQ[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) aI[u] = aQ[u] = 0;
        for (int s = r; s < G.S; s += 16) {
            const int idx = e * G.S + s;
            const int prn = P.prn[idx];
            if (prn <= 0) continue;
            const size_t cp = (size_t)idx * G.CP1 + c;
            const double cst = P.cstep[idx], d = P.dstep[idx];
            const uint32_t cib = P.cp_ib[cp];
            const CodeEnd ce = code_walk(P.cp_x[cp], (int)(cib & 0xffffu), cst, 1.0 / cst, 16 * g, 1 << 30, [](int, double, int, int) {});
            double x = ce.x;
            double p = carr_walk_track(P.cp_p[cp], d, 1.0 / __builtin_fabs(d), 16 * g, 16 * g + 1, 16 * g + 1, [](int, double) {}).p;
            {
                // ... and both chains walked on to the END of the chunk: they must arrive at the next checkpoint, bit for bit (the
                // end-of-epoch state behind the last chunk).  The listed groups are scattered over the batch by the bits of the
                // phases, so this is a random sample of ~20 000 (chunk, channel) pairs per 120 s batch -- the only check the CODE
                // checkpoints get on this kernel family (k_synth's replay checks every one), and one more on the carrier's beside
                // k_verify_carr's rotation.  A mismatch ends the batch in gal_synth_finish's repair path.
                int nR = G.N - c * G.R;
                nR = nR > G.R ? G.R : nR;
                const int rem = nR - 16 * g;
                const CodeEnd ce2 = code_walk(ce.x, ce.ibit, cst, 1.0 / cst, rem, 1 << 30, [](int, double, int, int) {});
                const uint32_t fl_end = (cib >> 16) | (uint32_t)ce.flipped | (uint32_t)ce2.flipped;
                const double p2 = carr_walk_track(p, d, 1.0 / __builtin_fabs(d), rem, rem + 1, rem + 1, [](int, double) {}).p;
                const bool bad = d2u(ce2.x) != d2u(P.cp_x[cp + 1]) || ((uint32_t)ce2.ibit | (fl_end << 16)) != P.cp_ib[cp + 1] ||
                                 d2u(p2) != d2u(P.cp_p[cp + 1]);
                if (bad) atomicAdd(&P.ctr[CTR_MISMATCH], 1);
            }
            // Everything the 16 samples read from memory, fetched in one go (a load per sample and table would make this a
            // chain of ~50 dependent memory round trips): the symbol in force and its successor (:497-506: at most one code wrap
            // inside 16 samples), the two stream words the half chips in front of the wrap can fall into and the two behind it
            int ib0 = ce.ibit, fl0 = (int)(cib >> 16) | ce.flipped;
            if (x >= 4092.0) {  // a wrap pending at the group's first sample (:491) is taken here: it may land anywhere in the
                x -= 4092.0;    // period (an epoch may start with code_phase0 up to 6138), a wrap by stepping lands at its start
                ib0 += 1;
                if (ib0 >= GAL_N_SYM_PAGE) {
                    ib0 = 0;
                    fl0 = 1;
                }
            }
            int ib1 = ib0 + 1, fl1 = fl0;
            if (ib1 >= GAL_N_SYM_PAGE) {
                ib1 = 0;
                fl1 = 1;
            }
            const uint32_t *pg0 = (fl0 ? P.page_next : P.page_cur) + (size_t)idx * GAL_PAGE_WORDS;
            const uint32_t *pg1 = (fl1 ? P.page_next : P.page_cur) + (size_t)idx * GAL_PAGE_WORDS;
            const uint32_t *str = P.str + (size_t)(prn - 1) * STR_WORDS;
            int wb = (int)(x * 2.0) >> 4;
            wb = wb > STR_WORDS - 1 ? STR_WORDS - 1 : wb;
            const uint32_t pw0 = pg0[ib0 >> 5], pw1 = pg1[ib1 >> 5];
            const uint32_t wa0 = str[wb], wa1 = str[wb + 1 < STR_WORDS ? wb + 1 : wb], wh0 = str[0], wh1 = str[1];
            // data symbol (:517) and secondary code (:518) of both symbols: 1 = the factor -1
            const int db0 = (int)((pw0 >> (ib0 & 31)) & 1u), sb0 = (int)((P.cs25 >> (ib0 % 25)) & 1u);
            const int db1 = (int)((pw1 >> (ib1 & 31)) & 1u), sb1 = (int)((P.cs25 >> (ib1 % 25)) & 1u);
            bool wrapped = false;
            for (int u = 0; u < cnt; ++u) {
                if (x >= 4092.0) {  // :491-507
                    x -= 4092.0;
                    wrapped = true;
                }
                const int k = ((int)(511.0 * p)) & 511;  // :509-510
                const int icode = (int)(x * 2.0);        // :512
                // stream field of half chip h: bit 0 = E1B ^ E1C chip, bit 1 = E1C chip ^ (h & 1); a chip bit 1 is the
                // value -1, and the BOC(1,1) sub-carrier makes the even half chip -chip, the odd one +chip
                // (src/gal-sig.cpp:9-233)
                const int wi = icode >> 4;
                const uint32_t w = wrapped ? (wi == 0 ? wh0 : wh1) : (wi == wb ? wa0 : wa1);
                const uint32_t fld = (w >> (2 * (icode & 15))) & 3u;
                const uint32_t cbit = (fld >> 1) ^ ((uint32_t)icode & 1u), bbit = (fld & 1u) ^ cbit;
                const int sub = (icode & 1) ? 1 : -1;
                const int E1B_chip = sub * (bbit ? -1 : 1), E1C_chip = sub * (cbit ? -1 : 1);
                const int databit = (wrapped ? db1 : db0) ? -1 : 1;   // :517
                const int secCode = (wrapped ? sb1 : sb0) ? -1 : 1;   // :518
                if (P.signal == 1) {
                    // CBOC(6,1,1/11) (include/galsynth.h, GAL_CFG_CBOC; not in the reference): the code arrays carry sc_A = the
                    // BOC(1,1) sub-carrier; sc_B from (int)(12 x), first half period negative; sc_A sc_B turns a value that carries
                    // sc_A into one that carries sc_B.  Tables: 2 TA, 2 TB (int16 pairs), so the halved sums below are exact
                    const int i12 = (int)(x * 12.0);
                    const int ab = ((icode ^ i12) & 1) ? -1 : 1;
                    const int Ba = E1B_chip * databit, Ca = E1C_chip * secCode;
                    const int ta = s_lut[k], tb = s_lut[512 + k];
                    aI[u] += ((Ba - Ca) / 2) * (int)(short)(ta & 0xffff) + ab * ((Ba + Ca) / 2) * (int)(short)(tb & 0xffff);
                    aQ[u] += ((Ba - Ca) / 2) * (int)(short)((uint32_t)ta >> 16) + ab * ((Ba + Ca) / 2) * (int)(short)((uint32_t)tb >> 16);
                } else {
                    const int v = E1B_chip * databit - E1C_chip * secCode;  // :520-521, in {-2, 0, 2}
                    const int ent2 = s_lut[k];  // int16 pair (2 cos, 2 sin)
                    aI[u] += (v / 2) * (int)(short)(ent2 & 0xffff);
                    aQ[u] += (v / 2) * (int)(short)((uint32_t)ent2 >> 16);
                }
                x = x + cst;           // :528
                p = carr_step(p, d);   // :531-532
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int I = aI[u], Q = aQ[u];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) {
                I += __shfl_xor(I, m, 16);
                Q += __shfl_xor(Q, m, 16);
            }
            if (r == u && u < cnt) iq[(size_t)er * G.N + n0 + u] = ((uint32_t)(uint16_t)(short)I) | ((uint32_t)(uint16_t)(short)Q << 16);
        }
    }
}

#endif  // GAL_GTU_MAIN (k_repair_g)

// ------------------------------------------------------------------------------------------------
// Blocks of a launch over `ne` epochs: every block takes a contiguous, equally long run of the launch's chunks (k_synth_g), cut so
// that block boundaries fall on epoch boundaries: ne x bpe blocks.  One block per epoch where that fills the device (M-SYN12: 1199
// blocks on 512 resident slots -- 2 blocks of 512 threads per CU: 61 KB of LDS, 128 VGPRs); a batch of fewer epochs is cut into the
// smallest bpe with ne x bpe >= slots that leaves every wave two chunks; epochs long enough that a block's tables are cheap beside
// its samples (16 chunks per wave and more: BASELINE config 4's 2442-chunk epochs) on until the launch is 4096 blocks.
// Measured and NOT taken (round 5; profiles/r05b_rounds_ab.log, same box, M-SYN12, kernel alone / pipelined step):
// a PERSISTENT grid of balanced ranges that ignore the epoch boundaries -- 512 blocks (one round of the slots) 0.939 / 1.103 ms,
// 1024 blocks 0.909 / 1.064, 1536 0.932 / 1.065, 2048 0.969 / 1.09, 3072 0.98 / 1.158 -- against one block per epoch 0.894 / 1.025:
// the partly empty third round costs less than it looks (the blocks left run on emptier SIMDs, faster), a block that crosses an
// epoch boundary pays a barrier and a second set of tables, and 512 blocks that live as long as the launch leave the next batch's
// walker kernels no SIMD to start on.  (GAL_TEST_HOOKS: GAL_G_ROUNDS / GAL_G_BPE select those layouts.)
static int sg_grid(const DevPlan *P, int ne, int nch = 0)
{
    const long long U = (long long)ne * P->nchunks;
    const bool wide = nch > SG_MAXCH;  // one block of 16 waves per CU instead of two of 8
    const int slots = (P->gslots > 0 ? P->gslots : 512) / (wide ? 2 : 1);
#ifdef SG_BALANCED_RANGES
    if (P->grounds > 0 && U >= (long long)slots * P->grounds * 32) return slots * P->grounds;  // balanced persistent ranges
#else
    (void)U;
#endif
    int bpe = P->gbpe > 0 ? P->gbpe : 1;                                                        // (hooks: fixed blocks per epoch)
    if (P->gbpe <= 0) {
        const int nw = wide ? 16 : 8;  // waves per block
        bpe = (slots + ne - 1) / ne;
        const int most = P->nchunks / (2 * nw) > 1 ? P->nchunks / (2 * nw) : 1;
        bpe = bpe > most ? most : bpe;
        while ((long long)ne * bpe < 4096 && P->nchunks / (bpe * 2 * nw) >= 16) bpe *= 2;
    }
    return ne * bpe;
}
// ... and the blocks per epoch of that grid (0: a hooks layout that ignores the epoch boundaries)
static int sg_bpe(const DevPlan *P, int ne, int nch = 0)
{
    return sg_grid(P, ne, nch) / ne;
}

// the wide and the bisection instances live in translation units of their own (GAL_GTU 1 / 2): (acc, mode) dispatched there
extern "C" int galk_launch_synth_g_wide(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, int mode, const uint8_t *act, const int *nact,
                                        uint32_t *iq, const SynGeom *G, int grid, hipStream_t st);
extern "C" int galk_launch_synth_g_search(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, int mode, const uint8_t *act, const int *nact,
                                          uint32_t *iq, const SynGeom *G, int grid, hipStream_t st);

#if GAL_GTU_WIDE
template <bool ACC, int MODE>
static int launch_wide_t(const DevPlan *P, const DevPlan *Pd, int nch, const uint8_t *act, const int *nact, uint32_t *iq, const SynGeom &G, int ngrid,
                         hipStream_t st)
{
    const dim3 grid(ngrid), wblock(SG_THREADS);  // blocks of 1024 threads, one per CU
#define GAL_WCASE(n) case n: hipLaunchKernelGGL((k_synth_g<n, ACC, MODE, 0>), grid, wblock, 0, st, Pd, G, act, nact, iq, P->gflist, P->gflist_cap); return 0;
    switch (nch) {
        GAL_WCASE(13) GAL_WCASE(14) GAL_WCASE(15) GAL_WCASE(16) GAL_WCASE(17) GAL_WCASE(18)
        GAL_WCASE(19) GAL_WCASE(20) GAL_WCASE(21) GAL_WCASE(22) GAL_WCASE(23) GAL_WCASE(24)
    default: return -1;
    }
#undef GAL_WCASE
}
extern "C" int galk_launch_synth_g_wide(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, int mode, const uint8_t *act, const int *nact,
                                        uint32_t *iq, const SynGeom *G, int grid, hipStream_t st)
{
#define GAL_MCASE(m) case m: return accumulate ? launch_wide_t<true, m>(P, Pd, nch, act, nact, iq, *G, grid, st) : launch_wide_t<false, m>(P, Pd, nch, act, nact, iq, *G, grid, st);
    switch (mode) {
        GAL_MCASE(1) GAL_MCASE(2) GAL_MCASE(3) GAL_MCASE(4)
    default: return -3;
    }
#undef GAL_MCASE
}
#endif  // GAL_GTU_WIDE

#if GAL_GTU_SEARCH
template <bool ACC, int MODE>
static int launch_search_t(const DevPlan *P, const DevPlan *Pd, int nch, const uint8_t *act, const int *nact, uint32_t *iq, const SynGeom &G, int ngrid,
                           hipStream_t st)
{
    const dim3 grid(ngrid), block(512);
#define GAL_SCASE(n) case n: hipLaunchKernelGGL((k_synth_g<n, ACC, MODE, 0, true>), grid, block, 0, st, Pd, G, act, nact, iq, P->gflist, P->gflist_cap); return 0;
    switch (nch) {
        GAL_SCASE(1) GAL_SCASE(2) GAL_SCASE(3) GAL_SCASE(4) GAL_SCASE(5) GAL_SCASE(6)
        GAL_SCASE(7) GAL_SCASE(8) GAL_SCASE(9) GAL_SCASE(10) GAL_SCASE(11) GAL_SCASE(12)
    default: return -1;
    }
#undef GAL_SCASE
}
extern "C" int galk_launch_synth_g_search(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, int mode, const uint8_t *act, const int *nact,
                                          uint32_t *iq, const SynGeom *G, int grid, hipStream_t st)
{
#define GAL_MCASE(m) case m: return accumulate ? launch_search_t<true, m>(P, Pd, nch, act, nact, iq, *G, grid, st) : launch_search_t<false, m>(P, Pd, nch, act, nact, iq, *G, grid, st);
    switch (mode) {
        GAL_MCASE(1) GAL_MCASE(2) GAL_MCASE(3) GAL_MCASE(4)
    default: return -3;
    }
#undef GAL_MCASE
}
#endif  // GAL_GTU_SEARCH

#if GAL_GTU_MAIN
template <bool ACC, int MODE, int SIG>
static int launch_synth_g_t(const DevPlan *P, const DevPlan *Pd, int nch, const uint8_t *act, const int *nact, uint32_t *iq, int e0,
                            int ne, hipStream_t st, const SynGeom &G)
{
    if constexpr (SIG == 0) {
        // crowded pattern thresholds: the bisection instances (BOC(1,1), <= 12 channels per launch); 13-24 channels: the wide ones
        if (P->rw_search) return galk_launch_synth_g_search(P, Pd, nch, ACC ? 1 : 0, MODE, act, nact, iq, &G, sg_grid(P, ne, nch), st);
        if (nch > SG_MAXCH) return galk_launch_synth_g_wide(P, Pd, nch, ACC ? 1 : 0, MODE, act, nact, iq, &G, sg_grid(P, ne, nch), st);
    }
#ifdef SG_FORCE_THREADS  // A/B builds
    const dim3 grid(sg_grid(P, ne, nch)), block(SG_FORCE_THREADS);
#else
    const dim3 grid(sg_grid(P, ne, nch)), block(P->gthreads >= 64 && P->gthreads <= SG_THREADS && (P->gthreads & 63) == 0 ? P->gthreads : 512);
#endif
#define GAL_CASE(n) case n: hipLaunchKernelGGL((k_synth_g<n, ACC, MODE, SIG>), grid, block, 0, st, Pd, G, act, nact, iq, P->gflist, P->gflist_cap); break;
    switch (nch) {  // one instance per channel count, in the CBOC mode too (round 4: 4 / 8 / 12 positions, a 9-SV batch paid for 12)
        GAL_CASE(1) GAL_CASE(2) GAL_CASE(3) GAL_CASE(4) GAL_CASE(5) GAL_CASE(6)
        GAL_CASE(7) GAL_CASE(8) GAL_CASE(9) GAL_CASE(10) GAL_CASE(11) GAL_CASE(12)
    default: return -1;
    }
#undef GAL_CASE
    return 0;
}

static SynGeom sg_geom(const DevPlan *P, int e0, int ne, int nch = 0)
{
    SynGeom G;
    G.e0 = e0;
    G.ne = ne;
    G.S = P->S; G.N = P->N; G.R = P->R; G.nchunks = P->nchunks; G.CP1 = P->CP1;
    G.blocks_per_epoch = ne > 0 ? sg_bpe(P, ne, nch) : 1;
    G.cls = 1;
    G.per = P->nchunks;
    return G;
}

__global__ void k_warm_g() {}
extern "C" void galk_warm_g(hipStream_t st) { hipLaunchKernelGGL(k_warm_g, dim3(1), dim3(64), 0, st); }

extern "C" int galk_launch_synth_g(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, const uint8_t *act, const int *nact,
                                   uint32_t *iq, int e0, int ne, hipStream_t st)
{
    if (P->R != SG_CHUNK) return -2;
    const SynGeom G = sg_geom(P, e0, ne, nch);
    if (P->signal == 1) {
#define SG_MODE_CASE(m) case m: return accumulate ? launch_synth_g_t<true, m, 1>(P, Pd, nch, act, nact, iq, e0, ne, st, G) \
                                                  : launch_synth_g_t<false, m, 1>(P, Pd, nch, act, nact, iq, e0, ne, st, G);
        switch (P->rw) {
            SG_MODE_CASE(1) SG_MODE_CASE(2) SG_MODE_CASE(3) SG_MODE_CASE(4)
        default: return -3;
        }
#undef SG_MODE_CASE
    }
#define SG_MODE_CASE(m) case m: return accumulate ? launch_synth_g_t<true, m, 0>(P, Pd, nch, act, nact, iq, e0, ne, st, G) \
                                                  : launch_synth_g_t<false, m, 0>(P, Pd, nch, act, nact, iq, e0, ne, st, G);
    switch (P->rw) {
        SG_MODE_CASE(1) SG_MODE_CASE(2) SG_MODE_CASE(3) SG_MODE_CASE(4)
    default: return -3;
    }
#undef SG_MODE_CASE
}

// behind the last synthesis launch of the batch, same stream
extern "C" void galk_launch_repair_g(const DevPlan *P, uint32_t *iq, int e0, hipStream_t st)
{
    const SynGeom G = sg_geom(P, e0, 0);
    // (512 blocks x 16 rows: a batch of the reference geometry lists ~2000 groups, one of BASELINE config 4's ~16 000 per 600 epochs;
    // blocks without a group leave at once)
    hipLaunchKernelGGL(k_repair_g, dim3(512), dim3(256), 0, st, *P, G, iq, P->gflist, P->gflist_cap);
}
#endif  // GAL_GTU_MAIN (launchers)
