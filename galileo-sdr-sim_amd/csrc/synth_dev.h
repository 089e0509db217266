// synth_dev.h -- device-side view of one planned batch (shared by synth_kernels.hip and synth_api.cpp).
#ifndef GAL_SYNTH_DEV_H_
#define GAL_SYNTH_DEV_H_

#include <stdint.h>

#include <hip/hip_runtime.h>

#include "../../include/galsynth.h"

enum { CTR_UNVERIFIED = 0, CTR_PASSES = 1, CTR_MISMATCH = 2, CTR_UNVER_NEXT = 3, CTR_WALKS = 4, CTR_SHIFTS = 5, CTR_REWALK_NEXT = 6,
       CTR_TICKET = 7,
       CTR_GFLAGS = 8,  // k_synth_g: groups listed for k_repair_g (the list's fill count)
       CTR_GOVER = 9,   // ... the list overflowed: gal_synth_finish repeats the batch with the exact-replay kernel
       CTR_COUNT = 10 };

// Everything lives in HBM; [E][S] arrays are indexed e * S + s.
struct DevPlan {
    int E;        // epochs in the batch
    int S;        // channel slots per epoch row
    int N;        // samples per epoch
    int R;        // samples replayed by one lane (chunk)
    int nchunks;  // ceil(N / R)
    int CP1;      // checkpoint row stride = nchunks + 1 (last entry = end-of-epoch state)
    int blocks_per_epoch;
    int cls;      // code-phase classes: chunk c is replayed by position (c % cls) * (nchunks / cls) + c / cls of its epoch
                  // (k_synth: lanes of one wave then share their code phase); 1 = natural order
    int W;        // carrier-walk legs per epoch
    int Lc;       // chunks per leg (leg length = Lc * R samples)
    int LEGS;     // E * W
    int LEGS_all; // LEGS of the PLAN (gal_synth_execute_range cuts E and LEGS to the executed prefix; what is laid out once per plan --
                  // the stitch's look-back records -- keeps the plan's strides whatever the cut)
    int Wc;       // code-walk legs per epoch (a power of two <= 64: the legs of an epoch are neighbouring lanes of one wave)
    int Lkc;      // chunks per code leg
    double delt;  // 1.0 / sample_rate, src/galileo-sdr.cpp:162
    uint32_t cs25;

    const uint32_t *page_init;  // [restart records][16]: the page in force at a (re)allocation (gal_chan_epoch_t::page_init), compact --
    const int *init_ix;         // [E][S] row of page_init for a record with GAL_CH_RESTART (round 6: the 176-byte records themselves are no
                                // longer uploaded, this was all the device read from them beside the SoA copies: half of a plan's upload)
    const gal_chan_state_t *state_in;  // [S]
    gal_chan_state_t *state_out;       // [S]

    // SoA copies (k_prep)
    int *prn;           // [E][S]
    uint32_t *flags;    // [E][S]
    int *ib0;           // [E][S]
    double *x0;         // [E][S] code phase at epoch start
    double *p0;         // [E][S] carrier phase for restarts
    double *cstep;      // [E][S] fl(f_code * delt)
    double *dstep;      // [E][S] fl(f_carr * delt)
    uint32_t *page_next;  // [E][S][16]
    uint32_t *page_cur;   // [E][S][16] page in force at epoch start (k_pages)
    uint8_t *flip_in;     // [E][S] symbol counter wrapped inside this epoch

    // carrier speculation (leg arrays are slot-major, [S][LEGS])
    double *pguess;      // [E][S] ideal-arithmetic phase at epoch start (host: synth_api.cpp, carrier_guesses)
    long long *gss_w;    // [E][S] ideal last wrap (or root) at or before the epoch start: global sample index
    double *gss_r;       // [E][S] ... and its residual
    long long *anc_w;    // anchor of the leg: global sample index ...
    double *anc_r;       // ... and the phase before that sample (a wrap residual, or the chain root)
    long long *clm_w;    // claim: last wrap seen by the leg's last walk (-1: none)
    double *clm_r;
    double *pend;        // phase after the leg's last sample (that walk)
    uint8_t *verified;
    uint8_t *dirty;      // 0 clean, 1 walk again from the new anchor, 2 translate by `shift` instead of walking
    double *marg;        // min distance of the leg's walked states to their binade boundaries (nco_walk.h)
    int8_t *tdir;        // rounding direction of the first tie the walk met (WalkOut::tdir), 0 = none
    long long *tpos;     // global sample index right after that tie step
    double *shift;       // pending translation (new anchor residual - walked anchor residual)
    uint8_t *risk;       // 1: the leg was accepted by a translation that used more than 1/256 of its binade margin: k_verify_carr
                         // re-walks such a leg in every batch, whatever the rotation says
    int ver_mod, ver_rem;  // k_verify_carr re-walks the legs i = ver_rem (mod ver_mod) of the executed epochs (1, 0: every leg)
    void *scanm;         // scratch of the stitch: tickets and look-back records (synth_kernels.hip: ScanM)
    int translate;       // 1 normal; 0: always re-walk (the all-walked fallback); 2: GAL_TEST_HOOKS builds only
    int tr_e0, tr_e1;    // legs of epochs outside [tr_e0, tr_e1) are never translated (gal_synth_execute_range)
    int hook_spoil;      // GAL_TEST_HOOKS builds only (GAL_GUESS_SPOIL): the first pass anchors leg GAL_HOOK_BAD_LEG of slot 0 one sample late
    int cp_e0;           // the walkers emit chunk checkpoints from this epoch on only (gal_synth_execute_range: epochs in
                         // front of the range are walked silently -- their states are needed, their checkpoints are not)

    // checkpoints, one per chunk + end state
    double *cp_x;     // [E][S][CP1]
    double *cp_p;     // [E][S][CP1]
    uint32_t *cp_ib;  // [E][S][CP1]  ibit | flipped << 16

    int *ctr;  // [CTR_COUNT]

    // tables
    const int *lut;       // [512] int16 pairs (2 cos, 2 sin), low half first; CBOC: [2][512] = 2 TA, 2 TB
    int signal;           // 0 BOC(1,1) (the reference), 1 CBOC(6,1,1/11) (GAL_CFG_CBOC)
    int rw;               // resampled-window body of k_synth: 1 every code step has 0.74 <= 2 f_code / fs < 1 (holds),
                          // 2 / 3 every code step has 2 f_code / fs <= 0.133 / 0.266 (<= 2 / 4 advances), 0 classic per-sample window index
    int rw_search;        // k_synth_g: 1 = find a group's pattern by bisection over the sorted thresholds (rates whose thresholds crowd), 0 = bin table
    int fam;              // synthesis kernel family: 0 k_synth (one chunk per lane, exact replay), 1 k_synth_g (one 16-sample group per
                          // lane from the chunk's checkpoint in closed form + k_repair_g for the undecided groups; synth_group.hip)
    int gbpe;             // k_synth_g: 0 (product): contiguous chunk ranges, see gslots / grounds; > 0 (GAL_TEST_HOOKS): blocks per epoch
    int gthreads;         // k_synth_g: threads per block (512)
    int gslots;           // k_synth_g: blocks the device holds at a time (2 per CU)
    int grounds;          // k_synth_g: a big launch is gslots x grounds blocks
    uint32_t *gflist;     // k_synth_g -> k_repair_g: the undecided groups, (epoch in range * nchunks + chunk) * 64 + group
    int gflist_cap;
    const uint32_t *str;  // [50][512] half-chip streams: bit 2h = E1B^E1C chip, bit 2h+1 = E1C chip ^ (h & 1)
};

// by-value geometry of the hot kernel (everything else it reaches through a device copy of DevPlan, so
// that rarely used pointers do not occupy SGPRs inside the sample loop)
struct SynGeom {
    int S, N, R, nchunks, CP1, blocks_per_epoch;
    int cls, per;  // lane order: position L of the epoch replays chunk (L % per) * cls + L / per, per = nchunks / cls
    int e0;  // first epoch of the executed range (the output buffer starts there)
    int ne;  // epochs of the launch (k_synth_g: its blocks cut ne x nchunks chunks among themselves)
};

#endif
