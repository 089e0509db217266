// synth_kernels.hip -- gfx950 (CDNA4) kernels of the Galileo E1B/C IQ synthesis engine.
//
// Replaces the per-sample loop of the reference, src/galileo-sdr.cpp:481-539.
// Pipeline per batch of epochs (walker chain on two high-priority streams of the handle, k_synth on the
// caller's stream, joined by events):
//   (host)         gal_synth_plan writes the SoA copies of the epoch records and the NCO steps c = f_code*delt,
//                  d = f_carr*delt (one rounding each, exactly the product the reference recomputes every sample,
//                  :528,:531) into the upload region
//   k_walk_code    one lane per (epoch, slot): exact closed-form walk of the code-phase chain, emitting a
//                  checkpoint (phase, symbol index, page-flip flag) every R samples           [nco_walk.h]
//   k_pages        which page is in force at each epoch start (pages change only when a symbol counter
//                  wraps inside the sample loop, :497-506)                [both on the second walker stream]
//   k_walk_carr / k_scanm, normally ONE pass
//                  the carrier chain runs unbroken across epochs, so it is evaluated speculatively on LEGS
//                  (8 per epoch): a leg is walked from its anchor = the last wrap event before it (first
//                  guess: drift-compensated ideal arithmetic); the stitcher accepts a leg only when its
//                  anchor is BITWISE the claim of the verified chain before it, otherwise re-anchors it at
//                  the predicted claim (rounded-add chains commute with shifts on the 2^-52 grid every wrap
//                  residual lives on, up to one predictable tie flip).  A re-anchored leg whose anchor only
//                  moved by less than its binade margin is TRANSLATED by the stitcher on the spot instead of
//                  walked again; the last block of a stitch publishes the pass and the end-of-batch phase.
//   k_synth<NCH>   the hot kernel: one lane replays R consecutive samples for ALL active channels with the
//                  reference's exact operation sequence from its checkpoint, accumulates packed
//                  (Q<<16)+I in a register, and stores int16 I/Q in 64-byte bursts.  PRN memory codes (as
//                  2-bit-per-half-chip streams) and the sin/cos LUT live in LDS.  Each lane finally checks
//                  its end state against the next checkpoint, so whatever the walkers and the translation
//                  produced is verified against genuine stepping on every run.
// No MFMA anywhere: this is FP64/integer ALU work bounded above by the 4 B/sample HBM write.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nco_walk.h"
#include "synth_dev.h"
#include "synth_common.h"

using namespace galnco;
using namespace galdev;

// The file is compiled as eight translation units, side by side (Makefile: -DGAL_TU=0..7), because the instantiations of
// k_synth take minutes in one go: TU 0 holds the walker kernels and the launch dispatcher -- the only part the
// GAL_TEST_HOOKS build changes --, TU 1..4 one family of k_synth each, (SIG, RW) = (0, 0), (0, 1), (0, 2), (1, 0), shared
// by both libraries; TU 5 the family (0, 3), TU 6 (1, 1): CBOC on resampled windows.  Without GAL_TU everything lands in one
// translation unit (kasm.sh (a tool of rounds 3-5: git history)).  (The default kernel of the reference geometry, k_synth_g, is synth_group.hip.)
#if !defined(GAL_TU)
#define GAL_TU_WALK 1
#define GAL_TU_SYNTH 1
#define GAL_TU_FAMILY(k) 1
#else
#define GAL_TU_WALK (GAL_TU == 0)
#define GAL_TU_SYNTH (GAL_TU != 0)
#define GAL_TU_FAMILY(k) (GAL_TU == (k))
#endif

// the walker kernels raise their waves' issue priority (A/B: -DGAL_WALK_PRIO=0 leaves it alone)
#ifndef GAL_WALK_PRIO
#define GAL_WALK_PRIO 3
#endif
#if GAL_WALK_PRIO > 0
#define GAL_WALK_SETPRIO() __builtin_amdgcn_s_setprio(GAL_WALK_PRIO)
#else
#define GAL_WALK_SETPRIO() ((void)0)
#endif

#ifdef GAL_TEST_HOOKS
#define GAL_HOOK_BAD_LEG 5  // (slot 0, epoch 0, leg 5) receives a wrong translation when P.translate == 2; P.translate == 3: code leg 1 of
                            // (slot 0, epoch 0) does (k_walk_code)
#endif

#if GAL_TU_WALK
// (The SoA copies of the epoch records -- prn, flags, ib0, x0, p0, cstep, dstep, page_next -- are written by
// gal_synth_plan on the host, into the upload region: round 2 had a kernel for it, one more launch in front of the chain.)
// ------------------------------------------------------------------------------------------------
// One lane per (slot, epoch, LEG): the code chain restarts every epoch from host-known values (src/gal-sig.cpp:336), and since
// round 5 an epoch's chain is cut into P.Wc legs (4 in a long batch, up to 16 where a batch of a few epochs is all latency) that
// are walked side by side -- rounds 1-4 walked an epoch's 351 dependent closed-form steps in ONE lane, 0.22 ms, the long pole of a
// lone handle's walker chain and of a one-epoch call.  Leg 0 starts from the epoch's own state; leg k > 0 from the last wrap in
// front of it as ideal arithmetic predicts it, and the legs of an epoch -- neighbouring lanes -- are then stitched from left to
// right: accepted as walked, TRANSLATED by what the anchor was off (a multiple of 2^-41; the walk's margin covers it), or walked
// again from the true anchor (nco_walk.h: code_leg_walk / code_ideal_anchor / code_leg_accept, with the host statement of this
// kernel, walk_host.cpp: galwalk_code_legs, tested against brute-force stepping in tests/test_walker_cpu.py).
__device__ __forceinline__ double shfl_up1_f64(double v)
{
    const uint64_t u = d2u(v);
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)u, 1), hi = (uint32_t)__shfl_up((int)(uint32_t)(u >> 32), 1);
    return u2d(((uint64_t)hi << 32) | lo);
}

__global__ void k_walk_code(DevPlan P)
{
    GAL_WALK_SETPRIO();  // latency-bound: win issue arbitration against a co-running k_synth
    // a wave = 64 / Wc consecutive epochs of ONE slot, the Wc legs of an epoch in neighbouring lanes (similar trip counts, idle slots
    // leave as whole waves)
    const int Wc = P.Wc;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.E * P.S * Wc) return;  // (Wc divides 64: a group of legs is inside one wave, and leaves as a whole)
    const int k = t & (Wc - 1);
    const int se = t / Wc;
    const int s = se / P.E;
    const int idx = (se - s * P.E) * P.S + s;
    if (P.prn[idx] <= 0) return;
    double *cpx = P.cp_x + (size_t)idx * P.CP1;
    uint32_t *cpi = P.cp_ib + (size_t)idx * P.CP1;
    const double c = P.cstep[idx];
    const double x0 = P.x0[idx];
    const int ib0 = P.ib0[idx];
    if (idx < P.cp_e0 * P.S) {
        // an epoch in front of the executed range: only its page flip matters (k_pages), no checkpoint is read
        if (k == 0) {
            const CodeEnd end = code_walk(x0, ib0, c, 1.0 / c, P.N, P.N, [](int, double, int, int) {});
            P.flip_in[idx] = (uint8_t)end.flipped;
        }
        return;
    }
    const int Lk = P.Lkc;  // chunks per leg
    const int n0 = k * Lk * P.R;
    int n1 = (k + 1) * Lk * P.R;
    n1 = n1 > P.N ? P.N : n1;
    const bool have = n0 < P.N;
    const bool last = have && (k == Wc - 1 || (k + 1) * Lk * P.R >= P.N);  // the leg that ends the epoch
    CodeEvent anc;
    anc.w = -1; anc.r = x0; anc.ib = ib0; anc.fl = 0;
    if (k > 0 && have) anc = code_ideal_anchor(x0, ib0, c, n0);
    CodeLeg L;
    L.claim = anc; L.x = x0; L.ibit = ib0; L.fl = 0; L.margin = 0.0; L.tpos = -1; L.tx = 0.0;
    const double inv_c = 1.0 / c;
    auto emit = [&](int ci, double x, int ib, int fl) {
        cpx[k * Lk + ci] = x;
        cpi[k * Lk + ci] = (uint32_t)ib | ((uint32_t)fl << 16);
    };
    if (have) L = code_leg_walk(anc, c, inv_c, n0, n1 - n0, P.R, emit);
    // ---- the stitch, leg by leg: lane k looks at the (by then true) claim of lane k - 1
    const bool tie = code_tie_prone(c);
    double shift = 0.0;   // what the leg's checkpoints are to be moved by ...
    int shift_from = 0;   // ... from this sample of the epoch on (translation across a tie step: the ones in front of it are rewritten)
    for (int step = 1; step < Wc; ++step) {
        CodeEvent prev;
        prev.w = __shfl_up(L.claim.w, 1);
        prev.r = shfl_up1_f64(L.claim.r);
        prev.ib = __shfl_up(L.claim.ib, 1);
        prev.fl = __shfl_up(L.claim.fl, 1);
        if (k != step || !have) continue;
        double dl;
        int how = code_leg_accept(anc, prev, L.margin, tie, L.tpos, &dl);
        // the all-walked fallback (gal_synth_finish after a checkpoint mismatch): no leg is accepted by translation, every one that
        // was not walked from the true anchor is walked again from it
        if (!P.translate && (how == 1 || how == 3)) how = 2;
#ifdef GAL_TEST_HOOKS
        if (P.translate == 3 && idx == 0 && k == 1 && (how == 0 || how == 1)) {  // (slot 0, epoch 0, code leg 1): a deliberately wrong shift
            how = 1;
            dl += 4.547473508864641e-13;  // 2^-41
        }
#endif
        if (how == 1) {
            shift = dl;
            shift_from = n0;
            L.x += dl;
            L.claim.r += dl;
        } else if (how == 3) {
            const double xt = code_leg_upto(prev, c, inv_c, n0, L.tpos, P.R, emit);
            shift = xt - L.tx;
            shift_from = L.tpos;
            L.x += shift;
            if (L.claim.w == anc.w) L.claim = prev;
            else L.claim.r += shift;
        } else if (how == 2) {
            anc = prev;
            L = code_leg_walk(anc, c, inv_c, n0, n1 - n0, P.R, emit);
        }
    }
    if (!have) return;
    if (shift != 0.0) {  // the leg's own checkpoints, eight read-modify-writes in flight
        const int nck = (n1 - n0 + P.R - 1) / P.R;
        for (int c0 = 0; c0 < nck; c0 += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (c0 + q < nck && n0 + (c0 + q) * P.R >= shift_from) ? cpx[k * Lk + c0 + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (c0 + q < nck && n0 + (c0 + q) * P.R >= shift_from) cpx[k * Lk + c0 + q] = v[q] + shift;
        }
    }
    if (last) {
        cpx[P.nchunks] = L.x;
        cpi[P.nchunks] = (uint32_t)L.ibit | ((uint32_t)L.fl << 16);
        // a wrap still pending after the last sample is discarded by the next epoch's overwrite
        // (SURVEY.md section 7.3-3), so only flips that happened inside the loop count
        P.flip_in[idx] = (uint8_t)L.fl;
    }
}

// ------------------------------------------------------------------------------------------------
// Carrier chain.  It runs unbroken across epochs, so it is evaluated speculatively on LEGS: leg
// i = e*W + w of slot s covers samples [w*L, (w+1)*L) of epoch e (L = Lc*R).  Leg arrays are slot-major,
// [s][i], so one wave can stitch a slot with coalesced loads.
//
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const uint64_t u = d2u(v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, lane);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), lane);
    return u2d(((uint64_t)hi << 32) | lo);
}

// 256 threads = one wave per SIMD: beside a running k_synth (which fills every SIMD's register file) a block can start
// as soon as ONE synthesis block retires; a 1024-thread block would have to wait for an entirely empty CU
#define GUESS_THREADS 256
// (The first guesses of the speculation -- ideal-arithmetic phase at every epoch start, ideal last wrap before it -- are computed by
// gal_synth_plan on the host since round 5, synth_api.cpp: carrier_guesses; rounds 1-4: a kernel here, k_carr_guess.)
__device__ __forceinline__ int d_residue_u52(double D) { return (int)((long long)(D * 4503599627370496.0) & 3LL); }

// TRANSLATED acceptance of a leg (called by the stitcher, k_scanm, for a leg whose anchor is the
// same wrap event as before with a residual that moved by `dl`, a multiple of 2^-52 smaller than the leg's binade
// margin): a walk from the new anchor would visit the same binades step by step, so every state it produces is the old
// one plus the shift, bit for bit (nco_walk.h: binade_margin; ties: WalkOut::tdir) -- the leg's 32 checkpoints, end
// phase and claim are shifted in place instead of walking it again.  k_synth's replay check covers it.
struct TrRec {       // a translation whose checkpoint updates are left to the caller (k_scanm: the block does them together)
    double dl, dl2;  // shift before / from the tie step on
    long long tp, A; // global sample index right after the tie step; of the leg's first sample
    size_t base;     // element offset of the leg's first checkpoint in cp_p
    int nck;         // checkpoints of the leg (0: nothing to do)
};

__device__ __forceinline__ void translate_leg(const DevPlan &P, const int s, const int i, double dl, TrRec *defer)
{
    const int e = i / P.W, w = i - e * P.W;
    const int idx = e * P.S + s;
    const size_t li = (size_t)s * P.LEGS + i;
    const long long A = (long long)e * P.N + (long long)w * (P.Lc * P.R);  // global index of the leg's first sample
#ifdef GAL_TEST_HOOKS
    if (P.translate == 2 && li == GAL_HOOK_BAD_LEG) dl += 4.440892098500626e-16;  // a deliberately wrong shift
#endif
    // an odd shift flips the first tie of the walk: from that wrap on the trajectory is off by dl2
    const int td = P.tdir[li];
    const bool flip = td != 0 && (d_residue_u52(dl) & 1);
    const double dl2 = flip ? dl - (double)td * 2.220446049250313e-16 : dl;
    const long long tp = flip ? P.tpos[li] : (long long)1 << 62;  // global index right after the tie step
    int nck = P.nchunks - w * P.Lc;
    nck = nck > P.Lc ? P.Lc : nck;
    const size_t base = (size_t)idx * P.CP1 + (size_t)w * P.Lc;
    double *cpp = P.cp_p + base;
    if (defer) {
        defer->dl = dl; defer->dl2 = dl2; defer->tp = tp; defer->A = A; defer->base = base; defer->nck = nck;
    } else {
        // (eight independent read-modify-writes in flight: a plain loop waits for every load)
        for (int c0 = 0; c0 < nck; c0 += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = c0 + k < nck ? cpp[c0 + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + k < nck) cpp[c0 + k] = v[k] + ((A + (long long)(c0 + k) * P.R >= tp) ? dl2 : dl);
        }
    }
    if (w == P.W - 1) cpp[nck] += dl2;  // (a tie step, if any, lies before the end of the leg)
    if (__builtin_fabs(dl) * 256.0 > P.marg[li]) P.risk[li] = 1;  // not overwhelmingly inside the margin: verified in every batch
    P.pend[li] += dl2;
    if (P.clm_w[li] >= 0) P.clm_r[li] += (P.clm_w[li] >= tp) ? dl2 : dl;
    P.marg[li] -= __builtin_fabs(dl) + 2.220446049250313e-16;
    if (flip) P.tdir[li] = (int8_t)-td;  // the shifted trajectory resolved that tie the other way
}

// k_walk_carr: one lane per (leg, slot).  A leg is walked from its ANCHOR -- the last wrap event at or before
// its first sample, (omega, r): "the phase before global sample omega is r", or the chain root -- first up to
// the leg start (no output, epoch by epoch because the step changes), then through the leg itself, emitting
// the chunk checkpoints.  It reports the last wrap it saw (its CLAIM) or none.  Anchoring at wraps is what
// makes the speculation robust: right after a wrap every phase is a multiple of 2^-52, so the difference
// between a guessed and the true trajectory survives every later rounding unchanged, whereas a mid-cycle
// phase (finer grid) would be re-rounded at each binade crossing.  first != 0: the anchor is the last wrap
// predicted by ideal arithmetic (k_carr_guess / ideal_last_wrap).
__global__ void k_walk_carr(DevPlan P, int first)
{
    GAL_WALK_SETPRIO();  // latency-bound: win issue arbitration against a co-running k_synth
    if (first) {
        // the batch's counters start here (no memset, no kernel in front of the chain: rounds 1-4 had k_carr_guess do it);
        // UNVERIFIED != 0 makes the stitch behind this pass work.  Nothing else in a first pass reads or writes them (the legs it
        // walks -- every active one -- are counted on the host)
        if (blockIdx.x == 0 && threadIdx.x < CTR_COUNT) P.ctr[threadIdx.x] = threadIdx.x == CTR_UNVERIFIED ? 1 : 0;
    } else if (P.ctr[CTR_UNVERIFIED] == 0) {
        return;  // converged: remaining enqueued passes are no-ops
    }
    // a wave = 64 consecutive legs of ONE slot: similar Doppler, hence similar trip counts (a wave runs as long
    // as its slowest lane), idle slots are whole waves that leave at once, and the leg arrays are read coalesced
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.LEGS * P.S) return;
    const int s = t / P.LEGS;
    const int i = t - s * P.LEGS;
    const int e = i / P.W, w = i - e * P.W;
    const int idx = e * P.S + s;
    if (P.prn[idx] <= 0) return;
    const size_t li = (size_t)s * P.LEGS + i;
    const int L = P.Lc * P.R;
    const long long A = (long long)e * P.N + (long long)w * L;  // global index of the leg's first sample
    long long cur;
    double p;
    if (first) {
        // first pass: anchor from ideal arithmetic (predicted last wrap at or before the leg start)
        int om;
        double rr;
        if (ideal_last_wrap(P.pguess[idx], eff_step(P.dstep[idx]), w * L, &om, &rr)) {  // (guess arrays: epoch-major, from the host)
            cur = (long long)e * P.N + om;
            p = rr;
        } else {
            cur = P.gss_w[idx];
            p = P.gss_r[idx];
        }
#ifdef GAL_TEST_HOOKS
        if (P.hook_spoil && li == GAL_HOOK_BAD_LEG) {
            // an anchor that is not a wrap EVENT: one sample further along the same trajectory.  The stitch re-anchors the leg at the
            // true event and pass two walks it again -- its claim was right all along, so nothing behind it moves (what a natural
            // misprediction of a wrap's sample index looks like; an anchor off the trajectory would poison one successor per pass)
            p = carr_step(p, P.dstep[(int)(cur / P.N) * P.S + s]);
            cur += 1;
        }
#endif
        P.anc_w[li] = cur;
        P.anc_r[li] = p;
        P.verified[li] = 0;
    } else {
        if (!P.dirty[li]) return;
        cur = P.anc_w[li];
        p = P.anc_r[li];
    }
    long long lw = -1;
    double lr = 0.0;
    double mg = 4.0;
    int tdir = 0;
    long long tpos = -1;
    while (cur < A) {  // anchor -> leg start
        const int ec = (int)(cur / P.N);
        long long seg_end = (long long)(ec + 1) * P.N;
        seg_end = seg_end > A ? A : seg_end;
        const int n = (int)(seg_end - cur);
        const double d = P.dstep[ec * P.S + s];
        if (d == 0.0) {
            // a carrier that stands still: `p += 0; p -= (long)p` leaves the phase alone (it is never -0.0: gal_synth_plan), no wrap, no
            // binade crossing.  A channel without Doppler has its anchor at the chain root, so every leg of it comes through here once
            // per epoch in front of it (M-SYN12 with one such channel: 1 ms of walker chain, one dependent load per epoch): the
            // still epochs behind this one are skipped eight at a time
            long long e2 = (long long)ec + 1;
            const long long eA = A / P.N;  // the epoch the leg lies in
            while (e2 + 8 <= eA) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = P.dstep[(size_t)(e2 + q) * P.S + s];
                bool still = true;
#pragma unroll
                for (int q = 0; q < 8; ++q) still = still && v[q] == 0.0;
                if (!still) break;
                e2 += 8;
            }
            const long long skip_to = e2 * P.N;
            cur = skip_to < A ? skip_to : A;
            continue;
        }
        const WalkOut o = carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, n, n, [](int, double) {});
        if (o.last_w >= 0) {
            lw = cur + o.last_w;
            lr = o.last_r;
        }
        mg = o.margin < mg ? o.margin : mg;
        if (!tdir && o.tdir) {
            tdir = o.tdir;
            tpos = cur + o.tpos;
        }
        p = o.p;
        cur = seg_end;
    }
    int n = P.N - w * L;
    n = n > L ? L : n;
    const double d = P.dstep[idx];
    double *cpp = P.cp_p + (size_t)idx * P.CP1 + (size_t)w * P.Lc;
    // (epochs in front of the executed range are walked without checkpoints: one stretch, nobody reads them)
    const WalkOut o = e < P.cp_e0 ? carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, n, n, [](int, double) {})
                                  : carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, P.R, 0, [&](int c, double v) { cpp[c] = v; });
    if (o.last_w >= 0) {
        lw = A + o.last_w;
        lr = o.last_r;
    }
    if (w == P.W - 1) P.cp_p[(size_t)idx * P.CP1 + P.nchunks] = o.p;
    P.pend[li] = o.p;
    P.clm_w[li] = lw;  // -1: no wrap between the anchor and the end of the leg
    P.clm_r[li] = lr;
    P.marg[li] = o.margin < mg ? o.margin : mg;
    if (!tdir && o.tdir) {
        tdir = o.tdir;
        tpos = A + o.tpos;
    }
    P.tdir[li] = (int8_t)tdir;
    P.tpos[li] = tpos;
    P.dirty[li] = 0;
    P.risk[li] = 0;  // walked, not translated
    if (!first) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(true);  // one atomic per wave
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) atomicAdd(&P.ctr[CTR_WALKS], __builtin_popcountll(m));
    }
}

// k_verify_carr: legs of the executed epochs walked once more, genuinely and in closed form, from their own first checkpoint:
// each checkpoint of the leg and the state it hands to the next leg must come out bit for bit (CTR_MISMATCH otherwise, which sends
// gal_synth_finish into the all-walked fallback).  This is what k_synth's exact replay establishes on its way; k_synth_g
// (synth_group.hip) never forms the exact phase, so batches that run it get this kernel on the walker stream, beside the synthesis.
// WHICH legs: EVERY leg of the executed epochs in every batch (ver_mod = 1) -- the default since round 6 (ADVICE r5 / VERDICT r5 item 3:
// nearly every leg is accepted by TRANSLATION, DESIGN.md section 3 -- a proof, with the leg's binade margin as its hypothesis -- and a
// product that re-checks a proof only now and then can hand out wrong samples with chain_mismatch == 0 should the proof ever fail).
// GAL_CFG_VERIFY_SAMPLED (opt-in; round 5's default): (a) the legs i = ver_rem (mod ver_mod) -- an eighth per batch, rotating with
// the handle's batch count, so every leg position is re-walked every eighth batch; (b) in EVERY batch the legs whose translation
// used more than 1/256 of their margin (P.risk; typical: 2^-26 of it) -- first nsel threads: the rotation's legs (compact: a wave =
// 64 selected legs of one slot); the threads behind them: one per eight legs, for (b).  What full verification costs (round 5, same
// box, profiles/r05f_verify_ab.log): the pipelined step 0.979 -> 1.010 ms, a single handle's 1.238 -> 1.297.
__device__ __forceinline__ void verify_carr_leg(const DevPlan &P, const int s, const int i)
{
    const int e = i / P.W, w = i - e * P.W;
    if (e < P.cp_e0) return;  // walked silently: no checkpoints
    const int idx = e * P.S + s;
    if (P.prn[idx] <= 0) return;
    const int L = P.Lc * P.R;
    int n = P.N - w * L;
    n = n > L ? L : n;
    if (n <= 0) return;
    const double d = P.dstep[idx];
    const double *cpp = P.cp_p + (size_t)idx * P.CP1 + (size_t)w * P.Lc;
    int bad = 0;
    const WalkOut o = carr_walk_track(cpp[0], d, 1.0 / __builtin_fabs(d), n, P.R, 0, [&](int c, double v) { bad += cpp[c] != v; });
    bad += cpp[(n + P.R - 1) / P.R] != o.p;  // the next leg's first checkpoint, or the end-of-epoch state
    if (bad) atomicAdd(&P.ctr[CTR_MISMATCH], bad);
}

__global__ void k_verify_carr(DevPlan P)
{
    if (P.ctr[CTR_UNVERIFIED] != 0) return;  // chain not complete: gal_synth_finish iterates and launches this again
    const int mod = P.ver_mod > 1 ? P.ver_mod : 1;
    const int per_slot = (P.LEGS + mod - 1) / mod;
    const int nsel = per_slot * P.S;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nsel) {
        const int s = t / per_slot;
        const int i = (t - s * per_slot) * mod + P.ver_rem;
        if (i < P.LEGS) verify_carr_leg(P, s, i);
        return;
    }
    // (b), sampled mode only: eight legs' flags per thread; a set one is next to never seen, and EVERY set one is re-walked (round 5
    // re-walked the first of the eight only: ADVICE r5)
    t -= nsel;
    const int total = P.LEGS * P.S;
    if (mod == 1 || t * 8 >= total) return;
#pragma unroll 1
    for (int q = 0; q < 8 && t * 8 + q < total; ++q) {
        const int li = t * 8 + q;
        if (!P.risk[li] || li % P.LEGS % mod == P.ver_rem) continue;
        const int s = li / P.LEGS;
        verify_carr_leg(P, s, li - s * P.LEGS);
    }
}

// k_verify_code: the same for the CODE chain (round 6; ADVICE r5: on the k_synth_g family its checkpoints were only checked where
// k_repair_g happened to walk, ~0.5 % of them per batch).  One lane per (slot, epoch, code leg) of the executed epochs -- k_walk_code's
// layout --, the leg walked once more from its own first checkpoint, genuinely and in closed form (code_walk: no speculation, no
// translation): every checkpoint of the leg, the state it hands to the next leg (or the end-of-epoch state) and -- leg 0 -- the epoch's
// host-given start state must come out bit for bit, symbol counter and page-flip flag included.  By induction from the epoch's
// start every code checkpoint of the batch is then what stepping src/galileo-sdr.cpp:491-507,528 sample by sample produces.
// Sampled mode (GAL_CFG_VERIFY_SAMPLED): the legs (epoch * Wc + leg) = ver_rem (mod ver_mod).
__global__ void k_verify_code(DevPlan P)
{
    const int Wc = P.Wc;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.E * P.S * Wc) return;
    const int k = t & (Wc - 1);
    const int se = t / Wc;
    const int s = se / P.E;
    const int e = se - s * P.E;
    if (e < P.cp_e0) return;  // epochs in front of the executed range carry no checkpoints
    const int idx = e * P.S + s;
    if (P.prn[idx] <= 0) return;
    const int mod = P.ver_mod > 1 ? P.ver_mod : 1;
    if (mod > 1 && (e * Wc + k) % mod != P.ver_rem) return;
    const int Lk = P.Lkc;
    const int n0 = k * Lk * P.R;
    if (n0 >= P.N) return;
    int n1 = (k + 1) * Lk * P.R;
    n1 = n1 > P.N ? P.N : n1;
    const double *cpx = P.cp_x + (size_t)idx * P.CP1 + (size_t)k * Lk;
    const uint32_t *cpi = P.cp_ib + (size_t)idx * P.CP1 + (size_t)k * Lk;
    const double c = P.cstep[idx];
    const uint32_t w0 = cpi[0];
    const int fl0 = (int)(w0 >> 16);
    int bad = 0;
    if (k == 0) bad += d2u(cpx[0]) != d2u(P.x0[idx]) || w0 != (uint32_t)P.ib0[idx];
    const CodeEnd end = code_walk(cpx[0], (int)(w0 & 0xffffu), c, 1.0 / c, n1 - n0, P.R, [&](int ci, double x, int ib, int fl) {
        bad += d2u(cpx[ci]) != d2u(x) || cpi[ci] != ((uint32_t)ib | ((uint32_t)(fl | fl0) << 16));
    });
    const int nck = (n1 - n0 + P.R - 1) / P.R;  // the next leg's first checkpoint, or the end-of-epoch state
    bad += d2u(cpx[nck]) != d2u(end.x) || cpi[nck] != ((uint32_t)end.ibit | ((uint32_t)(end.flipped | fl0) << 16));
    if (n1 >= P.N) bad += P.flip_in[idx] != (uint8_t)(end.flipped | fl0);  // what k_pages reads
    if (bad) atomicAdd(&P.ctr[CTR_MISMATCH], bad);
}

// k_scanm stitches the legs of a slot.  Sequential statement (what its three phases compute; walk_host.cpp::galwalk_spec_wrap
// runs the same statement on the host):
//     chain state: last claim (lc_w, lc_r), its pending correction D, allok
//     for each leg i:   root      -> (lc_w, lc_r) = (first sample, given phase), D = 0, allok = true
//                       link_ok   =  anchor_i bitwise == (lc_w, lc_r)  and the leg was walked from it
//                       allok    &=  link_ok ;   verified_i = allok
//                       new anchor = (lc_w, lc_r + D)            (predicted true value of that claim)
//                       t         =  new anchor - old anchor      (0 if it is a different wrap event)
//                       leg saw a wrap -> (lc_w, lc_r) = its claim, D = tie_flip(t)   (else inherited)
// A leg is accepted only through bitwise equality with the verified chain; the prediction (rounded-add chains
// commute with shifts by multiples of 2^-52 while the itinerary is unchanged, up to the tie flip) decides how
// many passes are needed, and -- for legs accepted by translation -- rests on nco_walk.h: binade_margin, with
// k_synth's replay check behind it.
struct ClaimState {
    int kind;  // 0 nothing yet, 1 defined, 2 chain broken (idle epoch)
    long long w;
    double r;
};

struct LegRec {
    bool act, root, dirty, hw;
    int tdir;
    long long A, aw, cw;
    double ar, cr, known;
};

__device__ __forceinline__ LegRec leg_load(const DevPlan &P, int s, int i, double start0)
{
    LegRec L;
    const int e = i / P.W, w = i - e * P.W;
    const int idx = e * P.S + s;
    const size_t li = (size_t)s * P.LEGS + i;
    L.act = P.prn[idx] > 0;
    const uint32_t fl = P.flags[idx];
    L.root = L.act && w == 0 && (e == 0 || (fl & GAL_CH_RESTART));
    L.known = (fl & GAL_CH_RESTART) ? P.p0[idx] : start0;
    L.A = (long long)e * P.N + (long long)w * (P.Lc * P.R);
    L.aw = P.anc_w[li];
    L.ar = P.anc_r[li];
    L.cw = P.clm_w[li];
    L.cr = P.clm_r[li];
    L.hw = L.act && L.cw >= 0;
    L.tdir = P.tdir[li];
    L.dirty = L.act && P.dirty[li] != 0;
    return L;
}

// The pending correction D travels along the chain as   D <- D + c[(D / 2^-52) mod 4]   per wrap-bearing leg:
// a leg that was walked from an anchor G below the claim in front of it is off by t = G + D, a multiple of
// 2^-52, all the way -- except that an ODD t flips the first tie its walk met (nco_walk.h: WalkOut::tdir),
// after which it is off by t - tdir * 2^-52.  Such maps (plus "reset to a constant") are closed under
// composition, so the fold over many legs is a 4-entry table and the block scan reproduces the sequential
// statement exactly.
#define GAL_U52 2.220446049250313e-16  // 2^-52

__device__ __forceinline__ int d_residue(double D) { return (int)((long long)(D * 4503599627370496.0) & 3LL); }

__device__ __forceinline__ double tie_flip(double t, int tdir)
{
    return (tdir != 0 && (d_residue(t) & 1)) ? t - (double)tdir * GAL_U52 : t;
}

struct DMap {
    int isconst;  // the legs reset D: result = K whatever came in
    double K;
    double c[4];  // else result = D + c[residue(D)]
};

// (bit masks, not c[r] and not a chain of selects, which the compiler turns back into c[r]: a register array indexed by a lane's
// value lives in scratch memory, and these maps sit on the latency path of the stitch kernels)
__device__ __forceinline__ double dmap_pick(const DMap &f, int r)
{
    const long long b0 = __double_as_longlong(f.c[0]), b1 = __double_as_longlong(f.c[1]), b2 = __double_as_longlong(f.c[2]),
                    b3 = __double_as_longlong(f.c[3]);
    const long long lo = (r & 1) ? b1 : b0, hi = (r & 1) ? b3 : b2;
    const long long m = -(long long)((r >> 1) & 1);
    return __longlong_as_double((hi & m) | (lo & ~m));
}

__device__ __forceinline__ double dmap_apply(const DMap &f, double D) { return f.isconst ? f.K : D + dmap_pick(f, d_residue(D)); }

__device__ __forceinline__ DMap dmap_combine(const DMap &a, const DMap &b)  // a then b
{
    if (b.isconst) return b;
    DMap r;
    if (a.isconst) {
        r.isconst = 1;
        r.K = dmap_apply(b, a.K);
        r.c[0] = r.c[1] = r.c[2] = r.c[3] = 0.0;
        return r;
    }
    r.isconst = 0;
    r.K = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const double mid = (double)m * GAL_U52 + a.c[m];
        r.c[m] = a.c[m] + dmap_pick(b, d_residue(mid));
    }
    return r;
}

// What one leg does to the chain, given the claim state `lc` in front of it (exact, from sweep 1).
struct LegOp {
    bool act, root, have, link_ok, hw, same;
    int tdir;
    long long nw;  // event the leg should be anchored at
    double base;   // its residual before the pending correction is added
    double G;      // gap: claim residual minus the anchor residual that was walked (same event only)
};

__device__ __forceinline__ LegOp leg_op(const DevPlan &P, int s, const LegRec &L, ClaimState &lc)
{
    LegOp o;
    o.act = L.act;
    o.root = L.root;
    o.have = false; o.link_ok = false; o.hw = false; o.same = false; o.tdir = L.tdir;
    o.nw = 0; o.base = 0.0; o.G = 0.0;
    if (!L.act) {
        lc.kind = 2;
        return o;
    }
    if (L.root) {
        lc.kind = 1;
        lc.w = L.A;
        lc.r = L.known;
    }
    o.have = lc.kind == 1;
    o.link_ok = o.have && !L.dirty && L.aw == lc.w && d2u(L.ar) == d2u(lc.r);
    o.nw = lc.w;
    o.base = lc.r;
    o.same = o.have && L.aw == lc.w;
    o.G = o.same ? lc.r - L.ar : 0.0;
    o.hw = L.hw;
    if (L.hw) {
        lc.w = L.cw;
        lc.r = L.cr;
    }
    return o;
}

// D after the leg, given D before it (the sequential statement's "D = D_leg" / inheritance / resets)
__device__ __forceinline__ double leg_d_out(const LegOp &o, double D)
{
    if (!o.act) return 0.0;
    if (o.root) D = 0.0;
    if (!o.hw) return D;
    if (!o.same) return 0.0;
    return tie_flip(o.G + D, o.tdir);
}

// What the stitch does to ONE leg once the true carries in front of it are known (phase 3 of k_scanm): verified, or re-anchored at the predicted true value `nr` of the claim in front of it -- and then
// either TRANSLATED on the spot (same wrap event, only its residual moved, and the move is provably itinerary-preserving:
// translate_leg) or marked for another walk.  (Round 2 left the translations to the next walker pass: one more kernel and
// a skipped stitch behind it in the chain k_synth waits for.)
__device__ __forceinline__ void stitch_apply_leg(const DevPlan &P, const int s, const int i, const LegRec &L, const LegOp &o,
                                                 const int allok, const double nr, int &unver, int &rewalk, int &shifts,
                                                 TrRec *defer = nullptr)
{
    const size_t li = (size_t)s * P.LEGS + i;
    if (allok) {
        P.verified[li] = 1;
        return;
    }
    ++unver;
    if (o.have && (L.aw != o.nw || d2u(L.ar) != d2u(nr))) {
        const double dl = nr - L.ar;  // both residuals are multiples of 2^-52: exact
        const int el = i / P.W;
        const bool tr = P.translate && el >= P.tr_e0 && el < P.tr_e1 && L.aw == o.nw &&
                        __builtin_fabs(dl) + 8.881784197001252e-16 /* 2^-50 */ < P.marg[li];
        P.anc_w[li] = o.nw;
        P.anc_r[li] = nr;
        P.dirty[li] = tr ? 0 : 1;
        if (tr) {
            translate_leg(P, s, i, dl, defer);
            ++shifts;
        }
        rewalk += tr ? 0 : 1;
    }
    rewalk += o.have ? 0 : 1;
}

// End of a stitch: the LAST block to get here (a ticket; every block has fenced its writes before taking one) publishes
// the count the next pass looks at and the end-of-batch carrier phase.  When the stitcher only TRANSLATED (no leg has to
// be walked again), the translated claims are by construction the anchors it predicted for their successors: the chain is
// complete (what stands behind this is the same argument as for a single translated leg, and k_synth's replay check).
__device__ __forceinline__ void stitch_publish(const DevPlan &P, const int t, const int nblocks, const int unver, const int rewalk,
                                               const int shifts, int *s_last)
{
    if (t == 0) {
        if (unver) atomicAdd(&P.ctr[CTR_UNVER_NEXT], unver);
        if (rewalk) atomicAdd(&P.ctr[CTR_REWALK_NEXT], rewalk);
        if (shifts) atomicAdd(&P.ctr[CTR_SHIFTS], shifts);
        __threadfence();
        *s_last = atomicAdd(&P.ctr[CTR_TICKET], 1) == nblocks - 1;
    }
    __syncthreads();
    if (!*s_last) return;
    __threadfence();
    if (t == 0) {
        const int unv = atomicAdd(&P.ctr[CTR_UNVER_NEXT], 0), rw = atomicAdd(&P.ctr[CTR_REWALK_NEXT], 0);
        P.ctr[CTR_UNVERIFIED] = rw == 0 ? 0 : unv;
        P.ctr[CTR_UNVER_NEXT] = 0;
        P.ctr[CTR_REWALK_NEXT] = 0;
        P.ctr[CTR_TICKET] = 0;
        P.ctr[CTR_PASSES] += 1;
    }
    if (t < P.S) {  // end-of-batch carrier phase per slot (final once the chain is verified; rewritten by later passes)
        const int prn = P.prn[(P.E - 1) * P.S + t];
        // (device-scope load: the leg may have been translated by a block on another CU a moment ago)
        const double pe = __hip_atomic_load(&P.pend[(size_t)t * P.LEGS + (P.LEGS - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        P.state_out[t].carr_phase = prn > 0 ? pe : 0.0;
    }
}

// ---- The stitch, in 256-thread blocks, one leg per thread.  (Rounds 2-4 had a 1024-thread block per slot for batches up to 512
// epochs, k_carr_scan: 16 waves and 72 KB of LDS on ONE CU -- beside a running synthesis kernel that means waiting for a CU to
// drain completely -- with 10-step scans through LDS; stitch_ab.sh (a tool of rounds 3-5: git history), one handle: 0.189 -> 0.179 ms per step at 1 epoch,
// 0.451 -> 0.419 at 128, 0.880 -> 0.731 at 512 with this kernel in its place.)  The legs of a slot are spread over B blocks (one
// wave per SIMD: such a block starts as soon as ONE synthesis block retires); a batch of up to 32 epochs has B = 1 and none of what
// follows.  What a single block would do with two block-wide scans is done with block-local scans plus a LOOK-BACK over the blocks
// in front, inside ONE launch (rounds 2-4, long batches: three launches -- claims, fold, apply -- with the block totals handed
// over through kernel boundaries).  A block publishes the aggregate of its legs as
// soon as its local scan is done; then its threads fetch the aggregates of ALL blocks in front of it, one record per thread (each
// waits until that record carries this launch's tag), and the waves fold them in order -- both the claim chain ("the last one
// that speaks") and the fold (segmented AND + D map) are associative, so this is the sequential statement.  No block waits for
// another's PREFIX, only for aggregates, which every block publishes before it waits for anything: no chain of waits.  Which
// legs a block takes is decided by a TICKET it draws when it starts, not by blockIdx: a block only ever waits for blocks with a
// lower ticket, and those are running or done -- no assumption about the order in which the hardware dispatches a grid over the
// XCDs.  Records carry the launch's tag, so nothing has to be cleared between passes.
//   What it costs (scanm_stamps.py (a tool of rounds 3-5: git history), 1199 epochs: 38 blocks per slot, 608 in all, 80 us alone on the device): the two
// exchanges are 15-25 us each -- an agent-scope store and the loads that wait for it cross the fabric between the XCDs, 2-3 us
// a trip and four trips per exchange -- which is what the two kernel boundaries cost before; the checkpoint shifts at the end
// are 17 us (80 MB read + written); the scans themselves 1 + 4 us.
#define SCANM_THREADS 256

struct ScanM {  // look-back records of the multi-block stitch: [S][Bs] each
    int B;                                            // blocks per slot of THIS launch (the executed prefix's legs)
    int Bs;                                           // record stride = blocks per slot of the PLAN: the sub-arrays keep their places whatever
                                                      // range of the plan a launch covers (laid out from the cut plan, a second
                                                      // range moved the status words over former payload words that nobody clears)
    uint32_t tag;                                     // of this launch (non-zero)
    int nap;                                          // 64-cycle naps between two looks at a record that is not there yet
    int lpb;                                          // legs per block: SCANM_THREADS (GAL_TEST_HOOKS: fewer, so that small batches have many blocks)
    uint32_t *cnt;                                    // [S] tickets drawn so far (B per launch and slot)
    uint32_t *st1, *st2;                              // tag of the launch whose claim / fold aggregate the record holds
    int *a1_kind; long long *a1_w; double *a1_r;      // claims: the block's last speaker
    int *a2_f; double *a2_K, *a2_c;                   // fold: f = fv | v << 1 | isconst << 2; c: [4]
};

struct FoldRec {
    int fv, v;
    DMap m;
};

__device__ __forceinline__ FoldRec fold_identity()
{
    FoldRec r;
    r.fv = 0; r.v = 1;
    r.m.isconst = 0; r.m.K = 0.0; r.m.c[0] = r.m.c[1] = r.m.c[2] = r.m.c[3] = 0.0;
    return r;
}

__device__ __forceinline__ FoldRec fold_combine(const FoldRec &a, const FoldRec &b)  // a then b
{
    FoldRec r;
    r.fv = a.fv | b.fv;
    r.v = b.fv ? b.v : (a.v & b.v);
    r.m = dmap_combine(a.m, b.m);
    return r;
}

template <class T>
__device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Every word of a record is written and read with agent-scope atomics (sc1: past the XCD's own L2), so record traffic needs no
// cache maintenance -- an agent-scope FENCE here would write back / invalidate the XCD's whole L2 in every wave of every block
// (measured: the fused kernel slower than the three launches it replaces).  What is needed is order: payload before tag on the
// writer's side, tag before payload on the reader's -- the memory counter drained, nothing moved across by the compiler.
#define SCANM_ORDER() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

#ifdef GAL_TEST_HOOKS
// where k_scanm's time goes (scanm_stamps.py (a tool of rounds 3-5: git history)): the 100 MHz wall clock at nine points of every block
#define SCANM_NSTAMP 9
#define SCANM_STAMP_BLOCKS 4096
__device__ unsigned long long g_scanm_stamp[SCANM_STAMP_BLOCKS * SCANM_NSTAMP];  // [block][stage]: plain stores, nothing shared
#define SCANM_STAMP(i)                                                                             \
    do {                                                                                           \
        const unsigned lin_ = blockIdx.y * gridDim.x + blockIdx.x;                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  /* (pins the stamp: nothing in flight across it) */ \
        if (threadIdx.x == 0 && lin_ < SCANM_STAMP_BLOCKS) g_scanm_stamp[lin_ * SCANM_NSTAMP + (i)] = wall_clock64(); \
        asm volatile("" ::: "memory");                                                             \
    } while (0)
#else
#define SCANM_STAMP(i) do { } while (0)
#endif

// wait until record `o` carries this launch's tag (the block that writes it drew its ticket before the one that waits: it is
// running or done)
__device__ __forceinline__ void scanm_wait(const uint32_t *st, size_t o, uint32_t tag, int nap)
{
    while (ld_agent(st + o) != tag)
        for (int n = 0; n < nap; ++n) __builtin_amdgcn_s_sleep(1);
    SCANM_ORDER();
}

// Wave-wide inclusive scans of the two chains by lane shuffles (six steps, no LDS round trips, no barriers), and the step across
// a block's four waves: each wave's total through LDS, one barrier.  This kernel runs once per batch, every CU executes its code
// cold, and scanm_stamps.py (a tool of rounds 3-5: git history) shows the time going where code is executed for the FIRST time, whatever it does (a block
// with nothing in front of it spends 11 us in a look-back of two barriers when the look-back's code is new, 1 us when the
// block-local scan has already fetched it): what counts is the number of instruction-cache lines on the path.  So the scans
// are real functions (__noinline__), called for the block's own legs and again for the aggregates of the blocks in front, and
// their loops stay rolled.
__device__ __forceinline__ ClaimState claim_shfl_up(const ClaimState &c, int off)
{
    ClaimState a;
    a.kind = __shfl_up(c.kind, off);
    a.w = __shfl_up(c.w, off);
    a.r = __shfl_up(c.r, off);
    return a;
}

__device__ __forceinline__ ClaimState claim_wave_scan(ClaimState c, int lane)  // "the last one that speaks", inclusive
{
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) {
        const ClaimState a = claim_shfl_up(c, off);
        if (lane >= off && c.kind == 0) c = a;
    }
    return c;
}

__device__ __forceinline__ FoldRec fold_shfl_up(const FoldRec &r, int off)
{
    FoldRec a;
    const int f = __shfl_up(r.fv | (r.v << 1) | (r.m.isconst << 2), off);
    a.fv = f & 1; a.v = (f >> 1) & 1; a.m.isconst = (f >> 2) & 1;
    a.m.K = __shfl_up(r.m.K, off);
#pragma unroll
    for (int m = 0; m < 4; ++m) a.m.c[m] = __shfl_up(r.m.c[m], off);
    return a;
}

__device__ __forceinline__ FoldRec fold_wave_scan(FoldRec r, int lane)
{
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) {
        const FoldRec a = fold_shfl_up(r, off);
        if (lane >= off) r = fold_combine(a, r);
    }
    return r;
}

__global__ __launch_bounds__(SCANM_THREADS) void k_scanm(DevPlan P, ScanM M)
{
    GAL_WALK_SETPRIO();
    if (P.ctr[CTR_UNVERIFIED] == 0) return;
    constexpr int NW = SCANM_THREADS / 64;
    __shared__ ClaimState s_ct[NW];   // the waves' totals, claim chain / fold
    __shared__ FoldRec s_ft[NW];
    __shared__ ClaimState s_carry1;   // what the blocks in front amount to
    __shared__ FoldRec s_carry2;
    __shared__ int s_unver, s_rewalk, s_shifts, s_last, s_ticket;
    const int s = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (M.B > 1) {
        if (t == 0) s_ticket = (int)atomicAdd(&M.cnt[s], 1u);  // (back to 0 at the end of the launch: below)
        __syncthreads();
    }
    const int b = M.B > 1 ? s_ticket : 0;
    const int i = b * M.lpb + t;  // my leg
    const bool in = t < M.lpb && i < P.LEGS;
    const size_t ob = (size_t)s * M.Bs + b;
    if (t == 0) {
        s_unver = 0;
        s_rewalk = 0;
        s_shifts = 0;
    }
    SCANM_STAMP(0);
    const double start0 = P.state_in[s].carr_phase;
    LegRec L;
    L.act = false; L.root = false; L.dirty = false; L.hw = false; L.tdir = 0; L.A = 0; L.aw = 0; L.cw = -1; L.ar = 0.0; L.cr = 0.0; L.known = 0.0;
    if (in) L = leg_load(P, s, i, start0);
    SCANM_STAMP(1);

    // ---- phase 1: the claim chain.  My leg's word, the block's "last one that speaks" scan, the block's aggregate out, the
    // claim state in front of the block in
    ClaimState lc = {0, 0, 0.0};  // (becomes: the claim state in front of MY leg)
    {
        ClaimState mine = {0, 0, 0.0};
        if (in) {
            if (!L.act) {
                mine.kind = 2;
            } else {
                if (L.root) { mine.kind = 1; mine.w = L.A; mine.r = L.known; }
                if (L.hw) { mine.kind = 1; mine.w = L.cw; mine.r = L.cr; }
            }
        }
        const ClaimState inc = claim_wave_scan(mine, lane);
        if (lane == 63) s_ct[wv] = inc;
        lc = claim_shfl_up(inc, 1);
        if (lane == 0) lc.kind = 0;
    }
    __syncthreads();
    {
        ClaimState wp = {0, 0, 0.0};  // the waves in front of mine, within the block
#pragma unroll 1
        for (int w2 = 0; w2 < NW; ++w2)
            if (w2 < wv && s_ct[w2].kind != 0) wp = s_ct[w2];
        if (lc.kind == 0) lc = wp;
    }
    SCANM_STAMP(2);
    if (t == 0 && b + 1 < M.B) {  // the block's aggregate out, for the blocks behind
        ClaimState agg = {0, 0, 0.0};
#pragma unroll 1
        for (int w2 = 0; w2 < NW; ++w2)
            if (s_ct[w2].kind != 0) agg = s_ct[w2];
        st_agent(M.a1_kind + ob, agg.kind);
        st_agent(M.a1_w + ob, agg.w);
        st_agent(M.a1_r + ob, agg.r);
        SCANM_ORDER();
        st_agent(M.st1 + ob, M.tag);
    }
    {
        // ... and the aggregates of the blocks in front in: one record per thread, scanned by the waves as above
        ClaimState carry = {0, 0, 0.0};
        for (int base = 0; base < b; base += SCANM_THREADS) {
            __syncthreads();  // (s_ct: everybody has read the previous round's totals)
            const int q = base + t;
            ClaimState rec = {0, 0, 0.0};
            if (q < b) {
                const size_t oq = (size_t)s * M.Bs + q;
                scanm_wait(M.st1, oq, M.tag, M.nap);
                rec.kind = ld_agent(M.a1_kind + oq);
                rec.w = ld_agent(M.a1_w + oq);
                rec.r = ld_agent(M.a1_r + oq);
            }
            rec = claim_wave_scan(rec, lane);
            if (lane == 63) s_ct[wv] = rec;
            __syncthreads();
            if (t == 0) {
#pragma unroll 1
                for (int w2 = 0; w2 < NW; ++w2)
                    if (s_ct[w2].kind != 0) carry = s_ct[w2];
            }
        }
        if (t == 0) s_carry1 = carry;
    }
    __syncthreads();
    SCANM_STAMP(3);
    if (lc.kind == 0) lc = s_carry1;

    // ---- phase 2: the fold (segmented AND of the links, D map evaluated on the four residues)
    LegOp o;
    o.act = false; o.root = false; o.have = false; o.link_ok = false; o.hw = false; o.same = false; o.tdir = 0; o.nw = 0; o.base = 0.0; o.G = 0.0;
    FoldRec pre;  // (becomes: the fold of everything in front of MY leg)
    {
        FoldRec f = fold_identity();
        if (in) {
            ClaimState lcc = lc;
            o = leg_op(P, s, L, lcc);
            int allok = 1, fv = 0;
            if (!o.act) {
                allok = 0;
                fv = 1;
            } else {
                if (o.root) fv = 1;
                allok = o.link_ok ? 1 : 0;
            }
            double D4[4] = {0.0, GAL_U52, 2.0 * GAL_U52, 3.0 * GAL_U52};
#pragma unroll
            for (int m = 0; m < 4; ++m) D4[m] = leg_d_out(o, D4[m]);
            f.fv = fv;
            f.v = allok;
            f.m.isconst = (!o.act || o.root || (o.hw && !o.same)) ? 1 : 0;
            f.m.K = D4[0];
#pragma unroll
            for (int m = 0; m < 4; ++m) f.m.c[m] = D4[m] - (double)m * GAL_U52;
        }
        const FoldRec inc = fold_wave_scan(f, lane);
        if (lane == 63) s_ft[wv] = inc;
        pre = fold_shfl_up(inc, 1);
        if (lane == 0) pre = fold_identity();
    }
    __syncthreads();
    {
        FoldRec wp = fold_identity();
#pragma unroll 1
        for (int w2 = 0; w2 < NW; ++w2)
            if (w2 < wv) wp = fold_combine(wp, s_ft[w2]);
        pre = fold_combine(wp, pre);
    }
    SCANM_STAMP(4);
    if (t == 0 && b + 1 < M.B) {
        FoldRec agg = s_ft[0];
#pragma unroll 1
        for (int w2 = 1; w2 < NW; ++w2) agg = fold_combine(agg, s_ft[w2]);
        st_agent(M.a2_f + ob, agg.fv | (agg.v << 1) | (agg.m.isconst << 2));
        st_agent(M.a2_K + ob, agg.m.K);
#pragma unroll
        for (int m = 0; m < 4; ++m) st_agent(M.a2_c + ob * 4 + m, agg.m.c[m]);
        SCANM_ORDER();
        st_agent(M.st2 + ob, M.tag);
    }
    {
        FoldRec carry = fold_identity();
        for (int base = 0; base < b; base += SCANM_THREADS) {
            __syncthreads();
            const int q = base + t;
            FoldRec rec = fold_identity();
            if (q < b) {
                const size_t oq = (size_t)s * M.Bs + q;
                scanm_wait(M.st2, oq, M.tag, M.nap);
                const int fl = ld_agent(M.a2_f + oq);
                rec.fv = fl & 1; rec.v = (fl >> 1) & 1; rec.m.isconst = (fl >> 2) & 1;
                rec.m.K = ld_agent(M.a2_K + oq);
#pragma unroll
                for (int m = 0; m < 4; ++m) rec.m.c[m] = ld_agent(M.a2_c + oq * 4 + m);
            }
            rec = fold_wave_scan(rec, lane);
            if (lane == 63) s_ft[wv] = rec;
            __syncthreads();
            if (t == 0) {
#pragma unroll 1
                for (int w2 = 0; w2 < NW; ++w2) carry = fold_combine(carry, s_ft[w2]);
            }
        }
        if (t == 0) s_carry2 = carry;
    }
    __syncthreads();

    SCANM_STAMP(5);
    // ---- phase 3: my leg with the true carries, translations on the spot
    pre = fold_combine(s_carry2, pre);
    int allok = pre.fv ? pre.v : 0;                          // nothing is verified before the first root
    double D = pre.m.isconst ? pre.m.K : pre.m.c[0];         // the prefix map applied to D = 0
    int unver = 0, rewalk = 0, shifts = 0;
    TrRec tr = {0.0, 0.0, 0, 0, 0, 0};
    if (in) {
        if (!o.act) {
            allok = 0;
        } else {
            if (o.root) {
                allok = 1;
                D = 0.0;
            }
            allok &= o.link_ok ? 1 : 0;
            const double nr = o.base + D;
            stitch_apply_leg(P, s, i, L, o, allok, nr, unver, rewalk, shifts, &tr);
        }
    }
    SCANM_STAMP(6);
    if (unver) atomicAdd(&s_unver, unver);
    if (rewalk) atomicAdd(&s_rewalk, rewalk);
    if (shifts) atomicAdd(&s_shifts, shifts);
    // The checkpoint shifts of the block's translated legs, done by the whole block: thread t takes checkpoint t % 32 (+ 32,
    // + 64 ...) of leg t / 32 (+ 8, + 16 ...), so a wave touches two runs of 256 contiguous bytes per access instead of 64
    // cache lines (one leg per lane, as the walker kernel did it in round 2: 16x the traffic, 77 us of the chain)
    __shared__ double s_dl[SCANM_THREADS], s_dl2[SCANM_THREADS];
    __shared__ long long s_tp[SCANM_THREADS], s_A[SCANM_THREADS];
    __shared__ size_t s_base[SCANM_THREADS];
    __shared__ int s_nck[SCANM_THREADS];
    s_dl[t] = tr.dl; s_dl2[t] = tr.dl2; s_tp[t] = tr.tp; s_A[t] = tr.A; s_base[t] = tr.base; s_nck[t] = tr.nck;
    __syncthreads();
    for (int c0 = 0; c0 < P.Lc; c0 += 32) {
        const int c = c0 + (t & 31);
        for (int l0 = 0; l0 < SCANM_THREADS; l0 += 64) {  // 8 legs per step, 8 steps in flight
            double v[8];
            bool on[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int leg = l0 + 8 * k + (t >> 5);
                on[k] = c < s_nck[leg];
                v[k] = on[k] ? P.cp_p[s_base[leg] + c] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int leg = l0 + 8 * k + (t >> 5);
                if (on[k]) P.cp_p[s_base[leg] + c] = v[k] + ((s_A[leg] + (long long)c * P.R >= s_tp[leg]) ? s_dl2[leg] : s_dl[leg]);
            }
        }
    }
    __syncthreads();
    SCANM_STAMP(7);
    stitch_publish(P, t, (int)(gridDim.x * gridDim.y), s_unver, s_rewalk, s_shifts, &s_last);
    if (s_last && t < P.S) M.cnt[t] = 0;  // (the last block through: every block has drawn its ticket)
    SCANM_STAMP(8);
}

// Page in force at the start of each epoch, src/galileo-sdr.cpp:497-506 + src/channel.cpp:88: the page
// changes only at a (re)allocation or when the symbol counter wrapped inside the previous epoch.
// `src` encodes where the page comes from:  -1: state_in,  2*e: page_next of epoch e,  2*e+1: page_init of epoch e.
// Every active epoch e leaves ONE event for the epochs after it -- 2e if its counter wrapped, else 2e+1 if it was
// (re)allocated -- so the page of epoch e is page_init(e) if it restarts, else the last event before it: a block-wide
// "last one that speaks" scan, one epoch per thread (round 1 walked the epochs serially in one wave: 0.13 ms on the
// critical path of a lone handle).
__global__ __launch_bounds__(GUESS_THREADS) void k_pages(DevPlan P)
{
    GAL_WALK_SETPRIO();  // latency-bound: win issue arbitration against a co-running k_synth
    __shared__ int s_ev[GUESS_THREADS];
    const int s = blockIdx.x;
    const int t = threadIdx.x;
    int carry = -1;  // block-uniform: last event of the tiles before
    for (int base = 0; base < P.E; base += GUESS_THREADS) {
        const int e = base + t;
        const bool in = e < P.E;
        const int idx = (in ? e : 0) * P.S + s;
        const int prn = in ? P.prn[idx] : 0;
        const bool restart = prn > 0 && (P.flags[idx] & GAL_CH_RESTART);
        const bool flip = prn > 0 && P.flip_in[idx] != 0;
        const int ev = flip ? 2 * e : (restart ? 2 * e + 1 : -2);  // -2: says nothing
        s_ev[t] = ev;
        __syncthreads();
        for (int off = 1; off < GUESS_THREADS; off <<= 1) {
            int v2 = -2;
            const bool take = t >= off && s_ev[t] == -2;
            if (take) v2 = s_ev[t - off];
            __syncthreads();
            if (take) s_ev[t] = v2;
            __syncthreads();
        }
        int before = carry;
        if (t > 0 && s_ev[t - 1] != -2) before = s_ev[t - 1];
        const int mine = restart ? 2 * e + 1 : before;
        if (in && prn > 0) {
            const uint32_t *src = mine < 0 ? P.state_in[s].page
                                  : (mine & 1) ? P.page_init + (size_t)P.init_ix[(size_t)(mine >> 1) * P.S + s] * GAL_PAGE_WORDS
                                               : P.page_next + ((size_t)(mine >> 1) * P.S + s) * GAL_PAGE_WORDS;
            uint32_t *dst = P.page_cur + (size_t)idx * GAL_PAGE_WORDS;
            uint32_t w[GAL_PAGE_WORDS];
#pragma unroll
            for (int k = 0; k < GAL_PAGE_WORDS; ++k) w[k] = src[k];
#pragma unroll
            for (int k = 0; k < GAL_PAGE_WORDS; ++k) dst[k] = w[k];
        }
        const int last = s_ev[GUESS_THREADS - 1];
        __syncthreads();
        if (last != -2) carry = last;
    }
    // end-of-batch state for the next call
    if (t < GAL_PAGE_WORDS) {
        const uint32_t *src = carry < 0 ? P.state_in[s].page
                              : (carry & 1) ? P.page_init + (size_t)P.init_ix[(size_t)(carry >> 1) * P.S + s] * GAL_PAGE_WORDS
                                            : P.page_next + ((size_t)(carry >> 1) * P.S + s) * GAL_PAGE_WORDS;
        P.state_out[s].page[t] = src[t];
    }
    if (t == 0) {
        const int prn = P.prn[(P.E - 1) * P.S + s];
        P.state_out[s].prn = prn > 0 ? prn : 0;
        P.state_out[s].reserved = 0;
    }
}

#endif  // GAL_TU_WALK

#if GAL_TU_SYNTH
// ------------------------------------------------------------------------------------------------
// Hot kernel.  Block = 256 threads = 4 waves = 4 tiles of 64 chunks of ONE epoch, so every per-epoch
// constant (NCO steps, active PRNs) is wave-uniform and lives in SGPRs.
//
// A lane replays its chunk in GROUPS of 16 samples.  Inside a group the per-sample work of every channel
// is BRANCH-FREE, so the 16 x 4 channel-steps of a part form one basic block and the compiler interleaves
// the channels' dependency chains (FP64 add/cvt latency and the LDS LUT read are hidden by ILP instead of
// by occupancy).  What used to be rare branches is hoisted to the group boundaries:
//   * chips: 16 samples advance the code by < 16 BOC half chips (plan() checks f_code/fs <= 0.5), so the
//     group prologue cuts a 16-half-chip window W (2 bits per half chip: "v != 0" and "v < 0") out of the
//     channel's half-chip stream in LDS and XORs the data/secondary signs of the current symbol onto it;
//     a sample then needs one shift-add and one bit-field extract to get its table selector.
//   * code wrap (x >= 4092, src/galileo-sdr.cpp:491-507): a wave-uniform test at the group start
//     (any lane, any of the part's channels within 16 samples of the wrap?) picks between a FAST group body
//     with no wrap handling at all and a SLOW one (`x -= ge ? 4092 : 0`, window spliced from the end of
//     this code period and the start of the next with the next symbol's signs; the symbol counter is
//     advanced in the group epilogue).  With the default chunk size (a divisor of the code period) the wraps
//     of all 64 lanes fall into the same group, so ~9 of 10 groups are fast.
// Per-lane persistent state per channel: y = 2x, p (FP64) and one packed word
//     st = ibit[8:0] | use_next_page[9] | sg[11:10] | sg_next[13:12] | (sg * 0x55)[23:16],  sg = (data^sec) | sec<<1.
// LDS: [NCH][512] half-chip stream words + 2 x 1024-entry carrier LUT (plain for positive, conjugate for negative
// Doppler), entries = the int16 PAIR (2 cos, 2 sin).  A sample's contribution is
//     (I, Q) += (2 cos, 2 sin) * v',   v' in {-1, 0, +1}  =  the SIGNED 2-bit window field of its half chip,
// ONE v_pk_mad_u16 (both halves multiplied by the low half of v'): the sign and the zero case of
// v = E1B d - E1C s (:520-525) cost nothing, and the accumulator IS the little-endian int16 I,Q word of the output
// (:536-537) -- exact while |I|, |Q| < 32768, i.e. up to 65 channels of amplitude 500.
#ifndef SYN_WAVES
#define SYN_WAVES 3  // waves per SIMD the register allocation aims at (LDS allows 4 blocks per CU)
#endif


// sign bits of symbol `ibit`: bit0 = data ^ secondary, bit1 = secondary (1 => factor -1);
// data symbol = page bit, secondary = CS25[ibit % 25] (src/galileo-sdr.cpp:517-518)
__device__ __forceinline__ uint32_t sym_signs(const DevPlan *Pd, int idx, int ibit, int use_next)
{
    const uint32_t *pg = (use_next ? Pd->page_next : Pd->page_cur) + (size_t)idx * GAL_PAGE_WORDS;
    const uint32_t dbit = (pg[ibit >> 5] >> (ibit & 31)) & 1u;
    const uint32_t sbit = (Pd->cs25 >> (ibit % 25)) & 1u;
    return (dbit ^ sbit) | (sbit << 1);
}

// st for symbol (ibit, nx) with the signs of it and of its successor
__device__ __forceinline__ uint32_t sym_state(const DevPlan *Pd, int idx, int ibit, int nx)
{
    if (ibit >= GAL_N_SYM_PAGE) {  // :497-506: next page
        ibit = 0;
        nx = 1;
    }
    int nib = ibit + 1, nnx = nx;
    if (nib >= GAL_N_SYM_PAGE) {
        nib = 0;
        nnx = 1;
    }
    const uint32_t sg = sym_signs(Pd, idx, ibit, nx);
    const uint32_t sgn = sym_signs(Pd, idx, nib, nnx);
    // byte 2: the sign pair of the current symbol on four half chips (sg * 0x55) -- one v_perm_b32 makes the 16-half-chip XOR
    // mask of it at every group start (GAL_SIGN_MASK_ST) instead of a field extract, a 24-bit multiply and a shift-or
    // (same-box A/B, four alternations: 1.233 -> 1.225 ms per pipelined step)
    return (uint32_t)ibit | ((uint32_t)nx << 9) | (sg << 10) | (sgn << 12) | ((sg * 0x55u) << 16);
}

struct ChanState {
    double y;     // TWICE the code phase, i.e. in BOC half chips (pre wrap-check).  Doubling is exact in binary
                  // floating point and commutes with rounding, so y_n == 2*x_n bit for bit when the step is
                  // doubled too; (int)y is then the reference's icode = (int)(code_phase*2) (:512) for free.
    double p;     // carrier phase, cycles, MIRRORED: the reference's carr_phase times the sign of this epoch's
                  // step, so that it is >= 0 whenever phase and step agree in sign (always, except during the
                  // first cycle after a Doppler sign change or a restart)
    uint32_t st;  // packed symbol state, see above
};

struct ChanGroup {  // live only inside one 16-sample group
    // 16-half-chip window, half chip ic0 + j at bits 2j (v != 0) and 2j+1 (v < 0), signs applied.  The
    // stream (synth_api.cpp) stores bit 2h = E1B^E1C chip, bit 2h+1 = E1C chip ^ (h & 1) for half chip h:
    // v = E1B*d - E1C*s is non-zero iff B^C^d^s, negative iff C^s (given non-zero), and the BOC(1,1)
    // sub-carrier boc[2c] = -chip, boc[2c+1] = +chip (src/gal-sig.cpp:198-213) flips the sign on odd h.
    uint32_t W;
    int m;       // bit offset of a sample's half chip in W = (2*icode + m) & 31; slow groups add 16 at the code
                 // wrap (2 * 8184 = 16 mod 32) and start from an EVEN multiple of 16 ... see group_begin_slow
    int mw;      // slow groups: value of m once the wrap has been taken (the epilogue compares)
};


template <int J>
__device__ __forceinline__ void group_begin_fast(const ChanState &c, ChanGroup &g, const uint32_t *s_str)
{
    const int ic0 = (int)c.y;  // y < 8184 - 16*cs2: no wrap before the group ends
    const uint32_t *wp = s_str + J * STR_PITCH + (ic0 >> 4);
    const uint32_t lo = wp[0], hi = wp[1];  // (the pad word makes wp[1] valid for the last word; ds_read2_b32)
    const uint32_t mask = GAL_SIGN_MASK_ST(c.st);
    g.W = window_signed(__builtin_amdgcn_alignbit(hi, lo, (uint32_t)ic0 << 1) ^ mask);  // v_alignbit uses shift[4:0]
    g.m = -2 * ic0;
}

// RESAMPLED window (k_synth<.., RW = 1>).  Sample u of a fast group reads half chip ic0 + g(u), g(u) = floor(f + u s),
// f = frac(y) at the group start, s = the code step in half chips (0.74 <= s < 1 here): g advances by one per sample
// except at <= 4 HOLDS, where two samples share a half chip.  Instead of evaluating (int)y, a shift-add and a
// field extract per SAMPLE, the group start looks the hold positions up and spreads the window once,
//     X = window with field u = half chip of SAMPLE u:   X <- (X & ~M_d) | ((X << 2) & M_d),  M_d = ~0 << 2 u_d,
// after which a sample costs one v_bfe_i32 with a constant offset and the code NCO advances once per group
// (GAL_ADV).  The pattern (M_1..M_4) depends on f only through the 15 thresholds T_u = 1 - frac(u s) it lies between;
// the block prologue tabulates them per channel: s_bin[floor(128 f)] = (the one threshold near that bin, 16 x the
// number of thresholds below the bin), s_pat[id] = masks for `id` thresholds <= f.  EXACTNESS: the sample's true half
// chip is (int)y_u of the SEQUENTIAL y_u, which lies within 16 ulp(8192) = 2^-36 of the real line y_0 + u s; f and the
// thresholds are rounded to float (2^-25 each); whenever f is within RW_DELTA = 2^-22 of the threshold of its bin, or
// the bin is near two thresholds, `unsafe` is raised and the whole group runs the per-sample slow body instead.
struct RwTmp {  // one channel's group-start temporaries between the phases below
    int ic0;
    float f;
    uint2 be;
    uint32_t lo, hi;
    uint4 M;
    // CBOC only: the BOC(6,1) half-period parity pattern of the group
    float f6;
    uint2 be6;
    uint32_t p6;  // bit 2u+1: parity of the half-period index of sample u
    int i12;
    uint32_t x, hw;  // the spread window (fields (B != C, sign bit) of SAMPLE u) and the sign correction that does not
                     // depend on the half period: 1 ^ parity of sample u's half chip, at bit 2u+1
};

// The group start in three phases, each run for the four channels of a part before the next one starts, so that the
// part waits ONCE for each round of LDS reads instead of once per channel: (A) addresses, bin entry + stream words in
// flight; (B) threshold compare, pattern masks in flight, signed window; (C) the spread.
template <int J, int BINS = RW_BINS, int PITCH = RW_BIN_PITCH>
__device__ __forceinline__ void rw_phase_a(const ChanState &c, RwTmp &t, const uint2 *s_bin)
{
    t.ic0 = (int)c.y;  // y < 8184 - 16*cs2: no wrap before the group ends
    t.f = (float)__builtin_amdgcn_fract(c.y);
    const int bi = (int)(t.f * (float)BINS);
    t.be = s_bin[J * PITCH + bi];
}

// CBOC: sample u's BOC(6,1) half period is (int)(6 y_u) = i12_0 + floor(f6 + u 6s), f6 = frac(6 y_0): its PARITY over the
// 16 samples depends on f6 only through the 15 thresholds 1 - frac(u 6s) -- the same look-up as for the chip holds, on
// a second pair of tables (s_bin6, s_pat6).  6 y is formed the way the per-sample step forms it (one rounded product).
template <int J>
__device__ __forceinline__ void rw_phase_a6(const ChanState &c, RwTmp &t, const uint2 *s_bin6)
{
    const double y6 = 6.0 * c.y;
    t.i12 = (int)y6;
    t.f6 = (float)__builtin_amdgcn_fract(y6);
    const int bi = (int)(t.f6 * (float)CB_BINS);
    t.be6 = s_bin6[J * CB_BIN_PITCH + bi];
}

template <int J>
__device__ __forceinline__ bool rw_phase_b(RwTmp &t, const uint32_t str0, const uint4 *s_pat)
{
    const float thr = __uint_as_float(t.be.x);
    const uint32_t po = t.be.y + (t.f >= thr ? 16u : 0u);
    t.M = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(s_pat + J * 16) + po);
    // stream words ic0 / 16 and the next one of row J (str0: LDS byte address of the rows, in an SGPR -- written as
    // asm because the compiler otherwise re-materialises the row offset in a VGPR for every group)
    uint32_t wa;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(wa) : "v"(t.ic0 >> 4), "s"(str0 + (uint32_t)(J * STR_PITCH * 4)));
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
    t.lo = ((lds_u32_ptr)(uintptr_t)wa)[0];
    t.hi = ((lds_u32_ptr)(uintptr_t)wa)[1];
    return !(__builtin_fabsf(t.f - thr) >= RW_DELTA);  // too close to call (a NaN threshold = undecidable bin)
}

template <int J>
__device__ __forceinline__ bool rw_phase_b6(RwTmp &t, const uint32_t *s_pat6)
{
    const float thr = __uint_as_float(t.be6.x);
    const uint32_t po = t.be6.y + (t.f6 >= thr ? 4u : 0u);  // be6.y = 4 x (thresholds below the bin): byte offset of the word
    // (pattern: parity relative to sample 0's half period, whose own parity is added here)
    t.p6 = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_pat6 + J * 16) + po) ^
           (((uint32_t)t.i12 & 1u) ? 0xAAAAAAAAu : 0u);
    return !(__builtin_fabsf(t.f6 - thr) >= RW_DELTA);
}

// all 16 fields = field d of w (a two's-complement 2-bit value 00 / 01 / 11)
__device__ __forceinline__ uint32_t rw_rep(const uint32_t w, const int d)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)w, 2 * d, 1);      // 0 / ~0: the field's low bit
    const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)w, 2 * d + 1, 1);  // ... its high bit
    return gal_bfi(0x55555555u, lo, hi);
}


// MODE 1: code step 0.74 .. 1 half chips per sample -- the window advances every sample except at <= 4 holds (masks
// M_d = ~0 << 2 u_d): spread.  MODE 2: code step <= 2/15 (sample rates from 15.4 MS/s) -- the window advances at <= 2
// samples of the group (masks A_d = ~0 << 2 u_d): X = field 0 everywhere, field 1 from the first advance on, field 2
// from the second.  MODE 3: the same with <= 4 advances, code step <= 4/15 (7.7 .. 15.4 MS/s).
template <int MODE>
__device__ __forceinline__ uint32_t rw_phase_c_m(const uint32_t mask, const RwTmp &t);
template <int MODE>
__device__ __forceinline__ uint32_t rw_phase_c(const ChanState &c, const RwTmp &t)
{
    return rw_phase_c_m<MODE>(GAL_SIGN_MASK_ST(c.st), t);
}
// (the same with the XOR mask of the symbol's signs supplied by the caller)
template <int MODE>
__device__ __forceinline__ uint32_t rw_phase_c_m(const uint32_t mask, const RwTmp &t)
{
    uint32_t x = window_signed(__builtin_amdgcn_alignbit(t.hi, t.lo, (uint32_t)t.ic0 << 1) ^ mask);
    if constexpr (MODE == 2) {
        const uint32_t w = x;
        x = gal_bfi(t.M.x, rw_rep(w, 1), rw_rep(w, 0));
        x = gal_bfi(t.M.y, rw_rep(w, 2), x);
    } else if constexpr (MODE == 3) {  // <= 4 advances: code step <= 4/15 half chips (7.7 .. 15.4 MS/s)
        const uint32_t w = x;
        x = gal_bfi(t.M.x, rw_rep(w, 1), rw_rep(w, 0));
        x = gal_bfi(t.M.y, rw_rep(w, 2), x);
        x = gal_bfi(t.M.z, rw_rep(w, 3), x);
        x = gal_bfi(t.M.w, rw_rep(w, 4), x);
    } else {
        x = rw_spread(x, t.M);
    }
    return x;
}

// CBOC group start after the chip-hold look-up.  (C1) the raw window -- fields (B != C, sign of C x secondary) per half
// chip, symbol signs applied -- is spread like the BOC(1,1) one, so that field u belongs to SAMPLE u; the half chip of
// sample u is ic0 + u - (holds before u), and its parity comes out of the hold masks.  Then the second look-up (rw_phase_a6 /
// b6: parity of the BOC(6,1) half period) runs with only two words per channel still live, and (D) forms the two words the
// samples read -- XA: field u = the (B - C) term's factor of sample u, 0 / +1 / -1 as a two's-complement 2-bit value (what
// the BOC(1,1) path calls the signed window); XB: the same for the (B + C) term, whose sign is that of C x secondary x
// BOC(1,1) sub-carrier x BOC(6,1) sub-carrier, i.e. bit 1 ^ 1 ^ parity(half chip) ^ parity(half period) (chan_step_cboc).
__device__ __forceinline__ void rw_phase_c1_cboc(const ChanState &c, RwTmp &t)
{
    const uint32_t mask = GAL_SIGN_MASK_ST(c.st);
    uint32_t x = __builtin_amdgcn_alignbit(t.hi, t.lo, (uint32_t)t.ic0 << 1) ^ mask;
    x = rw_spread(x, t.M);
    t.x = x;
    // bit 2u+1 of hp: (holds before u) & 1; of 0x88888888: u & 1; their XOR with ic0's parity is the half chip's parity, and
    // 0xAAAAAAAA (the "^ 1") ^ 0x88888888 = 0x22222222
    const uint32_t hp = (t.M.x ^ t.M.y ^ t.M.z ^ t.M.w) & 0xAAAAAAAAu;
    t.hw = hp ^ 0x22222222u ^ (((uint32_t)t.ic0 & 1u) ? 0xAAAAAAAAu : 0u);
}

__device__ __forceinline__ void rw_phase_d_cboc(const RwTmp &t, uint32_t &xa, uint32_t &xb)
{
    xa = window_signed(t.x);
    const uint32_t sb = t.x ^ t.hw ^ t.p6;   // (odd bits: the sign of the (B + C) factor)
    const uint32_t lo = ~t.x & 0x55555555u;  // B == C: this term is the one that is non-zero
    xb = lo | (sb & (lo << 1));
}

template <int J>
__device__ __forceinline__ void group_begin_slow(const ChanState &c, ChanGroup &g, const uint32_t *s_str)
{
    const bool pend = c.y >= 8184.0;  // wrap pending: the first sample of the group takes it (:491-507)
    const double ye = pend ? c.y - 8184.0 : c.y;
    const int ic0 = (int)ye;
    const int w = ic0 >> 4;
    const uint32_t lo = s_str[J * STR_PITCH + w];
    const uint32_t hi = s_str[J * STR_PITCH + w + 1];
    const uint32_t head = s_str[J * STR_PITCH];
    const uint32_t sg_nxt = (c.st >> 12) & 3u;
    const uint32_t sg_cur = pend ? sg_nxt : ((c.st >> 10) & 3u);
    uint32_t W = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(ic0 & 15) << 1) ^ GAL_SIGN_MASK(sg_cur);
    // half chips from jw on belong to the next code period: stream start, next symbol's signs
    const int jw = 8184 - ic0;
    const bool splice = jw < 16;
    const uint32_t sh = (uint32_t)(2 * jw) & 31u;
    const uint32_t keep = splice ? ((1u << sh) - 1u) : ~0u;
    const uint32_t tail = splice ? ((head ^ GAL_SIGN_MASK(sg_nxt)) << sh) : 0u;
    g.W = window_signed((W & keep) | tail);
    // a pending wrap is taken by the first sample, which switches m to mw like any other wrap
    g.mw = -2 * ic0 + (pend ? 0 : 16);
    g.m = g.mw - 16;
}


// One sample of one channel, src/galileo-sdr.cpp:509-532, branch-free:  (I, Q) += v' * (2 cos, 2 sin).
// cs2 = 2 * f_code * delt.  lutb = LDS byte address of entry k = 0 of this channel's carrier table (plain table for
// positive, conjugate table for negative Doppler: the phase is kept MIRRORED, see ChanState, so the reference's
// k = (int)(511 carr_phase) is +-(int)(511 p)); each table holds entries k in (-512, 512) as LUT[k & 511] resp.
// LUT[-k & 511], which is the two's-complement mask of :509-510.
__device__ __forceinline__ void chan_step(ChanState &c, const ChanGroup &g, const double cs2, const double ds,
                                          const uint32_t lutb, int &acc)
{
    // --- chip lookup, :512-521: icode = (int)(2x); signed 2-bit field of that half chip
    const int ic = (int)c.y;
    int off;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(off) : "v"(ic), "v"(g.m));
    const int v = __builtin_amdgcn_sbfe((int)g.W, (uint32_t)off, 2);  // v_bfe_i32 uses offset[4:0]
    // --- carrier LUT, :509-510: trunc toward zero
    const int k = (int)(511.0 * c.p);
    uint32_t a;
    // (spelled in asm: the combiner otherwise re-associates the shift-add into shifts, masks and an add3)
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(k), "s"(lutb));
    const int t = *(const __attribute__((address_space(3))) int *)(uintptr_t)a;
    gal_acc(acc, t, v);
    // --- NCO updates, :528-532 (mirrored: IEEE addition and truncation are sign-symmetric)
    c.y = c.y + cs2;
    c.p = carr_step(c.p, __builtin_fabs(ds));
}

// The fast group body's version: no code wrap inside the group, and the mirrored phase is non-negative (phase and
// step have the same sign, checked by the caller), so that `p += d; p -= (long)p` (:531-532) becomes
// |p| = fract(|p| + |d|): for 0 <= x < 2, x - floor(x) is the reference's x - trunc(x) and the subtraction is
// exact, hence the same bits with one instruction less.
__device__ __forceinline__ void chan_step_fast(ChanState &c, const ChanGroup &g, const double cs2, const double ds,
                                               const uint32_t lutb, int &acc)
{
    const int ic = (int)c.y;
    int off;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(off) : "v"(ic), "v"(g.m));
    const int v = __builtin_amdgcn_sbfe((int)g.W, (uint32_t)off, 2);
    const int k = (int)(511.0 * c.p);
    uint32_t a;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(k), "s"(lutb));
    const int t = *(const __attribute__((address_space(3))) int *)(uintptr_t)a;
    gal_acc(acc, t, v);
    c.y = c.y + cs2;
    c.p = __builtin_amdgcn_fract(c.p + __builtin_fabs(ds));
}

// one sample of one channel in a resampled group: chip value from field U of X, carrier as in chan_step_fast
__device__ __forceinline__ void chan_step_rw(ChanState &c, const uint32_t X, const int U, const double ds,
                                             const uint32_t lutb, int &acc)
{
    const int v = __builtin_amdgcn_sbfe((int)X, (uint32_t)(2 * U), 2);
    const int k = (int)(511.0 * c.p);
    uint32_t a;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a) : "v"(k), "s"(lutb));
    const int t = *(const __attribute__((address_space(3))) int *)(uintptr_t)a;
    gal_acc(acc, t, v);
    c.p = __builtin_amdgcn_fract(c.p + __builtin_fabs(ds));
}

// one sample of one channel in a resampled CBOC group: both factors from constant fields, one 8-byte table entry (TA[k],
// TB[k]), two packed multiply-adds (exactly one of the factors is non-zero); carrier as in chan_step_fast
__device__ __forceinline__ void chan_step_rw_cboc(ChanState &c, const uint32_t XA, const uint32_t XB, const int U, const double ds,
                                                  const uint32_t lutb, int &acc)
{
    const int va = __builtin_amdgcn_sbfe((int)XA, (uint32_t)(2 * U), 2);
    const int vb = __builtin_amdgcn_sbfe((int)XB, (uint32_t)(2 * U), 2);
    const int k = (int)(511.0 * c.p);
    uint32_t a;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(a) : "v"(k), "s"(lutb));
    typedef int gal_i2 __attribute__((ext_vector_type(2)));
    const gal_i2 tt = *(const __attribute__((address_space(3))) gal_i2 *)(uintptr_t)a;  // ds_read_b64
    gal_acc(acc, tt.x, va);
    gal_acc(acc, tt.y, vb);
    c.p = __builtin_amdgcn_fract(c.p + __builtin_fabs(ds));
}

// The same with the symbol advance of :491-507: x -= 4092 when x >= 4092 (subtracting +0.0 otherwise is exact);
// the symbol counter is advanced in the group epilogue.
__device__ __forceinline__ void chan_step_wrap(ChanState &c, ChanGroup &g, const double cs2, const double ds,
                                               const uint32_t lutb, int &acc)
{
    const bool ge = c.y >= 8184.0;
    c.y = c.y - (ge ? 8184.0 : 0.0);
    g.m = ge ? g.mw : g.m;
    chan_step(c, g, cs2, ds, lutb, acc);
}

__device__ __forceinline__ void group_end(ChanState &c, const ChanGroup &g, const DevPlan *Pd, const int idx)
{
    if (__builtin_expect(g.m == g.mw, 0)) {  // the code wrapped inside this group (once per 4 ms of signal)
        c.st = sym_state(Pd, idx, (int)(c.st & 0x1ffu) + 1, (int)((c.st >> 9) & 1u));
    }
}

// ---- CBOC(6,1,1/11), opt-in (GAL_CFG_CBOC).  The E1 OS ICD's composite sub-carrier
//     e_B (alpha sc_A + beta sc_B) - e_C (alpha sc_A - beta sc_B),   alpha = sqrt(10/11), beta = sqrt(1/11),
// discretised the way the reference discretises its BOC(1,1) (src/gal-sig.cpp:198-213: the first half of a
// sub-carrier period is negative): sc_A from (int)(2 x), sc_B from (int)(12 x).  With B = E1B chip x data and
// C = E1C chip x secondary, exactly one of (B - C), (B + C) is non-zero per sample, so a channel contributes
//     sc_A (B - C) TA[k]   or   sc_B (B + C) TB[k],      TA = lround(alpha LUT), TB = lround(beta LUT)
// -- integer arithmetic like the reference's (:520-525); the reference itself has no CBOC: include/galsynth.h
// (GAL_CFG_CBOC) carries the definition, the test suite's CPU checker restates it.  One plain per-sample loop, no windows:
// about 3.5x the instructions of the BOC(1,1) path.  lutb: LDS byte address of entry k = 0 of the channel's A table
// (plain or conjugate); the B tables follow 8 KB later.
// LAY = 1 (the resampled-window build of the mode): one 512-entry table of 8-byte entries (TA[k], TB[k]) per Doppler
// sign, indexed with k & 511 (plain: LUT[k & 511]; conjugate: entry j holds LUT[-j & 511], and (k & 511) = k mod 512).
template <int J, int LAY = 0>
__device__ __forceinline__ void chan_step_cboc(ChanState &c, const double cs2, const double ds, const uint32_t lutb,
                                               const uint32_t str0, const DevPlan *Pd, const int idx, int &acc)
{
    if (c.y >= 8184.0) {  // :491-507, checked before use
        c.y = c.y - 8184.0;
        c.st = sym_state(Pd, idx, (int)(c.st & 0x1ffu) + 1, (int)((c.st >> 9) & 1u));
    }
    const int h = (int)c.y;           // half chip = (int)(2 x)
    const int i12 = (int)(6.0 * c.y);  // BOC(6,1) half period = (int)(12 x): 6 y and 12 x are the same real number
    // (str0: LDS byte address of the stream rows)
    const uint32_t w = *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)(str0 + (uint32_t)(J * STR_PITCH + (h >> 4)) * 4u);
    // stream field of half chip h: bit0 = E1B ^ E1C chip, bit1 = E1C chip ^ (h & 1); sg = (data ^ sec) | sec << 1
    const uint32_t f = ((w >> ((h & 15) * 2)) ^ (c.st >> 10)) & 3u;
    const uint32_t nz = f & 1u;  // B != C: the (B - C) term, else the (B + C) term
    const uint32_t sign = nz ? (f >> 1) : ((f >> 1) ^ 1u ^ ((uint32_t)(h ^ i12) & 1u));
    const int k = (int)(511.0 * c.p);
    const uint32_t a = LAY == 1 ? lutb + (nz ? 0u : 4u) + (((uint32_t)k & 511u) << 3) : lutb + (nz ? 0u : 8192u) + (uint32_t)(k << 2);
    const int t = *(const __attribute__((address_space(3))) int *)(uintptr_t)a;
    gal_acc(acc, t, sign ? -1 : 1);
    c.y = c.y + cs2;
    c.p = carr_step(c.p, __builtin_fabs(ds));
}

// CBOC inside the group machinery of the BOC(1,1) path: a group that cannot reach the code wrap (same countdown) reads
// its half-chip fields from the 16-half-chip window, signs of the current symbol applied -- bit 0: B != C, i.e. the
// (B - C) term is the one that is non-zero; bit 1: the sign of C x secondary x BOC(1,1) sub-carrier -- and needs
// neither the wrap test nor the stream read of chan_step_cboc per sample.
template <int J>
__device__ __forceinline__ void group_begin_cboc(const ChanState &c, ChanGroup &g, const uint32_t *s_str)
{
    const int ic0 = (int)c.y;  // y < 8184 - 16*cs2: no wrap before the group ends
    const uint32_t *wp = s_str + J * STR_PITCH + (ic0 >> 4);
    const uint32_t lo = wp[0], hi = wp[1];
    g.W = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)ic0 << 1) ^ GAL_SIGN_MASK_ST(c.st);
    g.m = -2 * ic0;
}

__device__ __forceinline__ void chan_step_cboc_fast(ChanState &c, const ChanGroup &g, const double cs2, const double ds,
                                                    const uint32_t lutb, int &acc)
{
    const int h = (int)c.y;
    int off;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(off) : "v"(h), "v"(g.m));
    const uint32_t f = __builtin_amdgcn_ubfe(g.W, (uint32_t)off, 2);
    const int i12 = (int)(6.0 * c.y);
    // branch-free: bit 0 of f = 1: (B - C) term, sign = bit 1; = 0: (B + C) term, sign = bit 1 ^ 1 ^ parity(h ^ i12)
    uint32_t par = (uint32_t)(h ^ i12);
    asm("" : "+v"(par));  // (keeps the compiler from predicating the (int)(6y) path on bit 0: exec-mask juggling per sample)
    const uint32_t flip = ~(f | par) & 1u;
    const int v = 1 - 2 * (int)((f >> 1) ^ flip);
    const int k = (int)(511.0 * c.p);
    const uint32_t a = lutb + ((~f & 1u) << 13) + (uint32_t)(k << 2);  // B tables 8 KB behind the A tables
    const int t = *(const __attribute__((address_space(3))) int *)(uintptr_t)a;
    gal_acc(acc, t, v);
    c.y = c.y + cs2;
    c.p = __builtin_amdgcn_fract(c.p + __builtin_fabs(ds));  // mirrored phase non-negative in a fast group (GAL_SAFE)
}


#define GAL_CH_LIST(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11)
#define GAL_MAX_NCH 12
// ACC: add onto samples already in `iq` (second and later channel groups when > 12 channels are active)
// SIG: 0 = BOC(1,1) as the reference generates it, 1 = CBOC(6,1,1/11) (see chan_step_cboc)
// RW: 1, 2 = fast groups take their 16 chip values from a RESAMPLED window (rw_phase_a/b/c) instead of indexing the
//     window per sample; 1 needs 0.74 <= 2 f_code / fs < 1 on every channel of the batch (2.6 MS/s), 2 needs
//     2 f_code / fs <= 0.133 (15.4 MS/s and above: config 4's 25 MS/s); the host decides (DevPlan::rw)
template <int NCH, bool ACC, int SIG = 0, int RW = 0>
__global__ __launch_bounds__(SYN_BLOCK) __attribute__((amdgpu_waves_per_eu(SYN_WAVES))) void k_synth(const DevPlan *__restrict__ Pd, SynGeom G,
                                                     const uint8_t *__restrict__ act_all,
                                                     const int *__restrict__ nact_all, uint32_t *__restrict__ iq)
{
    static_assert(NCH <= GAL_MAX_NCH, "extend GAL_CH_LIST");
    __shared__ uint32_t s_str[NCH * STR_PITCH];
    // entry k + 512 of table 0: LUT[k & 511], of table 1: LUT[-k & 511]; CBOC: the same pair for TA, then for TB
    constexpr bool CBRW = SIG == 1 && RW != 0;  // CBOC on resampled windows: 8-byte (TA, TB) entries, k >= 0 tables only
    constexpr int LUT_TABLES = CBRW ? 1 : SIG == 1 ? 4 : 2;
    __shared__ int s_lut[LUT_TABLES * 1024 * (CBRW ? 2 : 1)];
    // RW: per channel the hold patterns of a 16-sample group (see rw_phase_a)
    constexpr int BINS = CBRW ? CB_BINS : RW_BINS, BPITCH = CBRW ? CB_BIN_PITCH : RW_BIN_PITCH;
    __shared__ uint2 s_bin[RW ? NCH * BPITCH : 1];
    __shared__ uint4 s_pat[RW ? NCH * 16 : 1];
    __shared__ float s_thr[RW ? NCH * 16 : 1];
    // CBOC: the same for the BOC(6,1) half-period parity (rw_phase_a6)
    __shared__ uint2 s_bin6[CBRW ? NCH * CB_BIN_PITCH : 1];
    __shared__ uint32_t s_pat6[CBRW ? NCH * 16 : 1];
    __shared__ float s_thr6[CBRW ? NCH * 16 : 1];
    // CBOC on resampled windows: the code steps (used once per 16-sample group) live here instead of in 24 VGPRs -- this
    // instantiation needs two spread words per channel and 8-byte table reads, and spilled inside the group loop without
    __shared__ double s_csl[CBRW ? GAL_MAX_NCH : 1];
    __shared__ double s_tie[GAL_MAX_NCH];
    __shared__ int s_rwbad;  // RW: a channel's group has more holds / advances than the pattern masks hold (the host's
                             // gate excludes it; if it happens all the same, every group of the block runs the slow body)

    const int er = blockIdx.x / G.blocks_per_epoch;  // epoch relative to the executed range
    const int tg = blockIdx.x - er * G.blocks_per_epoch;
    const int e = G.e0 + er;
    const int tid = threadIdx.x;

    // Everything the block needs before its sample loop is fetched in a few WIDE phases (all loads of a phase are
    // independent and in flight together): with one load per wait the ~60 dependent round trips to L2 / HBM of this
    // prologue cost ~40 us per block, 5 % of its run time.
    // ---- phase 0 (scalar): plan pointers, the epoch's active list (16-byte rows, zero-padded) and channel records
    const int *const p_lut = Pd->lut;
    const int *const p_prn = Pd->prn;
    const uint32_t *const p_str = Pd->str;
    const double *const p_cpx = Pd->cp_x, *const p_cpp = Pd->cp_p;
    const uint32_t *const p_cpi = Pd->cp_ib;
    const double *const p_cstep = Pd->cstep, *const p_dstep = Pd->dstep;
    const uint32_t *const p_pcur = Pd->page_cur, *const p_pnext = Pd->page_next;
    const uint32_t cs25 = Pd->cs25;
    const int nact = __builtin_amdgcn_readfirstlane(nact_all[e]);
    if (ACC && nact == 0) return;  // nothing to add in this epoch (an accumulating launch behind k_synth_g: most epochs of most batches)
    const uint4 aw = *reinterpret_cast<const uint4 *>(act_all + (size_t)e * GAL_ACT_ROW);
    const uint32_t awv[4] = {aw.x, aw.y, aw.z, aw.w};
    // slot index e * S + act[j] of position j (idle positions alias slot act[0]: loads stay in bounds, results unused)
#define GAL_IX(j) const int ix##j = __builtin_amdgcn_readfirstlane(e * G.S + (int)((awv[(j) >> 2] >> (8 * ((j) & 3))) & 0xffu));
    GAL_CH_LIST(GAL_IX)
#undef GAL_IX
    int ixs[GAL_MAX_NCH];
#define GAL_IXS(j) ixs[j] = ix##j;
    GAL_CH_LIST(GAL_IXS)
#undef GAL_IXS

    // ---- phase 1: LDS tables.  Per thread 2 stream words per channel + 8 LUT entries, all loads first, then the stores
    {
        uint32_t w0[NCH], w1[NCH];
        constexpr int LUT_PER_THREAD = (int)(sizeof(s_lut) / sizeof(int)) / SYN_BLOCK;
        int lv[LUT_PER_THREAD];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int prn = p_prn[ixs[j]];
            prn = prn < 1 ? 1 : prn;  // idle position: any valid row, zeroed below
            const uint32_t *src = p_str + (size_t)(prn - 1) * STR_WORDS;
            w0[j] = src[tid];
            w1[j] = src[tid + SYN_BLOCK];
        }
#pragma unroll
        for (int q = 0; q < LUT_PER_THREAD; ++q) {
            const int i = tid + q * SYN_BLOCK;
            if constexpr (CBRW) {  // int i = (table (plain / conjugate) x 512 + k) x 2 + (0: TA, 1: TB)
                const int k = (i >> 1) & 511;
                lv[q] = p_lut[((i & 1) << 9) + (((i >> 10) ? -k : k) & 511)];
            } else {
                const int k = (i & 1023) - 512;
                lv[q] = p_lut[((i >> 11) << 9) + ((((i >> 10) & 1) ? -k : k) & 511)];
            }
        }
        // idle positions (epochs with fewer than NCH active channels) run the same branch-free group code on an
        // all-zero state and an all-zero stream: window 0 -> every field 0 -> no contribution; steps 0 keep the state
        // at rest
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const bool on = j < nact;
            s_str[j * STR_PITCH + tid] = on ? w0[j] : 0u;
            s_str[j * STR_PITCH + tid + SYN_BLOCK] = on ? w1[j] : 0u;
            if (tid == 0) s_str[j * STR_PITCH + STR_WORDS] = on ? w0[j] : 0u;  // pad word = word 0
        }
#pragma unroll
        for (int q = 0; q < LUT_PER_THREAD; ++q) s_lut[tid + q * SYN_BLOCK] = lv[q];
    }
    [[maybe_unused]] double rw_s[NCH];  // RW: the channels' code steps in half chips (wave-uniform)
    if constexpr (RW) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) rw_s[j] = j < nact ? uniform_f64(2.0 * p_cstep[ixs[j]]) : 0.0;
    }
    // step of channel jj, jj a per-thread value (a select chain over the uniform values: no LDS round trip)
    auto rw_step_of = [&](const int jj) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < NCH; ++j) v = jj == j ? rw_s[j] : v;
        return v;
    };
    if constexpr (RW) {
        // ---- hold patterns, step A: the 15 thresholds T_u = 1 - frac(u s) of each channel, sorted: thread
        // (channel, u) ranks its own; thread (channel, 16) writes the sentinel and the tie binade of the code step
        if (tid == 0) s_rwbad = 0;
        if constexpr (CBRW) {
            if (tid < NCH) s_csl[tid] = rw_step_of(tid);
        }
        if (tid < NCH * 16) {
            const int j = tid >> 4, u = (tid & 15) + 1;
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
                // y + s rounds to the grid q = 2^(e-52) of y's binade e; with M the 53-bit significand of s and
                // es its exponent, s / q = M 2^(es-e) has the fractional part 1/2 -- a tie, whose rounding direction
                // depends on y -- exactly in the binade e = es + 1 + ctz(M).  Everywhere else fl(y + s) = y + RN_q(s).
                double tl = 1e300;
                if (s > 0.0) {
                    const uint64_t b = d2u(s);
                    const uint64_t M = (b & 0xfffffffffffffULL) | 0x10000000000000ULL;
                    const int es = (int)((b >> 52) & 0x7ff);
                    tl = u2d((uint64_t)(es + 1 + __builtin_ctzll(M)) << 52);
                }
                s_tie[j] = tl;
            }
            if constexpr (CBRW) {  // the same ranking for the BOC(6,1) half periods: step 6 s per sample
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
    }
    __syncthreads();
    if constexpr (RW) {
        // ---- step B: the bin tables (NCH x 129 entries over all threads) and the 16 patterns of each channel
        for (int t = tid; t < NCH * (BINS + 1); t += SYN_BLOCK) {
            const int j = t / (BINS + 1), b = t - j * (BINS + 1);
            const float lo = (float)b * (1.0f / BINS) - RW_EDGE, hi = (float)(b + 1) * (1.0f / BINS) + RW_EDGE;
            int cnt = 0, idb = 0;
            float thr = 4.0f;  // "no threshold near this bin": never reached, never close
            for (int i = 0; i < 15; ++i) {
                const float th = s_thr[j * 16 + i];
                idb += th < lo;
                const bool in = th >= lo && th < hi;
                cnt += in;
                thr = in ? th : thr;
            }
            if (cnt >= 2 || b == BINS) thr = __builtin_nanf("");  // undecidable here: the group runs the slow body
            if (j >= nact) { thr = 4.0f; idb = 0; }
            s_bin[j * BPITCH + b] = make_uint2(__float_as_uint(thr), (uint32_t)idb * 16u);
        }
        if constexpr (CBRW) {
            for (int t = tid; t < NCH * (CB_BINS + 1); t += SYN_BLOCK) {
                const int j = t / (CB_BINS + 1), b = t - j * (CB_BINS + 1);
                const float lo = (float)b * (1.0f / CB_BINS) - RW_EDGE, hi = (float)(b + 1) * (1.0f / CB_BINS) + RW_EDGE;
                int cnt = 0, idb = 0;
                float thr = 4.0f;
                for (int i = 0; i < 15; ++i) {
                    const float th = s_thr6[j * 16 + i];
                    idb += th < lo;
                    const bool in = th >= lo && th < hi;
                    cnt += in;
                    thr = in ? th : thr;
                }
                if (cnt >= 2 || b == CB_BINS) thr = __builtin_nanf("");
                if (j >= nact) { thr = 4.0f; idb = 0; }
                s_bin6[j * CB_BIN_PITCH + b] = make_uint2(__float_as_uint(thr), (uint32_t)idb * 4u);
            }
            if (tid < NCH * 16) {
                // pattern `id` (id thresholds <= f6): bit 2u+1 = parity of floor(f6 + u 6s), the number of half periods
                // sample u lies beyond sample 0's
                const int j = tid >> 4, id = tid & 15;
                const double s6 = 6.0 * rw_step_of(j);
                const double Tlo = id ? (double)s_thr6[j * 16 + id - 1] : 0.0;
                double Thi = (double)s_thr6[j * 16 + id];
                Thi = Thi > 1.0 ? 1.0 : Thi;
                const double f = 0.5 * (Tlo + Thi);
                uint32_t w = 0u;
                for (int u = 1; u <= 15; ++u)
                    if ((long long)__builtin_floor(f + (double)u * s6) & 1LL) w |= 2u << (2 * u);
                s_pat6[tid] = j < nact ? w : 0u;
            }
        }
        if (tid < NCH * 16) {
            const int j = tid >> 4, id = tid & 15;  // id = number of thresholds <= f
            const double s = rw_step_of(j);
            const double Tlo = id ? (double)s_thr[j * 16 + id - 1] : 0.0;
            double Thi = (double)s_thr[j * 16 + id];
            Thi = Thi > 1.0 ? 1.0 : Thi;
            const double f = 0.5 * (Tlo + Thi);
            uint32_t m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
            int d = 0;
            double gp = 0.0;  // floor(f), f < 1
            for (int u = 1; u <= 15; ++u) {
                const double g = __builtin_floor(f + (double)u * s);
                if ((RW >= 2) ? (g != gp) : (g == gp)) {  // RW 1: sample u HOLDS the half chip of sample u - 1; 2, 3: ADVANCES
                    const uint32_t m = ~0u << (2 * u);
                    m0 = d == 0 ? m : m0; m1 = d == 1 ? m : m1; m2 = d == 2 ? m : m2; m3 = d == 3 ? m : m3;
                    ++d;
                }
                gp = g;
            }
            if (j >= nact) m0 = m1 = m2 = m3 = 0u;
            else if (d > (RW == 2 ? 2 : 4)) s_rwbad = 1;  // (form 1: 4 holds, form 2: 2 advances, form 3: 4 advances)
            s_pat[tid] = make_uint4(m0, m1, m2, m3);
        }
        __syncthreads();
    }
    [[maybe_unused]] const bool rw_off = RW != 0 && __builtin_amdgcn_readfirstlane(s_rwbad) != 0;

    // Position L of the epoch -> chunk c.  When the chunk length divides the code period (cls chunks per period),
    // the positions are ordered by CODE-PHASE CLASS c % cls first: all chunks of a class start at the same code phase,
    // so a wave (64 consecutive positions = 2-3 classes) meets a channel's code wrap in a group only if one of ITS
    // classes is the one that wraps in this chunk -- about one wave in four instead of every wave, and when it does,
    // many of its lanes need the slow body, not one in ten.
    const int L = (tg * (SYN_BLOCK / 64) + (tid >> 6)) * 64 + (tid & 63);
    if (L >= G.nchunks) return;
    const int c = (L % G.per) * G.cls + L / G.per;  // (identity for cls == 1)
    const int n0 = c * G.R;
    int nsteps = G.N - n0;
    if (nsteps > G.R) nsteps = G.R;

    // Per-channel state as individually named scalars (macro-expanded), NOT arrays: hipcc turns small
    // per-thread arrays into wide vector registers and copies whole tuples around every conditional update.
    // (The arrays below live only in this prologue, fully unrolled.)
#define GAL_HI(x) ((uint32_t)(d2u(x) >> 32))
#define GAL_MIRROR_BITS(p, ds) p = u2d(((uint64_t)(GAL_HI(p) ^ (GAL_HI(ds) & 0x80000000u)) << 32) | (uint32_t)d2u(p));
    double yv[NCH], pv[NCH], csv[NCH], dsv[NCH];
    uint32_t stv[NCH];
    {
        // ---- phase 2: the chunk's checkpoints (per lane) and the epoch's NCO steps (scalar)
        double cx[NCH];
        uint32_t cib[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const size_t cp = (size_t)ixs[j] * G.CP1 + c;
            cx[j] = p_cpx[cp];
            pv[j] = p_cpp[cp];
            cib[j] = p_cpi[cp];  // ibit | flipped << 16
            csv[j] = p_cstep[ixs[j]];
            dsv[j] = p_dstep[ixs[j]];
        }
        // ---- phase 3: page words of the current and of the next symbol (sym_state, branch-free)
        uint32_t wa[NCH], wb[NCH];
        int ib[NCH], nx[NCH], nib[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int b = (int)(cib[j] & 0xffffu), x = (int)(cib[j] >> 16);
            const bool over = b >= GAL_N_SYM_PAGE;  // :497-506: next page
            b = over ? 0 : b;
            x = over ? 1 : x;
            int nb = b + 1, nnx = x;
            const bool over2 = nb >= GAL_N_SYM_PAGE;
            nb = over2 ? 0 : nb;
            nnx = over2 ? 1 : nnx;
            const size_t po = (size_t)ixs[j] * GAL_PAGE_WORDS;
            wa[j] = (x ? p_pnext : p_pcur)[po + (b >> 5)];
            wb[j] = (nnx ? p_pnext : p_pcur)[po + (nb >> 5)];
            ib[j] = b; nx[j] = x; nib[j] = nb;
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const bool on = j < nact;
            const uint32_t d0 = (wa[j] >> (ib[j] & 31)) & 1u, s0b = (cs25 >> (ib[j] % 25)) & 1u;
            const uint32_t d1 = (wb[j] >> (nib[j] & 31)) & 1u, s1b = (cs25 >> (nib[j] % 25)) & 1u;
            const uint32_t sg = (d0 ^ s0b) | (s0b << 1), sgn = (d1 ^ s1b) | (s1b << 1);
            stv[j] = on ? ((uint32_t)ib[j] | ((uint32_t)nx[j] << 9) | (sg << 10) | (sgn << 12) | ((sg * 0x55u) << 16)) : 0u;
            csv[j] = on ? uniform_f64(2.0 * csv[j]) : 0.0;
            dsv[j] = on ? uniform_f64(dsv[j]) : 0.0;
            yv[j] = on ? 2.0 * cx[j] : 0.0;
            double pm = pv[j];
            GAL_MIRROR_BITS(pm, dsv[j])  // mirrored: see ChanState
            pv[j] = on ? pm : 0.0;
        }
    }
#define GAL_DECL(j)                                                                         \
    ChanState ch##j = {0.0, 0.0, 0u};                                                       \
    double cs##j = 0.0, ds##j = 0.0;                                                        \
    if (j < NCH) {                                                                          \
        ch##j.y = yv[j < NCH ? j : 0]; ch##j.p = pv[j < NCH ? j : 0]; ch##j.st = stv[j < NCH ? j : 0]; \
        cs##j = csv[j < NCH ? j : 0]; ds##j = dsv[j < NCH ? j : 0];                         \
    }
    GAL_CH_LIST(GAL_DECL)
#undef GAL_DECL
    // LDS byte address of s_lut[512] (entry k = 0 of the plain table), wave-uniform
    typedef const __attribute__((address_space(3))) int *lds_int_ptr;
    // (cast first, offset second: the generic-pointer offset in between is not always folded away by the compiler)
    // (CBOC on resampled windows: 512-entry tables of 8-byte entries, entry k = 0 first)
    const uint32_t lut0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_int_ptr)s_lut + (CBRW ? 0u : 512u * 4u));
    uint32_t *out = iq + (size_t)er * G.N + n0;  // iq holds the executed range only
    // 64-byte bursts: a lane stores 16 samples back to back so that a half cache line leaves the CU whole
    // (16-byte pieces ~3000 cycles apart were measured at 2.9x the algorithmic HBM write traffic, 64-byte
    // bursts at 1.2x: tools/wrcal.hip).
    const bool vec_ok = ((((size_t)er * G.N + n0) & 15) == 0);

#ifdef GAL_CBOC_PLAIN
    constexpr bool kCbocPlain = SIG == 1;
#else
    constexpr bool kCbocPlain = false;
#endif
    if constexpr (kCbocPlain) {
        // ---- CBOC: plain per-sample loop, four samples per 16-byte store (the first version; A/B builds only)
#define GAL_SGN4(j) [[maybe_unused]] const uint32_t sg4##j = lut0 + (((uint32_t)(d2u(ds##j) >> 32) >> 31) << 12);
        GAL_CH_LIST(GAL_SGN4)
#undef GAL_SGN4
        typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
        const uint32_t str0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32_ptr)s_str);
#define GAL_STEP_C(j) if (j < NCH && j < nact) chan_step_cboc<j>(ch##j, cs##j, ds##j, sg4##j, str0, Pd, ix##j, acc);
        int s0 = 0;
        for (; s0 + 4 <= nsteps; s0 += 4) {
            int o[4];
            if (ACC) {
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = (int)out[s0 + u];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int acc = o[u];
                GAL_CH_LIST(GAL_STEP_C)
                o[u] = acc;
            }
            if (vec_ok && ((s0 & 3) == 0)) {
                *reinterpret_cast<uint4 *>(out + s0) = make_uint4((uint32_t)o[0], (uint32_t)o[1], (uint32_t)o[2], (uint32_t)o[3]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) out[s0 + u] = (uint32_t)o[u];
            }
        }
        for (; s0 < nsteps; ++s0) {
            int acc = ACC ? (int)out[s0] : 0;
            GAL_CH_LIST(GAL_STEP_C)
            out[s0] = (uint32_t)acc;
        }
#undef GAL_STEP_C
    } else {
    // one threshold for all channels: y below it cannot reach the wrap within 16 samples (the code rates of
    // the channels differ by parts per million, so the largest step serves all)
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr0;
    [[maybe_unused]] const uint32_t str0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u32_ptr0)s_str);
    double csmax = 0.0;
#define GAL_CSMAX(j) if (j < NCH) csmax = cs##j > csmax ? cs##j : csmax;
    GAL_CH_LIST(GAL_CSMAX)
#undef GAL_CSMAX
    // A part's group may take the FAST body iff none of its codes can reach the wrap within the group's 16 samples,
    // i.e. y < 8184 - 16 csmax at the group start, and no mirrored phase is negative.  Instead of testing that per
    // channel per group, every part keeps a per-lane COUNTDOWN sf = number of further groups for which it provably
    // holds: y after i groups is y + 16 i cs up to ~1e-9 (65 groups of 16 roundings at ulp 2^-40), far inside the
    // 2^-10 taken off the threshold and the 2^-30 taken off the reciprocal; fast groups leave p >= 0.  The countdown
    // is recomputed after every slow group (which is correct whether or not a wrap occurs: being conservative
    // costs time only).
    const double thr2 = uniform_f64(8184.0 - 16.0 * csmax - 0.0009765625);
    const double inv16 = uniform_f64(csmax > 0.0 ? (1.0 - 9.313225746154785e-10) / (16.0 * csmax) : 0.0);
    [[maybe_unused]] const double tieb = uniform_f64(16.0 * csmax + 0.0009765625);
#define GAL_SF_DECL(j) [[maybe_unused]] int sf##j = 0;
    GAL_CH_LIST(GAL_SF_DECL)
#undef GAL_SF_DECL

// (no `j < nact` tests inside the group loop: see the zero-filled stream rows above)
// RW: the once-per-group code advance (GAL_ADV) assumes fl(y + s) = y + RN_q(s) within a binade, which holds in every
// binade but the channel's tie binade [tl, 2 tl) (s_tie, block prologue): groups that could touch it run the slow body
#define GAL_ROOM(j) if (j < NCH) { const double r = thr2 - ch##j.y; room = r < room ? r : room; negp |= (int)GAL_HI(ch##j.p); \
        if constexpr (RW) { const double tl = s_tie[j], yy = ch##j.y;                                                       \
            const double r2 = yy < tl ? (tl - tieb) - yy : (yy < 2.0 * tl ? -1.0 : 1048576.0);                              \
            room = r2 < room ? r2 : room; } }
#define GAL_SAFE(a, b, c, d)                                                     \
    if (a < NCH) {                                                               \
        double room = 1048576.0;                                                 \
        int negp = 0;                                                            \
        GAL_ROOM(a) GAL_ROOM(b) GAL_ROOM(c) GAL_ROOM(d)                          \
        const int n = (int)(room * inv16); /* room < 0 -> n <= 0 */              \
        sf##a = ((negp < 0) | rw_off) ? 0 : n; /* rw_off: pattern overflow, every group of the block slow */ \
    }
#define GAL_BEGIN_F(j) if (j < NCH) { if constexpr (SIG == 1) group_begin_cboc<j>(ch##j, gr##j, s_str); else group_begin_fast<j>(ch##j, gr##j, s_str); }
#define GAL_BEGIN_S(j) if (j < NCH) group_begin_slow<j>(ch##j, gr##j, s_str);
/* LDS byte address of entry k = 0 of the channel's table: plain (ds >= 0) or conjugate (ds < 0); scalar ALU */
#define GAL_SGN4(j) const uint32_t sg4##j = lut0 + (((uint32_t)(d2u(ds##j) >> 32) >> 31) << 12);
#define GAL_STEP_F(j) if (j < NCH) { if constexpr (SIG == 1) { if (full || j < nact) chan_step_cboc_fast(ch##j, gr##j, cs##j, ds##j, sg4##j, acc); } \
                                    else chan_step_fast(ch##j, gr##j, cs##j, ds##j, sg4##j, acc); }
// CBOC slow groups: the per-sample step that tests the wrap and reads the stream itself (idle positions skipped: a CBOC
// channel never contributes zero)
/* the channel's code step: a register, or (CBOC on resampled windows) an LDS read */
typedef const volatile __attribute__((address_space(3))) double *lds_vf64_ptr;
#define GAL_CS(j) (CBRW ? ((lds_vf64_ptr)s_csl)[(j) < GAL_MAX_NCH ? (j) : 0] : cs##j)
#define GAL_STEP_C(j) if (j < NCH && j < nact) chan_step_cboc<j, CBRW ? 1 : 0>(ch##j, csl##j, ds##j, sg4##j, str0, Pd, ix##j, acc);
#define GAL_STEP_S(j) if (j < NCH) chan_step_wrap(ch##j, gr##j, cs##j, ds##j, sg4##j, acc);
#define GAL_END(j) if (j < NCH) group_end(ch##j, gr##j, Pd, ix##j);
// pin the step: without this the instruction selector floats the pure-arithmetic parts of all 16 steps apart
// (all NCO chains first, all accumulates last) and spills hundreds of values
#define GAL_PIN(a, b, c, d)                                                                              \
    asm volatile("" : "+v"(acc), "+v"(ch##a.y), "+v"(ch##a.p), "+v"(ch##b.y), "+v"(ch##b.p),             \
                      "+v"(ch##c.y), "+v"(ch##c.p), "+v"(ch##d.y), "+v"(ch##d.p));
// The channels are replayed four at a time over the whole group (accumulating into o[]): four independent
// dependency chains give the scheduler enough ILP to cover FP64 and LDS latency, while only four channels'
// group temporaries are live at once.  sched_barrier keeps the parts apart.  `near` is wave-uniform after the
// ballot: no lane of the wave has any of the four codes within 16 samples of its wrap -> fast body.
#define GAL_RW_A(j) if (j < NCH) rw_phase_a<j, BINS, BPITCH>(ch##j, rt##j, s_bin);
#define GAL_RW_B(j) if (j < NCH) unsafe##j = rw_phase_b<j>(rt##j, str0, s_pat);
/* CBOC: spread first, then the half-period look-up (few temporaries live across it) */
#define GAL_RW_C1(j) if (j < NCH) rw_phase_c1_cboc(ch##j, rt##j);
#define GAL_RW_A6(j) if (j < NCH) rw_phase_a6<j>(ch##j, rt##j, s_bin6);
#define GAL_RW_B6(j) if (j < NCH) unsafe##j = rw_phase_b6<j>(rt##j, s_pat6);
/* CBOC: an idle position (all-zero stream row) has B == C everywhere, so its (B + C) word is cleared by hand */
#define GAL_RW_C(j) if (j < NCH) { if constexpr (CBRW) { rw_phase_d_cboc(rt##j, gx##j, gxb##j); gxb##j = j < nact ? gxb##j : 0u; } \
                                   else gx##j = rw_phase_c<RW>(ch##j, rt##j); }
#define GAL_STEP_R(j) if (j < NCH) { if constexpr (CBRW) chan_step_rw_cboc(ch##j, gx##j, gxb##j, u, ds##j, sg4##j, acc); \
                                     else chan_step_rw(ch##j, gx##j, u, ds##j, sg4##j, acc); }
#define GAL_PIN_R(a, b, c, d) asm volatile("" : "+v"(acc), "+v"(ch##a.p), "+v"(ch##b.p), "+v"(ch##c.p), "+v"(ch##d.p));
// RW: the code NCO over the 16 samples of a fast group in three instructions.  Within a binade (and outside the tie
// binade, which GAL_ROOM keeps away) every sequential step adds the same S = RN_q(cs2) = fl(y + cs2) - y, and y + 16 S
// is a multiple of q below the binade's end, hence exact: fma(S, 16, y) IS the 16th sequential sum.  If that value
// has left y's binade the group may have crossed the boundary (coarser grid behind it): then, and only then, the 16
// additions are made one by one (wave-uniform branch; the lanes of a wave share their code phase class).
#define GAL_ADV(j)                                                                            \
    if (j < NCH) {                                                                            \
        const double y0 = ch##j.y;                                                            \
        const double cstep_ = GAL_CS(j);                                                      \
        const double y1 = y0 + cstep_;                                                        \
        const double S = y1 - y0;                                                             \
        double y16 = __builtin_fma(S, 16.0, y0);                                              \
        if (__builtin_amdgcn_ballot_w64((GAL_HI(y16) ^ GAL_HI(y0)) > 0xfffffu) != 0) {       \
            y16 = y0;                                                                         \
            _Pragma("unroll") for (int q = 0; q < SYN_GROUP; ++q) y16 = y16 + cstep_;        \
        }                                                                                     \
        ch##j.y = y16;                                                                        \
    }
#define GAL_LAST_OF(a, b, c, d) ((d) < NCH ? (d) : (c) < NCH ? (c) : (b) < NCH ? (b) : (a))
#define GAL_PART(a, b, c, d)                                                     \
    if (a < NCH) {                                                               \
        ChanGroup gr##a = {0u, 0, 1}, gr##b = {0u, 0, 1};                        \
        ChanGroup gr##c = {0u, 0, 1}, gr##d = {0u, 0, 1};                        \
        const bool near = (GSZ != SYN_GROUP) | (sf##a < 1);                      \
        bool fast = __builtin_amdgcn_ballot_w64(near) == 0;                      \
        if constexpr (RW != 0) {                                                 \
            [[maybe_unused]] uint32_t gx##a = 0u, gx##b = 0u, gx##c = 0u, gx##d = 0u; \
            [[maybe_unused]] uint32_t gxb##a = 0u, gxb##b = 0u, gxb##c = 0u, gxb##d = 0u; \
            [[maybe_unused]] RwTmp rt##a = {}, rt##b = {}, rt##c = {}, rt##d = {}; \
            if (fast) {                                                          \
                [[maybe_unused]] bool unsafe##a = false, unsafe##b = false, unsafe##c = false, unsafe##d = false; \
                GAL_RW_A(a) GAL_RW_A(b) GAL_RW_A(c) GAL_RW_A(d)                  \
                GAL_RW_B(a) GAL_RW_B(b) GAL_RW_B(c) GAL_RW_B(d)                  \
                fast = (__builtin_amdgcn_ballot_w64(unsafe##a) | __builtin_amdgcn_ballot_w64(unsafe##b) | \
                        __builtin_amdgcn_ballot_w64(unsafe##c) | __builtin_amdgcn_ballot_w64(unsafe##d)) == 0; \
                if constexpr (CBRW) {                                            \
                    if (fast) {                                                  \
                        GAL_RW_C1(a) GAL_RW_C1(b) GAL_RW_C1(c) GAL_RW_C1(d)      \
                        GAL_RW_A6(a) GAL_RW_A6(b) GAL_RW_A6(c) GAL_RW_A6(d)      \
                        GAL_RW_B6(a) GAL_RW_B6(b) GAL_RW_B6(c) GAL_RW_B6(d)      \
                        fast = (__builtin_amdgcn_ballot_w64(unsafe##a) | __builtin_amdgcn_ballot_w64(unsafe##b) | \
                                __builtin_amdgcn_ballot_w64(unsafe##c) | __builtin_amdgcn_ballot_w64(unsafe##d)) == 0; \
                    }                                                            \
                }                                                                \
            }                                                                    \
            if (fast) {                                                          \
                sf##a -= 1;                                                      \
                GAL_RW_C(a) GAL_RW_C(b) GAL_RW_C(c) GAL_RW_C(d)                  \
                GAL_SGN4(a) GAL_SGN4(b) GAL_SGN4(c) GAL_SGN4(d)                  \
                _Pragma("unroll") for (int u = 0; u < GSZ; ++u)                  \
                {                                                                \
                    int acc = o[u];                                              \
                    GAL_STEP_R(a) GAL_STEP_R(b) GAL_STEP_R(c) GAL_STEP_R(d)      \
                    if (u & 1) { GAL_PIN_R(a, b, c, d) }                         \
                    o[u] = acc;                                                  \
                }                                                                \
                GAL_ADV(a) GAL_ADV(b) GAL_ADV(c) GAL_ADV(d)                      \
            }                                                                    \
        } else if (fast) {                                                       \
            sf##a -= 1;                                                          \
            GAL_BEGIN_F(a) GAL_BEGIN_F(b) GAL_BEGIN_F(c) GAL_BEGIN_F(d)          \
            GAL_SGN4(a) GAL_SGN4(b) GAL_SGN4(c) GAL_SGN4(d)                      \
            /* CBOC: a channel never contributes zero, so idle positions must be skipped; the usual epoch has all */ \
            /* positions of the part active: one test per group instead of one per channel and sample             */ \
            if (SIG != 1 || GAL_LAST_OF(a, b, c, d) < nact) {                    \
                constexpr bool full = true;                                      \
                _Pragma("unroll") for (int u = 0; u < GSZ; ++u)                  \
                {                                                                \
                    int acc = o[u];                                              \
                    GAL_STEP_F(a) GAL_STEP_F(b) GAL_STEP_F(c) GAL_STEP_F(d)      \
                    if (u & 1) { GAL_PIN(a, b, c, d) } /* 2 steps per scheduling unit: measured best (1: -2 %, 4: spills) */ \
                    o[u] = acc;                                                  \
                }                                                                \
            } else {                                                             \
                constexpr bool full = false;                                     \
                _Pragma("unroll") for (int u = 0; u < GSZ; ++u)                  \
                {                                                                \
                    int acc = o[u];                                              \
                    GAL_STEP_F(a) GAL_STEP_F(b) GAL_STEP_F(c) GAL_STEP_F(d)      \
                    if (u & 1) { GAL_PIN(a, b, c, d) }                           \
                    o[u] = acc;                                                  \
                }                                                                \
            }                                                                    \
        }                                                                        \
        if (!fast) {                                                             \
            if constexpr (SIG == 1) {                                            \
                GAL_SGN4(a) GAL_SGN4(b) GAL_SGN4(c) GAL_SGN4(d)                  \
                [[maybe_unused]] const double csl##a = GAL_CS(a), csl##b = GAL_CS(b), csl##c = GAL_CS(c), csl##d = GAL_CS(d); \
                _Pragma("unroll") for (int u = 0; u < GSZ; ++u)                  \
                {                                                                \
                    int acc = o[u];                                              \
                    GAL_STEP_C(a) GAL_STEP_C(b) GAL_STEP_C(c) GAL_STEP_C(d)      \
                    GAL_PIN(a, b, c, d)                                          \
                    o[u] = acc;                                                  \
                }                                                                \
            } else {                                                             \
                GAL_BEGIN_S(a) GAL_BEGIN_S(b) GAL_BEGIN_S(c) GAL_BEGIN_S(d)      \
                GAL_SGN4(a) GAL_SGN4(b) GAL_SGN4(c) GAL_SGN4(d)                  \
                _Pragma("unroll") for (int u = 0; u < GSZ; ++u)                  \
                {                                                                \
                    int acc = o[u];                                              \
                    GAL_STEP_S(a) GAL_STEP_S(b) GAL_STEP_S(c) GAL_STEP_S(d)      \
                    GAL_PIN(a, b, c, d)                                          \
                    o[u] = acc;                                                  \
                }                                                                \
                GAL_END(a) GAL_END(b) GAL_END(c) GAL_END(d)                      \
            }                                                                    \
            if (GSZ == SYN_GROUP) { GAL_SAFE(a, b, c, d) }                       \
        }                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                       \
    }

// Balanced parts: a part with a single channel has no ILP to hide its FP64/LDS latency, so 5, 6, 7, 9 and 10
// channels are cut 3+2, 3+3, 4+3, 3+3+3 and 4+3+3 instead of 4+1, 4+2, 4+3, 4+4+1 and 4+4+2 (positions >= NCH
// are compiled out; 10 and 11 serve as the dummies).
#define GAL_FOR_PARTS(M)                                                                         \
    if constexpr (NCH == 5) { M(0, 1, 2, 11) M(3, 4, 10, 11) }                                    \
    else if constexpr (NCH == 6) { M(0, 1, 2, 11) M(3, 4, 5, 11) }                                \
    else if constexpr (NCH == 9) { M(0, 1, 2, 11) M(3, 4, 5, 11) M(6, 7, 8, 11) }                 \
    else if constexpr (NCH == 10) { M(0, 1, 2, 3) M(4, 5, 6, 11) M(7, 8, 9, 11) }                 \
    else { M(0, 1, 2, 3) M(4, 5, 6, 7) M(8, 9, 10, 11) }
#define GAL_ALL_PARTS GAL_FOR_PARTS(GAL_PART)

    GAL_FOR_PARTS(GAL_SAFE)

    int s0 = 0;
    for (; s0 + SYN_GROUP <= nsteps; s0 += SYN_GROUP) {
        constexpr int GSZ = SYN_GROUP;
        int o[SYN_GROUP];  // the int16 pairs I,Q of the group's samples, as they go to memory
        if (ACC) {  // second and later channel groups of a sample (> 12 active channels): continue from what is stored
            if (vec_ok) {
#pragma unroll
                for (int q = 0; q < SYN_GROUP / 4; ++q) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(out + s0 + 4 * q);
                    o[4 * q] = (int)v.x; o[4 * q + 1] = (int)v.y; o[4 * q + 2] = (int)v.z; o[4 * q + 3] = (int)v.w;
                }
            } else {
#pragma unroll
                for (int u = 0; u < SYN_GROUP; ++u) o[u] = (int)out[s0 + u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < SYN_GROUP; ++u) o[u] = 0;
        }
        GAL_ALL_PARTS
        if (vec_ok) {
#pragma unroll
            for (int q = 0; q < SYN_GROUP / 4; ++q)
                *reinterpret_cast<uint4 *>(out + s0 + 4 * q) =
                    make_uint4((uint32_t)o[4 * q], (uint32_t)o[4 * q + 1], (uint32_t)o[4 * q + 2], (uint32_t)o[4 * q + 3]);
        } else {
#pragma unroll
            for (int u = 0; u < SYN_GROUP; ++u) out[s0 + u] = (uint32_t)o[u];
        }
    }
    for (; s0 < nsteps; ++s0) {  // ragged tail: groups of one sample
        constexpr int GSZ = 1;
        int o[1];
        o[0] = ACC ? (int)out[s0] : 0;
        GAL_ALL_PARTS
        out[s0] = (uint32_t)o[0];
    }
#undef GAL_ALL_PARTS
#undef GAL_FOR_PARTS
#undef GAL_SAFE
#undef GAL_ROOM
#undef GAL_PART
#undef GAL_LAST_OF
#undef GAL_ADV
#undef GAL_RW_A
#undef GAL_RW_B
#undef GAL_RW_C
#undef GAL_RW_B6
#undef GAL_RW_A6
#undef GAL_RW_C1
#undef GAL_STEP_R
#undef GAL_PIN_R
#undef GAL_PIN
#undef GAL_SGN4
#undef GAL_BEGIN_F
#undef GAL_BEGIN_S
#undef GAL_STEP_F
#undef GAL_STEP_C
#undef GAL_STEP_S
#undef GAL_END

    }  // SIG == 0

    // --- chain self-check: replayed end state must equal the walker's next checkpoint bit for bit (all loads first).
    {
        const double *const q_cpx = Pd->cp_x, *const q_cpp = Pd->cp_p;
        const uint32_t *const q_cpi = Pd->cp_ib;
        double ex[NCH], ep[NCH];
        uint32_t ei[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const size_t cp = (size_t)ixs[j] * G.CP1 + c + 1;
            ex[j] = q_cpx[cp];
            ep[j] = q_cpp[cp];
            ei[j] = q_cpi[cp];
        }
        int bad = 0;
#define GAL_CHECK(j)                                                                  \
    if (j < NCH && j < nact) {                                                        \
        const uint32_t v = ei[j < NCH ? j : 0];                                       \
        bad += d2u(ch##j.y) != d2u(2.0 * ex[j < NCH ? j : 0]);                        \
        { double pm = ep[j < NCH ? j : 0]; GAL_MIRROR_BITS(pm, ds##j)                 \
          bad += ch##j.p != pm; } /* numeric: the mirrored form may leave -0.0 for +0.0 */ \
        bad += (ch##j.st & 0x3ffu) != ((v & 0x1ffu) | ((v >> 16) << 9));              \
    }
        GAL_CH_LIST(GAL_CHECK)
#undef GAL_CHECK
#undef GAL_MIRROR_BITS
#undef GAL_HI
        // ... and the carrier must enter this epoch exactly where it left the previous one (:531-532)
        if (c == 0 && e > 0) {
#define GAL_LINK(j)                                                                                  \
    if (j < NCH && j < nact) {                                                                       \
        const int idx = ix##j;                                                                       \
        if (!(Pd->flags[idx] & GAL_CH_RESTART)) {                                                    \
            const double p_in = q_cpp[(size_t)idx * G.CP1];                                          \
            const double p_prev = q_cpp[(size_t)(idx - G.S) * G.CP1 + G.nchunks];                    \
            bad += d2u(p_in) != d2u(p_prev);                                                         \
        }                                                                                            \
    }
            GAL_CH_LIST(GAL_LINK)
#undef GAL_LINK
        }
        if (bad) atomicAdd(&Pd->ctr[CTR_MISMATCH], bad);
    }
}

#endif  // GAL_TU_SYNTH

// ------------------------------------------------------------------------------------------------
// Launchers (called from synth_api.cpp, which is plain C++ and does not see <<<>>>).
#if GAL_TU_WALK

extern "C" void galk_launch_walk_code(const DevPlan *P, hipStream_t st)
{
    const int n = P->E * P->S * P->Wc;  // one lane per leg of the code chain
    hipLaunchKernelGGL(k_walk_code, dim3((n + 63) / 64), dim3(64), 0, st, *P);
}

extern "C" void galk_launch_verify_carr(const DevPlan *P, hipStream_t st)
{
    const int mod = P->ver_mod > 1 ? P->ver_mod : 1;
    const int nsel = (P->LEGS + mod - 1) / mod * P->S;  // the rotation's legs, then (mod > 1) one thread per eight legs for the risky ones
    const int n = nsel + (mod > 1 ? (P->LEGS * P->S + 7) / 8 : 0);
    hipLaunchKernelGGL(k_verify_carr, dim3((n + 255) / 256), dim3(256), 0, st, *P);
}

extern "C" void galk_launch_verify_code(const DevPlan *P, hipStream_t st)
{
    const int n = P->E * P->S * P->Wc;
    hipLaunchKernelGGL(k_verify_code, dim3((n + 63) / 64), dim3(64), 0, st, *P);
}

extern "C" void galk_launch_walk_carr(const DevPlan *P, int first, hipStream_t st)
{
    const int n = P->LEGS * P->S;
    hipLaunchKernelGGL(k_walk_carr, dim3((n + 63) / 64), dim3(64), 0, st, *P, first);
}

#ifdef GAL_TEST_HOOKS
extern "C" int galk_scanm_stamps(unsigned long long *out, int reset)  // out: SCANM_STAMP_BLOCKS x SCANM_NSTAMP words
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scanm_stamp), sizeof(g_scanm_stamp)) != hipSuccess) return -1;
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_scanm_stamp)) != hipSuccess || hipMemset(d, 0, sizeof(g_scanm_stamp)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// legs per block of the stitch (GAL_TEST_HOOKS: GAL_SCAN_BLOCK_LEGS = 1 ... 256 gives the small batches of the randomised soaks
// many blocks per slot, i.e. the look-back)
static int scanm_lpb()
{
#ifdef GAL_TEST_HOOKS
    if (const char *e = getenv("GAL_SCAN_BLOCK_LEGS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= SCANM_THREADS) return v;
    }
#endif
    return SCANM_THREADS;
}

extern "C" int galk_scanm_blocks(int legs) { return (legs + scanm_lpb() - 1) / scanm_lpb(); }

// scratch: the stitch's look-back records (bytes for S slots; see ScanM).  The ticket counters and the two status arrays come
// first: gal_synth_plan clears them once (galk_scanm_status_bytes), after that the launches' tags keep the passes apart.
extern "C" size_t galk_scanm_status_bytes(int S, int legs) { return 256 + ((size_t)S * galk_scanm_blocks(legs) * 4 + 255) / 256 * 256 * 2; }
extern "C" size_t galk_scanm_bytes(int S, int legs)
{
    const size_t SB = (size_t)S * galk_scanm_blocks(legs);
    return galk_scanm_status_bytes(S, legs) + ((SB * 4 + 255) / 256 * 256) * 2 + ((SB * 8 + 255) / 256 * 256) * 3 + ((SB * 32 + 255) / 256 * 256) + 4096;
}

extern "C" void galk_launch_carr_scan(const DevPlan *P, uint32_t tag, hipStream_t st)
{
    ScanM M;
    M.lpb = scanm_lpb();
    M.B = galk_scanm_blocks(P->LEGS);
    M.Bs = galk_scanm_blocks(P->LEGS_all > P->LEGS ? P->LEGS_all : P->LEGS);
    M.tag = tag ? tag : 1u;
    M.nap = 1;
#ifdef GAL_TEST_HOOKS
    if (const char *e = getenv("GAL_SCANM_NAP")) M.nap = atoi(e);
#endif
    const size_t SB = (size_t)P->S * M.Bs;
    char *p = (char *)P->scanm;
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) / 256 * 256; return q; };
    M.cnt = (uint32_t *)take(256);  // (S <= 64)
    M.st1 = (uint32_t *)take(SB * 4); M.st2 = (uint32_t *)take(SB * 4);
    M.a1_kind = (int *)take(SB * 4); M.a2_f = (int *)take(SB * 4);
    M.a1_w = (long long *)take(SB * 8); M.a1_r = (double *)take(SB * 8); M.a2_K = (double *)take(SB * 8);
    M.a2_c = (double *)take(SB * 32);
    hipLaunchKernelGGL(k_scanm, dim3(M.B, P->S), dim3(SCANM_THREADS), 0, st, *P, M);
}

extern "C" void galk_launch_pages(const DevPlan *P, hipStream_t st)
{
    hipLaunchKernelGGL(k_pages, dim3(P->S), dim3(GUESS_THREADS), 0, st, *P);
}

// The batch's completion record, written by the DEVICE straight into pinned host memory behind the synthesis kernel: the
// counters (walker passes, replay check), the end-of-batch channel state and -- last, behind a system-scope fence -- the
// batch's sequence number, which gal_synth_finish polls.  (Two hipMemcpyAsync + hipStreamSynchronize before: two blit
// kernels and the runtime's wait path between the end of k_synth and the host seeing it.)
__global__ void k_publish(const int *__restrict__ ctr, const uint32_t *__restrict__ state, const int state_words,
                          int *h_ctr, uint32_t *h_state, uint32_t *h_flag, const uint32_t seq)
{
    const int t = threadIdx.x;
    if (t < CTR_COUNT) h_ctr[t] = ctr[t];
    for (int i = t; i < state_words; i += blockDim.x) h_state[i] = state[i];
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" void galk_launch_publish(const DevPlan *P, int *h_ctr, void *h_state, uint32_t *h_flag, uint32_t seq, hipStream_t st)
{
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, st, P->ctr, (const uint32_t *)P->state_out,
                       (int)(sizeof(gal_chan_state_t) / 4) * P->S, h_ctr, (uint32_t *)h_state, h_flag, seq);
}

#endif  // GAL_TU_WALK

#if GAL_TU_SYNTH
template <bool ACC, int SIG, int RW = 0>
static int launch_synth_t(const DevPlan *P, const DevPlan *Pd, int nch, const uint8_t *act, const int *nact,
                          uint32_t *iq, int e0, int ne, hipStream_t st)
{
    const dim3 grid(ne * P->blocks_per_epoch), block(SYN_BLOCK);
    SynGeom G;
    G.e0 = e0;
    G.S = P->S; G.N = P->N; G.R = P->R; G.nchunks = P->nchunks; G.CP1 = P->CP1; G.blocks_per_epoch = P->blocks_per_epoch;
    G.cls = P->cls > 0 ? P->cls : 1;
    G.per = P->nchunks / G.cls;
#define GAL_CASE(n) case n: hipLaunchKernelGGL((k_synth<n, ACC, SIG, RW>), grid, block, 0, st, Pd, G, act, nact, iq); break;
    if constexpr (SIG == 1) {
        // the opt-in CBOC mode is built for 4, 8 and 12 positions only (a third of the compile time of this file went
        // into its 24 instantiations): positions beyond the active count are idle and skipped like in any epoch with
        // fewer channels than the launch was sized for
        if (nch < 1 || nch > 12) return -1;
        switch ((nch + 3) / 4 * 4) {
            GAL_CASE(4) GAL_CASE(8) GAL_CASE(12)
        default: return -1;
        }
    } else {
        switch (nch) {
            GAL_CASE(1) GAL_CASE(2) GAL_CASE(3) GAL_CASE(4) GAL_CASE(5) GAL_CASE(6)
            GAL_CASE(7) GAL_CASE(8) GAL_CASE(9) GAL_CASE(10) GAL_CASE(11) GAL_CASE(12)
        default: return -1;
        }
    }
#undef GAL_CASE
    return 0;
}

// One launcher and one no-op kernel per k_synth family (each family is its own code object in the product build; HIP
// loads a code object at the first launch of any of its kernels -- ~10 ms for the larger ones --, which galk_warm moves
// from the caller's first batch into gal_synth_create).
#define GAL_FAMILY(K, SIG, RW)                                                                                       \
    __global__ void k_warm_f##K() {}                                                                                 \
    extern "C" void galk_warm_f##K(hipStream_t st) { hipLaunchKernelGGL(k_warm_f##K, dim3(1), dim3(64), 0, st); }    \
    extern "C" int galk_launch_synth_f##K(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate,               \
                                          const uint8_t *act, const int *nact, uint32_t *iq, int e0, int ne,         \
                                          hipStream_t st)                                                             \
    {                                                                                                                \
        return accumulate ? launch_synth_t<true, SIG, RW>(P, Pd, nch, act, nact, iq, e0, ne, st)                      \
                          : launch_synth_t<false, SIG, RW>(P, Pd, nch, act, nact, iq, e0, ne, st);                    \
    }
#if GAL_TU_FAMILY(1)
GAL_FAMILY(1, 0, 0)
#endif
#if GAL_TU_FAMILY(2)
GAL_FAMILY(2, 0, 1)
#endif
#if GAL_TU_FAMILY(3)
GAL_FAMILY(3, 0, 2)
#endif
#if GAL_TU_FAMILY(4)
GAL_FAMILY(4, 1, 0)
#endif
#if GAL_TU_FAMILY(5)
GAL_FAMILY(5, 0, 3)
#endif
#if GAL_TU_FAMILY(6)
GAL_FAMILY(6, 1, 1)
#endif
#undef GAL_FAMILY
#endif  // GAL_TU_SYNTH

#if GAL_TU_WALK
#define GAL_FAMILY_DECL(K)                                                                                           \
    extern "C" void galk_warm_f##K(hipStream_t st);                                                                  \
    extern "C" int galk_launch_synth_f##K(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate,               \
                                          const uint8_t *act, const int *nact, uint32_t *iq, int e0, int ne, hipStream_t st);
GAL_FAMILY_DECL(1) GAL_FAMILY_DECL(2) GAL_FAMILY_DECL(3) GAL_FAMILY_DECL(4) GAL_FAMILY_DECL(5) GAL_FAMILY_DECL(6)
#undef GAL_FAMILY_DECL

__global__ void k_warm() {}
extern "C" void galk_touch(hipStream_t st) { hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, st); }
// signal: 0 = BOC(1,1), 1 = CBOC; ratio = 2 * 1.023e6 / sample rate = the nominal code step in half chips per sample.  Only the
// families a handle of this configuration can launch are loaded now (the walkers, the classic body, and the resampled-window
// form whose gate the rate can pass: synth_api.cpp, gal_synth_plan); should a batch need another one after all, HIP loads it
// at that launch.
extern "C" void galk_warm(hipStream_t st, int signal, double ratio)
{
    hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, st);
    const bool rw1 = ratio >= 0.70 && ratio <= 1.02;
    if (signal == 1) {
        galk_warm_f4(st);
        if (rw1) galk_warm_f6(st);
        return;
    }
    galk_warm_f1(st);
    if (rw1) galk_warm_f2(st);
    if (ratio <= 0.14) galk_warm_f3(st);
    else if (ratio <= 0.28) galk_warm_f5(st);
}

extern "C" int galk_launch_synth(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, const uint8_t *act,
                                 const int *nact, uint32_t *iq, int e0, int ne, hipStream_t st)
{
    if (P->signal == 1)
        return P->rw == 1 ? galk_launch_synth_f6(P, Pd, nch, accumulate, act, nact, iq, e0, ne, st)
                          : galk_launch_synth_f4(P, Pd, nch, accumulate, act, nact, iq, e0, ne, st);
    if (P->rw == 1) return galk_launch_synth_f2(P, Pd, nch, accumulate, act, nact, iq, e0, ne, st);
    if (P->rw == 2) return galk_launch_synth_f3(P, Pd, nch, accumulate, act, nact, iq, e0, ne, st);
    if (P->rw == 3) return galk_launch_synth_f5(P, Pd, nch, accumulate, act, nact, iq, e0, ne, st);
    return galk_launch_synth_f1(P, Pd, nch, accumulate, act, nact, iq, e0, ne, st);
}
#endif  // GAL_TU_WALK
