// nco_walk.h -- exact closed-form evaluation of the reference's sequential FP64 NCO recurrences.
//
// The reference advances two phases per channel per sample (src/galileo-sdr.cpp:528-532):
//     code_phase += f_code * delt;                       (wrap check at the top of the next sample, :491-507)
//     carr_phase += f_carr * delt;  carr_phase -= (long)carr_phase;
// Both are chains of *rounded* double additions, so x0 + n*step is not bit-exact.
// What IS exact: while x stays inside one binade [2^k, 2^(k+1)) it sits on that binade's ulp grid
// g = 2^(k-52), and   fl(x + d) == x + RN_g(d)   for every such x, where RN_g(d) is d rounded to a
// multiple of g (ties: to even, valid once x/g is even -- which one genuine tie step guarantees).
// n such steps therefore collapse into ONE exact fma: x_n = fma(n, RN_g(d), x).  A genuine single
// step is taken at every binade crossing and at every wrap.  Magnitude-decreasing runs (phase and
// step of opposite sign) are the mirror image inside a binade.
//
// The same functions are compiled for the device (hipcc) and for the host (g++, CPU unit tests
// against brute-force stepping in tests/test_walker_cpu.py).  Compile with -ffp-contract=off.
#ifndef GAL_NCO_WALK_H_
#define GAL_NCO_WALK_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define GAL_HD __host__ __device__ __forceinline__
#else
#define GAL_HD static inline
#endif

namespace galnco {

GAL_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
GAL_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

GAL_HD double fma_exact(double a, double b, double c) { return __builtin_fma(a, b, c); }

static constexpr uint64_t kSign = 0x8000000000000000ull;
static constexpr uint64_t kExpMask = 0x7ff0000000000000ull;

// One genuine carrier step, src/galileo-sdr.cpp:531-532: `p += d; p -= (long)p`.  trunc(p) equals
// (double)(long)p for every p except p == -0.0 (trunc keeps the sign of zero), and -0.0 can only come out
// of the addition when p and d are both -0.0, which gal_synth_plan() excludes by canonicalising a -0.0
// start phase to +0.0 (the int16 output is the same either way; only the sign of a zero state differed).
GAL_HD double carr_step(double p, double d)
{
    p = p + d;
    p = p - __builtin_trunc(p);
    return p;
}

struct Batch {
    int n;       // steps that may be taken in closed form (>= 0)
    double inc;  // x_j = fma(j, inc, x) exactly for 0 <= j <= n
};

// How many genuine steps `x <- fl(x + d)` starting at x can be replaced by x + j*inc, with every
// visited state strictly inside x's binade and below `cap` in magnitude (cap = 1.0 for the carrier
// phase, 4092.0 for the code phase: no wrap can trigger inside a batch).  Conservative: may return
// fewer steps than possible, never more.
GAL_HD Batch nco_batch(double x, double d, int n_max, double cap, double inv_ad)
{
    // Straight-line on purpose: on the GPU 64 lanes walk 64 different chains, so every early return
    // would be executed by somebody; everything is computed and the verdict is a select at the end.
    const uint64_t xb = d2u(x), db = d2u(d);
    const uint64_t xa = xb & ~kSign, da = db & ~kSign;
    const int ex = (int)(xa >> 52), ed = (int)(da >> 52);
    // x must sit above d's binade and on a normal grid
    bool ok = (n_max > 0) & (ex > ed) & (ex >= 54) & (ex < 0x7ff);
    const int exs = ok ? ex : 1023;                     // keep the arithmetic below finite when !ok
    const double pk = u2d((uint64_t)exs << 52);         // 2^k
    const double g = u2d((uint64_t)(exs - 52) << 52);   // ulp of the binade
    const double ad = u2d(da), ax = ok ? u2d(xa) : 1.0;
    const double dk = (ad + pk) - pk;                   // RN_g(|d|), ties to even
    const double rem = ad - dk;                         // exact
    const bool tie = (rem + rem == g) | (rem + rem == -g);
    ok &= !(tie & ((xa & 1ull) != 0));                  // tie: x/g must be even (one genuine step makes it so)
    const bool asc = ((xb ^ db) & kSign) == 0;
    double lim = pk + pk;
    lim = lim > cap ? cap : lim;
    // room left: up to the binade ceiling (or the wrap cap) when |x| grows, down to the floor when it shrinks
    const double t = asc ? (lim - g) - ax : ax - (pk + g);
    const bool fixed = dk == 0.0;                       // |d| <= g/2: x is a fixed point of the rounded add
    // t / dk through the caller's reciprocal of |d| (one division per walk instead of one per batch):
    // |dk - |d|| <= g/2 and n < 2^20 keep the estimate within 2^-20 of the true quotient, so it can only
    // overshoot floor(t/dk) by one, which the exact remainder test below catches (n*dk is a multiple of
    // g below 2^(k+1), hence exactly representable); an undershoot is merely conservative.
    double q = t * inv_ad;
    q = q > (double)n_max ? (double)n_max : q;
    q = (ok & !fixed & (t >= dk)) ? q : 0.0;
    int n = (int)q;
    n -= (fma_exact(-(double)n, dk, t) < 0.0) ? 1 : 0;
    n = n < 0 ? 0 : n;
    n = (ok & fixed & (t >= 0.0)) ? n_max : n;          // not at the binade floor, where the grid below is finer
    Batch b;
    b.n = n;
    const double sdk = asc ? dk : -dk;
    b.inc = (n == 0 || fixed) ? 0.0 : ((xb & kSign) ? -sdk : sdk);
    return b;
}

// ---------------------------------------------------------------------------------------------
// Carrier chain: N samples with constant step d.  `emit(c, p)` receives the phase BEFORE sample
// c*R for c = 0 .. ceil(N/R)-1; the return value is the phase after sample N-1 (what the next epoch
// starts from).  inv_ad = 1.0 / fabs(d) (any value if d == 0).
// Loop shape (uniform across lanes): [checkpoint at the iteration start?] [closed-form batch; checkpoints inside it are
// emitted from the closed form] [one genuine step unless the walk is over].
// carr_walk with wrap tracking: additionally reports the LAST wrap inside the walk -- the local sample
// index right after the wrapping step and the residual phase there -- which is what the speculative
// stitcher (synth_kernels.hip: k_walk_carr / k_scanm) uses as a leg's hand-over state: right after a
// wrap every phase is a multiple of 2^-52, so differences between neighbouring trajectories survive all
// later roundings (binade crossings coarsen the grid only up to 2^-53).
struct WalkOut {
    double p;       // phase after the last sample
    int last_w;     // local index after the last wrapping step, -1 if the walk never wrapped
    double last_r;  // phase at last_w
    double margin;  // min distance of any visited state to the boundaries of its own binade (see below)
    int tdir;       // first wrap whose rounded add was a TIE (sum exactly between two grid points of [1,2)):
                    // +1 / -1 = the add rounded up / down (rounded minus exact sum), 0 = no tie in this walk.
                    // A trajectory shifted by an ODD multiple of 2^-52 resolves that tie the other way: from
                    // there on it is off by the shift minus tdir * 2^-52 (an even multiple: later ties agree).
    int tpos;       // local index right after that step (valid when tdir != 0)
};

// Distance of the states of one closed-form batch to their binade's boundaries.  a and b are the first and
// the last state of the batch (the batch never leaves the binade, so the extremes are at its ends; a == b
// for an empty batch).  Used for TRANSLATED acceptance in the stitcher: if a whole trajectory is shifted by
// delta (a multiple of 2^-52, |delta| below this margin) every rounded add sees the same binade, hence the
// same rounding grid, and -- outside tie epochs -- yields the same result shifted by delta, bit for bit.
GAL_HD double binade_margin(double a, double b)
{
    const uint64_t ua = d2u(a) & ~kSign, ub = d2u(b) & ~kSign;
    const uint64_t ulo = ua < ub ? ua : ub, uhi = ua < ub ? ub : ua;
    const double lo = u2d(ulo), hi = u2d(uhi);
    const double pk = u2d(ulo & 0xfff0000000000000ull);  // binade floor (0 for zero / subnormal: margin 0)
    const double m1 = lo - pk, m2 = (pk + pk) - hi;
    return m1 < m2 ? m1 : m2;
}

// Can a wrap step with this carrier step land exactly between two grid points of [1,2) (spacing 2^-52)?
// Phases in [0.5,1) are multiples of 2^-53, so the sum is an odd multiple of 2^-53 only if the step is a
// multiple of 2^-53 itself: an odd one ties at every wrap, an even one at most at the first wrap of its epoch
// (the phase it inherits may carry the 2^-53 bit; after a wrap everything is a multiple of 2^-52).  How such a
// tie resolves depends on the parity of the phase, which a shift by an odd multiple of 2^-52 flips -- legs that
// touch such an epoch are never accepted by translation.  Steps of a quarter cycle or more may wrap from a
// lower binade: treated as tie-prone wholesale.
GAL_HD bool tie_step(double d)
{
    const double ad = d < 0.0 ? -d : d;
    if (!(ad < 0.25)) return true;
    const double t53 = ad * 9007199254740992.0;  // * 2^53, exact
    return t53 == (double)(long long)t53;
}

// The carrier walk proper.  Loop shape (uniform across lanes): [checkpoint at the iteration start?] [closed-form batch
// inside the current binade; checkpoints inside it come out of the same closed form] [one genuine step unless over].
// The batch is a specialisation of nco_batch for the case that matters -- phase and step of the same sign
// (|p| grows towards the wrap), 2^-30 <= |d| -- with everything that depends only on d hoisted out of the
// loop; other states (a phase still running against a step that changed sign, degenerate steps) take the
// general nco_batch.  Same results as stepping sample by sample, bit for bit (tests/test_walker_cpu.py).
template <class Emit>
GAL_HD WalkOut carr_walk_track(double p, double d, double inv_ad, int N, int R, int cp0, Emit emit)
{
    // cp0: local index of the first checkpoint (checkpoints at cp0, cp0+R, ...); pass cp0 >= N for none
    int i = 0;
    int next_cp = cp0, c = 0;
    WalkOut o;
    o.last_w = -1;
    o.last_r = 0.0;
    o.tdir = 0;
    o.tpos = -1;
    const bool tieprone = tie_step(d);
    const uint64_t db = d2u(d), da = db & ~kSign;
    const double ad = u2d(da);
    const uint32_t ed = (uint32_t)(da >> 52);
    const uint32_t dsign = (uint32_t)(db >> 63);
    const bool lean_d = (ad >= 9.313225746154785e-10) && (ad < 1.0);  // 2^-30 <= |d| < 1
    // the one binade in which |d| is an odd multiple of half an ulp (round-to-even ties inside a batch)
    const uint32_t e_tie = ed + 1u + (uint32_t)__builtin_ctzll(da | (1ull << 52));
    double mg = 4.0;
    // one genuine step with everything that hangs on it (wrap bookkeeping, tie record)
#define GAL_GENUINE_STEP()                                                                           \
    {                                                                                                \
        const double q_ = p + d;                                                                     \
        const double t_ = __builtin_trunc(q_);                                                       \
        const bool wrapped_ = t_ != 0.0;                                                             \
        if (tieprone && wrapped_ && o.tdir == 0) { /* rare: only steps that are multiples of 2^-53 */ \
            const double bv_ = q_ - p; /* TwoSum: err = (p + d) - q exactly */                       \
            const double err_ = (p - (q_ - bv_)) + (d - bv_);                                        \
            if (err_ == 1.1102230246251565e-16) o.tdir = -1; /* exact sum above q: rounded down */   \
            if (err_ == -1.1102230246251565e-16) o.tdir = 1; /* rounded up */                        \
            if (o.tdir) o.tpos = i + 1;                                                              \
        }                                                                                            \
        p = q_ - t_; /* == carr_step(p, d) */                                                        \
        ++i;                                                                                         \
        o.last_w = wrapped_ ? i : o.last_w;                                                          \
        o.last_r = wrapped_ ? p : o.last_r;                                                          \
    }
    // ---- general iterations: until phase and step have the same sign (at most one carrier cycle, after a
    //      Doppler sign change or a restart), and throughout for degenerate steps
    while (i < N) {
        const uint64_t pb = d2u(p);
        if (lean_d && (((uint32_t)(pb >> 63) == dsign) || (pb & ~kSign) == 0)) break;
        if (next_cp == i) {
            emit(c, p);
            ++c;
            next_cp += R;
        }
        const Batch b = nco_batch(p, d, N - i, 1.0, inv_ad);
        const double a0 = p;
        // checkpoints INSIDE the batch come out of the same closed form (x_j = fma(j, inc, x) exactly for j <= n): the
        // walk does not stop for them
        while (next_cp <= i + b.n && next_cp < N) {
            emit(c, fma_exact((double)(next_cp - i), b.inc, a0));
            ++c;
            next_cp += R;
        }
        p = fma_exact((double)b.n, b.inc, p);
        const double m = binade_margin(a0, p);
        mg = m < mg ? m : mg;
        i += b.n;
        if (i < N) GAL_GENUINE_STEP()
    }
    // ---- lean iterations: |p| only grows until the wrap, which keeps the sign -- the regime is permanent
    const double sd = dsign ? -1.0 : 1.0;
    while (i < N) {
        if (next_cp == i) {
            emit(c, p);
            ++c;
            next_cp += R;
        }
        const uint64_t pa = d2u(p) & ~kSign;
        const uint32_t ea = (uint32_t)(pa >> 52);
        const bool can = ea > ed;                      // above the step's binade (hence normal)
        const uint64_t pkb = (uint64_t)ea << 52;
        const double pk = u2d(pkb);                    // binade floor 2^k (0 for a zero / subnormal phase)
        const double top = u2d(pkb | 0x000fffffffffffffull);  // largest double of the binade
        const double a = u2d(pa);
        const double dk = (ad + pk) - pk;              // RN_g(|d|), ties to even; > 0 because |d| >= 2^-30 > g/2
        const bool odd_tie = (ea == e_tie) & ((uint32_t)pa & 1u);  // a tie binade needs x/g even
        const double t = top - a;                      // room to the binade ceiling, exact, >= 0
        // t / dk through the caller's reciprocal of |d|: |dk - |d|| <= g/2 keeps the estimate within
        // 2^-33 / |d| <= 2^-3 of the true quotient, so it overshoots floor(t/dk) by at most one, which the
        // exact remainder test catches (n*dk is a multiple of g below 2^(k+1): exactly representable)
        double q = t * inv_ad;
        const double qmax = (double)(N - i);
        q = q > qmax ? qmax : q;
        int n = (int)q;
        n -= (fma_exact(-(double)n, dk, t) < 0.0) ? 1 : 0;
        n = n < 0 ? 0 : n;
        n = (can & !odd_tie) ? n : 0;
        const double nd = (double)n;
        const double room = fma_exact(-nd, dk, t);     // ceiling minus the last state of the batch (>= 0)
        const double m1 = a - pk;                      // first state of the batch above the binade floor
        mg = m1 < mg ? m1 : mg;
        mg = room < mg ? room : mg;
        while (next_cp <= i + n && next_cp < N) {      // checkpoints inside the batch: same closed form, no stop
            emit(c, fma_exact((double)(next_cp - i) * sd, dk, p));
            ++c;
            next_cp += R;
        }
        p = fma_exact(nd * sd, dk, p);
        i += n;
        if (i < N) GAL_GENUINE_STEP()
    }
#undef GAL_GENUINE_STEP
    const double m = binade_margin(p, p);  // the state handed over
    o.margin = m < mg ? m : mg;
    o.p = p;
    return o;
}

// Carrier chain: N samples with constant step d.  `emit(c, p)` receives the phase BEFORE sample c*R for
// c = 0 .. ceil(N/R)-1; the return value is the phase after sample N-1 (what the next epoch starts from).
// inv_ad = 1.0 / fabs(d) (any value if d == 0).
template <class Emit>
GAL_HD double carr_walk(double p, double d, double inv_ad, int N, int R, Emit emit)
{
    return carr_walk_track(p, d, inv_ad, N, R, 0, emit).p;
}

// Code chain: N samples, step c > 0, wrap at 4092 checked BEFORE each sample's use
// (src/galileo-sdr.cpp:491-507).  `emit(cidx, x, ibit, flipped)` receives the PRE-check state before
// sample cidx*R.  Returns the pre-check state after sample N-1 through the reference parameters.
struct CodeEnd {
    double x;
    int ibit;
    int flipped;
};

template <class Emit>
GAL_HD CodeEnd code_walk(double x, int ibit, double cstep, double inv_c, int N, int R, Emit emit)
{
    int i = 0;
    int next_cp = 0, c = 0;
    int flipped = 0;
    // The batch is nco_batch specialised for this chain -- phase and step positive, the phase only grows until the wrap,
    // cap 4092 -- with everything that depends only on the step hoisted out of the loop (the walk is one long dependency
    // chain per lane: its length in instructions IS its run time).  Steps outside [2^-20, 4) chips take the general
    // nco_batch: same results either way, bit for bit (tests/test_walker_cpu.py against brute-force stepping).
    const uint64_t cb = d2u(cstep);
    const uint32_t ed = (uint32_t)(cb >> 52);
    // (2^-20: what gal_synth_plan admits.  Below about 2^-30 the quotient estimate t * inv_c -- the reciprocal of the step, not of
    // its rounded value dk -- can overshoot floor(t / dk) by more than the one step the remainder test takes back, in this batch
    // and in the general nco_batch alike: such steps are taken one by one)
    const bool lean = (cstep >= 9.5367431640625e-07) && (cstep < 4.0) && (x >= 0.0);
    // the one binade in which the step is an odd multiple of half an ulp (round-to-even ties inside a batch)
    const uint32_t e_tie = ed + 1u + (uint32_t)__builtin_ctzll(cb | (1ull << 52));
    while (i < N) {
        if (next_cp == i) {
            emit(c, x, ibit, flipped);
            ++c;
            next_cp += R;
        }
        const bool ge = x >= 4092.0;
        x = x - (ge ? 4092.0 : 0.0);
        ibit += ge ? 1 : 0;
        const bool flip = ibit >= 500;
        ibit = flip ? 0 : ibit;
        flipped |= flip ? 1 : 0;
        int n;
        double inc;
        if (lean) {
            const uint64_t xb = d2u(x);
            const uint32_t ea = (uint32_t)(xb >> 52);
            const bool can = ea > ed;                          // above the step's binade (hence normal)
            const uint64_t pkb = (uint64_t)ea << 52;
            const double pk = u2d(pkb);                        // binade floor 2^k
            // largest state a batch may visit: the last double of the binade, or the last one below the wrap
            const double top = ea == 1034u ? 4091.9999999999995 : u2d(pkb | 0x000fffffffffffffull);
            const double dk = (cstep + pk) - pk;               // RN_g(step), ties to even; > 0 (step > g / 2 by far)
            const bool odd_tie = (ea == e_tie) & ((uint32_t)xb & 1u);  // a tie binade needs x / g even
            const double t = top - x;                          // room, exact
            // t / dk through the caller's reciprocal of the step: overshoots floor(t / dk) by at most one, which the
            // exact remainder test catches (n * dk is a multiple of g below 2^(k+1): exactly representable)
            double q = t * inv_c;
            const double qmax = (double)(N - i);
            q = q > qmax ? qmax : q;
            n = (int)q;
            n -= (fma_exact(-(double)n, dk, t) < 0.0) ? 1 : 0;
            n = n < 0 ? 0 : n;
            n = (can & !odd_tie & (t >= 0.0)) ? n : 0;
            inc = dk;
        } else if (cstep >= 9.5367431640625e-07) {
            const Batch b = nco_batch(x, cstep, N - i, 4092.0, inv_c);
            n = b.n;
            inc = b.inc;
        } else {  // below 2^-20 (never in the product: gal_synth_plan's limit) no quotient estimate is trusted: genuine steps only
            n = 0;
            inc = 0.0;
        }
        // checkpoints inside the batch come out of its closed form (every state of a batch is below the wrap, so the
        // pre-check state IS the state): the walk stops at binade crossings and wraps only -- 13 times per code period
        // instead of once per chunk on top of that
        while (next_cp <= i + n && next_cp < N) {
            emit(c, fma_exact((double)(next_cp - i), inc, x), ibit, flipped);
            ++c;
            next_cp += R;
        }
        x = fma_exact((double)n, inc, x);
        i += n;
        if (i < N) {
            x = x + cstep;
            ++i;
        }
    }
    CodeEnd r;
    r.x = x;
    r.ibit = ibit;
    r.flipped = flipped;
    return r;
}


// ---------------------------------------------------------------------------------------------
// The code chain of ONE epoch in LEGS (round 5; rounds 1-4: one lane walked the epoch's 351 dependent closed-form steps, the long
// pole of a one-epoch call and of a lone handle's walker chain).  Leg 0 starts from the epoch's own start state, which the
// caller supplies; leg k > 0 is walked from an ANCHOR -- the last wrap (:491-507) strictly in front of its first sample, "after the
// check at the top of sample w the phase is r and the symbol counter ib" -- guessed in ideal arithmetic.  A post-wrap phase
// x - 4092 is a multiple of 2^-41 (the ulp of [2048, 4096), where every wrap happens), and every binade below 4096 has a grid that
// 2^-41 is a multiple of: a trajectory shifted by delta = true residual - guessed residual therefore produces every state shifted
// by delta, bit for bit, as long as (a) every state stays in its binade and on its side of the wrap threshold -- code_walk_track
// records the MARGIN, the smallest distance of a visited state to a binade boundary or to 4092 -- and (b) no rounding is a tie
// whose resolution depends on the parity of x / ulp, which delta can flip only in [2048, 4096) and only if the step is a
// multiple of 2^-42 (code_tie_prone: such epochs -- one in ~4000 -- are not translated, their legs are walked one after the other).
// The stitch (code_leg_accept) is the carrier chain's (synth_kernels.hip: k_scanm) in small: the legs of an epoch sit in
// neighbouring lanes, leg k looks at the VERIFIED claim of leg k - 1.
struct CodeTrack {
    double x;        // pre-check state after the last sample
    int ibit;
    int flipped;     // the symbol counter passed 500 during this walk
    int last_w;      // local index of the last sample at whose top a wrap fired, -1: none
    double last_r;   // post-check state there (x - 4092)
    int last_ib;     // symbol counter behind that wrap
    int last_fl;     // `flipped` behind that wrap
    double margin;   // smallest distance of a visited state to a boundary of its binade or to the wrap threshold
    int tpos;        // tie-prone steps: local index of the first state that FOLLOWS a state in [2048, 4096) inside that binade, -1: none
    double tx;       // ... and that state (see code_tie_prone)
};

// Steps that are an ODD multiple of 2^-42: in [2048, 4096) (ulp 2^-41) every addition is then a tie, resolved to the even
// neighbour -- which depends on the parity of x / 2^-41, and a shift by an odd multiple of 2^-41 flips that.  (A multiple of 2^-41
// adds exactly there; below 2048 every binade's ulp divides 2^-42 and the shift is an even multiple of it: no tie hangs on it.)
// What saves the translation: the FIRST addition inside [2048, 4096) leaves x / 2^-41 even whatever it was, so from the second state
// in that binade on (CodeTrack::tpos) two trajectories differ by a constant even multiple again -- the stitch walks the few
// hundred samples from the true anchor to that state once more and reads the new shift off it (code_leg_accept: 3).
GAL_HD bool code_tie_prone(double cstep)
{
    const double t42 = cstep * 4398046511104.0;  // * 2^42, exact
    if (!(cstep < 4.0)) return true;
    return t42 == (double)(long long)t42 && ((long long)t42 & 1LL);
}

// code_walk with the bookkeeping of a speculative leg.  Same states as code_walk (the batches are the same closed forms), bit for
// bit; checkpoints at local samples cp0, cp0 + R, ... (cp0 >= N: none).  Steps outside [2^-20, 4) chips: margin 0 (never translated).
template <class Emit>
GAL_HD CodeTrack code_walk_track(double x, int ibit, double cstep, double inv_c, int N, int R, int cp0, Emit emit)
{
    int i = 0;
    int next_cp = cp0, c = 0;
    CodeTrack o;
    o.flipped = 0;
    o.last_w = -1;
    o.last_r = 0.0;
    o.last_ib = 0;
    o.last_fl = 0;
    o.tpos = -1;
    o.tx = 0.0;
    const uint64_t cb = d2u(cstep);
    const uint32_t ed = (uint32_t)(cb >> 52);
    const bool lean = (cstep >= 9.5367431640625e-07) && (cstep < 4.0) && (x >= 0.0);
    const uint32_t e_tie = ed + 1u + (uint32_t)__builtin_ctzll(cb | (1ull << 52));
    double mg = lean ? 8192.0 : 0.0;
    while (i < N) {
        if (next_cp == i) {
            emit(c, x, ibit, o.flipped);
            ++c;
            next_cp += R;
        }
        const bool in_top = x >= 2048.0 && x < 4092.0;  // (this state is added to inside [2048, 4096): what follows it is even)
        const bool ge = x >= 4092.0;
        x = x - (ge ? 4092.0 : 0.0);
        ibit += ge ? 1 : 0;
        const bool flip = ibit >= 500;
        ibit = flip ? 0 : ibit;
        o.flipped |= flip ? 1 : 0;
        o.last_w = ge ? i : o.last_w;
        o.last_r = ge ? x : o.last_r;
        o.last_ib = ge ? ibit : o.last_ib;
        o.last_fl = ge ? o.flipped : o.last_fl;
        // (a wrapped state sits x above the threshold it has just passed: a shift below -x would not have wrapped here)
        mg = (ge && x < mg) ? x : mg;
        int n;
        double inc;
        if (lean) {
            const uint64_t xb = d2u(x);
            const uint32_t ea = (uint32_t)(xb >> 52);
            const bool can = ea > ed;
            const uint64_t pkb = (uint64_t)ea << 52;
            const double pk = u2d(pkb);
            const double top = ea == 1034u ? 4091.9999999999995 : u2d(pkb | 0x000fffffffffffffull);
            const double dk = (cstep + pk) - pk;
            const bool odd_tie = (ea == e_tie) & ((uint32_t)xb & 1u);
            const double t = top - x;
            double q = t * inv_c;
            const double qmax = (double)(N - i);
            q = q > qmax ? qmax : q;
            n = (int)q;
            n -= (fma_exact(-(double)n, dk, t) < 0.0) ? 1 : 0;
            n = n < 0 ? 0 : n;
            n = (can & !odd_tie & (t >= 0.0)) ? n : 0;
            inc = dk;
            // the batch's first state above its binade's floor, its last one below the ceiling (or the wrap threshold); a state at or
            // below the step's own binade (no batch) is within one step of zero: it is the residual of a wrap, counted above
            const double m1 = x - pk;
            const double room = fma_exact(-(double)n, dk, t);
            mg = (can && m1 < mg) ? m1 : mg;
            mg = (can && room < mg) ? room : mg;
        } else if (cstep >= 9.5367431640625e-07) {
            const Batch b = nco_batch(x, cstep, N - i, 4092.0, inv_c);
            n = b.n;
            inc = b.inc;
        } else {
            n = 0;
            inc = 0.0;
        }
        while (next_cp <= i + n && next_cp < N) {
            emit(c, fma_exact((double)(next_cp - i), inc, x), ibit, o.flipped);
            ++c;
            next_cp += R;
        }
        if (in_top && o.tpos < 0 && i + 1 <= N) {  // the state behind this one: by the batch's closed form or by the genuine step
            o.tpos = i + 1;
            o.tx = n >= 1 ? fma_exact(1.0, inc, x) : x + cstep;
        }
        x = fma_exact((double)n, inc, x);
        i += n;
        if (i < N) {
            x = x + cstep;
            ++i;
            if (lean) {  // the state a genuine step lands on: above its binade's floor, below its ceiling
                const uint64_t ub = d2u(x) & 0x7ff0000000000000ull;
                const double fl_ = u2d(ub), m1 = x - fl_, m2 = (fl_ + fl_) - x;
                const double m = m1 < m2 ? m1 : m2;
                const double m4 = x < 4092.0 ? 4092.0 - x : x - 4092.0;
                mg = m < mg ? m : mg;
                mg = m4 < mg ? m4 : mg;
            }
        }
    }
    o.x = x;
    o.ibit = ibit;
    o.margin = mg;
    return o;
}

// An anchor / a claim of the code chain: the wrap at the top of local sample w (w = -1: the epoch's start state, PRE-check) left the
// phase r, the symbol counter ib and the flip flag fl.
struct CodeEvent {
    int w;
    double r;
    int ib, fl;
};

// Ideal-arithmetic guess of the last wrap strictly in front of local sample n (n >= 1): y(m) = x0 + m c is the unwrapped pre-check
// phase before sample m, wrap j fires at the top of the first sample with y >= 4092 j.  Residual snapped to the 2^-41 grid every
// true residual lives on.  Returns the epoch's start when there is none.
GAL_HD CodeEvent code_ideal_anchor(double x0, int ib0, double cstep, int n)
{
    CodeEvent a;
    a.w = -1;
    a.r = x0;
    a.ib = ib0;
    a.fl = 0;
    const double yl = x0 + (double)(n - 1) * cstep;
    const double j = __builtin_floor(yl * (1.0 / 4092.0));
    if (!(j >= 1.0) || !(cstep > 0.0)) return a;
    double w = __builtin_ceil((4092.0 * j - x0) / cstep);
    w = w < 0.0 ? 0.0 : w;
    w = w > (double)(n - 1) ? (double)(n - 1) : w;
    const double res = (x0 + w * cstep) - 4092.0 * j;
    a.w = (int)w;
    a.r = (res + 3072.0) - 3072.0;
    const int ib = ib0 + (int)j;
    a.fl = ib >= 500 ? 1 : 0;
    a.ib = ib >= 500 ? ib - 500 : ib;
    return a;
}

// What the stitch does with leg k once the claim `prev` of the legs in front of it is TRUE: 0 = the leg was walked from that very
// anchor (exact as it stands), 1 = same wrap, residual off by *delta, and the walk's margin covers it: every state of the leg is
// the walked one + *delta, 3 = the same up to sample tpos, with a shift to be read off the true state there from it on (tie-prone
// step, delta an odd multiple of 2^-41, and the walk reached a second state in [2048, 4096): code_leg_upto), 2 = walk it again
// from `prev`.
GAL_HD int code_leg_accept(const CodeEvent &anchor, const CodeEvent &prev, double margin, bool tie_prone, int tpos, double *delta)
{
    *delta = 0.0;
    if (anchor.w != prev.w || anchor.ib != prev.ib || anchor.fl != prev.fl) return 2;
    if (d2u(anchor.r) == d2u(prev.r)) return 0;
    if (anchor.w < 0) return 2;  // (two different start states: cannot happen, the start is given)
    const double dl = prev.r - anchor.r;  // both multiples of 2^-41 below one step: exact
    const double adl = dl < 0.0 ? -dl : dl;
    // (2^-39: the shift behind a tie step is delta +- 2^-41, and the margins are a last place short of the true distances)
    if (!(adl + 1.8189894035458565e-12 /* 2^-39 */ < margin)) return 2;
    *delta = dl;
    const bool odd = ((long long)(dl * 2199023255552.0 /* 2^41 */) & 1LL) != 0;
    return (tie_prone && odd && tpos >= 0) ? 3 : 1;
}

// One leg of the code chain, walked from `anchor`: silently up to the leg's first sample n0 (no checkpoints), then through its nl
// samples, emitting the pre-check state of every chunk start (emit(chunk within the leg, x, ibit, flipped since the epoch start)).
struct CodeLeg {
    CodeEvent claim;  // the last wrap seen (the anchor itself if none): what the next leg must be anchored at
    double x;         // pre-check state after the leg's last sample
    int ibit, fl;
    double margin;
    int tpos;         // sample of the epoch whose pre-check state is the first one behind an addition inside [2048, 4096), -1: none
    double tx;        // ... and that state as walked
};

template <class Emit>
GAL_HD CodeLeg code_leg_walk(const CodeEvent &anchor, double cstep, double inv_c, int n0, int nl, int R, Emit emit)
{
    const int start = anchor.w < 0 ? 0 : anchor.w;
    const int ns = n0 - start;
    const CodeTrack t1 = code_walk_track(anchor.r, anchor.ib, cstep, inv_c, ns, R, ns, [](int, double, int, int) {});
    const int fl1 = anchor.fl | t1.flipped;
    const CodeTrack t2 = code_walk_track(t1.x, t1.ibit, cstep, inv_c, nl, R, 0,
                                         [&](int c, double x, int ib, int fl) { emit(c, x, ib, fl1 | fl); });
    CodeLeg o;
    o.claim = anchor;
    if (t1.last_w >= 0) {
        o.claim.w = start + t1.last_w;
        o.claim.r = t1.last_r;
        o.claim.ib = t1.last_ib;
        o.claim.fl = anchor.fl | t1.last_fl;
    }
    if (t2.last_w >= 0) {
        o.claim.w = n0 + t2.last_w;
        o.claim.r = t2.last_r;
        o.claim.ib = t2.last_ib;
        o.claim.fl = fl1 | t2.last_fl;
    }
    o.x = t2.x;
    o.ibit = t2.ibit;
    o.fl = fl1 | t2.flipped;
    o.margin = t1.margin < t2.margin ? t1.margin : t2.margin;
    o.tpos = t1.tpos >= 0 ? start + t1.tpos : (t2.tpos >= 0 ? n0 + t2.tpos : -1);
    o.tx = t1.tpos >= 0 ? t1.tx : t2.tx;
    return o;
}

// The true state at sample `upto` of the epoch (its pre-check state) from a TRUE anchor in front of it, emitting the checkpoints
// of the chunk starts in [n0, upto) on the way (the leg's own, where upto lies inside it): the stitch's second look at the stretch
// up to CodeLeg::tpos of a tie-prone leg.
template <class Emit>
GAL_HD double code_leg_upto(const CodeEvent &anchor, double cstep, double inv_c, int n0, int upto, int R, Emit emit)
{
    const int start = anchor.w < 0 ? 0 : anchor.w;
    const int s_end = upto < n0 ? upto : n0;
    const CodeTrack t1 = code_walk_track(anchor.r, anchor.ib, cstep, inv_c, s_end - start, R, s_end - start, [](int, double, int, int) {});
    if (upto <= n0) return t1.x;
    const int fl1 = anchor.fl | t1.flipped;
    const CodeTrack t2 = code_walk_track(t1.x, t1.ibit, cstep, inv_c, upto - n0, R, 0,
                                         [&](int c, double x, int ib, int fl) { emit(c, x, ib, fl1 | fl); });
    return t2.x;
}

// ---------------------------------------------------------------------------------------------
// Mean advance per sample of the ROUNDED carrier chain: inside the binade [2^-j, 2^-j+1) every step adds
// RN_g(d) (g the binade's ulp) rather than d, and a phase sweeping (0,1) spends the fraction 2^-j of its
// steps there.  Used only for the first-pass GUESSES of the speculative stitcher: with the systematic part of
// the rounding drift removed, guessed and true wrap residuals differ by a random walk of a few 2^-53 per wrap
// instead of up to 2^-54 per SAMPLE, so almost every leg can be accepted by translation.
GAL_HD double eff_step(double d)
{
    const double ad = d < 0.0 ? -d : d;
    if (!(ad > 1e-12) || !(ad < 0.25)) return d;
    double acc = 0.0, w = 0.5, pk = 0.5;
    for (int j = 1; j <= 40; ++j) {
        if (pk <= ad) break;              // below the step's own binade nothing is rounded
        acc += w * ((ad + pk) - pk);      // RN to the ulp of [pk, 2pk)
        w *= 0.5;
        pk *= 0.5;
    }
    acc += 2.0 * w * ad;                  // the remaining fraction of the sweep adds d itself
    return d < 0.0 ? -acc : acc;
}

// ---------------------------------------------------------------------------------------------
// Ideal-arithmetic prediction of the last wrap at or before local sample `a` of an epoch that starts at
// phase p (|p| < 1) with step d: the unreduced phase p + n*d crosses the k-th integer at
// omega_k = ceil((k - p) / d); the wrap residual is snapped to the 2^-52 grid every true residual lives on.
// Returns false if there is no wrap in [1, a].  Only a GUESS for the speculative stitcher
// (synth_kernels.hip: k_carr_guess / k_walk_carr), which verifies everything bitwise.
GAL_HD bool ideal_last_wrap(double p, double d, int a, int *omega, double *r)
{
    const double xa = p + (double)a * d;
    const double ka = __builtin_trunc(xa);
    if (ka == 0.0 || d == 0.0) return false;
    double w = __builtin_ceil((ka - p) / d);
    w = w > (double)a ? (double)a : w;
    w = w < 1.0 ? 1.0 : w;
    const double res = (p + w * d) - ka;
    *omega = (int)w;
    *r = (res + 1.5) - 1.5;
    return true;
}

}  // namespace galnco
#endif  // GAL_NCO_WALK_H_
