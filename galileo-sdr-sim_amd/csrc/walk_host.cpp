// walk_host.cpp -- host build of csrc/nco_walk.h for CPU unit tests (tests/test_walker_cpu.py).
// Not part of the product path: the engine runs these routines on the GPU (synth_kernels.hip).  The
// brute-force functions below are the reference recurrences stepped one sample at a time
// (src/galileo-sdr.cpp:491-507, :528-532) and serve as the ground truth for the closed forms.
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <vector>

#include "nco_walk.h"

using namespace galnco;

extern "C" {

// closed-form carrier walk; cp[0..nchunks-1] = phase before sample c*R, returns end phase
double galwalk_carr(double p, double d, int N, int R, double *cp, int *n_iters)
{
    int iters = 0;
    (void)iters;
    double pe = carr_walk(p, d, 1.0 / __builtin_fabs(d), N, R, [&](int c, double v) { if (cp) cp[c] = v; });
    if (n_iters) *n_iters = 0;
    return pe;
}

double galwalk_carr_brute(double p, double d, int N, int R, double *cp)
{
    for (int i = 0; i < N; ++i) {
        if (cp && i % R == 0) cp[i / R] = p;
        p = p + d;
        p = p - (double)(long)p;
    }
    return p;
}

// counts loop iterations of the closed-form walker (cost model / regression guard)
long galwalk_carr_iters(double p, double d, int N)
{
    long it = 0;
    int i = 0;
    while (i < N) {
        ++it;
        const Batch b = nco_batch(p, d, N - i, 1.0, 1.0 / __builtin_fabs(d));
        if (b.n) p = fma_exact((double)b.n, b.inc, p);
        i += b.n;
        if (i < N) {
            p = carr_step(p, d);
            ++i;
        }
    }
    return it;
}

void galwalk_code(double x, int ibit, double c, int N, int R, double *cpx, uint32_t *cpi, double *xend,
                  int *ibend, int *flipped)
{
    CodeEnd e = code_walk(x, ibit, c, 1.0 / c, N, R, [&](int k, double v, int ib, int fl) {
        if (cpx) cpx[k] = v;
        if (cpi) cpi[k] = (uint32_t)ib | ((uint32_t)fl << 16);
    });
    *xend = e.x;
    *ibend = e.ibit;
    *flipped = e.flipped;
}

void galwalk_code_brute(double x, int ibit, double c, int N, int R, double *cpx, uint32_t *cpi, double *xend,
                        int *ibend, int *flipped)
{
    int fl = 0;
    for (int i = 0; i < N; ++i) {
        if (i % R == 0) {
            if (cpx) cpx[i / R] = x;
            if (cpi) cpi[i / R] = (uint32_t)ibit | ((uint32_t)fl << 16);
        }
        if (x >= 4092.0) {
            x -= 4092.0;
            ibit++;
            if (ibit >= 500) {
                ibit = 0;
                fl = 1;
            }
        }
        x = x + c;
    }
    *xend = x;
    *ibend = ibit;
    *flipped = fl;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Host emulation of the WRAP-ANCHORED leg pipeline (what the GPU runs: k_walk_carr / k_carr_scan).
// A leg is walked from its ANCHOR = the last wrap event at or before its first sample, (omega, r) with
// "phase before global sample omega is r" (or the chain root), through to its own end; it reports the
// last wrap it saw (its CLAIM) or "none".  The stitcher accepts a leg only if its anchor is bitwise the
// claim chain's value, otherwise re-anchors it at the predicted claim (claim + the claiming leg's own
// anchor correction).  Returns passes needed (-1: not converged within max_passes).
extern "C" {
int galwalk_spec_wrap(int E, int W, int L, int N, const int *prn, const uint32_t *flags, const double *p0,
                                 const double *dstep, double start0, int max_passes, double *pend_out,
                                 long *walks, int *unver_hist, int nthreads)
{
    const int LEGS = E * W;
    std::vector<double> pg(E, 0.0), rs(LEGS, 0.0), rc(LEGS, 0.0), pend(LEGS, 0.0);
    std::vector<long long> ws(LEGS, 0), wc(LEGS, -1);
    std::vector<uint8_t> ver(LEGS, 0), dirty(LEGS, 0), hw(LEGS, 0);
    std::vector<long long> gw(E, 0);
    std::vector<double> gr(E, 0.0);
    {   // k_carr_guess: ideal phase at every epoch start and the ideal last wrap (or root) before it
        double p = 0.0, lr = 0.0;
        long long lw = 0;
        for (int e = 0; e < E; ++e) {
            if (prn[e] <= 0) continue;
            if ((flags[e] & 1u) || e == 0) { p = (flags[e] & 1u) ? p0[e] : start0; lw = (long long)e * N; lr = p; }
            pg[e] = p;
            gw[e] = lw;
            gr[e] = lr;
            int om;
            double rr;
            if (ideal_last_wrap(p, dstep[e], N, &om, &rr)) { lw = (long long)e * N + om; lr = rr; }
            p = p + (double)N * dstep[e];
            p = p - __builtin_trunc(p);
        }
    }
    auto leg_start = [&](int i) { return (long long)(i / W) * N + (long long)(i % W) * L; };
    long nwalk = 0;
    int pass = 0;
    for (; pass < max_passes; ++pass) {
        const int first = pass == 0;
        for (int i = 0; i < LEGS; ++i) {
            const int e = i / W, w = i % W;
            if (prn[e] <= 0) continue;
            const long long A = leg_start(i);
            if (first) {  // anchor predicted by ideal arithmetic
                int om;
                double rr;
                if (ideal_last_wrap(pg[e], dstep[e], w * L, &om, &rr)) { ws[i] = (long long)e * N + om; rs[i] = rr; }
                else { ws[i] = gw[e]; rs[i] = gr[e]; }
                ver[i] = 0;
            } else if (!dirty[i]) continue;
            long long cur = ws[i];
            double p = rs[i];
            long long lw = -1;
            double lr = 0.0;
            while (cur < A) {  // anchor -> leg start, epoch by epoch (the step changes at epoch boundaries)
                const int ec = (int)(cur / N);
                long long seg_end = (long long)(ec + 1) * N;
                if (seg_end > A) seg_end = A;
                const int n = (int)(seg_end - cur);
                const double d = dstep[ec];
                const WalkOut o = carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, n, n, [](int, double) {});
                if (o.last_w >= 0) { lw = cur + o.last_w; lr = o.last_r; }
                p = o.p;
                cur = seg_end;
            }
            int n = N - w * L;
            if (n > L) n = L;
            const double d = dstep[e];
            const WalkOut o = carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, n, n, [](int, double) {});
            if (o.last_w >= 0) { lw = A + o.last_w; lr = o.last_r; }
            pend[i] = o.p;
            hw[i] = lw >= 0;
            wc[i] = lw;
            rc[i] = lr;
            dirty[i] = 0;
            ++nwalk;
        }
        int unver = 0;
        // one leg of the stitcher's sequential statement (== leg_advance in synth_kernels.hip)
        struct Chain { int kind; long long w; double r; double D; int allok; int fv, fd; };
        auto advance = [&](int i, Chain &c, bool apply) {
            const int e = i / W, w = i % W;
            if (prn[e] <= 0) { c.kind = 2; c.allok = 0; c.D = 0.0; c.fv = 1; c.fd = 1; return; }
            const bool root = w == 0 && (e == 0 || (flags[e] & 1u));
            if (root) { c.kind = 1; c.w = leg_start(i); c.r = (flags[e] & 1u) ? p0[e] : start0; c.D = 0.0; c.allok = 1; c.fv = 1; c.fd = 1; }
            const bool have = c.kind == 1;
            const bool link_ok = have && !dirty[i] && ws[i] == c.w && d2u(rs[i]) == d2u(c.r);
            c.allok &= link_ok ? 1 : 0;
            double Du = c.D;
            {   // tie epochs quantise phase differences to multiples of 2^-51 at every wrap
                long long ea = c.w > 0 ? (c.w - 1) / N : 0;
                ea = ea < E ? ea : E - 1;
                const double dp = dstep[ea];
                const double t53 = dp * 9007199254740992.0;
                const bool tie = (t53 == (double)(long long)t53) && (((long long)t53) & 1LL);
                if (tie) Du = (Du + 3.0) - 3.0;
            }
            const long long nw = c.w;
            const double nr = c.r + Du;
            const bool same = have && ws[i] == nw;
            const double Dleg = same ? nr - rs[i] : 0.0;
            if (apply) {
                if (c.allok) ver[i] = 1;
                else {
                    ++unver;
                    if (have && (ws[i] != nw || d2u(rs[i]) != d2u(nr))) { ws[i] = nw; rs[i] = nr; dirty[i] = 1; }
                }
            }
            if (hw[i]) { c.w = wc[i]; c.r = rc[i]; c.D = Dleg; if (!same) c.fd = 1; }
        };
        if (nthreads <= 0) {
            Chain c = {0, 0, 0.0, 0.0, 0, 0, 0};
            for (int i = 0; i < LEGS; ++i) advance(i, c, true);
        } else {
            // the kernel's three sweeps: K consecutive legs per thread, block scans of the partial results
            const int T = nthreads, K = (LEGS + T - 1) / T;
            std::vector<Chain> agg1(T), agg2(T);
            for (int t = 0; t < T; ++t) {  // sweep 1: last claim
                Chain m = {0, 0, 0.0, 0.0, 0, 0, 0};
                for (int i = t * K; i < std::min(LEGS, (t + 1) * K); ++i) {
                    const int e = i / W, w = i % W;
                    if (prn[e] <= 0) { m.kind = 2; continue; }
                    if (w == 0 && (e == 0 || (flags[e] & 1u))) { m.kind = 1; m.w = leg_start(i); m.r = (flags[e] & 1u) ? p0[e] : start0; }
                    if (hw[i]) { m.kind = 1; m.w = wc[i]; m.r = rc[i]; }
                }
                agg1[t] = m;
            }
            for (int t = 1; t < T; ++t)
                if (agg1[t].kind == 0) agg1[t] = agg1[t - 1];  // inclusive "last one that speaks"
            for (int t = 0; t < T; ++t) {  // sweep 2: fold from neutral carries
                Chain c = {0, 0, 0.0, 0.0, 1, 0, 0};
                if (t > 0) { c.kind = agg1[t - 1].kind; c.w = agg1[t - 1].w; c.r = agg1[t - 1].r; }
                for (int i = t * K; i < std::min(LEGS, (t + 1) * K); ++i) advance(i, c, false);
                agg2[t] = c;
            }
            for (int t = 1; t < T; ++t) {  // inclusive segmented AND / SUM
                Chain a = agg2[t - 1], b = agg2[t];
                Chain r = b;
                r.fv = a.fv | b.fv;
                r.allok = b.fv ? b.allok : (a.allok & b.allok);
                r.fd = a.fd | b.fd;
                r.D = b.fd ? b.D : a.D + b.D;
                agg2[t] = r;
            }
            for (int t = 0; t < T; ++t) {  // sweep 3: apply with the true carries
                Chain c = {0, 0, 0.0, 0.0, 0, 0, 0};
                if (t > 0) {
                    c.kind = agg1[t - 1].kind; c.w = agg1[t - 1].w; c.r = agg1[t - 1].r;
                    c.allok = agg2[t - 1].fv ? agg2[t - 1].allok : 0;
                    c.D = agg2[t - 1].D;
                }
                for (int i = t * K; i < std::min(LEGS, (t + 1) * K); ++i) advance(i, c, true);
            }
        }
        if (unver_hist) unver_hist[pass] = unver;
        if (unver == 0) { ++pass; break; }
    }
    if (walks) *walks = nwalk;
    memcpy(pend_out, pend.data(), sizeof(double) * LEGS);
    for (int i = 0; i < LEGS; ++i)
        if (prn[i / W] > 0 && !ver[i]) return -1;
    return pass;
}
}  // extern "C"
