// walk_host.cpp -- host build of csrc/nco_walk.h for CPU unit tests (tests/test_walker_cpu.py).
// Not part of the product path: the engine runs these routines on the GPU (synth_kernels.hip).  The
// brute-force functions below are the reference recurrences stepped one sample at a time
// (src/galileo-sdr.cpp:491-507, :528-532) and serve as the ground truth for the closed forms.
#include <cstdint>
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

// Host emulation of the device pipeline k_carr_guess -> (k_walk_carr, k_carr_scan)* for ONE slot.
// dstep[E], flags[E], p0[E], prn[E] (S = 1).  Fills pend_out[E]; returns the number of walk passes
// needed, or -1 if max_passes was not enough.
int galwalk_spec_chain(int E, int N, const int *prn, const uint32_t *flags, const double *p0, const double *dstep,
                       double start0, int max_passes, double *pst_out, double *pend_out, long *walks)
{
    std::vector<double> pst(E, 0.0), pend(E, 0.0);
    std::vector<uint8_t> ver(E, 0), dirty(E, 0);
    carr_guess_slot(0, E, 1, N, prn, flags, p0, dstep, start0, pst.data(), ver.data(), dirty.data());
    long nwalk = 0;
    int pass = 0;
    for (; pass < max_passes; ++pass) {
        for (int e = 0; e < E; ++e) {
            if (prn[e] <= 0 || !dirty[e]) continue;
            pend[e] = carr_walk(pst[e], dstep[e], 1.0 / __builtin_fabs(dstep[e]), N, N, [](int, double) {});
            dirty[e] = 0;
            ++nwalk;
        }
        const int unver = carr_scan_slot(0, E, 1, prn, flags, p0, start0, pst.data(), pend.data(), ver.data(),
                                         dirty.data(), pass == 0);
        if (unver == 0) {
            ++pass;
            break;
        }
    }
    if (walks) *walks = nwalk;
    memcpy(pst_out, pst.data(), sizeof(double) * E);
    memcpy(pend_out, pend.data(), sizeof(double) * E);
    for (int e = 0; e < E; ++e)
        if (prn[e] > 0 && !ver[e]) return -1;
    return pass;
}

// Host emulation of the LEG pipeline (k_carr_guess, k_walk_carr, k_carr_scan of synth_kernels.hip) for
// ONE slot: epoch e is split into W legs of L samples.  Same arithmetic as the kernels, evaluated
// sequentially.  Returns the number of passes (walk + scan) needed, -1 if max_passes was not enough.
// pend_out[E*W]; *walks = total leg walks.
int galwalk_spec_legs(int E, int W, int L, int N, const int *prn, const uint32_t *flags, const double *p0,
                      const double *dstep, double start0, int max_passes, double *pend_out, long *walks,
                      int *unver_hist, int mode)
{
    const int LEGS = E * W;
    std::vector<double> pg(E, 0.0), pst(LEGS, 0.0), pend(LEGS, 0.0);
    std::vector<uint8_t> ver(LEGS, 0), dirty(LEGS, 0);
    {
        double p = 0.0;
        for (int e = 0; e < E; ++e) {
            if (prn[e] <= 0) continue;
            if (flags[e] & 1u) p = p0[e];
            else if (e == 0) p = start0;
            pg[e] = p;
            p = p + (double)N * dstep[e];
            p = p - (double)(long long)p;
        }
    }
    long nwalk = 0;
    int pass = 0;
    for (; pass < max_passes; ++pass) {
        const int first = pass == 0, jacobi = pass == 0;
        for (int i = 0; i < LEGS; ++i) {
            const int e = i / W, w = i % W;
            if (prn[e] <= 0) continue;
            const double d = dstep[e];
            if (first) {
                const double x = pg[e] + (double)(w * L) * d;
                pst[i] = x - (double)(long long)x;
                ver[i] = 0;
            } else if (!dirty[i]) continue;
            int n = N - w * L;
            if (n > L) n = L;
            pend[i] = carr_walk(pst[i], d, 1.0 / __builtin_fabs(d), n, n, [](int, double) {});
            dirty[i] = 0;
            ++nwalk;
        }
        int unver = 0;
        bool c_act = false, c_ver = false;
        double c_D = 0.0;
        for (int i = 0; i < LEGS; ++i) {
            const int e = i / W, w = i % W;
            const bool act = prn[e] > 0;
            if (!act) { c_act = false; c_ver = false; c_D = 0.0; continue; }
            const bool root = w == 0 && (e == 0 || (flags[e] & 1u));
            const double known = (flags[e] & 1u) ? p0[e] : start0;
            const double cur = pst[i];
            const double pprev = i > 0 ? pend[i - 1] : 0.0;
            const bool prev_act = c_act;
            const bool link_ok = !dirty[i] && (root ? d2u(cur) == d2u(known) : (prev_act && d2u(cur) == d2u(pprev)));
            const double G = root ? known - cur : (prev_act ? pprev - cur : 0.0);
            const bool head = root || !prev_act;
            const bool v = head ? link_ok : (link_ok && c_ver);
            double D = head ? G : G + c_D;
            if (jacobi) D = G;
            const double D_prev = c_D;
            if (v) ver[i] = 1;
            else {
                ++unver;
                double Du = D_prev;
                if (mode == 1) Du = (D_prev + 1.5) - 1.5;
                if (mode == 3) Du = (D_prev + 1.5) - 1.5;
                if (mode >= 2 && i > 0) {
                    // predecessor leg in a "tie epoch" (step is an odd multiple of 2^-53: every wrap rounds a
                    // tie, which quantises phase differences to multiples of 2^-51)
                    const double dp = dstep[(i - 1) / W];
                    const double t53 = dp * 9007199254740992.0;  // dp * 2^53, exact
                    const bool tie = (t53 == (double)(long long)t53) && (((long long)t53) & 1LL);
                    if (tie) Du = (D_prev + 3.0) - 3.0;
                }
                const double nstart = root ? known : (prev_act ? pprev + (jacobi ? 0.0 : Du) : cur);
                if (d2u(nstart) != d2u(cur)) { pst[i] = nstart; dirty[i] = 1; }
            }
            c_act = true; c_ver = v; c_D = D;
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
