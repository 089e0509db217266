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
    double pe = carr_walk(p, d, N, R, [&](int c, double v) { if (cp) cp[c] = v; });
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
        const Batch b = nco_batch(p, d, N - i, 1.0);
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
    CodeEnd e = code_walk(x, ibit, c, N, R, [&](int k, double v, int ib, int fl) {
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
            pend[e] = carr_walk(pst[e], dstep[e], N, N, [](int, double) {});
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

}  // extern "C"
