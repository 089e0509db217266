// walk_host.cpp -- host build of csrc/nco_walk.h for CPU unit tests (tests/test_walker_cpu.py).
// Not part of the product path: the engine runs these routines on the GPU (synth_kernels.hip).  The
// brute-force functions below are the reference recurrences stepped one sample at a time
// (src/galileo-sdr.cpp:491-507, :528-532) and serve as the ground truth for the closed forms.
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
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

// ONE carrier cycle from the wrap residual r (the phase right after a wrap, or any start phase): samples to the next wrap, the
// residual there, the walk's binade margin (every start within +-margin of r takes the same itinerary: same sample count, residual
// shifted by the same amount) and the closed-form iterations it cost.  For tools/carrier_table_prototype.py (DESIGN.md section 9:
// the wrap-residual -> next-residual map as a piecewise translation).  Returns 0 if no wrap happens within n_max samples.
int galwalk_cycle(double r, double d, int n_max, int *n_out, double *r_out, double *margin_out, long *iters_out)
{
    const WalkOut o = carr_walk_track(r, d, 1.0 / __builtin_fabs(d), n_max, n_max + 1, n_max + 1, [](int, double) {});
    if (o.last_w < 0) return 0;
    // (n_max is chosen by the caller so that exactly one wrap lies inside: last_w is the first)
    if (n_out) *n_out = o.last_w;
    if (r_out) *r_out = o.last_r;
    if (margin_out) *margin_out = o.margin;
    if (iters_out) *iters_out = galwalk_carr_iters(r, d, o.last_w);
    return 1;
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

// The code chain of one epoch in `legs` legs, as k_walk_code runs it on the device (synth_kernels.hip): every leg walked from its
// ideal-arithmetic anchor, then the stitch from leg to leg -- accepted as it stands, translated, or walked again from the true
// anchor.  stats[0..3]: legs accepted as walked / translated / walked again / translated across a tie.  force_tie != 0: never translate.
void galwalk_code_legs(double x0, int ib0, double c, int N, int R, int legs, double *cpx, uint32_t *cpi, double *xend, int *ibend,
                       int *flipped, int *stats, int force_tie)
{
    const int nchunks = (N + R - 1) / R;
    const int Lk = (nchunks + legs - 1) / legs;
    const bool tie = code_tie_prone(c);
    std::vector<CodeEvent> anc(legs);
    std::vector<CodeLeg> leg(legs);
    std::vector<int> have(legs, 0);
    auto walk = [&](int k) {
        const int n0 = k * Lk * R;
        const int n1 = std::min(N, (k + 1) * Lk * R);
        leg[k] = code_leg_walk(anc[k], c, 1.0 / c, n0, n1 - n0, R, [&](int ci, double x, int ib, int fl) {
            cpx[k * Lk + ci] = x;
            cpi[k * Lk + ci] = (uint32_t)ib | ((uint32_t)fl << 16);
        });
    };
    for (int k = 0; k < legs; ++k) {  // (on the device: all legs at once)
        if (k * Lk * R >= N) continue;
        have[k] = 1;
        if (k == 0) {
            anc[k].w = -1; anc[k].r = x0; anc[k].ib = ib0; anc[k].fl = 0;
        } else {
            anc[k] = code_ideal_anchor(x0, ib0, c, k * Lk * R);
        }
        walk(k);
    }
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    int last = 0;
    for (int k = 1; k < legs; ++k) {
        if (!have[k]) continue;
        double dl;
        int how = code_leg_accept(anc[k], leg[k - 1].claim, leg[k].margin, tie, leg[k].tpos, &dl);
        if (force_tie && how != 0) how = 2;
        stats[how] += 1;
        const int n0 = k * Lk * R, n1 = std::min(N, (k + 1) * Lk * R);
        if (how == 1) {
            for (int ci = 0; ci * R < n1 - n0; ++ci) cpx[k * Lk + ci] += dl;
            leg[k].x += dl;
            leg[k].claim.r += dl;
        } else if (how == 3) {
            // true state at tpos from the true anchor (the checkpoints in front of it come out of this walk), the rest shifted by
            // what the two trajectories differ by there
            const int tp = leg[k].tpos;
            const double xt = code_leg_upto(leg[k - 1].claim, c, 1.0 / c, n0, tp, R, [&](int ci, double x, int ib, int fl) {
                cpx[k * Lk + ci] = x;
                cpi[k * Lk + ci] = (uint32_t)ib | ((uint32_t)fl << 16);
            });
            const double dl2 = xt - leg[k].tx;
            for (int ci = 0; ci * R < n1 - n0; ++ci)
                if (n0 + ci * R >= tp) cpx[k * Lk + ci] += dl2;
            leg[k].x += dl2;  // (tpos lies inside the walk: the end is behind it)
            if (leg[k].claim.w == anc[k].w) leg[k].claim = leg[k - 1].claim;  // no wrap seen: the anchor's event, true value
            else leg[k].claim.r += dl2;                                      // (the first wrap behind the anchor lies behind tpos)
        } else if (how == 2) {
            anc[k] = leg[k - 1].claim;
            walk(k);
        }
        last = k;
    }
    *xend = leg[last].x;
    *ibend = leg[last].ibit;
    *flipped = leg[last].fl;
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
// Host emulation of the WRAP-ANCHORED leg pipeline (what the GPU runs: k_walk_carr / k_scanm).
// A leg is walked from its ANCHOR = the last wrap event at or before its first sample, (omega, r) with
// "phase before global sample omega is r" (or the chain root), through to its own end; it reports the
// last wrap it saw (its CLAIM) or "none".  The stitcher accepts a leg only if its anchor is bitwise the
// claim chain's value, otherwise re-anchors it at the predicted claim (claim + the claiming leg's own
// anchor correction).  Returns passes needed (-1: not converged within max_passes).
extern "C" {
int galwalk_spec_wrap(int E, int W, int L, int N, const int *prn, const uint32_t *flags, const double *p0,
                                 const double *dstep, double start0, int max_passes, double *pend_out,
                                 long *walks, int *unver_hist, int nthreads, int R, double *cp_out, int translate,
                                 long *shifts)
{
    // R > 0: checkpoints every R samples inside each leg go to cp_out[leg * (L / R) + c] (L % R == 0);
    // translate != 0: legs whose anchor residual merely moved are shifted instead of walked again
    // (synth_kernels.hip: k_walk_carr dirty == 2 / k_scanm phase 3)
    const int LEGS = E * W;
    const int Lc = R > 0 ? L / R : 0;
    const bool use_eff = getenv("GALWALK_NO_EFF") == nullptr;
    std::vector<double> marg(LEGS, 0.0), shift(LEGS, 0.0);
    std::vector<int> tdir(LEGS, 0);
    std::vector<long long> tpos(LEGS, -1);
    long nshift = 0;
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
            const double de = use_eff ? eff_step(dstep[e]) : dstep[e];
            if (ideal_last_wrap(p, de, N, &om, &rr)) { lw = (long long)e * N + om; lr = rr; }
            p = p + (double)N * de;
            p = p - __builtin_trunc(p);
        }
    }
    auto leg_start = [&](int i) { return (long long)(i / W) * N + (long long)(i % W) * L; };
    long nwalk = 0;
    int pass = 0;
    int mode = 0, rewalk = 0;  // k_carr_publish: translation-only pass pending -> its scan is skipped
    for (; pass < max_passes; ++pass) {
        const int first = pass == 0;
        for (int i = 0; i < LEGS; ++i) {
            const int e = i / W, w = i % W;
            if (prn[e] <= 0) continue;
            const long long A = leg_start(i);
            if (first) {  // anchor predicted by ideal arithmetic
                int om;
                double rr;
                if (ideal_last_wrap(pg[e], use_eff ? eff_step(dstep[e]) : dstep[e], w * L, &om, &rr)) { ws[i] = (long long)e * N + om; rs[i] = rr; }
                else { ws[i] = gw[e]; rs[i] = gr[e]; }
                ver[i] = 0;
            } else if (!dirty[i]) continue;
            int n = N - w * L;
            if (n > L) n = L;
            if (!first && dirty[i] == 2) {  // translated acceptance
                const double dl = shift[i];
                const bool flip = tdir[i] != 0 && ((long long)(dl * 4503599627370496.0) & 1LL);
                const double dl2 = flip ? dl - (double)tdir[i] * 2.220446049250313e-16 : dl;
                const long long tp = flip ? tpos[i] : (long long)1 << 62;
                if (cp_out) for (int c = 0; c * R < n; ++c) cp_out[(size_t)i * Lc + c] += (A + (long long)c * R >= tp) ? dl2 : dl;
                pend[i] += dl2;
                if (getenv("GALWALK_DEBUG")) {
                    long long cur = ws[i];
                    double p = rs[i];
                    double mg2 = 4.0;
                    while (cur < A) {
                        const int ec = (int)(cur / N);
                        long long seg_end = (long long)(ec + 1) * N;
                        if (seg_end > A) seg_end = A;
                        const int nn = (int)(seg_end - cur);
                        const WalkOut o = carr_walk_track(p, dstep[ec], 1.0 / __builtin_fabs(dstep[ec]), nn, nn, nn, [](int, double) {});
                        mg2 = std::min(mg2, o.margin);
                        p = o.p;
                        cur = seg_end;
                    }
                    const WalkOut o = carr_walk_track(p, dstep[e], 1.0 / __builtin_fabs(dstep[e]), n, n, n, [](int, double) {});
                    if (d2u(o.p) != d2u(pend[i]))
                        fprintf(stderr, "TRANSLATE MISMATCH leg %d (e %d w %d) anchor %lld r_new %a dl %a marg(before) %a marg_newwalk %a A %lld d %a pend_tr %a pend_walk %a pass %d\n",
                                i, e, w, ws[i], rs[i], dl, marg[i], std::min(mg2, o.margin), A, dstep[e], pend[i], o.p, pass);
                }
                if (hw[i]) rc[i] += (wc[i] >= tp) ? dl2 : dl;
                marg[i] -= __builtin_fabs(dl) + 2.220446049250313e-16;
                if (flip) tdir[i] = -tdir[i];
                dirty[i] = 0;
                ++nshift;
                continue;
            }
            long long cur = ws[i];
            double p = rs[i];
            long long lw = -1;
            double lr = 0.0;
            double mg = 4.0;
            int td = 0;
            long long tpo = -1;
            while (cur < A) {  // anchor -> leg start, epoch by epoch (the step changes at epoch boundaries)
                const int ec = (int)(cur / N);
                long long seg_end = (long long)(ec + 1) * N;
                if (seg_end > A) seg_end = A;
                const int n = (int)(seg_end - cur);
                const double d = dstep[ec];
                const WalkOut o = carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, n, n, [](int, double) {});
                if (o.last_w >= 0) { lw = cur + o.last_w; lr = o.last_r; }
                mg = std::min(mg, o.margin);
                if (!td && o.tdir) { td = o.tdir; tpo = cur + o.tpos; }
                p = o.p;
                cur = seg_end;
            }
            const double d = dstep[e];
            const WalkOut o = (cp_out && R > 0)
                ? carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, R, 0, [&](int c, double v) { cp_out[(size_t)i * Lc + c] = v; })
                : carr_walk_track(p, d, 1.0 / __builtin_fabs(d), n, n, n, [](int, double) {});
            if (o.last_w >= 0) { lw = A + o.last_w; lr = o.last_r; }
            marg[i] = std::min(mg, o.margin);
            if (!td && o.tdir) { td = o.tdir; tpo = A + o.tpos; }
            tdir[i] = td;
            tpos[i] = tpo;
            pend[i] = o.p;
            hw[i] = lw >= 0;
            wc[i] = lw;
            rc[i] = lr;
            dirty[i] = 0;
            ++nwalk;
        }
        if (mode == 1) {  // every pending leg was translated onto the anchor the stitcher predicted
            for (int i = 0; i < LEGS; ++i) ver[i] = 1;
            if (unver_hist) unver_hist[pass] = 0;
            ++pass;
            break;
        }
        int unver = 0;
        rewalk = 0;
        // ---- stitcher (== leg_op / leg_d_out / DMap of synth_kernels.hip)
        struct Lc { int kind; long long w; double r; };
        struct Op { bool act, root, have, link_ok, hw, same; int tdir; long long nw; double base, G; };
        auto leg_op = [&](int i, Lc &lc) {
            Op o = {false, false, false, false, false, false, tdir[i], 0, 0.0, 0.0};
            const int e = i / W, w = i % W;
            o.act = prn[e] > 0;
            if (!o.act) { lc.kind = 2; return o; }
            o.root = w == 0 && (e == 0 || (flags[e] & 1u));
            if (o.root) { lc.kind = 1; lc.w = leg_start(i); lc.r = (flags[e] & 1u) ? p0[e] : start0; }
            o.have = lc.kind == 1;
            o.link_ok = o.have && !dirty[i] && ws[i] == lc.w && d2u(rs[i]) == d2u(lc.r);
            o.nw = lc.w;
            o.base = lc.r;
            o.same = o.have && ws[i] == lc.w;
            o.G = o.same ? lc.r - rs[i] : 0.0;
            o.hw = hw[i] != 0;
            if (o.hw) { lc.w = wc[i]; lc.r = rc[i]; }
            return o;
        };
        auto odd52 = [](double t) { return (int)((long long)(t * 4503599627370496.0) & 1LL); };
        auto tie_flip = [&](double t, int td) { return (td != 0 && odd52(t)) ? t - (double)td * 2.220446049250313e-16 : t; };
        auto d_out = [&](const Op &o, double D) {
            if (!o.act) return 0.0;
            if (o.root) D = 0.0;
            if (!o.hw) return D;
            if (!o.same) return 0.0;
            return tie_flip(o.G + D, o.tdir);
        };
        auto apply_leg = [&](int i, const Op &o, int &allok, double &D) {
            if (!o.act) { allok = 0; D = 0.0; return; }
            if (o.root) { allok = 1; D = 0.0; }
            allok &= o.link_ok ? 1 : 0;
            const double nr = o.base + D;
            if (allok) ver[i] = 1;
            else {
                ++unver;
                if (o.have && (ws[i] != o.nw || d2u(rs[i]) != d2u(nr))) {
                    const double dl = nr - rs[i];
                    const bool tr = translate && ws[i] == o.nw &&
                                    __builtin_fabs(dl) + 8.881784197001252e-16 < marg[i];
                    if (translate && getenv("GALWALK_DEBUG2") && !tr)
                        fprintf(stderr, "REWALK leg %d same %d tdir %d odd %d dl %a marg %a\n", i, (int)(ws[i] == o.nw), tdir[i], odd52(dl), dl, marg[i]);
                    ws[i] = o.nw; rs[i] = nr; dirty[i] = tr ? 2 : 1;
                    if (tr) shift[i] = dl;
                    rewalk += tr ? 0 : 1;
                }
                rewalk += o.have ? 0 : 1;
            }
            D = d_out(o, D);
        };
        if (nthreads <= 0) {
            Lc lc = {0, 0, 0.0};
            int allok = 0;
            double D = 0.0;
            for (int i = 0; i < LEGS; ++i) {
                // leg_op reads the OLD anchors; apply_leg overwrites them afterwards
                const Op o = leg_op(i, lc);
                apply_leg(i, o, allok, D);
            }
        } else {
            const double U = 2.220446049250313e-16;  // 2^-52
            auto resid = [&](double D) { return (int)((long long)(D * 4503599627370496.0) & 3LL); };
            struct Agg { int kind; long long w; double r; int fv, v, ic; double K, c[4]; };
            const int T = nthreads, K = (LEGS + T - 1) / T;
            std::vector<Agg> ag(T);
            for (int t = 0; t < T; ++t) {  // sweep 1
                Agg m = {0, 0, 0.0, 0, 1, 0, 0.0, {0, 0, 0, 0}};
                for (int i = t * K; i < std::min(LEGS, (t + 1) * K); ++i) {
                    const int e = i / W, w = i % W;
                    if (prn[e] <= 0) { m.kind = 2; continue; }
                    if (w == 0 && (e == 0 || (flags[e] & 1u))) { m.kind = 1; m.w = leg_start(i); m.r = (flags[e] & 1u) ? p0[e] : start0; }
                    if (hw[i]) { m.kind = 1; m.w = wc[i]; m.r = rc[i]; }
                }
                ag[t] = m;
            }
            std::vector<Lc> lcin(T);
            {
                Lc run = {0, 0, 0.0};
                for (int t = 0; t < T; ++t) {
                    lcin[t] = run;
                    if (ag[t].kind != 0) run = {ag[t].kind, ag[t].w, ag[t].r};
                }
            }
            for (int t = 0; t < T; ++t) {  // sweep 2: fold allok and the D map on the four residues
                Lc lc = lcin[t];
                int allok = 1, fv = 0, ic = 0;
                double D4[4] = {0.0, U, 2 * U, 3 * U};
                for (int i = t * K; i < std::min(LEGS, (t + 1) * K); ++i) {
                    const Op o = leg_op(i, lc);
                    if (!o.act) { allok = 0; fv = 1; }
                    else { if (o.root) { allok = 1; fv = 1; } allok &= o.link_ok ? 1 : 0; }
                    if (!o.act || o.root || (o.hw && !o.same)) ic = 1;
                    for (int m = 0; m < 4; ++m) D4[m] = d_out(o, D4[m]);
                }
                ag[t].fv = fv; ag[t].v = allok; ag[t].ic = ic; ag[t].K = D4[0];
                for (int m = 0; m < 4; ++m) ag[t].c[m] = D4[m] - m * U;
            }
            // exclusive prefix of (allok, D map), then sweep 3
            int pfv = 0, pv = 1, pic = 0;
            double pK = 0.0, pc[4] = {0, 0, 0, 0};
            for (int t = 0; t < T; ++t) {
                Lc lc = lcin[t];
                int allok = (t > 0 && pfv) ? pv : 0;
                double D = t > 0 ? (pic ? pK : pc[0]) : 0.0;
                for (int i = t * K; i < std::min(LEGS, (t + 1) * K); ++i) {
                    const Op o = leg_op(i, lc);
                    apply_leg(i, o, allok, D);
                }
                // fold thread t into the prefix: (p then ag[t])
                const Agg &b = ag[t];
                const int nfv = pfv | b.fv, nv = b.fv ? b.v : (pv & b.v);
                if (b.ic) { pic = 1; pK = b.K; }
                else if (pic) { pK = pK + b.c[resid(pK)]; }
                else { double nc[4]; for (int m = 0; m < 4; ++m) { const double mid = m * U + pc[m]; nc[m] = pc[m] + b.c[resid(mid)]; } for (int m = 0; m < 4; ++m) pc[m] = nc[m]; }
                pfv = nfv; pv = nv;
            }
        }
        if (unver_hist) unver_hist[pass] = unver;
        if (unver == 0) { ++pass; break; }
        if (rewalk == 0 && translate) mode = 1;
    }
    if (walks) *walks = nwalk;
    if (shifts) *shifts = nshift;
    memcpy(pend_out, pend.data(), sizeof(double) * LEGS);
    for (int i = 0; i < LEGS; ++i)
        if (prn[i / W] > 0 && !ver[i]) return -1;
    return pass;
}
}  // extern "C"
