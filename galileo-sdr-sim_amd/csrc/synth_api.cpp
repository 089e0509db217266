// synth_api.cpp -- C ABI (include/galsynth.h) over the gfx950 kernels in synth_kernels.hip.
// Host-side plumbing only: validation, HBM arena, kernel sequencing.  There is deliberately no CPU
// implementation behind these entry points: without a usable GPU they fail with GAL_E_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <chrono>
#include <sched.h>
#include <time.h>
#include <vector>

#define GAL_SYNTH_NO_SIZED_MACROS 1
#include "nco_walk.h"
#include "synth_dev.h"
#include "e1_tables.inc"

// Smallest distance between two of the 15 thresholds T_u = 1 - frac(u s) of the 16-sample hold pattern (k_synth,
// rw_phase_a): the kernel's bin table (128 bins of the group-start fraction) decides a lane only if its bin holds ONE
// threshold, so every pair must be more than a bin (plus the registration margin) apart -- true for 2.6 MS/s (0.017),
// false where 2.046 MHz / fs is close to a fraction with a denominator below 16 (2.5 MS/s: 9/11, 2.728 MS/s: 3/4).
// ... and, for k_synth_g (whose group-start phase is approximate: 0 and 1 are thresholds of sample 0's own half chip there), the
// smallest distance of a pattern threshold to 0 or 1
static double rw_threshold_edge(double s)
{
    double g = 1.0;
    for (int u = 1; u <= 15; ++u) {
        const double us = (double)u * s;
        const double t = 1.0 - (us - std::floor(us));
        g = std::min(g, std::min(t, 1.0 - t));
    }
    return g;
}

static double rw_threshold_gap(double s)
{
    double T[15];
    for (int u = 1; u <= 15; ++u) {
        const double us = (double)u * s;
        const double t = 1.0 - (us - std::floor(us));
        int i = u - 1;
        while (i > 0 && T[i - 1] > t) { T[i] = T[i - 1]; --i; }
        T[i] = t;
    }
    double g = 1.0;
    for (int i = 1; i < 15; ++i) g = std::min(g, T[i] - T[i - 1]);
    return g;
}
// Form of the resampled windows a code step of cs2 half chips per sample takes (k_synth<.., RW>, k_synth_g<.., MODE>): 1 = holds
// (the reference's 2.6 MS/s), 2 = at most two advances per 16 samples (from 15.4 MS/s), 3 = at most four (from 7.7 MS/s), 4 = any
// pattern of holds (2.77 .. 7.7 MS/s; k_synth_g only: k_synth runs its classic windows there), 0 = none.
// ONE definition for gal_synth_plan's gate and for gal_synth_create's choice of the code objects to load up front.
static int rw_mode_of(double cs2)
{
    return (cs2 >= 0.74 && cs2 < 0.9999) ? 1 : (cs2 >= 0.0083 && cs2 <= 0.133) ? 2 : (cs2 > 0.133 && cs2 <= 0.266) ? 3
           : (cs2 > 0.266 && cs2 < 0.74) ? 4 : 0;
}
static constexpr double kRwMinGap = 1.0 / 128.0 + 4e-6;
// the CBOC mode keeps two bin tables per channel (chip holds, BOC(6,1) half-period parity) of 64 bins each
static constexpr double kRwMinGapCboc = 1.0 / 64.0 + 4e-6;
#ifdef GAL_TEST_HOOKS
// tests/test_walker_cpu.py checks the gate against an independent evaluation (no device needed)
extern "C" double gal_hooks_rw_threshold_gap(double s) { return rw_threshold_gap(s); }
extern "C" double gal_hooks_rw_min_gap(void) { return kRwMinGap; }
#endif

extern "C" {
void galk_warm(hipStream_t st, int signal, double ratio);
void galk_warm_g(hipStream_t st);
int galk_launch_synth_g(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, const uint8_t *act, const int *nact,
                        uint32_t *iq, int e0, int ne, hipStream_t st);
void galk_launch_repair_g(const DevPlan *P, uint32_t *iq, int e0, hipStream_t st);
void galk_touch(hipStream_t st);
void galk_launch_walk_code(const DevPlan *P, hipStream_t st);
void galk_launch_walk_carr(const DevPlan *P, int first, hipStream_t st);
void galk_launch_carr_scan(const DevPlan *P, uint32_t tag, hipStream_t st);
size_t galk_scanm_status_bytes(int S, int legs);
void galk_launch_verify_carr(const DevPlan *P, hipStream_t st);
void galk_launch_verify_code(const DevPlan *P, hipStream_t st);
void galk_launch_pages(const DevPlan *P, hipStream_t st);
void galk_launch_publish(const DevPlan *P, int *h_ctr, void *h_state, uint32_t *h_flag, uint32_t seq, hipStream_t st);
int galk_scanm_blocks(int legs);
size_t galk_scanm_bytes(int S, int legs);
int galk_launch_synth(const DevPlan *P, const DevPlan *Pd, int nch, int accumulate, const uint8_t *act,
                      const int *nact, uint32_t *iq, int e0, int ne, hipStream_t st);
}

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t err__ = (expr);                                                                 \
        if (err__ != hipSuccess)                                                                   \
            return fail(GAL_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__),   \
                        __FILE__, __LINE__);                                                       \
    } while (0)

int16_t g_cos[512], g_sin[512];
bool g_tables_ready = false;

void init_tables()
{
    if (g_tables_ready) return;
    // quarter-wave expansion of the 512-entry table of include/constants.h:216-284
    for (int k = 0; k < 512; ++k) {
        int q;
        if (k < 128) q = kCosQ[k];
        else if (k < 256) q = -kCosQ[255 - k];
        else if (k < 384) q = -kCosQ[k - 256];
        else q = kCosQ[511 - k];
        g_cos[k] = (int16_t)q;
    }
    for (int k = 0; k < 512; ++k) g_sin[k] = g_cos[(k - 128) & 511];
    g_tables_ready = true;
}

constexpr int kKernelMaxChan = 12;  // channels one k_synth launch replays per lane
constexpr size_t kRunHostStagedBytes = 32u << 20;  // gal_synth_run_host: batches up to here land in pinned memory of the handle
constexpr int kActRow = 32;         // bytes per epoch in a channel group's active-position list (synth_common.h: GAL_ACT_ROW)
constexpr int kGroupMaxChan = 24;   // channels one k_synth_g launch takes (BOC(1,1); its wide instances: synth_group.hip)
constexpr int kGroupChunk = 1024;   // k_synth_g: samples per wave iteration = chunk length of its batches (synth_group.hip: SG_CHUNK)
constexpr int kGroupSyms = 64;      // ... symbol masks per channel and epoch (SG_SYMS)
constexpr int kGroupListMin = 1 << 16;  // ... least capacity of the undecided-group list (a 120 s batch lists ~2000 of 19.5 M groups);
                                        // a plan's list holds 0.5 % of its groups + this
constexpr int kVerifyRotation = 8;  // GAL_CFG_VERIFY_SAMPLED: k_verify_carr / k_verify_code re-walk every eighth leg position per batch (default:
                                    // all of them).  Same box, M-SYN12, pipelined step / one handle / k_synth_g beside it, carrier legs only
                                    // (profiles/r05f_verify_ab.log): none 0.974 / 1.230 / 0.842 ms; every leg 1.010 / 1.297 / 0.890; every 4th
                                    // 0.985 / 1.258 / 0.865; 8th 0.979 / 1.238 / 0.846; 16th 0.976 / 1.234 / 0.845
constexpr int kSmallPlanEpochs = 32;  // plans up to here are latency-bound calls (see commit_staged: enqueued passes)
constexpr int kDefaultPasses = 3;   // carrier passes enqueued up front for a NEW plan: walk + stitch (which translates on the spot), two spare --
                                    // no-op launches in front of k_synth when the chain is complete after one, as it is for four fresh
                                    // scenarios in five (tools/fresh_plan_probe.py, 32 seeds of M-SYN12: 26 x 1 pass, 4 x 2, 2 x 3; a batch that
                                    // needs more than were enqueued pays gal_synth_finish's iteration and a SECOND synthesis, 3.4 ms
                                    // instead of 1.9).  A plan that is executed again enqueues what its last execute needed (the chain
                                    // is deterministic); rounds 3-5 kept that count per HANDLE, right for a bench that re-executes one
                                    // resident plan and wrong for a caller with new parameters every batch
}  // namespace

// First guesses of the speculative carrier walk (synth_kernels.hip: k_walk_carr, first pass): per slot and epoch the IDEAL-arithmetic
// phase at the epoch start (mean advance of the rounded chain, nco_walk.h: eff_step) and the ideal last wrap event -- global sample
// index and residual -- at or before that epoch start, or the chain root.  Nothing here has to be exact: the stitcher accepts a leg
// only through bitwise equality with the verified chain; the guesses decide how many legs can be accepted by translation.  Epoch-major
// [E][S] (a prefix computation: a range execute's cut plan sees the same values).  Rounds 1-4 ran this as a kernel (k_carr_guess, a
// block scan per slot: 22 us at the head of the chain every batch waits for); on the host it is ~100 flops per record at plan time.
// One slot's chain of guesses, stepped epoch by epoch inside gal_synth_plan's single pass over the records (round 6; round 5: a
// function of its own over the staged arrays, slot by slot -- a strided second pass, 0.35 of a 1199-epoch plan's 0.9 ms).
struct GuessChain {
    double ph = 0.0;   // guessed phase after the epoch before, in the REFERENCE'S representation: in (-1, 1), keeping its sign until it
                       // crosses zero; a wrap keeps the sign (`p += d; p -= (long)p`, src/galileo-sdr.cpp:531-532)
    int kind = 0;      // last event before this epoch: 0 nothing yet, 1 defined (a wrap or a root), 2 chain broken (idle epoch)
    long long ev_w = 0;
    double ev_r = 0.0;
    // epoch e of the slot: on = the record is active, restart = GAL_CH_RESTART, p0 its start phase (canonical), dstep = fl(f_carr delt),
    // start_in = the carried phase of state_in; writes the record's three guesses
    void step(const int e, const int N, const bool on, const bool restart, const double p0, const double dstep, const double start_in,
              double *pguess, long long *gss_w, double *gss_r)
    {
        if (!on) {  // idle epochs leave the phase alone and break the chain of events (never read; the staging buffer is not cleared)
            kind = 2;
            *pguess = 0.0;
            *gss_w = 0;
            *gss_r = 0.0;
            return;
        }
        const bool reset = restart || e == 0;
        const double d = galnco::eff_step(dstep);
        const double mine = reset ? (restart ? p0 : start_in) : ph;
        // The phase at the END of the epoch.  A phase and a step of different signs (the epochs behind a Doppler sign change: a
        // satellite at culmination) run towards zero, and crossing zero is NOT a wrap -- trunc() of a phase in (-1, 1) is 0 -- so
        // the phase takes the step's sign only once it has crossed; wraps keep it.  (Rounds 1-5 gave the carried phase the sign of
        // the NEXT epoch's step: 0.3 in front of a negative step became -0.7, the device predicted a wrap 0.3 / |d| samples on that
        // never happens, and every leg anchored at it went into a second and third pass: tools/fresh_plan_probe.py, the four seeds of
        // 32 whose Doppler crosses zero.)
        const double full = (double)N * d;
        double y = mine + (full - std::trunc(full));  // (only the fraction matters; keeps the sum small)
        y = y - std::trunc(y);
        const bool mixed = mine != 0.0 && d != 0.0 && ((mine < 0.0) != (d < 0.0));
        const bool crossed = !mixed || std::fabs(full) > std::fabs(mine);
        if (y != 0.0 && d != 0.0) {
            const bool neg = crossed ? d < 0.0 : mine < 0.0;
            if ((y < 0.0) != neg) y += neg ? -1.0 : 1.0;
        }
        ph = y;
        const bool use_root = reset || kind != 1;  // (a chain without a root is rejected by gal_synth_plan)
        *pguess = mine;
        *gss_w = use_root ? (long long)e * N : ev_w;
        *gss_r = use_root ? mine : ev_r;
        // the last event up to the END of this epoch: a wrap inside it (ideal_last_wrap counts the integers the unreduced phase
        // passes in the step's direction -- zero is none), else its root, else what came before
        int om;
        double rr;
        if (galnco::ideal_last_wrap(mine, d, N, &om, &rr)) {
            kind = 1;
            ev_w = (long long)e * N + om;
            ev_r = rr;
        } else if (reset) {
            kind = 1;
            ev_w = (long long)e * N;
            ev_r = mine;
        }
    }
};

// What a plan produces on the host.  gal_synth_plan[_async] fills one of these (and the pinned staging buffer) WITHOUT touching what
// the batch in flight still needs; commit_staged() makes it the handle's plan and enqueues its upload -- at once if nothing is in
// flight, else at the next gal_synth_execute (behind the gal_synth_finish of the batch in flight): plan(k+1) under execute(k) on ONE
// handle (round 6).  Until the commit the device pointers of P are OFFSETS into the arena.
struct StagedPlan {
    DevPlan P{};
    int n_groups = 0, all_first = 0, all_count = 0, n_exact_records = 0, nact_max = 0;
    std::vector<int> group_nch, group_kind;
    std::vector<int64_t> act_prefix;
    size_t o_plan = 0, o_act = 0, o_nact = 0, up_bytes = 0, total = 0;
    size_t o_cpx = 0, o_ancw = 0, zero_end = 0, o_clmw = 0, clmw_bytes = 0, o_scanm = 0, scanm_clear = 0;
    int R = 0, nchunks = 0;
    float ms_plan = 0.0f;
};

struct gal_synth {
    StagedPlan staged;            // the plan made while a batch was in flight (or being committed)
    bool staged_pending = false;  // ... waits for its commit
    gal_synth_cfg_t cfg{};
    int device = 0;
    int n_cu = 256;  // compute units of the device (MI355X: 256)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;  // code-chain walk + page resolution run beside the carrier passes
    // the walker chain runs on the handle's own HIGH-PRIORITY stream: k_synth fills every SIMD's register file,
    // so walker workgroups of the next batch only get on when a synthesis wave retires -- with priority they
    // are first in line then, instead of queueing behind the pending synthesis workgroups of other handles
    hipStream_t walk_stream = nullptr;
    hipEvent_t ev_walk = nullptr;
    hipEvent_t ev_ver = nullptr;  // k_verify_carr done (k_synth_g batches)
    hipEvent_t ev_verc = nullptr; // k_verify_code done (k_synth_g batches; second walker stream)
    hipEvent_t ev_ctr = nullptr;  // the first carrier walk of the batch in flight is done: it resets the batch's counters at its start, and
                                  // k_verify_code, on the other walker stream, must not count a mismatch in front of that
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_prep = nullptr, ev_aux = nullptr;
    hipEvent_t ev_upd = nullptr;  // the plan's upload and memsets are complete (what the walkers of its first execute wait for)

    // tables in HBM
    int *d_lut = nullptr;
    uint32_t *d_str = nullptr;   // [50][512] half-chip streams (2 bits per BOC half chip)
    DevPlan *d_plan = nullptr;   // device copy of P, inside the arena (the hot kernel reads rarely used fields through it)
    char *h_up = nullptr;        // pinned staging buffer of plan(): the arena's upload region, byte for byte
    size_t h_up_bytes = 0;

    // arena for the planned batch
    void *arena = nullptr;
    size_t arena_bytes = 0;
    DevPlan P{};   // the planned batch
    DevPlan Pw{};  // what the walker kernels of the batch in flight see: P cut to the epochs [0, end of the executed range)
    bool planned = false;
    bool executed = false;
    bool in_flight = false;  // execute() enqueued, finish() not yet called
    int n_groups = 0;  // channel groups (each <= kKernelMaxChan) -> synth launches per execute
    std::vector<int> group_nch;
    std::vector<int> group_kind;     // 1: k_synth_g, 0: k_synth (k_synth_g batches: the records that are not fit for it, accumulating)
    int all_first = 0, all_count = 0;  // k_synth_g batches with records of kind 0: the groups that hold ALL records (list-overflow path)
    int n_exact_records = 0;         // records of the batch that take the exact-replay launch behind k_synth_g
    uint8_t *d_act = nullptr;  // [groups][E][kActRow]
    int *d_nact = nullptr;     // [groups][E]
    int nact_max = 0;
    uint32_t *last_iq = nullptr;
    int range_e0 = 0, range_ne = 0;  // epoch range of the last execute
    void *own_iq = nullptr;
    size_t own_iq_bytes = 0;
    void *own_pin = nullptr;  // gal_synth_run_host, small batches: pinned landing buffer of the device->host copy
    size_t own_pin_bytes = 0;
    int *h_ctr = nullptr;  // pinned: [CTR_COUNT] counters, [CTR_COUNT] spare, then the completion flag (k_publish)
    uint32_t *h_flag = nullptr;  // = (uint32_t *)(h_ctr + 2 * CTR_COUNT): sequence number of the last batch whose record is complete
    uint32_t seq = 0;            // sequence number of the batch in flight
    uint32_t scan_tag = 0;       // tag of the last stitch launched (k_scanm's look-back records carry it)
    size_t scanm_off = ~(size_t)0, scanm_clear = 0;  // where the stitch's records lie in the arena, as last cleared
    int64_t legs_walked = 0, legs_translated = 0, n_fallbacks = 0;  // last finish(): carrier legs walked / translated
    std::vector<int64_t> act_prefix;  // [E + 1] active records in the epochs before e (a first walker pass walks W legs of each)
    int64_t first_pass_legs = 0;      // legs walked by the first passes of the batch in flight (every active leg; not counted on the device)
    gal_chan_state_t *h_state = nullptr;  // pinned [S]
    bool state_fetched = false;           // h_state holds the state of the batch in flight
    gal_synth_stats_t stats{};
    int enq_passes = kDefaultPasses;  // carrier passes the next execute enqueues (see kDefaultPasses)
    int small_need = 2;               // ... what the handle's last plan of <= kSmallPlanEpochs epochs needed (its first: one spare)
    // k_synth's resampled-window body: per slot the last code step whose hold-pattern thresholds were examined and
    // their smallest distance (rw_threshold_gap) -- reused only for the identical step
    std::vector<double> rw_s0, rw_g0, rw_e0;
    // ... and, since round 6, for every step within rw_rad of it: the thresholds T_u = 1 - frac(u s) move by u |ds| <= 15 |ds| and no
    // u s crosses an integer while 15 |ds| stays below the distance of every T_u to 0 and 1 (rw_e0), so inside
    // |ds| < min((e0 - min) / 15, (g0 - min) / 14) both gates keep their verdict.  A Doppler moves the step of the reference geometry by
    // 2e-6 of itself, the radius is 6e-4: one evaluation per slot instead of one per record (14 388 x 0.3 us of a 1199-epoch plan)
    std::vector<double> rw_rad;
    bool host_only = false;       // GAL_TEST_HOOKS, gal_hooks_plan_host_ms: gal_synth_plan's host work without a device (timing on CPU)
    hipEvent_t ev_up0 = nullptr, ev_up1 = nullptr;  // around the upload of the last plan (timed: stats.ms_h2d)
    bool upload_pending = false;  // gal_synth_plan_async: the upload is enqueued, nobody has waited for it yet
    bool upload_timed = false;
    bool upload_unordered = false;  // ... and no execute has put its walkers behind it yet
    float ms_plan = 0.0f;         // host time of the last gal_synth_plan (validation, lists, staging; without the wait for the upload)
    double prev_wait_us = 0.0;  // how long the last gal_synth_finish waited for its batch (paces the next one's naps)
    int g_holdoff = 0;  // batches for which k_synth_g is not used although it could be: its last batch listed too many groups
};

// The stream the handle works on: the caller's (gal_synth_set_stream), or one of its own, made at first need.
static hipStream_t handle_stream(gal_synth *h)
{
    if (!h->stream) {
        if (!h->own_stream && hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        h->stream = h->own_stream;
    }
    return h->stream;
}

extern "C" {

#ifdef GAL_TEST_HOOKS
// libgalsynth_hooks.so: the same sources with the fault-injection hooks of the repair-path tests compiled in
// (GAL_WALK_LEGS / GAL_WALK_TRANSLATE / GAL_WALK_PASSES environment variables).  Never shipped, never benchmarked.
const char *gal_synth_version(void) { return "galsynth 0.4 (gfx950, HIP) +testhooks"; }
#else
const char *gal_synth_version(void) { return "galsynth 0.4 (gfx950, HIP)"; }
#endif
const char *gal_synth_last_error(void) { return g_err; }

int gal_synth_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strstr(prop.gcnArchName, "gfx950")) ++ok;
    }
    return ok;
}

const uint32_t *gal_tables_e1b(void) { return &kE1B[0][0]; }
const uint32_t *gal_tables_e1c(void) { return &kE1C[0][0]; }
const int16_t *gal_tables_cos512(void) { init_tables(); return g_cos; }
const int16_t *gal_tables_sin512(void) { init_tables(); return g_sin; }
uint32_t gal_tables_cs25(void) { return kCS25; }

// GAL_CREATE_TIMING=1: where gal_synth_create spends its time (stderr)
static void create_stage(const char *what)
{
    static const bool on = getenv("GAL_CREATE_TIMING") != nullptr;
    if (!on) return;
    static thread_local auto tl = std::chrono::steady_clock::now();
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[create] %-34s +%7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tl).count());
    tl = t;
}

// GAL_PLAN_TIMING=1: where gal_synth_plan spends its host time (stderr)
static void plan_stage(const char *what)
{
    static const bool on = getenv("GAL_PLAN_TIMING") != nullptr;
    if (!on) return;
    static thread_local auto tl = std::chrono::steady_clock::now();
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[plan] %-34s +%8.1f us\n", what, std::chrono::duration<double, std::micro>(t - tl).count());
    tl = t;
}

int gal_synth_create(const gal_synth_cfg_t *cfg, gal_synth_t **out)
{
    create_stage("enter");
    if (!cfg || !out) return fail(GAL_E_INVAL, "gal_synth_create: null argument");
    *out = nullptr;
    if (!(cfg->sample_rate > 0.0) || cfg->samples_per_epoch < 4 || cfg->n_slots < 1 ||
        cfg->n_slots > GAL_ENGINE_MAX_CHAN)
        return fail(GAL_E_INVAL, "gal_synth_create: bad cfg (rate %g, samples/epoch %d, slots %d)",
                    cfg->sample_rate, cfg->samples_per_epoch, cfg->n_slots);
    if (cfg->chunk_samples < 0 || (cfg->chunk_samples & 3))
        return fail(GAL_E_INVAL, "gal_synth_create: chunk_samples must be a multiple of 4");
    init_tables();

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(GAL_E_DEVICE, "no HIP device: the synthesis engine has no CPU fallback");
    create_stage("hipGetDeviceCount (HIP start-up)");
    int dev = cfg->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(GAL_E_INVAL, "device %d out of range (%d devices)", dev, ndev);
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(GAL_E_DEVICE, "device %d is %s; this library carries gfx950 code only", dev,
                    prop.gcnArchName);

    create_stage("set device + properties");
    gal_synth *h = new (std::nothrow) gal_synth();
    if (!h) return fail(GAL_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->device = dev;
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto bail = [&](int code) {
        gal_synth_destroy(h);
        return code;
    };
    // Streams cost 7-17 ms each to create (a hardware queue apiece): the handle's own stream is made when a batch is planned
    // without the caller having set one (handle_stream), the walker streams here -- two high-priority ones; a plain second
    // stream only where priorities are not to be had or not wanted (GAL_CFG_SINGLE_STREAM)
    {
        bool have = false;
        if (!(cfg->flags & GAL_CFG_SINGLE_STREAM)) {
            int least = 0, greatest = 0;
            hipStream_t ws = nullptr, as = nullptr;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
                hipStreamCreateWithPriority(&ws, hipStreamNonBlocking, greatest) == hipSuccess &&
                hipStreamCreateWithPriority(&as, hipStreamNonBlocking, greatest) == hipSuccess &&
                hipEventCreateWithFlags(&h->ev_walk, hipEventDisableTiming) == hipSuccess) {
                h->walk_stream = ws;
                h->aux_stream = as;
                have = true;
            } else {  // no stream priorities here: everything stays on the caller's stream (still correct)
                (void)hipGetLastError();
                if (ws) hipStreamDestroy(ws);
                if (as) hipStreamDestroy(as);
            }
        }
        if (!have && hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess)
            return bail(fail(GAL_E_DEVICE, "hipStreamCreate failed"));
    }
    create_stage("streams");
    for (auto &e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(fail(GAL_E_DEVICE, "hipEventCreate failed"));
    if (hipEventCreate(&h->ev_up0) != hipSuccess || hipEventCreate(&h->ev_up1) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_prep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_upd, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ver, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_verc, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_ctr, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_aux, hipEventDisableTiming) != hipSuccess)
        return bail(fail(GAL_E_DEVICE, "hipEventCreate failed"));
    // (coherent = fine-grained: k_publish writes both from the device while the host polls the flag behind h_ctr)
    if (hipHostMalloc((void **)&h->h_ctr, (2 * CTR_COUNT + 16) * sizeof(int), hipHostMallocCoherent) != hipSuccess ||
        hipHostMalloc((void **)&h->h_state, sizeof(gal_chan_state_t) * GAL_ENGINE_MAX_CHAN,
                      hipHostMallocCoherent) != hipSuccess)
        return bail(fail(GAL_E_NOMEM, "pinned host allocation failed"));
    h->h_flag = (uint32_t *)(h->h_ctr + 2 * CTR_COUNT);
    *h->h_flag = 0;

    create_stage("events + pinned record");
    if (hipMalloc((void **)&h->d_lut, 2 * 512 * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&h->d_str, 50 * 512 * sizeof(uint32_t)) != hipSuccess)
        return bail(fail(GAL_E_NOMEM, "table allocation failed"));
    {
        // per PRN 8184 BOC half chips x 2 bits: bit 2h = (E1B ^ E1C) chip, bit 2h+1 = E1C chip ^ (h & 1)
        // (synth_kernels.hip, ChanGroup); the upper half of word 511 is unused
        std::vector<uint32_t> str(50 * 512, 0u);
        for (int prn = 0; prn < 50; ++prn)
            for (int hc = 0; hc < 2 * GAL_CODE_LEN; ++hc) {
                const int chip = hc >> 1;
                const uint32_t b = (kE1B[prn][chip >> 5] >> (chip & 31)) & 1u, c = (kE1C[prn][chip >> 5] >> (chip & 31)) & 1u;
                const uint32_t two = (b ^ c) | ((c ^ (uint32_t)(hc & 1)) << 1);
                str[prn * 512 + (hc >> 4)] |= two << (2 * (hc & 15));
            }
        if (hipMemcpy(h->d_str, str.data(), str.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(GAL_E_DEVICE, "table upload failed"));
    }
    int lut[2 * 512];
    // v = E1B d - E1C s is 0 or +-2 (src/galileo-sdr.cpp:520-525): the factor 2 lives in the table, the kernel
    // multiplies by v / 2.  Entry = the int16 pair (2 cos, 2 sin), low half first: the layout of an output sample.
    auto pair = [](int c, int s) {
        return (int)(((uint32_t)(uint16_t)(int16_t)(2 * s) << 16) | (uint32_t)(uint16_t)(int16_t)(2 * c));
    };
    if (cfg->flags & GAL_CFG_CBOC) {
        // CBOC(6,1,1/11): TA = lround(alpha LUT), TB = lround(beta LUT) (the same expressions as the oracle's)
        const double alpha = std::sqrt(10.0 / 11.0), beta = std::sqrt(1.0 / 11.0);
        for (int k = 0; k < 512; ++k) {
            lut[k] = pair((int)std::lround(alpha * (double)g_cos[k]), (int)std::lround(alpha * (double)g_sin[k]));
            lut[512 + k] = pair((int)std::lround(beta * (double)g_cos[k]), (int)std::lround(beta * (double)g_sin[k]));
        }
    } else {
        for (int k = 0; k < 512; ++k) lut[k] = lut[512 + k] = pair(g_cos[k], g_sin[k]);
    }
    if (hipMemcpy(h->d_lut, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(GAL_E_DEVICE, "table upload failed"));
    create_stage("tables (hipMalloc + uploads)");
    // code-object load now, not inside the first batch (the families this configuration can launch)
    galk_warm(nullptr, (cfg->flags & GAL_CFG_CBOC) ? 1 : 0, 2.0 * 1.023e6 / cfg->sample_rate);
    {
        // k_synth_g's code object, if a batch of this configuration can take it: the plan's own gate (rw_mode_of) on the nominal code step and on the steps +-1e-4 around it (Doppler moves a channel's step by a
        // few 1e-6 of itself)
        const double ratio = 2.0 * 1.023e6 / cfg->sample_rate;
        bool g_possible = false;
        for (const double f : {1.0 - 1e-4, 1.0, 1.0 + 1e-4}) {
            const int m = rw_mode_of(ratio * f);
            g_possible = g_possible || m != 0;
        }
        if (g_possible && !(cfg->flags & GAL_CFG_EXACT_REPLAY) && cfg->chunk_samples <= 0) galk_warm_g(nullptr);
    }
    create_stage("warm launches enqueued");
    // First use of the handle's own streams: HIP creates a stream's hardware queue at its first use, and which queues
    // the walker streams get -- their own, or one shared with streams other libraries created in the meantime -- decides
    // how well the chain runs beside another handle's synthesis ("Hardware queues").  From here on they are
    // fixed: a caller who creates its handles before it initialises RCCL & co. keeps them to itself.
    if (h->walk_stream) {
        galk_touch(h->walk_stream);
        galk_touch(h->aux_stream);
        if (hipStreamSynchronize(h->walk_stream) != hipSuccess || hipStreamSynchronize(h->aux_stream) != hipSuccess) {
            gal_synth_destroy(h);
            return fail(GAL_E_DEVICE, "kernel launch failed on device %d: %s", dev, hipGetErrorString(hipGetLastError()));
        }
    }
    if (hipStreamSynchronize(nullptr) != hipSuccess) {
        gal_synth_destroy(h);
        return fail(GAL_E_DEVICE, "kernel launch failed on device %d: %s", dev, hipGetErrorString(hipGetLastError()));
    }
    create_stage("code objects loaded, streams used");
#ifdef GAL_TEST_HOOKS
    if (const char *env = getenv("GAL_SCAN_TAG0")) h->scan_tag = (uint32_t)strtoul(env, nullptr, 0);  // the tag wrap, within reach of a test
#endif
    *out = h;
    return GAL_OK;
}

int gal_synth_destroy(gal_synth_t *h)
{
    if (!h) return GAL_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->arena) hipFree(h->arena);
    if (h->own_iq) hipFree(h->own_iq);
    if (h->own_pin) hipHostFree(h->own_pin);
    if (h->d_lut) hipFree(h->d_lut);
    if (h->d_str) hipFree(h->d_str);
    if (h->h_up) hipHostFree(h->h_up);
    if (h->h_ctr) hipHostFree(h->h_ctr);
    if (h->h_state) hipHostFree(h->h_state);
    for (auto &e : h->ev)
        if (e) hipEventDestroy(e);
    if (h->ev_prep) hipEventDestroy(h->ev_prep);
    if (h->ev_upd) hipEventDestroy(h->ev_upd);
    if (h->ev_up0) hipEventDestroy(h->ev_up0);
    if (h->ev_up1) hipEventDestroy(h->ev_up1);
    if (h->ev_aux) hipEventDestroy(h->ev_aux);
    if (h->ev_walk) hipEventDestroy(h->ev_walk);
    if (h->ev_ver) hipEventDestroy(h->ev_ver);
    if (h->ev_verc) hipEventDestroy(h->ev_verc);
    if (h->ev_ctr) hipEventDestroy(h->ev_ctr);
    if (h->walk_stream) hipStreamDestroy(h->walk_stream);
    if (h->aux_stream) hipStreamDestroy(h->aux_stream);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return GAL_OK;
}

int gal_synth_set_stream(gal_synth_t *h, void *hip_stream)
{
    if (!h) return fail(GAL_E_INVAL, "null handle");
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;  // (null: handle_stream makes one at first need)
    return GAL_OK;
}

size_t gal_synth_output_bytes(const gal_synth_t *h)
{
    if (!h) return 0;
    if (h->staged_pending) return (size_t)h->staged.P.E * (size_t)h->staged.P.N * 4u;  // (the plan the next execute runs)
    if (!h->planned) return 0;
    return (size_t)h->P.E * (size_t)h->P.N * 4u;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The staged plan becomes the handle's: arena (grown if need be), device pointers, the ONE host->device copy out of the pinned
// staging buffer, the memsets of what must start clean.  Everything the device needs is laid out in the staging buffer exactly as
// in the arena.  (Round 1 issued 4 pageable copies and 11 memsets one by one: 0.2 ms per plan.  Round 6: one pass over the caller's
// records fills the buffer -- no memset of it, no copy of the 176-byte records --, and the upload is not waited for unless the
// caller asked: gal_synth_plan is resident on return, gal_synth_plan_async's upload is waited for by the walkers on the device.)
static int commit_staged(gal_synth *h, const bool wait)
{
    StagedPlan &sp = h->staged;
    if (!h->host_only && sp.total > h->arena_bytes) {
        if (h->arena) hipFree(h->arena);
        h->arena = nullptr;
        h->arena_bytes = 0;
        if (hipMalloc(&h->arena, sp.total) != hipSuccess) return fail(GAL_E_NOMEM, "hipMalloc of %zu bytes failed", sp.total);
        h->arena_bytes = sp.total;
        h->scanm_off = ~(size_t)0;  // (new memory: the stitch's records have to be cleared)
    }
    char *const base = (char *)h->arena;
    DevPlan &P = sp.P;
    {
        // offsets -> addresses (every pointer of the plan into the arena; lut and str are the handle's tables, absolute already)
        auto rb = [&](auto *&ptr) { ptr = (std::remove_reference_t<decltype(ptr)>)(base + (size_t)(uintptr_t)ptr); };
        rb(P.page_init); rb(P.init_ix); rb(P.state_in); rb(P.state_out); rb(P.prn); rb(P.flags); rb(P.ib0); rb(P.x0); rb(P.p0);
        rb(P.cstep); rb(P.dstep); rb(P.page_next); rb(P.page_cur); rb(P.flip_in); rb(P.pguess); rb(P.gss_w); rb(P.gss_r);
        rb(P.anc_w); rb(P.anc_r); rb(P.clm_w); rb(P.clm_r); rb(P.pend); rb(P.verified); rb(P.dirty); rb(P.risk); rb(P.marg);
        rb(P.shift); rb(P.tpos); rb(P.tdir); rb(P.scanm); rb(P.cp_x); rb(P.cp_p); rb(P.cp_ib); rb(P.ctr); rb(P.gflist);
    }
    h->P = P;
    h->d_plan = (DevPlan *)(base + sp.o_plan);
    h->d_act = (uint8_t *)(base + sp.o_act);
    h->d_nact = (int *)(base + sp.o_nact);
    h->n_groups = sp.n_groups; h->all_first = sp.all_first; h->all_count = sp.all_count; h->n_exact_records = sp.n_exact_records;
    h->group_nch = std::move(sp.group_nch);
    h->group_kind = std::move(sp.group_kind);
    h->act_prefix = std::move(sp.act_prefix);
    h->nact_max = sp.nact_max;
    h->ms_plan = sp.ms_plan;
    h->staged_pending = false;
    h->planned = false;
    h->executed = false;
    char *const up = h->h_up;
    memcpy(up + sp.o_plan, &h->P, sizeof(DevPlan));
    if (!h->host_only) {
        hipStream_t st_up = handle_stream(h);
        if (!st_up) return fail(GAL_E_DEVICE, "hipStreamCreate failed");
        HIP_TRY(hipEventRecord(h->ev_up0, st_up));
        HIP_TRY(hipMemcpyAsync(base, up, sp.up_bytes, hipMemcpyHostToDevice, st_up));
        HIP_TRY(hipEventRecord(h->ev_up1, st_up));  // (the staging buffer is free from here on; the copy is what ms_h2d times)
        h->upload_timed = true;
        // what must start at zero: the leg records that a stitch may read before a walk has written them.  The checkpoint arrays
        // (98 MB of a 1199-epoch plan, 7 GB of config 4's) need no clearing: every entry a kernel USES -- those of active records
        // of the executed epochs -- is written by the walkers of the same execute first; idle positions alias a slot whose values
        // are read and dropped.  (Rounds 1-5 cleared them with every plan: ~0.05 ms of fill kernels per fresh M-SYN12 plan beside
        // the other handle's synthesis.  Checked by running the GPU suite on a build that fills them with NaN bit patterns
        // instead: GAL_TEST_HOOKS, GAL_ARENA_POISON=1.)
#ifdef GAL_TEST_HOOKS
        if (getenv("GAL_ARENA_POISON")) HIP_TRY(hipMemsetAsync(base + sp.o_cpx, 0xff, sp.o_ancw - sp.o_cpx, st_up));
#endif
        HIP_TRY(hipMemsetAsync(base + sp.o_ancw, 0, sp.zero_end - sp.o_ancw, st_up));
        HIP_TRY(hipMemsetAsync(base + sp.o_clmw, 0xff, sp.clmw_bytes, st_up));
        // (the stitch's tickets and look-back records: no word there may look like a tag this handle is still going to hand out.
        // Its own records never do -- tags only grow -- so this is for a region that held something else: a new layout)
        if (h->scanm_off != sp.o_scanm || h->scanm_clear != sp.scanm_clear) {
            HIP_TRY(hipMemsetAsync(base + sp.o_scanm, 0, sp.scanm_clear, st_up));
            h->scanm_off = sp.o_scanm;
            h->scanm_clear = sp.scanm_clear;
        }
        // what the walker streams of the next execute wait for (they do not wait for the caller's stream otherwise)
        HIP_TRY(hipEventRecord(h->ev_upd, st_up));
        h->upload_pending = true;
        h->upload_unordered = true;
        if (wait) {
            HIP_TRY(hipStreamSynchronize(st_up));
            h->upload_pending = false;
            h->upload_unordered = false;
        }
    }
    memset(&h->stats, 0, sizeof(h->stats));
    h->stats.n_epochs = h->P.E;
    h->stats.n_active_max = h->nact_max;
    h->stats.chunk_samples = sp.R;
    h->stats.chunks_per_epoch = sp.nchunks;
    // carrier passes enqueued up front: kDefaultPasses for a new plan -- but a plan of a few epochs (one-epoch calls, INTEGRATION.md
    // option B) is all latency, every no-op pass is two more launches in front of its synthesis, and no random small batch needs
    // more than one pass (tools/find_multi_pass_batch.py: 0 of 3000): such plans enqueue what the handle's last small batch needed
    h->enq_passes = h->P.E <= kSmallPlanEpochs ? std::max(1, h->small_need) : kDefaultPasses;
    h->planned = true;
    return GAL_OK;
}

// wait: return when the batch is resident in HBM (gal_synth_plan); else as soon as its upload is enqueued (gal_synth_plan_async)
static int plan_impl(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs, const gal_chan_state_t *state_in, const bool wait)
{
    if (h && h->in_flight && wait)
        return fail(GAL_E_STATE, "gal_synth_plan while a batch is in flight: call gal_synth_finish first (gal_synth_plan_async may be "
                                 "called with a batch in flight: its upload then waits for the next gal_synth_execute)");
    if (!h || !params || n_epochs < 1) return fail(GAL_E_INVAL, "gal_synth_plan: bad argument");
    const auto t_plan0 = std::chrono::steady_clock::now();
    plan_stage("enter");
    if (!h->host_only) HIP_TRY(hipSetDevice(h->device));
    const int E = n_epochs, S = h->cfg.n_slots, N = h->cfg.samples_per_epoch;
    h->staged_pending = false;  // (a staged plan that was never executed is replaced)
    StagedPlan sp;

    // ---- the upload region's fixed part (sizes that depend on E and S only), so that ONE pass over the caller's records can validate
    // them, list the active ones and write the SoA copies, the NCO steps and the carrier guesses (round 5: three passes and a copy of
    // the records); the variable part -- page_init table, active lists -- is laid out behind it once the pass knows their sizes
    const size_t ES = (size_t)E * S;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    const size_t o_plan = take(sizeof(DevPlan));
    const size_t o_state_in = take(sizeof(gal_chan_state_t) * S);
    // SoA copies of the records and the NCO steps (no kernel in front of the walker chain)
    const size_t o_prn = take(ES * 4), o_flags = take(ES * 4), o_ib0 = take(ES * 4);
    const size_t o_x0 = take(ES * 8), o_p0 = take(ES * 8), o_cstep = take(ES * 8), o_dstep = take(ES * 8);
    const size_t o_pnext = take(ES * GAL_PAGE_WORDS * 4);
    // first guesses of the speculative carrier walk (GuessChain: host, O(E * S); rounds 1-4: a kernel in front of the chain)
    const size_t o_pguess = take(ES * 8), o_gssw = take(ES * 8), o_gssr = take(ES * 8);
    // (the 176-byte records themselves stay on the host: beside the SoA copies the device only ever read page_init of the records
    // that (re)allocate a channel -- a compact table and an index row; rounds 1-5 uploaded all of them, 3.4 of 6.2 MB)
    const size_t o_initix = take(ES * 4);
    const size_t o_pinit = off;  // [restart records][16]; then the active lists
    {
        const size_t max_groups = 3 * (size_t)((S + kKernelMaxChan - 1) / kKernelMaxChan) + 2;
        const size_t need = o_pinit + align_up(ES * GAL_PAGE_WORDS * 4, 256) + max_groups * (align_up((size_t)E * kActRow, 256) + align_up((size_t)E * 4, 256)) + 4096;
        // (the staging buffer is free again once the upload of the plan before has been read out of it)
        if (h->upload_pending) {
            HIP_TRY(hipEventSynchronize(h->ev_up1));
            h->upload_pending = false;
        }
        if (need > h->h_up_bytes) {
            if (h->host_only) free(h->h_up);
            else if (h->h_up) hipHostFree(h->h_up);
            h->h_up = nullptr;
            h->h_up_bytes = 0;
            const size_t cap = need + need / 4;
            if (h->host_only) h->h_up = (char *)malloc(cap);
            else if (hipHostMalloc((void **)&h->h_up, cap, hipHostMallocDefault) != hipSuccess) h->h_up = nullptr;
            if (!h->h_up) return fail(GAL_E_NOMEM, "pinned staging allocation of %zu bytes failed", cap);
            h->h_up_bytes = cap;
        }
    }
    char *const up = h->h_up;
    int *const u_prn = (int *)(up + o_prn), *const u_ib0 = (int *)(up + o_ib0), *const u_initix = (int *)(up + o_initix);
    uint32_t *const u_flags = (uint32_t *)(up + o_flags), *const u_pnext = (uint32_t *)(up + o_pnext), *const u_pinit = (uint32_t *)(up + o_pinit);
    double *const u_x0 = (double *)(up + o_x0), *const u_p0 = (double *)(up + o_p0);
    double *const u_cstep = (double *)(up + o_cstep), *const u_dstep = (double *)(up + o_dstep);
    double *const u_pguess = (double *)(up + o_pguess), *const u_gssr = (double *)(up + o_gssr);
    long long *const u_gssw = (long long *)(up + o_gssw);
    gal_chan_state_t *const u_state = (gal_chan_state_t *)(up + o_state_in);
    if (state_in) memcpy(u_state, state_in, sizeof(gal_chan_state_t) * S);
    else memset(u_state, 0, sizeof(gal_chan_state_t) * S);
    for (int i = 0; i < S; ++i)
        if (u_state[i].carr_phase == 0.0) u_state[i].carr_phase = 0.0;  // -0.0 canonicalised to +0.0 (see carr_step in nco_walk.h)
    std::vector<GuessChain> guess(S);

    // ---- validate the batch, build the active-channel lists, split the records (host, O(E*S), one pass)
    std::vector<uint8_t> act_all((size_t)E * S, 0);
    std::vector<int> nact_all(E, 0);
    int nact_max = 0;
    std::vector<int> cur_prn(S, 0);
    for (int s = 0; s < S; ++s) cur_prn[s] = (state_in && state_in[s].prn > 0) ? state_in[s].prn : 0;
    // What every RECORD (channel-epoch) could run on, decided per record since round 5 (rounds 2-4: per batch -- one still
    // carrier sent all twelve channels to the exact-replay kernel; the reference treats channels independently,
    // src/galileo-sdr.cpp:487-534):
    //   rec_mode  the form of the resampled windows its code step takes (rw_mode_of) if its pattern thresholds are more than a bin
    //             apart from each other (k_synth's fast body needs that), else 0
    //   rec_g     ... and k_synth_g in that form: thresholds also a bin away from 0 and 1 (its group-start phase is approximate), and
    //             a carrier step of 0 or in [2^-40, 120 / (16 x 511)] cycles per sample -- at most 120 table entries per group (the
    //             table's extension behind a wrap), and a phase that either moves or stands still for good: one that creeps past an
    //             index boundary would have thousands of groups in a row listed for the exact replay
    std::vector<uint8_t> rec_mode((size_t)E * S, 0), rec_gm((size_t)E * S, 0), rec_gb((size_t)E * S, 0);
    double cs2_max = 0.0;  // largest code step of the batch, half chips per sample
    if ((int)h->rw_s0.size() != S) { h->rw_s0.assign(S, 0.0); h->rw_g0.assign(S, 0.0); h->rw_e0.assign(S, 0.0); h->rw_rad.assign(S, 0.0); }
    int n_restart = 0;  // records with GAL_CH_RESTART: their page_init goes up in a compact table
    const double delt = 1.0 / h->cfg.sample_rate;
    const bool cboc = (h->cfg.flags & GAL_CFG_CBOC) != 0;
    const double min_gap = cboc ? kRwMinGapCboc : kRwMinGap;
    for (int e = 0; e < E; ++e) {
        int n = 0;
        for (int s = 0; s < S; ++s) {
            const size_t i = (size_t)e * S + s;
            const gal_chan_epoch_t &r = params[i];
            {
                // the SoA copies (every record, idle ones too: the device reads prn of all).  src/galileo-sdr.cpp:528,531: the product
                // is rounded to double before it is added -- one IEEE multiplication, the same bits on the host as in the reference's loop
                const bool on = r.prn > 0, restart = on && (r.flags & GAL_CH_RESTART);
                u_prn[i] = r.prn;
                u_flags[i] = r.flags;
                u_ib0[i] = r.ibit0;
                u_x0[i] = r.code_phase0;
                u_p0[i] = r.carr_phase0 == 0.0 ? 0.0 : r.carr_phase0;
                u_cstep[i] = r.f_code * delt;
                u_dstep[i] = r.f_carr * delt;
                memcpy(u_pnext + i * GAL_PAGE_WORDS, r.page_next, sizeof(r.page_next));
                u_initix[i] = restart ? n_restart : 0;
                if (restart) memcpy(u_pinit + (size_t)n_restart * GAL_PAGE_WORDS, r.page_init, sizeof(r.page_init));
                guess[s].step(e, N, on, (r.flags & GAL_CH_RESTART) != 0, u_p0[i], u_dstep[i], u_state[s].carr_phase, u_pguess + i, u_gssw + i,
                              u_gssr + i);
            }
            if (r.prn <= 0) {
                cur_prn[s] = 0;
                continue;
            }
            if (r.prn > GAL_NUM_PRN) return fail(GAL_E_INVAL, "epoch %d slot %d: PRN %d out of range", e, s, r.prn);
            if (r.ibit0 < 0 || r.ibit0 >= GAL_N_SYM_PAGE)
                return fail(GAL_E_INVAL, "epoch %d slot %d: ibit0 %d out of range", e, s, r.ibit0);
            // the reference produces code_phase0 in [0, 4092) (src/gal-sig.cpp:336); a pending wrap is accepted,
            // but not one that is followed by a second wrap within a few samples (k_synth: one wrap per group)
            if (!(r.code_phase0 >= 0.0) || !(r.code_phase0 < 1.5 * GAL_CODE_LEN) || !std::isfinite(r.f_code) ||
                !(r.f_code > 0.0) || !std::isfinite(r.f_carr))
                return fail(GAL_E_INVAL, "epoch %d slot %d: bad phase/frequency", e, s);
            if (!(std::fabs(r.f_carr) < h->cfg.sample_rate) || !(r.f_code < h->cfg.sample_rate * 4000.0))
                return fail(GAL_E_INVAL, "epoch %d slot %d: NCO step out of range", e, s);
            // 16 samples must fit the 16-half-chip window of k_synth: 15 * (2 f_code / fs) + 1 <= 16
            if (!(r.f_code / h->cfg.sample_rate <= 0.5) || !(r.f_code / h->cfg.sample_rate >= 1.0 / 1048576.0))
                return fail(GAL_E_INVAL, "epoch %d slot %d: f_code / sample_rate outside [2^-20, 0.5]", e, s);
            if (r.flags & GAL_CH_RESTART) {
                if (!(std::fabs(r.carr_phase0) < 1.0))
                    return fail(GAL_E_INVAL, "epoch %d slot %d: carr_phase0 must be in (-1,1)", e, s);
            } else if (cur_prn[s] != r.prn) {
                return fail(GAL_E_INVAL,
                            "epoch %d slot %d: PRN %d continues without GAL_CH_RESTART but the slot held PRN %d", e,
                            s, r.prn, cur_prn[s]);
            }
            cur_prn[s] = r.prn;
            n_restart += (r.flags & GAL_CH_RESTART) ? 1 : 0;
            const double ad = std::fabs(r.f_carr * delt);
            // (a step of exactly zero is fine: the phase of the whole epoch is the checkpoint's, k_synth_g's loader lanes know its
            // index exactly; a step below 2^-40 that is not zero creeps over index boundaries for thousands of groups on end)
            const bool carr_ok = ad == 0.0 || (ad >= 9.094947017729282e-13 && 511.0 * ad * 16.0 <= 120.0);
            const double cs2 = 2.0 * (r.f_code * delt);
            cs2_max = std::max(cs2_max, cs2);
            const int mode = rw_mode_of(cs2);
            if (mode != 0) {
                // (the distance is not a continuous function of the step -- when some u s crosses an integer its threshold jumps from
                // 0 to 1 -- so a cached value is extrapolated only inside the radius in which nothing can cross: see rw_rad)
                if (!(std::fabs(cs2 - h->rw_s0[s]) <= h->rw_rad[s])) {
                    h->rw_s0[s] = cs2;
                    h->rw_g0[s] = rw_threshold_gap(cs2);
                    h->rw_e0[s] = rw_threshold_edge(cs2);
                    double rad = std::min((h->rw_e0[s] - min_gap) / 15.0, (h->rw_g0[s] - min_gap) / 14.0);
                    if (cboc) {  // ... and the pattern of the BOC(6,1) half periods: 6 s per sample
                        const double e6 = rw_threshold_edge(6.0 * cs2), g6 = rw_threshold_gap(6.0 * cs2);
                        h->rw_e0[s] = std::min(h->rw_e0[s], e6);
                        h->rw_g0[s] = std::min(h->rw_g0[s], g6);
                        rad = std::min(rad, std::min((e6 - min_gap) / 90.0, (g6 - min_gap) / 84.0));
                    }
                    // (both gates pass with room: the verdict holds for every step this close; else only for this very step)
                    h->rw_rad[s] = rad > 0.0 ? 0.98 * rad : 0.0;
                }
                const bool bins_ok = h->rw_g0[s] > min_gap && h->rw_e0[s] > min_gap;
                if (h->rw_g0[s] > min_gap) rec_mode[i] = (uint8_t)mode;
                // k_synth_g: through its bin tables where the thresholds are a bin apart from each other and from 0 and 1; where they
                // crowd (round 6: 4.092 / 8.184 / 16.368 MS/s and other rates at which 2 f_code / fs is near a small fraction) through the
                // bisection instances (BOC(1,1) only: the CBOC mode's second pattern has bin tables only)
                if (carr_ok && (bins_ok || !cboc)) {
                    rec_gm[i] = (uint8_t)mode;
                    rec_gb[i] = bins_ok ? 1 : 0;
                }
            }
            act_all[(size_t)e * S + n] = (uint8_t)s;
            ++n;
        }
        nact_all[e] = n;
        if (n > nact_max) nact_max = n;
    }
    plan_stage("validation + lists + SoA + guesses");
    // the batch: k_synth's fast body needs ONE window form on every record (P.rw); k_synth_g takes the records that are fit for it
    // in the form most of them have, the others go to an accumulating exact-replay launch behind it (classic windows)
    int rw_mode = 0;
    bool rw_ok = nact_max > 0;
    long long g_count[5] = {0, 0, 0, 0, 0}, n_records = 0;
    for (int e = 0; e < E; ++e)
        for (int k = 0; k < nact_all[e]; ++k) {
            const size_t i = (size_t)e * S + act_all[(size_t)e * S + k];
            ++n_records;
            if (rec_mode[i] == 0 || (rw_mode != 0 && rec_mode[i] != rw_mode)) rw_ok = false;
            if (rw_mode == 0) rw_mode = rec_mode[i];
            if (rec_gm[i]) g_count[rec_gm[i]] += 1;
        }
    int g_mode = 1;
    for (int m = 2; m <= 4; ++m)
        if (g_count[m] > g_count[g_mode]) g_mode = m;
    const long long n_grec = g_count[g_mode];
    bool g_search = false;  // some record of the group kernel's has pattern thresholds that crowd: the bisection instances for the batch
    for (size_t i = 0; i < (size_t)E * S && !g_search; ++i) g_search = rec_gm[i] == g_mode && !rec_gb[i];
#ifdef GAL_TEST_HOOKS
    if (getenv("GAL_G_SEARCH") && !cboc) g_search = true;  // (the soak runs every rate through the bisection instances this way)
#endif
    // epochs in which some record is not fit for k_synth_g in that form: the accumulating exact-replay launch behind it costs about
    // what a whole k_synth_g launch costs PER EPOCH IT HAS WORK IN (its blocks run the slow body at this chunk length: measured 1.0 ms
    // over 1199 epochs for one channel, gated_cost.py (a tool of rounds 3-5: git history)), the other epochs' blocks leave at once
    int n_exact_epochs = 0;
    for (int e = 0; e < E; ++e) {
        bool any = false;
        for (int k = 0; k < nact_all[e] && !any; ++k) {
            const size_t i = (size_t)e * S + act_all[(size_t)e * S + k];
            any = rec_gm[i] != g_mode;
        }
        n_exact_epochs += any ? 1 : 0;
    }
    if (state_in) {
        for (int s = 0; s < S; ++s)
            if (state_in[s].prn > 0 && !(std::fabs(state_in[s].carr_phase) < 1.0))
                return fail(GAL_E_INVAL, "state_in slot %d: carr_phase must be in (-1,1)", s);
    }

    // ---- chunking
    int R = h->cfg.chunk_samples;
    // k_synth_g (synth_group.hip) where it applies: BOC(1,1) on resampled windows of the hold form, automatic chunking, an epoch
    // that spans fewer symbols than the kernel's mask table holds, list entries that fit 32 bits
    // (and not while the handle is held off it: a batch whose code phase sits ON a pattern threshold group after group -- zero
    // Doppler with the phase on the 1 / 1300 lattice of 2 x 1.023 / 2.6, a synthetic input -- lists a percent of its groups, and
    // k_repair_g then costs more than the exact-replay kernel; the next 8 batches of the handle take that one)
    if (h->g_holdoff > 0) h->g_holdoff -= 1;
    // (... so splitting pays while fewer than half of the epochs have such records: k_synth alone takes 1.6 x k_synth_g's time)
    bool fam_g = n_grec > 0 && 2 * n_exact_epochs <= E && R <= 0 && !(h->cfg.flags & GAL_CFG_EXACT_REPLAY) &&
                 nact_max > 0 && h->g_holdoff == 0 &&
                 (double)N * cs2_max / (2.0 * GAL_CODE_LEN) + 4.0 < (double)kGroupSyms &&
                 (double)E * (double)((N + kGroupChunk - 1) / kGroupChunk) * 64.0 < 4294967296.0;
#ifdef GAL_TEST_HOOKS
    if (getenv("GAL_SYNTH_RW")) fam_g = false;  // the forced window forms are k_synth's
    if (const char *env = getenv("GAL_SYNTH_FAMILY")) fam_g = fam_g && atoi(env) != 0;
#endif
    if (fam_g) R = kGroupChunk;
    else if (R <= 0) {
        int target = (int)((N + 512) / 1024);
        target = (target + 63) / 64 * 64;
        if (target < 64) target = 64;
        R = (N + target - 1) / target;
        R = (R + 15) / 16 * 16;  // 64-byte store bursts stay aligned
        if (N >= 65536) {
            // k_synth takes its slow group body whenever any lane of a wave is within 16 samples of a code
            // wrap.  Lanes are consecutive chunks, so when the chunk length divides the code period (10400
            // samples at 2.6 MS/s -> R = 1040) all wraps of a wave fall into the same group.  Pick the multiple
            // of 16 near 1024 that minimises (slow groups + idle lanes).
            const double period = (double)GAL_CODE_LEN * h->cfg.sample_rate / 1.023e6;
            // Small batches: with chunks of ~1024 samples a batch of E epochs is E blocks of four waves -- below 256
            // epochs some CUs get nothing and the call lasts as long as ONE lane needs for its 1040 samples x all channels
            // (0.29 ms for a one-epoch call).  Shorter chunks spread the same samples over more lanes: aim at one block
            // per CU (65536 chunks in the batch), not below ~416 samples per chunk: every chunk is also a checkpoint the
            // walkers have to store, and for a one-epoch call 208-sample chunks lose more in the walker chain (0.33 ms)
            // than they win in the kernel (0.08 ms) -- 416: 0.26 + 0.13 ms (per_epoch_breakdown.py (a tool of rounds 3-5: git history)).
            double centre = 1024.0;
            int lo = 768, hi = 1536;
            const double want = (double)E * (double)N / 65536.0;
            if (want < 1024.0) {
                centre = want < 416.0 ? 416.0 : want;
                lo = (int)(0.75 * centre) / 16 * 16;
                hi = ((int)(1.5 * centre) + 15) / 16 * 16;
                if (lo < 64) lo = 64;
            }
            double best = 1e30;
            for (int cand = lo; cand <= hi; cand += 16) {
                const double q = std::floor(period / cand + 0.5);
                if (q < 1.0) continue;
                const double spread = 64.0 / q * std::fabs(period - q * cand);  // samples, over one wave
                double slow = 4.0 * (1.0 + spread / 16.0) * 16.0 / cand;        // 4 channels per part
                if (slow > 1.0) slow = 1.0;
                const int nck = (N + cand - 1) / cand;
                const double waste = 1.0 - (double)N / ((double)((nck + 63) / 64 * 64) * cand);
                const double cost = 0.3 * slow + waste + 0.02 * std::fabs(cand - centre) / centre;
                if (cost < best) {
                    best = cost;
                    R = cand;
                }
            }
        }
    }
    if (R < 4) R = 4;
    const int nchunks = (N + R - 1) / R;
    const int tiles = (nchunks + 63) / 64;
    // carrier-walk legs per epoch.  The walk and the verification of a batch take as long as ONE leg (a lane's dependent chain of
    // closed-form steps), whatever the batch's length, so a batch that does not fill the chip with 8 legs per epoch gets shorter
    // ones: 32 up to 256 epochs, 16 up to 512 (same box, bench.py --epochs E, one handle / two: 32 epochs 0.357 / 0.219 ms per step
    // with 8 legs, 0.215 / 0.129 with 32; 128 epochs 0.442 / 0.254 -> 0.295 / 0.182; 256 epochs 0.542 / 0.329 -> 0.400 / 0.300;
    // 512 epochs 0.714 / 0.493 -> 0.622 / 0.477 with 16, 0.626 / 0.512 with 32; 1199 epochs: 16 legs cost the pipelined step
    // 15 % -- profiles/r05s_walk_legs_ab.log, r05c_walk_ab.log).  One-epoch calls (INTEGRATION.md option B): k_walk_carr /
    // k_verify_carr 124 / 121 -> 38 / 36 us with 32 legs (round 4).
    int legs = E <= 256 ? 32 : E <= 512 ? 16 : 8;
#ifdef GAL_TEST_HOOKS
    if (const char *env = getenv("GAL_WALK_LEGS")) legs = atoi(env) > 0 ? atoi(env) : legs;
#endif
    int Lc = (nchunks + legs - 1) / legs;
    if (Lc < 1) Lc = 1;
    const int W = (nchunks + Lc - 1) / Lc;
    const size_t LEGS = (size_t)E * W;

    // ---- channel groups (one synth launch each; later groups accumulate onto the first).  Per group: one 32-byte row per epoch
    // (<= kKernelMaxChan positions for k_synth, which reads one word quad; <= kGroupMaxChan for k_synth_g's wide instances; zero-padded).  k_synth_g batches: first the groups of the
    // records that are fit for it (kind 1), then -- if the batch has any -- the groups of the others (kind 0: k_synth on classic
    // windows, accumulating), then ONCE MORE all records in groups of kind 0: what gal_synth_finish's list-overflow path launches.
    std::vector<int> grp_nch, grp_kind;
    // k_synth_g launches of <= 12 channels, later ones accumulating (rounds 3-5), or its WIDE instances, up to 24 channels in one launch
    // (round 6).  Measured, config 4's geometry, 24 SVs at 25 MS/s, kernel alone / pipelined step
    // (tools/wide_ab2.sh, profiles/r06_wide_ab.log): 5999 epochs 82.6 / 99.0 ms wide against 84.2 / 100.5 narrow (+1.9 % / +1.5 %: what the
    // read-modify-write of the second launch costs an issue-bound kernel); 600 epochs 10.1-10.3 / 10.5-10.7 ms wide against 9.14 / 10.3
    // narrow -- one block of 16 waves per CU has nobody to hide its table build and its tail behind (two narrow launches overlap each
    // other's).  So: wide where a block is a whole long epoch and the launch has many rounds of them, narrow otherwise.
    bool narrow_g = !(E >= 2048 && (N + kGroupChunk - 1) / kGroupChunk >= 1024) || g_search;  // (no wide bisection instances)
#ifdef GAL_TEST_HOOKS
    if (getenv("GAL_G_NARROW")) narrow_g = true;
    if (getenv("GAL_G_WIDE") && !g_search) narrow_g = false;
#endif
    std::vector<uint8_t> act_g;
    std::vector<int> nact_g;
    auto add_groups = [&](const int kind, auto &&take_record) {
        // positions of every epoch that take_record() accepts, split evenly over the launches
        std::vector<uint8_t> pos((size_t)E * S, 0);
        std::vector<int> cnt(E, 0);
        int most = 0;
        for (int e = 0; e < E; ++e) {
            for (int k = 0; k < nact_all[e]; ++k) {
                const uint8_t s = act_all[(size_t)e * S + k];
                if (take_record((size_t)e * S + s)) pos[(size_t)e * S + cnt[e]++] = s;
            }
            most = std::max(most, cnt[e]);
        }
        const int max_ch = (kind == 1 && !cboc && !narrow_g) ? kGroupMaxChan : kKernelMaxChan;
        const int ng = most == 0 ? (kind == 1 || grp_nch.empty() ? 1 : 0) : (most + max_ch - 1) / max_ch;
        const size_t g0 = grp_nch.size();
        grp_nch.resize(g0 + ng, 0);
        grp_kind.resize(g0 + ng, kind);
        act_g.resize((g0 + ng) * (size_t)E * kActRow, 0);
        nact_g.resize((g0 + ng) * (size_t)E, 0);
        for (int e = 0; e < E && ng > 0; ++e) {
            const int n = cnt[e];
            const int per = (n + ng - 1) / ng;  // about the same number of channels in every launch
            int k = 0;
            for (int g = 0; g < ng; ++g) {
                int m = n - k < per ? n - k : per;
                if (m < 0) m = 0;
                for (int j = 0; j < m; ++j) act_g[((g0 + g) * (size_t)E + e) * kActRow + j] = pos[(size_t)e * S + k + j];
                nact_g[(g0 + g) * (size_t)E + e] = m;
                grp_nch[g0 + g] = std::max(grp_nch[g0 + g], m);
                k += m;
            }
        }
        return ng;
    };
    int n_exact_records = 0;
    if (fam_g) {
        sp.n_groups = add_groups(1, [&](size_t i) { return rec_gm[i] == g_mode; });
        sp.n_groups += add_groups(0, [&](size_t i) { return rec_gm[i] != g_mode; });
        n_exact_records = (int)(n_records - n_grec);
        sp.all_first = sp.n_groups;
        sp.all_count = n_exact_records ? add_groups(0, [](size_t) { return true; }) : 0;  // (no exact records: the kind-1 groups ARE all)
    } else {
        sp.n_groups = add_groups(0, [](size_t) { return true; });
        sp.all_first = 0;
        sp.all_count = 0;
    }
    sp.group_nch = grp_nch;
    sp.group_kind = grp_kind;
    sp.n_exact_records = n_exact_records;
    const int n_groups = (int)grp_nch.size();  // (rows in the upload, not launches per execute)

    plan_stage("gate, chunking, groups");
    // ---- arena layout: the variable part of the upload region, then the device-only arrays
    (void)take((size_t)std::max(n_restart, 1) * GAL_PAGE_WORDS * 4);  // o_pinit
    const size_t CP1 = (size_t)nchunks + 1;
    const size_t o_act = take((size_t)n_groups * E * kActRow), o_nact = take((size_t)n_groups * E * 4);
    const size_t up_bytes = off;
    if (up_bytes > h->h_up_bytes) return fail(GAL_E_STATE, "gal_synth_plan: staging buffer bound exceeded (%zu > %zu)", up_bytes, h->h_up_bytes);
    // checkpoints (not cleared), then the zeroed region (ONE memset): leg records that are read before they are written
    const size_t o_cpx = take(ES * CP1 * 8), o_cpp = take(ES * CP1 * 8), o_cpi = take(ES * CP1 * 4);
    const size_t o_ancw = take(LEGS * S * 8), o_ancr = take(LEGS * S * 8), o_clmr = take(LEGS * S * 8);
    const size_t o_pend = take(LEGS * S * 8), o_ver = take(LEGS * S), o_dirty = take(LEGS * S), o_risk = take(LEGS * S);
    const size_t zero_end = off;
    const size_t o_clmw = take(LEGS * S * 8);  // "no claim" = -1
    const size_t o_state_out = take(sizeof(gal_chan_state_t) * S);
    const size_t o_pcur = take(ES * GAL_PAGE_WORDS * 4);
    const size_t o_flip = take(ES);
    const size_t o_marg = take(LEGS * S * 8), o_shift = take(LEGS * S * 8), o_tpos = take(LEGS * S * 8),
                 o_tdir = take(LEGS * S);
    const size_t o_ctr = take(CTR_COUNT * 4);
    const size_t g_groups = (size_t)E * (size_t)((N + 15) / 16);
    const size_t g_cap = std::min<size_t>((size_t)kGroupListMin + g_groups / 200, (size_t)1 << 30);
    const size_t o_gflist = take(fam_g ? g_cap * 4 : 16);
    const size_t o_scanm = take(galk_scanm_bytes(S, (int)LEGS));  // the stitch's look-back records (synth_kernels.hip: ScanM)

    const size_t total = off;
    plan_stage("layout + buffers");
    char *const base = nullptr;  // (offsets: commit_staged adds the arena's address)
    DevPlan &P = sp.P;
    P.E = E; P.S = S; P.N = N; P.R = R; P.nchunks = nchunks; P.CP1 = (int)CP1;
    P.blocks_per_epoch = (tiles + 3) / 4;
    {
        // chunks per code period, if the chunk length divides the period (to a hundredth of a sample) and the
        // epoch holds whole classes; otherwise natural order
        const double period = (double)GAL_CODE_LEN * h->cfg.sample_rate / 1.023e6;
        const int cls = (int)std::floor(period / R + 0.5);
        P.cls = (cls >= 2 && std::fabs(period - (double)cls * R) < 0.01 && nchunks % cls == 0) ? cls : 1;
    }
    P.W = W; P.Lc = Lc; P.LEGS = (int)LEGS; P.LEGS_all = (int)LEGS;
    {
        // the code chain of an epoch in legs (k_walk_code): 4 side by side in a long batch -- 88 + <= 13 dependent closed-form steps
        // per lane instead of 351 --, 16 where a batch of a few epochs is all latency
        int wc = E * 8 < 256 ? 16 : 4;
#ifdef GAL_TEST_HOOKS
        if (const char *env = getenv("GAL_CODE_LEGS")) wc = atoi(env) > 0 ? atoi(env) : wc;
#endif
        while (wc > 1 && (wc & (wc - 1))) wc &= wc - 1;  // a power of two
        wc = wc > 64 ? 64 : wc;
        P.Wc = wc;
        P.Lkc = (nchunks + wc - 1) / wc;
    }
    P.delt = 1.0 / h->cfg.sample_rate;
    P.cs25 = kCS25;
    P.page_init = (const uint32_t *)(base + o_pinit); P.init_ix = (const int *)(base + o_initix);
    P.state_in = (const gal_chan_state_t *)(base + o_state_in);
    P.state_out = (gal_chan_state_t *)(base + o_state_out);
    P.prn = (int *)(base + o_prn); P.flags = (uint32_t *)(base + o_flags); P.ib0 = (int *)(base + o_ib0);
    P.x0 = (double *)(base + o_x0); P.p0 = (double *)(base + o_p0);
    P.cstep = (double *)(base + o_cstep); P.dstep = (double *)(base + o_dstep);
    P.page_next = (uint32_t *)(base + o_pnext); P.page_cur = (uint32_t *)(base + o_pcur);
    P.flip_in = (uint8_t *)(base + o_flip);
    P.pguess = (double *)(base + o_pguess);
    P.gss_w = (long long *)(base + o_gssw); P.gss_r = (double *)(base + o_gssr);
    P.anc_w = (long long *)(base + o_ancw); P.anc_r = (double *)(base + o_ancr);
    P.clm_w = (long long *)(base + o_clmw); P.clm_r = (double *)(base + o_clmr);
    P.pend = (double *)(base + o_pend);
    P.verified = (uint8_t *)(base + o_ver); P.dirty = (uint8_t *)(base + o_dirty); P.risk = (uint8_t *)(base + o_risk);
    P.ver_mod = 1; P.ver_rem = 0;
    P.marg = (double *)(base + o_marg); P.shift = (double *)(base + o_shift); P.tpos = (long long *)(base + o_tpos);
    P.tdir = (int8_t *)(base + o_tdir);
    P.scanm = (void *)(base + o_scanm);
    P.translate = 1;
    P.hook_spoil = 0;
#ifdef GAL_TEST_HOOKS
    // one first-pass anchor that is WRONG by a sample (slot 0, leg GAL_HOOK_BAD_LEG): the stitch re-anchors the leg at another event, a
    // second pass walks it again -- the batch that needs more than one carrier pass, which the tests of the repair paths want (since
    // round 6's guesses no random small batch does: tools/find_multi_pass_batch.py)
    if (getenv("GAL_GUESS_SPOIL")) P.hook_spoil = 1;
#endif
    P.tr_e0 = 0;
    P.tr_e1 = E;
    P.cp_e0 = 0;
#ifdef GAL_TEST_HOOKS
    // 0 = never translate, 2 = translate with one deliberately wrong shift (exercises the fallback)
    if (const char *env = getenv("GAL_WALK_TRANSLATE")) P.translate = atoi(env);
#endif
    P.cp_x = (double *)(base + o_cpx); P.cp_p = (double *)(base + o_cpp); P.cp_ib = (uint32_t *)(base + o_cpi);
    P.ctr = (int *)(base + o_ctr);
    P.fam = fam_g ? 1 : 0;
    P.rw_search = (fam_g && g_search) ? 1 : 0;
    P.gflist = (uint32_t *)(base + o_gflist);
    P.gflist_cap = (int)g_cap;
#ifdef GAL_TEST_HOOKS
    if (const char *env = getenv("GAL_G_LIST_CAP")) P.gflist_cap = std::max(1, std::min((int)g_cap, atoi(env)));  // overflow path
#endif
    {
        // k_synth_g: every block takes a contiguous run of the launch's chunks, cut on epoch boundaries (synth_group.hip: sg_grid)
        P.gthreads = 512;
        P.gbpe = 0;
        P.gslots = 2 * h->n_cu;
        P.grounds = 0;
#ifdef GAL_TEST_HOOKS
        if (const char *env = getenv("GAL_G_THREADS")) P.gthreads = atoi(env);
        if (const char *env = getenv("GAL_G_BPE")) P.gbpe = atoi(env) > 0 ? atoi(env) : 0;
        if (const char *env = getenv("GAL_G_ROUNDS")) P.grounds = atoi(env) > 0 ? atoi(env) : P.grounds;
        if (const char *env = getenv("GAL_G_SLOTS")) P.gslots = atoi(env) > 0 ? atoi(env) : P.gslots;
#endif
    }

    P.lut = h->d_lut; P.str = h->d_str;
    P.signal = (h->cfg.flags & GAL_CFG_CBOC) ? 1 : 0;
    // k_synth_g: the form its records have; k_synth alone: its fast body if EVERY record has the same form, else classic windows
    // (k_synth's own fast body: forms 1-3 for BOC(1,1), form 1 for the CBOC mode)
    P.rw = fam_g ? g_mode : (rw_ok && rw_mode <= (cboc ? 1 : 3) ? rw_mode : 0);
#ifdef GAL_TEST_HOOKS
    // 0: classic windows (A/B runs); 11 / 12 / 13: force form 1 / 2 / 3 whatever the gate says (the kernel's own safety nets
    // -- undecidable bins, pattern overflow -- must then keep the output exact)
    if (const char *env = getenv("GAL_SYNTH_RW")) {
        const int v = atoi(env);
        P.rw = v == 0 ? 0 : (v >= 11 && v <= 13) ? (P.signal == 0 || v == 11 ? v - 10 : 0) : P.rw;
    }
#endif

    // ---- the staging buffer is complete but for the device copy of the plan (its pointers are final at the commit)
    if (n_restart == 0) memset(u_pinit, 0, GAL_PAGE_WORDS * 4);
    memcpy(up + o_act, act_g.data(), act_g.size());
    memcpy(up + o_nact, nact_g.data(), nact_g.size() * sizeof(int));
    plan_stage("lists copied");
    sp.nact_max = nact_max;
    sp.act_prefix.assign((size_t)E + 1, 0);
    for (int e = 0; e < E; ++e) sp.act_prefix[e + 1] = sp.act_prefix[e] + nact_all[e];
    sp.o_plan = o_plan; sp.o_act = o_act; sp.o_nact = o_nact; sp.up_bytes = up_bytes; sp.total = total;
    sp.o_cpx = o_cpx; sp.o_ancw = o_ancw; sp.zero_end = zero_end; sp.o_clmw = o_clmw; sp.clmw_bytes = LEGS * S * 8;
    sp.o_scanm = o_scanm; sp.scanm_clear = galk_scanm_status_bytes(S, (int)LEGS);
    sp.R = R; sp.nchunks = nchunks;
    sp.ms_plan = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
    h->staged = std::move(sp);
    h->staged_pending = true;
    // nothing in flight: the plan is the handle's at once, its upload enqueued (and waited for by gal_synth_plan).  A batch in flight
    // (gal_synth_plan_async only): the arena still belongs to it -- the commit happens in the next gal_synth_execute
    return h->in_flight ? GAL_OK : commit_staged(h, wait);
}

int gal_synth_plan(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs, const gal_chan_state_t *state_in)
{
    return plan_impl(h, params, n_epochs, state_in, true);
}

int gal_synth_plan_async(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs, const gal_chan_state_t *state_in)
{
    return plan_impl(h, params, n_epochs, state_in, false);
}

#ifdef GAL_TEST_HOOKS
// The HOST side of gal_synth_plan on its own -- validation, lists, staging; no device, no upload -- `reps` times over the same
// records: milliseconds per plan (tests/test_plan_host.py: the staged arrays against numpy; tools: what a fresh plan costs the host).
static gal_synth *g_host_plan = nullptr;
double gal_hooks_plan_host_ms(const gal_synth_cfg_t *cfg, const gal_chan_epoch_t *params, int32_t n_epochs, const gal_chan_state_t *state_in,
                              int32_t reps)
{
    gal_synth *&g = g_host_plan;
    if (g) {
        free(g->h_up);
        delete g;
        g = nullptr;
    }
    if (!cfg || !params) return -1.0;
    g = new gal_synth();
    g->cfg = *cfg;
    g->host_only = true;
    init_tables();
    double best = 1e30;
    for (int r = 0; r < (reps > 0 ? reps : 1); ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        if (plan_impl(g, params, n_epochs, state_in, false) != GAL_OK) return -1.0;
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return best;
}
// ... and where an array of that plan lies in its staging buffer (host_only: the arena's base is 0, the plan's device pointers are
// offsets into the upload region); "family": (const void *)(1 + P.fam), "groups": 1 + the launches per execute
const void *gal_hooks_plan_host_array(const char *name)
{
    const gal_synth *g = g_host_plan;
    if (!g || !g->planned || !name) return nullptr;
    const DevPlan &P = g->P;
    const struct { const char *n; const void *p; } tab[] = {
        {"prn", P.prn}, {"flags", P.flags}, {"ib0", P.ib0}, {"x0", P.x0}, {"p0", P.p0}, {"cstep", P.cstep}, {"dstep", P.dstep},
        {"page_next", P.page_next}, {"pguess", P.pguess}, {"gss_w", P.gss_w}, {"gss_r", P.gss_r}, {"init_ix", P.init_ix},
        {"page_init", P.page_init}, {"state_in", P.state_in}, {"act", g->d_act}, {"nact", g->d_nact}};
    for (const auto &t : tab)
        if (!strcmp(t.n, name)) return g->h_up + (size_t)(uintptr_t)t.p;
    if (!strcmp(name, "family")) return (const void *)(uintptr_t)(1 + P.fam);
    if (!strcmp(name, "groups")) return (const void *)(uintptr_t)(1 + g->n_groups);
    if (!strcmp(name, "search")) return (const void *)(uintptr_t)(1 + P.rw_search);
    if (!strcmp(name, "form")) return (const void *)(uintptr_t)(1 + P.rw);
    return nullptr;
}
#endif

// verify_here: k_synth_g does not verify the carrier checkpoints itself (k_synth's exact replay does, on its way); k_verify_carr
// does, in front of it on the same stream (repair paths, handles without a walker stream) -- the first launch of a batch runs it
// on the walker stream instead, beside the synthesis (gal_synth_execute_range)
static int enqueue_synth(gal_synth *h, uint32_t *iq, bool verify_here)
{
    if (verify_here && h->P.fam == 1 && h->nact_max != 0) {
        // (in front of the synthesis on its own stream: gal_synth_finish's repair paths and handles without a walker stream --
        // nothing to hide it behind, and the repair paths have reason to look at every leg)
        DevPlan Pv = h->Pw;
        if (h->stats.synth_runs > 1) Pv.ver_mod = 1, Pv.ver_rem = 0;
        galk_launch_verify_carr(&Pv, h->stream);
        galk_launch_verify_code(&Pv, h->stream);
    }
    if (h->nact_max == 0) {  // nothing is transmitted in this batch: the reference's loop stores zeros (:536-537)
        HIP_TRY(hipMemsetAsync(iq, 0, (size_t)h->range_ne * (size_t)h->P.N * 4u, h->stream));
        return GAL_OK;
    }
    // k_synth_g batches: its groups, then the accumulating exact-replay groups of the records it cannot take (classic windows); after
    // a list overflow (P.fam set to 0 by gal_synth_finish) and in k_synth batches: every record on k_synth
    const bool overflow_set = h->P.fam == 0 && h->all_count > 0;
    const int g_first = overflow_set ? h->all_first : 0, g_count = overflow_set ? h->all_count : h->n_groups;
    DevPlan Px = h->P;  // what an exact-replay launch of a k_synth_g batch sees
    if (h->P.fam == 1 || overflow_set) Px.rw = 0;
    for (int k = 0; k < g_count; ++k) {
        const int g = g_first + k;
        const uint8_t *act = h->d_act + (size_t)g * h->P.E * kActRow;
        const int *nact = h->d_nact + (size_t)g * h->P.E;
        const bool on_g = h->P.fam == 1 && h->group_kind[g] == 1;
        const int rc = on_g ? galk_launch_synth_g(&h->P, h->d_plan, h->group_nch[g], k > 0, act, nact, iq, h->range_e0, h->range_ne, h->stream)
                            : galk_launch_synth(h->P.fam == 1 || overflow_set ? &Px : &h->P, h->d_plan, h->group_nch[g], k > 0, act, nact, iq,
                                                h->range_e0, h->range_ne, h->stream);
        if (rc) return fail(GAL_E_INVAL, "no synthesis kernel for %d channels per group", h->group_nch[g]);
    }
    // the groups k_synth_g could not decide (chip pattern or table index within the rounding drift of a boundary), exactly
    if (h->P.fam == 1) {
        HIP_TRY(hipEventRecord(h->ev[3], h->stream));  // (k_synth_g's time and k_repair_g's are reported apart)
        galk_launch_repair_g(&h->P, iq, h->range_e0, h->stream);
    }
    HIP_TRY(hipGetLastError());
    return GAL_OK;
}

// a plan made while the batch before was in flight becomes the handle's now (that batch must have been finished)
static int ensure_committed(gal_synth *h)
{
    if (!h->staged_pending) return GAL_OK;
    if (h->in_flight) return fail(GAL_E_STATE, "gal_synth_execute while a batch is in flight: call gal_synth_finish first");
    HIP_TRY(hipSetDevice(h->device));
    return commit_staged(h, false);
}

int gal_synth_execute(gal_synth_t *h, int16_t *iq_dev)
{
    if (!h) return fail(GAL_E_INVAL, "gal_synth_execute: null argument");
    if (const int rc = ensure_committed(h)) return rc;
    if (!h->planned) return fail(GAL_E_STATE, "gal_synth_execute before gal_synth_plan");
    return gal_synth_execute_range(h, iq_dev, 0, h->P.E);
}

int gal_synth_execute_range(gal_synth_t *h, int16_t *iq_dev, int32_t first_epoch, int32_t n_epochs)
{
    if (!h || !iq_dev) return fail(GAL_E_INVAL, "gal_synth_execute: null argument");
    if (const int rc = ensure_committed(h)) return rc;
    if (!h->planned) return fail(GAL_E_STATE, "gal_synth_execute before gal_synth_plan");
    if (first_epoch < 0 || n_epochs < 1 || first_epoch + n_epochs > h->P.E)
        return fail(GAL_E_INVAL, "gal_synth_execute_range: epochs [%d, %d) outside the planned batch of %d", first_epoch,
                    first_epoch + n_epochs, h->P.E);
    if (h->in_flight)
        return fail(GAL_E_STATE, "gal_synth_execute while a batch is in flight: call gal_synth_finish first");
    if (((uintptr_t)iq_dev) & 15) return fail(GAL_E_INVAL, "iq_dev must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    // (the range is recorded only once every check has passed: finish()'s repair paths re-synthesise it)
    if (h->executed && (first_epoch != h->range_e0 || n_epochs != h->range_ne)) h->enq_passes = std::max(h->enq_passes, kDefaultPasses);  // another range of the plan: its own pass count
    h->range_e0 = first_epoch;
    h->range_ne = n_epochs;
    // TRANSLATED carrier legs are covered by k_synth's replay check, which works by induction from the chain root
    // and therefore only vouches for epochs it actually replays.  Outside the executed range a leg is accepted
    // through a genuine walk + bitwise stitch only (k_scanm: tr_e0 / tr_e1), so the carrier state a range
    // starts from never rests on an unchecked translation.
    // (the walker kernels take the plan by value; k_synth's device copy does not use these fields)
    // The walkers see the plan cut to [0, first_epoch + n_epochs): the carrier chain never restarts, so the epochs in
    // front of the range must be walked for the state the range starts from -- silently, no checkpoints (cp_e0) --
    // and the epochs behind it not at all.  (Slot-major scratch arrays are indexed with the cut plan's strides by every
    // kernel of this batch; checkpoint and per-epoch arrays are epoch-major and keep their places.)
    h->Pw = h->P;
    h->Pw.E = first_epoch + n_epochs;
    h->Pw.LEGS = h->Pw.E * h->P.W;
    h->Pw.tr_e0 = first_epoch;
    h->Pw.tr_e1 = first_epoch + n_epochs;
    h->Pw.cp_e0 = first_epoch;
    // k_verify_carr / k_verify_code (k_synth_g batches): every leg of both chains (default); GAL_CFG_VERIFY_SAMPLED: an eighth of the leg
    // positions per batch, rotating with the handle's batch count, plus the carrier legs whose translation was not overwhelmingly
    // inside its margin (synth_kernels.hip)
    h->Pw.ver_mod = (h->cfg.flags & GAL_CFG_VERIFY_SAMPLED) ? kVerifyRotation : 1;
#ifdef GAL_TEST_HOOKS
    if (const char *env = getenv("GAL_VERIFY_MOD")) h->Pw.ver_mod = atoi(env) > 0 ? atoi(env) : h->Pw.ver_mod;  // timing experiments
#endif
    h->Pw.ver_rem = (int)((h->seq + 1u) % (uint32_t)h->Pw.ver_mod);
    hipStream_t st = handle_stream(h);
    if (!st) return fail(GAL_E_DEVICE, "hipStreamCreate failed");
    const DevPlan *P = &h->Pw;
    // ws: the walker chain (the handle's high-priority stream).  It does not wait for the caller's stream: everything the
    // walkers read was uploaded by gal_synth_plan, which returns after its copies have completed, and everything they
    // write is scratch of this handle, whose last reader (k_synth of the previous batch) has completed because
    // gal_synth_finish has been called (in_flight above).  k_synth itself stays in order on the caller's stream.
    hipStream_t ws = h->walk_stream ? h->walk_stream : st;
    if (h->upload_unordered) {  // gal_synth_plan_async: the walkers read what the upload brings
        if (ws != st) HIP_TRY(hipStreamWaitEvent(ws, h->ev_upd, 0));
        h->upload_unordered = false;
    }
    HIP_TRY(hipEventRecord(h->ev[0], ws));
    HIP_TRY(hipEventRecord(h->ev_prep, ws));
    // Speculative carrier walk, the chain k_synth waits for: first guesses (which also reset the batch's counters), then
    // h->enq_passes passes of walk + stitch (the stitch translates on the spot and its last block publishes the pass and
    // the end-of-batch phase), enqueued back to back, then the synthesis kernel -- all asynchronously: the host does not
    // wait here, so several handles can be kept in flight (the latency-bound walk of one batch then runs beside the
    // issue-bound synthesis of another).  gal_synth_finish() looks at the counters; in the rare case that the chain was
    // not verified by then it iterates further and repeats the synthesis.
    // (a batch of a few epochs: the code chain of ONE epoch -- 550 dependent closed-form steps, 115 us -- is the longer of the two
    // since the carrier legs are short there, so it goes first)
    const bool code_first = h->Pw.E * 8 < 256 && h->aux_stream && h->aux_stream != ws;
    if (code_first) {
        HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_prep, 0));
        galk_launch_walk_code(P, h->aux_stream);
        galk_launch_pages(P, h->aux_stream);
        HIP_TRY(hipEventRecord(h->ev_aux, h->aux_stream));
    }
    int n_passes = h->enq_passes;
#ifdef GAL_TEST_HOOKS
    if (const char *env = getenv("GAL_WALK_PASSES")) n_passes = atoi(env) > 0 ? atoi(env) : n_passes;
#endif
    h->first_pass_legs = h->act_prefix[h->Pw.E] * h->P.W;
    if (h->scan_tag > 0xFFF00000u) {
        // the stitch's 32-bit launch tags are about to wrap (2 per batch: after ~2^31 batches of one handle): records of old launches
        // would then carry tags that are handed out again -- clear the status words once and start over
        HIP_TRY(hipMemsetAsync((char *)h->P.scanm, 0, h->scanm_clear, ws));
        h->scan_tag = 0;
    }
    for (int pass = 0; pass < n_passes; ++pass) {
        galk_launch_walk_carr(P, pass == 0, ws);  // (the first one also resets the batch's counters)
        if (pass == 0 && ws != st) HIP_TRY(hipEventRecord(h->ev_ctr, ws));
        galk_launch_carr_scan(P, ++h->scan_tag, ws);
    }
    // the code chain (restarts every epoch, no speculation) and the page resolution do not depend on the carrier chain:
    // they run on a second stream beside it and join before k_synth.  (Enqueued after the carrier passes: every launch
    // call in front of those delays the critical path by the 5-8 us the call takes.)
    if (!code_first) {
        HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_prep, 0));
        galk_launch_walk_code(P, h->aux_stream);
        galk_launch_pages(P, h->aux_stream);
        HIP_TRY(hipEventRecord(h->ev_aux, h->aux_stream));
    }
    // k_synth_g batches: the carrier checkpoints are verified by a kernel of their own (k_verify_carr), on the walker stream
    // behind the chain and beside the synthesis; the completion record waits for both
    bool verify_beside = h->P.fam == 1 && ws != st && h->nact_max != 0;
#ifdef GAL_TEST_HOOKS
    const bool no_verify = getenv("GAL_G_NOVERIFY") != nullptr;  // timing experiments only: what k_verify_carr costs the pipeline
    if (no_verify) verify_beside = false;
#else
    const bool no_verify = false;
#endif
    if (ws != st) {
        HIP_TRY(hipEventRecord(h->ev_walk, ws));
        HIP_TRY(hipStreamWaitEvent(st, h->ev_walk, 0));
        if (verify_beside) {
            galk_launch_verify_carr(P, ws);
            HIP_TRY(hipEventRecord(h->ev_ver, ws));
            // ... and the code checkpoints, on the second walker stream behind the code walk that wrote them (ev_aux is recorded):
            // it does not wait for the carrier chain -- only for the first carrier walk, whose first block resets the counters
            // this kernel counts its mismatches in
            HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_ctr, 0));
            galk_launch_verify_code(P, h->aux_stream);
            HIP_TRY(hipEventRecord(h->ev_verc, h->aux_stream));
        }
    }
    HIP_TRY(hipStreamWaitEvent(st, h->ev_aux, 0));
    HIP_TRY(hipEventRecord(h->ev[1], st));
    int rc = enqueue_synth(h, (uint32_t *)iq_dev, !verify_beside && !no_verify);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev[2], st));
    if (verify_beside) {
        HIP_TRY(hipStreamWaitEvent(st, h->ev_ver, 0));
        HIP_TRY(hipStreamWaitEvent(st, h->ev_verc, 0));
    }
    // counters (walker passes + replay check) and the end-of-batch state, behind the synthesis: nothing in front of
    // k_synth that it does not need (finish()'s repair paths fetch both again)
    h->seq += 1;
    if (h->seq == 0) h->seq = 1;
    galk_launch_publish(P, h->h_ctr, h->h_state, h->h_flag, h->seq, st);
    h->state_fetched = true;
    h->last_iq = (uint32_t *)iq_dev;
    h->stats.synth_runs = 1;
    h->executed = true;
    h->in_flight = true;
    return GAL_OK;
}

size_t gal_synth_stats_size(void) { return sizeof(gal_synth_stats_t); }

// the caller's struct may be older (shorter) than the library's: never more than its own sizeof is written
static void copy_stats(const gal_synth *h, void *stats, size_t stats_bytes)
{
    if (stats) memcpy(stats, &h->stats, std::min(stats_bytes, sizeof(gal_synth_stats_t)));
}

// The un-sized symbols are what binaries compiled against the 0.2 header call (0.3 and later headers turn the names into macros over
// the _n entry points): they fill the OLDEST published struct -- 40 bytes, up to synth_runs -- and nothing behind it.
static constexpr size_t kStatsBytesV02 = 40;
static_assert(offsetof(gal_synth_stats_t, kernel_family) == kStatsBytesV02, "gal_synth_stats_t 0.2 ends in front of kernel_family");

int gal_synth_finish(gal_synth_t *h, gal_chan_state_t *state_out, gal_synth_stats_t *stats)
{
    return gal_synth_finish_n(h, state_out, stats, kStatsBytesV02);
}

int gal_synth_finish_n(gal_synth_t *h, gal_chan_state_t *state_out, void *stats, size_t stats_bytes)
{
    if (!h) return fail(GAL_E_INVAL, "null handle");
    if (!h->executed) return fail(GAL_E_STATE, "gal_synth_finish before gal_synth_execute");
    h->in_flight = false;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = handle_stream(h);
    if (!st) return fail(GAL_E_DEVICE, "hipStreamCreate failed");
    const DevPlan *P = &h->Pw;
    // The batch is complete when its record has arrived in pinned memory (k_publish runs behind k_synth on the handle's
    // stream): poll the sequence number; the stream itself is looked at now and then, so that a failed launch or a
    // device fault ends the wait with its error instead of hanging it.
    {
        // Waiting without burning a core and without adding latency: 50 us of pause-spinning; then naps (nanosleep: ~70 us each
        // with the kernel's timer slack) for as long as the batch is EXPECTED to need -- 70 % of what the handle's previous
        // finish() had to wait, minus a nap; then sched_yield (other runnable threads of the process -- the CLI's producer and
        // writer, other handles' callers -- get the core, a lone caller sees the record at once).  A handle's batches are alike,
        // so in steady state the naps cover most of the wait and the yields only its last fifth.
        const uint32_t want = h->seq;
        const auto t_begin = std::chrono::steady_clock::now();
        unsigned spins = 0;
        double waited_us = 0.0;
        const double nap_until_us = 0.7 * h->prev_wait_us - 150.0;
        while (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) != want) {
            const bool look = (++spins & 255u) == 0;
            if (!look) {
                __builtin_ia32_pause();
                continue;
            }
            waited_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
            bool napped = false;
            if (waited_us >= 50.0) {
                if (waited_us < nap_until_us) {
                    timespec ts{0, 50000};
                    nanosleep(&ts, nullptr);
                    napped = true;
                } else {
                    sched_yield();
                }
            }
            if ((spins & 4095u) == 0 || napped) {  // a failed launch or a device fault ends the wait with its error
                const hipError_t q = hipStreamQuery(st);
                if (q == hipErrorNotReady) continue;
                HIP_TRY(q);
                // the stream has drained: the record must be there (same memory, read once more), or the kernel never ran
                if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) != want) {
                    HIP_TRY(hipStreamSynchronize(st));
                    if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) != want)
                        return fail(GAL_E_DEVICE, "the batch's completion record never arrived");
                }
                break;
            }
        }
        h->prev_wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    }
    float ms_walk = 0, ms_synth = 0;
    hipEventElapsedTime(&ms_walk, h->ev[0], h->ev[1]);
    hipEventElapsedTime(&ms_synth, h->ev[1], h->ev[2]);
    float ms_repair = 0;
    if (h->P.fam == 1 && h->nact_max != 0) {
        hipEventElapsedTime(&ms_synth, h->ev[1], h->ev[3]);
        hipEventElapsedTime(&ms_repair, h->ev[3], h->ev[2]);
    }
    int *ctr_walk = h->h_ctr, *ctr_end = h->h_ctr;  // counters after the synthesis kernel
    if (ctr_walk[CTR_UNVERIFIED] != 0) {
        // stragglers (itinerary mismatches / tie epochs beyond the enqueued passes): iterate from the host
        // until every leg is verified, then redo the end state and the synthesis with the exact checkpoints
        const int max_passes = h->cfg.max_walk_passes > 0 ? h->cfg.max_walk_passes : 64 + P->LEGS;
        while (ctr_walk[CTR_UNVERIFIED] != 0) {
            if (ctr_walk[CTR_PASSES] >= max_passes)
                return fail(GAL_E_CHAIN, "carrier walk did not converge in %d passes (%d legs unverified)",
                            ctr_walk[CTR_PASSES], ctr_walk[CTR_UNVERIFIED]);
            for (int k = 0; k < 2; ++k) {
                galk_launch_walk_carr(P, 0, st);
                galk_launch_carr_scan(P, ++h->scan_tag, st);
            }
            HIP_TRY(hipMemcpyAsync(ctr_walk, P->ctr, CTR_COUNT * sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        HIP_TRY(hipMemsetAsync(P->ctr + CTR_MISMATCH, 0, sizeof(int), st));
        HIP_TRY(hipMemsetAsync(P->ctr + CTR_GFLAGS, 0, 2 * sizeof(int), st));
        h->state_fetched = false;
        HIP_TRY(hipEventRecord(h->ev[1], st));
        h->stats.synth_runs += 1;
        int rc = enqueue_synth(h, h->last_iq, true);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev[2], st));
        HIP_TRY(hipMemcpyAsync(ctr_end, P->ctr, CTR_COUNT * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        float extra = 0;
        hipEventElapsedTime(&extra, h->ev[1], h->ev[2]);
        ms_walk += ms_synth;  // the first, speculative synthesis was wasted work
        ms_synth = extra;
    }
    if (ctr_end[CTR_MISMATCH] != 0 && P->translate) {
        // The replay kernel found a checkpoint that genuine stepping does not reproduce.  The only unverified-
        // by-walking inputs are the TRANSLATED legs of the two chains: redo both with every leg walked from its true anchor
        // (the carrier's verified bitwise by the stitch), then the synthesis with every leg of both re-checked.  (Never observed
        // outside the fault-injection hooks; kept so that a flaw in the translation argument costs time, not correctness -- which
        // holds as long as every batch is fully verified: the default, see GAL_CFG_VERIFY_SAMPLED.)
        DevPlan Pw = *P;
        Pw.translate = 0;
        const int max_passes = h->cfg.max_walk_passes > 0 ? h->cfg.max_walk_passes : 64 + P->LEGS;
        // (the code chain as well: its legs behind the first of an epoch are accepted by translation too -- k_walk_code with
        // translate == 0 walks every leg from its true anchor --, and the pages hang on its flip flags)
        galk_launch_walk_code(&Pw, st);
        galk_launch_pages(&Pw, st);
        int first = 1;  // (the first walk of a chain resets the batch's counters)
        do {
            if (!first && ctr_walk[CTR_PASSES] >= max_passes)
                return fail(GAL_E_CHAIN, "carrier walk did not converge in %d passes", ctr_walk[CTR_PASSES]);
            for (int k = 0; k < 2; ++k) {
                galk_launch_walk_carr(&Pw, first, st);
                galk_launch_carr_scan(&Pw, ++h->scan_tag, st);
                if (first) h->first_pass_legs += h->act_prefix[h->Pw.E] * h->P.W;
                first = 0;
            }
            HIP_TRY(hipMemcpyAsync(ctr_walk, P->ctr, CTR_COUNT * sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        } while (ctr_walk[CTR_UNVERIFIED] != 0);
        h->state_fetched = false;
        HIP_TRY(hipEventRecord(h->ev[1], st));
        h->stats.synth_runs += 1;
        int rc = enqueue_synth(h, h->last_iq, true);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev[2], st));
        HIP_TRY(hipMemcpyAsync(ctr_end, P->ctr, CTR_COUNT * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        float extra = 0;
        hipEventElapsedTime(&extra, h->ev[1], h->ev[2]);
        ms_walk += ms_synth;
        ms_synth = extra;
        h->n_fallbacks += 1;
    }
    if (h->P.fam == 1 && ctr_end[CTR_GOVER] != 0) {
        // more undecided groups than the list holds (a batch the gate should have kept away, e.g. hold patterns that overflow
        // the masks): the batch once more with the exact-replay kernel, which takes any input
        h->P.fam = 0;
        h->Pw.fam = 0;
        HIP_TRY(hipMemsetAsync(P->ctr + CTR_MISMATCH, 0, sizeof(int), st));
        HIP_TRY(hipEventRecord(h->ev[1], st));
        h->stats.synth_runs += 1;
        int rc = enqueue_synth(h, h->last_iq, false);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev[2], st));
        HIP_TRY(hipMemcpyAsync(ctr_end, P->ctr, CTR_COUNT * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        float extra = 0;
        hipEventElapsedTime(&extra, h->ev[1], h->ev[2]);
        ms_walk += ms_synth;
        ms_synth = extra;
    }
    h->stats.kernel_family = h->P.fam;
    h->stats.repaired_groups = h->P.fam == 1 ? ctr_end[CTR_GFLAGS] : 0;
    h->stats.exact_records = h->P.fam == 1 ? h->n_exact_records : 0;
    if (h->P.fam == 1 && (double)ctr_end[CTR_GFLAGS] > 0.004 * (double)h->range_ne * (double)((h->P.N + 15) / 16) + 4096.0)
        h->g_holdoff = 9;  // (this batch is exact like any other; the handle's next 8 go to the exact-replay kernel)
    h->stats.walk_passes = ctr_end[CTR_PASSES];
    h->enq_passes = std::max(1, std::min(ctr_end[CTR_PASSES], 8));  // (of THIS plan, executed again; a new plan starts from kDefaultPasses)
    if (h->P.E <= kSmallPlanEpochs) h->small_need = h->enq_passes;
    h->h_ctr[CTR_MISMATCH] = ctr_end[CTR_MISMATCH];
    h->stats.chain_mismatch = h->h_ctr[CTR_MISMATCH];
    h->stats.ms_walk = ms_walk;
    h->stats.ms_synth = ms_synth;
    h->stats.ms_repair = h->stats.synth_runs == 1 ? ms_repair : 0.0f;
    h->stats.window_mode = h->P.rw | ((h->P.fam == 1 && h->P.rw_search) ? 16 : 0);
    h->stats.ms_plan = h->ms_plan;
    if (h->upload_timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev_up0, h->ev_up1) == hipSuccess) h->stats.ms_h2d = ms;
        else (void)hipGetLastError();
    }
    h->legs_walked = h->first_pass_legs + ctr_end[CTR_WALKS];  // (a first pass walks every active leg: counted here, not on the device)
    h->legs_translated = ctr_end[CTR_SHIFTS];
    copy_stats(h, stats, stats_bytes);
    if (state_out) {
        if (!h->state_fetched)
            HIP_TRY(hipMemcpy(h->h_state, P->state_out, sizeof(gal_chan_state_t) * P->S, hipMemcpyDeviceToHost));
        memcpy(state_out, h->h_state, sizeof(gal_chan_state_t) * P->S);
    }
    if (h->stats.chain_mismatch != 0)
        return fail(GAL_E_CHAIN, "replay kernel disagreed with the NCO walker at %d chunk boundaries",
                    h->stats.chain_mismatch);
    return GAL_OK;
}

int gal_synth_walk_counts(const gal_synth_t *h, int64_t *legs_walked, int64_t *legs_translated, int64_t *fallbacks)
{
    if (!h) return fail(GAL_E_INVAL, "null handle");
    if (legs_walked) *legs_walked = h->legs_walked;
    if (legs_translated) *legs_translated = h->legs_translated;
    if (fallbacks) *fallbacks = h->n_fallbacks;
    return GAL_OK;
}

int gal_synth_run_host(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs,
                       const gal_chan_state_t *state_in, int16_t *iq_host, gal_chan_state_t *state_out,
                       gal_synth_stats_t *stats)
{
    return gal_synth_run_host_n(h, params, n_epochs, state_in, iq_host, state_out, stats, kStatsBytesV02);
}

int gal_synth_run_host_n(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs,
                         const gal_chan_state_t *state_in, int16_t *iq_host, gal_chan_state_t *state_out,
                         void *stats, size_t stats_bytes)
{
    if (!h || !iq_host) return fail(GAL_E_INVAL, "gal_synth_run_host: null argument");
    int rc = gal_synth_plan(h, params, n_epochs, state_in);
    if (rc) return rc;
    const size_t bytes = gal_synth_output_bytes(h);
    if (bytes > h->own_iq_bytes) {
        if (h->own_iq) hipFree(h->own_iq);
        h->own_iq = nullptr;
        h->own_iq_bytes = 0;
        if (hipMalloc(&h->own_iq, bytes) != hipSuccess) return fail(GAL_E_NOMEM, "hipMalloc of %zu bytes failed", bytes);
        h->own_iq_bytes = bytes;
    }
    // Small batches (one-epoch calls, INTEGRATION.md option B) are latency: their copy is enqueued behind the batch at once, into
    // pinned memory of the handle, so that it runs while gal_synth_finish is still on its way back; what is left behind finish is
    // one memcpy into the caller's (pageable) buffer.  (hipMemcpy into pageable memory, issued only after finish, cost such a
    // call 100 us between the completion record and the start of its copy.)
    const bool staged = bytes <= (size_t)kRunHostStagedBytes;
    if (staged && bytes > h->own_pin_bytes) {
        if (h->own_pin) hipHostFree(h->own_pin);
        h->own_pin = nullptr;
        h->own_pin_bytes = 0;
        if (hipHostMalloc(&h->own_pin, bytes, hipHostMallocDefault) != hipSuccess)
            return fail(GAL_E_NOMEM, "hipHostMalloc of %zu bytes failed", bytes);
        h->own_pin_bytes = bytes;
    }
    rc = gal_synth_execute(h, (int16_t *)h->own_iq);
    if (rc) return rc;
    hipStream_t st = handle_stream(h);
    if (staged) HIP_TRY(hipMemcpyAsync(h->own_pin, h->own_iq, bytes, hipMemcpyDeviceToHost, st));
    rc = gal_synth_finish_n(h, state_out, stats, stats_bytes);
    if (rc) return rc;
    if (!staged) {
        HIP_TRY(hipMemcpy(iq_host, h->own_iq, bytes, hipMemcpyDeviceToHost));
        return GAL_OK;
    }
    // finish() repeats the synthesis when the speculative chain was not verified in time (rare): the copy then has to be repeated too
    if (h->stats.synth_runs != 1) HIP_TRY(hipMemcpyAsync(h->own_pin, h->own_iq, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(iq_host, h->own_pin, bytes);
    return GAL_OK;
}

}  // extern "C"
