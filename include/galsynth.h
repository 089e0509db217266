/*
 * galsynth.h -- C ABI of the MI355X Galileo E1B/C baseband IQ synthesis engine.
 *
 * This is the drop-in boundary for the reference's per-sample hot loop.  The reference
 * (harshadms/galileo-sdr-sim) exposes no plugin/FFI interface for this path: the loop is inline in
 * galileo_task() at src/galileo-sdr.cpp:481-539.  The entry points below are what a maintainer
 * would call where that loop stands: everything the loop READS is carried by
 * gal_chan_epoch_t / gal_chan_state_t, everything it WRITES is the interleaved int16 IQ buffer
 * (src/galileo-sdr.cpp:536-537, the `ishort` wire format of gnss-sdr_Galileo_E1_ishort.conf:14-21)
 * plus the per-channel state that survives an epoch (carr_phase and page, §7.3-3 of SURVEY.md).
 *
 * Plain C, plain pointers and sizes; no torch / HIP types in the signatures (a hipStream_t is passed
 * as void*).  One handle per GPU/stream; a handle is thread-compatible, not thread-safe.
 * All functions return GAL_OK (0) or a negative gal_status_t; gal_synth_last_error() gives text.
 */
#ifndef GALSYNTH_H_
#define GALSYNTH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAL_MAX_CHAN 16          /* reference MAX_CHAN, include/constants.h:10 (engine accepts up to 64) */
#define GAL_ENGINE_MAX_CHAN 64   /* packed I/Q accumulation stays inside int16 up to 65 channels       */
#define GAL_N_SYM_PAGE 500       /* symbols per page, include/constants.h:34                            */
#define GAL_PAGE_WORDS 16        /* 500 symbols, bit i of word i>>5 (LSB first) = symbol i              */
#define GAL_CODE_LEN 4092        /* CA_SEQ_LEN_E1, include/constants.h:124                              */
#define GAL_NUM_PRN 50

typedef enum gal_status {
    GAL_OK = 0,
    GAL_E_INVAL = -1,     /* bad argument                                                   */
    GAL_E_NOMEM = -2,     /* host or device allocation failed                               */
    GAL_E_DEVICE = -3,    /* HIP runtime error / no usable GPU (there is NO CPU fallback)   */
    GAL_E_STATE = -4,     /* call sequence error (execute before plan, ...)                 */
    GAL_E_CHAIN = -5,     /* NCO chain self-check failed (see gal_synth_stats_t)            */
    GAL_E_IO = -6,
    GAL_E_BUSY = -7,      /* a resource another instance holds (galscen: the UDP position port)   */
    GAL_E_EMPTY = -8      /* galscen: the duration holds no epoch, (int)(10 d + 0.5) < 2: nothing to generate (the reference
                             opens its sink and closes it again, src/galileo-sdr.cpp:438)        */
} gal_status_t;

/* gal_chan_epoch_t.flags */
#define GAL_CH_RESTART 1u /* channel (re)allocated before this epoch: carrier phase := carr_phase0, page := page_init
                             (src/channel.cpp:81-99) */

/*
 * What the hot loop reads for ONE channel slot in ONE 0.1 s epoch -- the values
 * computeCodePhase() (src/gal-sig.cpp:308-347) leaves in channel_t (include/structures.h:140-162)
 * right before `for (isamp...)` (src/galileo-sdr.cpp:481), plus the two pages the loop may use.
 */
typedef struct gal_chan_epoch {
    int32_t  prn;          /* 1..50; 0 = idle slot (chan[i].prn, src/galileo-sdr.cpp:489)                      */
    int32_t  ibit0;        /* chan.ibit at epoch start, 0..499 (src/gal-sig.cpp:334,338)                         */
    uint32_t flags;        /* GAL_CH_*                                                                           */
    uint32_t reserved;
    double   f_carr;       /* Hz, chan.f_carr (src/gal-sig.cpp:318); |f_carr| < sample_rate                     */
    double   f_code;       /* Hz, chan.f_code (src/gal-sig.cpp:320); 2^-20 <= f_code / sample_rate <= 0.5      */
    double   code_phase0;  /* chips, chan.code_phase at epoch start (src/gal-sig.cpp:336), in [0, 6138)         */
    double   carr_phase0;  /* cycles; used only with GAL_CH_RESTART (src/channel.cpp:98-99)                     */
    uint32_t page_next[GAL_PAGE_WORDS]; /* page generateINavMsg(grx_of_this_epoch) would produce: installed when
                                           ibit wraps 499->0 inside this epoch (src/galileo-sdr.cpp:497-506)    */
    uint32_t page_init[GAL_PAGE_WORDS]; /* page in force at epoch start; used only with GAL_CH_RESTART          */
} gal_chan_epoch_t;        /* 176 bytes */

/* State that survives an epoch (and a call): carr_phase and the current page, per slot. */
typedef struct gal_chan_state {
    double   carr_phase;               /* chan.carr_phase after the last sample (src/galileo-sdr.cpp:531-532) */
    uint32_t page[GAL_PAGE_WORDS];     /* chan.page                                                            */
    int32_t  prn;                      /* PRN the state belongs to (0 = none)                                   */
    int32_t  reserved;
} gal_chan_state_t;        /* 80 bytes */

typedef struct gal_synth_cfg {
    double  sample_rate;        /* Hz; reference: 2.6e6 (include/constants.h:96). delt = 1.0 / sample_rate     */
    int32_t samples_per_epoch;  /* reference: NUM_IQ_SAMPLES = 260000 (include/constants.h:82)                 */
    int32_t n_slots;            /* channel slots per epoch record row; reference MAX_CHAN = 16                 */
    int32_t device;             /* HIP device ordinal; -1 = current device                                     */
    int32_t chunk_samples;      /* 0 = auto (~1024); samples replayed by one lane (multiple of 4)              */
    int32_t max_walk_passes;    /* 0 = default (64 + carrier legs in the plan); cap on speculative carrier-walk
                                   passes before gal_synth_finish gives up with GAL_E_CHAIN                    */
    uint32_t flags;             /* GAL_CFG_*                                                                   */
    int32_t reserved[2];
} gal_synth_cfg_t;

/* gal_synth_cfg_t.flags */
#define GAL_CFG_CBOC 2u          /* opt-in: CBOC(6,1,1/11), the composite sub-carrier of the E1 OS ICD, instead of the
                                    BOC(1,1) the reference generates (src/gal-sig.cpp:198-233; the reference has no such
                                    mode).  Definition, per channel and sample, with x = code phase in chips,
                                    B = E1B chip x data symbol, C = E1C chip x secondary code (all +-1), sc_A = +1 if
                                    (int)(2 x) is odd else -1 (the reference's sboc convention), sc_B likewise from
                                    (int)(12 x), k = ((int)(511 carr_phase)) & 511:
                                        I += sc_A (B - C) TAcos[k] + sc_B (B + C) TBcos[k]      (Q with the sin tables)
                                    TA = lround(sqrt(10/11) x cosTable512 / sinTable512), TB = lround(sqrt(1/11) x ...);
                                    everything else (wrap, symbol, page, NCO updates, int16 store) as src/galileo-sdr.cpp:481-539 */
#define GAL_CFG_EXACT_REPLAY 4u  /* always synthesise with the exact-replay kernel (every lane steps the reference's two NCO
                                    recurrences sample by sample and checks its end state against the next checkpoint), also where
                                    the default kernel of the reference geometry -- 16-sample groups from closed-form start
                                    states, undecided groups replayed exactly -- could run.  Same bits either way; this one is
                                    slower and carries the replay self-check (gal_synth_stats_t.kernel_family says which ran) */
#define GAL_CFG_VERIFY_ALL 8u    /* accepted and ignored: what it asked for is the default since 0.4 (see GAL_CFG_VERIFY_SAMPLED)     */
#define GAL_CFG_VERIFY_SAMPLED 16u /* batches of the default kernel (kernel_family 1), which never forms an exact phase itself.  DEFAULT
                                    (flag clear): every carrier leg and every code leg of the executed epochs is walked once more from
                                    its own first checkpoint, genuinely, in every batch, and every checkpoint must come out bit for
                                    bit (k_verify_carr, k_verify_code) -- what the exact-replay kernel establishes on its way.
                                    Flag set (round 5's default; ~3-5 % less per step): an eighth of the leg positions of both
                                    chains per batch, rotating with the handle's batch count -- every (epoch, leg) position of a
                                    repeated plan is re-walked once per N = 8 batches --, plus every carrier leg whose translation
                                    used more than 1/256 of its margin, plus the chunks k_repair_g walks to their ends.  A wrong
                                    translation (a proof would have to be wrong) is then caught within 8 batches instead of in
                                    the batch it happens in                                                  */
#define GAL_CFG_SINGLE_STREAM 1u /* enqueue every kernel on the handle's stream (no internal high-priority walker
                                    streams): for callers that capture or serialise the stream themselves       */

typedef struct gal_synth_stats {
    int32_t walk_passes;        /* carrier-walker passes the last execute needed                               */
    int32_t chain_mismatch;     /* #chunk boundaries where the replay kernel disagreed with the walker (must be 0) */
    int32_t n_epochs;
    int32_t n_active_max;       /* max simultaneously active channels in the plan                              */
    int32_t chunk_samples;
    int32_t chunks_per_epoch;
    float   ms_walk;            /* device time of the walker kernels, last execute (0 if timing disabled)      */
    float   ms_synth;           /* device time of the synthesis kernel, last execute                           */
    int32_t window_mode;        /* fast body of the synthesis kernel: 1 / 2 / 3 / 4 resampled windows (one chip-pattern look-up
                                   per 16 samples; 1: 0.74 <= 2 f_code / fs < 1, as at the reference's 2.6 MS/s; 2: 2 f_code / fs
                                   <= 0.133, sample rates from 15.4 MS/s; 3: <= 0.266, from 7.7 MS/s; 4: what lies between,
                                   2.77 .. 7.7 MS/s (kernel_family 1 only); all need well separated pattern thresholds --
                                   a rate at which 2 f_code / fs is within 1e-3 of a fraction with a denominator up to 15,
                                   e.g. 4.092 MS/s, has none on the exact-replay kernel), 0 per-sample window index (any rate).
                                   + 16 (0.4, kernel_family 1): the thresholds crowd and a group's pattern was found by bisection
                                   over them instead of through the bin table.  Same bits either way.                       */
    int32_t synth_runs;         /* synthesis launches the last batch took: 1, or 2 when gal_synth_finish() had to repeat
                                   it (carrier chain not complete when the kernel was started, or the replay check failed) */
    int32_t kernel_family;      /* 0: exact replay, one chunk of ~1000 samples per lane (any rate, any signal); 1: one 16-sample
                                   group per lane, start states in closed form from the chunk's exact checkpoint, groups whose
                                   chip pattern or table index hangs on the rounding history replayed exactly afterwards -- the
                                   default wherever gal_synth_plan's gate admits the batch: automatic chunking, every code step
                                   in one form of the resampled windows (window_mode 1 ... 4: any rate from 2.05 MS/s up; pattern
                                   thresholds that crowd -- 4.092, 8.184 MS/s ... -- are searched by bisection since 0.4, BOC(1,1)
                                   only), every carrier step 0 or in [2^-40, 0.0147] cycles per sample;
                                   BOC(1,1) and the CBOC mode in all four forms                                              */
    int32_t repaired_groups;    /* family 1: 16-sample groups that were replayed exactly (about 1 in 10 000)                */
    float   ms_repair;          /* family 1: device time of that replay (k_repair_g, behind the synthesis kernel; not in ms_synth) */
    int32_t exact_records;      /* family 1: records (channel-epochs) of the batch that are not fit for the group kernel -- a carrier
                                   that stands still or is faster than 120 table entries per 16 samples, pattern thresholds that
                                   crowd, another window form than the batch's -- and took an accumulating exact-replay launch
                                   behind it (0 in every scenario of the reference's geometry seen so far)                      */
    float   ms_plan;            /* 0.4: host time of the gal_synth_plan[_async] call behind this batch (validation, lists, staging)    */
    float   ms_h2d;             /* 0.4: device time of its host->device copy (one copy of the SoA records out of pinned memory)        */
} gal_synth_stats_t;
/* The struct only ever GROWS AT ITS END (0.2: 40 bytes, walk_passes .. synth_runs; 0.3: 56; 0.4: 64).  gal_synth_finish and
 * gal_synth_run_host are function-like macros over the _n entry points below, which copy min(the caller's sizeof, the library's)
 * bytes: a caller compiled against an older header never gets more than its own struct holds.  The plain SYMBOLS of those two
 * names stay exported for binaries built against 0.2 and fill exactly those 40 bytes.  Caveat of the macros: the two names cannot
 * be used as function pointers -- take &gal_synth_finish_n / &gal_synth_run_host_n (or define GAL_SYNTH_NO_SIZED_MACROS before
 * including this header and get the 40-byte symbols).  gal_synth_stats_size() = the library's sizeof. */

typedef struct gal_synth gal_synth_t;

/* Library / build identification; never fails. */
const char *gal_synth_version(void);
/* Text of the last error on this thread. */
const char *gal_synth_last_error(void);
/* Number of usable gfx950 devices (0 if none). */
int gal_synth_device_count(void);

int gal_synth_create(const gal_synth_cfg_t *cfg, gal_synth_t **out);
int gal_synth_destroy(gal_synth_t *h);

/* Use `hip_stream` (a hipStream_t) for all work of this handle; NULL = the handle's own stream. */
int gal_synth_set_stream(gal_synth_t *h, void *hip_stream);

/*
 * Upload the per-epoch parameters of a batch: params[e * n_slots + s], e < n_epochs (host memory).
 * state_in (host, n_slots entries, may be NULL for a fresh run) gives carr_phase/page for channels
 * that continue from a previous batch (records without GAL_CH_RESTART in their first active epoch).
 * After this call the batch is resident in HBM.  GAL_E_STATE while a batch is in flight (gal_synth_plan_async may be called then).
 */
int gal_synth_plan(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs,
                   const gal_chan_state_t *state_in);

/*
 * The same without waiting: the host work of the plan -- validation, lists, the SoA split into the handle's pinned staging buffer --
 * is done when it returns (`params` and `state_in` may be reused).
 *   - Nothing in flight on the handle: the upload is ENQUEUED on the handle's stream; the walkers of the next
 *     gal_synth_execute[_range] wait for it on the device.
 *   - A batch IN FLIGHT (gal_synth_execute called, gal_synth_finish not yet): allowed -- this is plan(k+1) under execute(k).  The
 *     plan is staged on the host only (the arena still belongs to the batch in flight); the next gal_synth_execute[_range], which
 *     must come behind the gal_synth_finish of that batch, makes it the handle's plan, enqueues the upload and goes on.  Until then
 *     gal_synth_finish / gal_synth_walk_counts speak of the batch in flight; gal_synth_output_bytes of the staged plan.
 * This is how a caller with fresh parameters for every batch -- the reference computes them between epochs,
 * src/galileo-sdr.cpp:450-479 -- keeps two handles busy: finish(A); execute(A); plan_async(A, the scenario after next), the same
 * for B (bench.py: configs.fresh_plan).  One thread at a time per handle.  Errors of the upload itself surface in
 * gal_synth_execute / gal_synth_finish.
 */
int gal_synth_plan_async(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs,
                         const gal_chan_state_t *state_in);

/* Bytes of IQ the planned batch produces: n_epochs * samples_per_epoch * 4. */
size_t gal_synth_output_bytes(const gal_synth_t *h);

/*
 * Run the hot path for the planned batch: NCO walk + per-sample synthesis, writing
 * interleaved int16 I,Q (little endian) to iq_dev (DEVICE memory, 16-byte aligned).  Asynchronous: the
 * walker chain runs on the handle's own high-priority streams and starts at once (its inputs were uploaded by
 * gal_synth_plan, which is synchronous; it does not wait for other work on the handle's stream), the synthesis kernel
 * runs in order on the handle's stream; may be called repeatedly for the same plan.  Several
 * handles may be in flight on different streams: the latency-bound walk of one batch then runs beside the
 * synthesis kernel of another (bench.py --pipeline).
 */
int gal_synth_execute(gal_synth_t *h, int16_t *iq_dev);

/*
 * The same for the epochs [first_epoch, first_epoch + n_epochs) of the planned batch only: iq_dev receives
 * n_epochs * samples_per_epoch * 4 bytes.  This is how ONE scenario is cut into contiguous epoch ranges for several GPUs
 * without any exchange (bench.py --shard scenario): every rank plans the whole scenario and executes its own range.
 * The carrier chain never restarts, so the NCO walk covers the epochs [0, first_epoch + n_epochs): those in front of the
 * range silently (their states are needed, their checkpoints are not), those behind it not at all -- a rank's walker
 * work grows with the prefix it needs, not with the plan.  gal_synth_finish then returns the channel state at the END OF
 * THE EXECUTED RANGE (= the end of the plan for a range that reaches it, in particular for gal_synth_execute).
 */
int gal_synth_execute_range(gal_synth_t *h, int16_t *iq_dev, int32_t first_epoch, int32_t n_epochs);

/* Wait for the batch (its completion record -- counters, end state, a sequence number -- is written by the device into
 * pinned host memory behind the synthesis kernel and polled here; work the caller has enqueued on the stream BEHIND
 * gal_synth_execute is not waited for), check the chain self-check, return the end-of-batch channel state (host,
 * n_slots entries, may be NULL) and statistics (may be NULL).  The IQ in iq_dev is FINAL ONLY AFTER THIS CALL HAS
 * RETURNED GAL_OK: the carrier chain is evaluated speculatively, and if the speculation was not verified in time
 * (or the replay check disagreed) finish() repeats the synthesis into iq_dev.  Do not enqueue copies out of
 * iq_dev between execute() and finish(). */
int gal_synth_finish(gal_synth_t *h, gal_chan_state_t *state_out, gal_synth_stats_t *stats);
int gal_synth_finish_n(gal_synth_t *h, gal_chan_state_t *state_out, void *stats, size_t stats_bytes);
size_t gal_synth_stats_size(void);

/* Diagnostics of the last gal_synth_finish(): carrier-chain legs evaluated by walking, legs accepted by
 * translation (csrc/nco_walk.h: binade_margin), and how often (since create) the replay check forced the
 * all-walked fallback.  Any pointer may be NULL. */
int gal_synth_walk_counts(const gal_synth_t *h, int64_t *legs_walked, int64_t *legs_translated, int64_t *fallbacks);

/* Convenience: plan + execute into an internal device buffer + copy to host `iq_host` + finish. */
int gal_synth_run_host(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs,
                       const gal_chan_state_t *state_in, int16_t *iq_host,
                       gal_chan_state_t *state_out, gal_synth_stats_t *stats);
int gal_synth_run_host_n(gal_synth_t *h, const gal_chan_epoch_t *params, int32_t n_epochs,
                         const gal_chan_state_t *state_in, int16_t *iq_host,
                         gal_chan_state_t *state_out, void *stats, size_t stats_bytes);
#ifndef GAL_SYNTH_NO_SIZED_MACROS
#define gal_synth_finish(h, state_out, stats) gal_synth_finish_n((h), (state_out), (stats), sizeof(gal_synth_stats_t))
#define gal_synth_run_host(h, params, n_epochs, state_in, iq_host, state_out, stats) \
    gal_synth_run_host_n((h), (params), (n_epochs), (state_in), (iq_host), (state_out), (stats), sizeof(gal_synth_stats_t))
#endif

/* Signal tables as the engine uses them (for tests and for the oracle to share DATA, not code). */
const uint32_t *gal_tables_e1b(void);   /* [50][128] */
const uint32_t *gal_tables_e1c(void);   /* [50][128] */
const int16_t  *gal_tables_cos512(void);/* [512]     */
const int16_t  *gal_tables_sin512(void);/* [512]     */
uint32_t        gal_tables_cs25(void);

#ifdef __cplusplus
}
#endif
#endif /* GALSYNTH_H_ */
