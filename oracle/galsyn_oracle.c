/*
 * galsyn_oracle.c -- CPU restatement of the reference's per-sample synthesis loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (galileo-sdr-sim_amd/, the C-ABI library, the
 * CLI) may include, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker.
 *
 * It follows harshadms/galileo-sdr-sim @2024_10_08 operation for operation:
 *   - sample loop, channel loop, accumulate, store ......... src/galileo-sdr.cpp:481-539
 *   - code expansion hex -> chips -> BOC(1,1) half chips ... src/gal-sig.cpp:9-233
 *   - tables (carrier LUT, CS25) ............................ include/constants.h:213-284
 *   - state overwritten at epoch start ..................... src/gal-sig.cpp:308-347 (values arrive
 *     in gal_chan_epoch_t), carrier phase / page carried ... src/galileo-sdr.cpp:531-532, :505
 * Compile with -O2 -ffp-contract=off (x86-64 baseline has no FMA, so the reference never fuses).
 *
 * Parity pins:
 *   0. THE WHOLE PROGRAM: oracle/_ref/ref_task is the reference's file-sink program compiled from the reference's own text
 *      (ref_task_harness.cpp: seven header lines that name UHD / Boost and the USRP sender omitted, nothing rewritten, no stand-in;
 *      reference flags).  It reproduces every recorded md5 G1..G9; tests/test_ref_task.py and tools/ref_task_fuzz.py compare
 *      front-end rows -> this oracle with what it writes, on scenarios no recorded md5 covers.
 *   1. THE LOOP ITSELF: oracle/_ref/libref_loop.so is src/galileo-sdr.cpp:481-539 compiled from the reference's own text
 *      (cut out at build time by oracle/Makefile, reference flags, no stand-in header; ref_loop_harness.cpp supplies only
 *      the locals the fragment names).  tests/test_ref_loop.py: this file hashes equal to it, epoch by epoch, on the G1
 *      rows and on two adversarial kernel-boundary batches (committed SHA-256s, tests/golden/ref_loop_sha256.npz), and is
 *      compared with it int16 by int16 on random batches wherever the library is present.
 *   2. Tables: oracle/_ref/ref_tables_dump (include/constants.h compiled here), tests/test_tables.py.
 *   3. End to end: md5s of the reference BINARY's output files G1..G9 (tests/golden/reference_md5.json; builds with
 *      stand-in Boost/UHD headers, by the survey and three judges): front-end rows -> this oracle -> same md5
 *      (tests/test_golden_scenarios.py) -- and the same rows through libref_loop.so give the same md5s
 *      (profiles/r04_ref_loop_all_md5.log).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/galsynth.h"
#include "../galileo-sdr-sim_amd/csrc/e1_tables.inc" /* DATA shared with the engine, not code */

#define HALF_CHIPS (2 * GAL_CODE_LEN)

typedef struct {
    int    prn;
    short  ca_E1B[HALF_CHIPS];
    short  ca_E1C[HALF_CHIPS];
    double f_carr, f_code;
    int    page[GAL_N_SYM_PAGE];
    double carr_phase;
    double code_phase;
    int    ibit;
} ochan_t;

static int o_cos[512], o_sin[512];
static char o_cs25[25];
static int o_tables_ready = 0;
/* CBOC(6,1,1/11) opt-in mode (NOT in the reference, which generates BOC(1,1) only, src/gal-sig.cpp:198-233):
 * integer carrier tables scaled by alpha = sqrt(10/11) and beta = sqrt(1/11) of the E1 OS ICD's composite
 * sub-carrier, so that the accumulation stays integer like the reference's (src/galileo-sdr.cpp:520-525). */
static int o_cosA[512], o_sinA[512], o_cosB[512], o_sinB[512];

static void o_init_tables(void)
{
    int k;
    if (o_tables_ready) return;
    for (k = 0; k < 512; k++) {
        int c;
        if (k < 128) c = kCosQ[k];
        else if (k < 256) c = -kCosQ[255 - k];
        else c = (511 - k) < 128 ? kCosQ[511 - k] : -kCosQ[255 - (511 - k)];
        o_cos[k] = c;
    }
    for (k = 0; k < 512; k++) o_sin[k] = o_cos[(k - 128) & 511];
    for (k = 0; k < 25; k++) o_cs25[k] = (char)((kCS25 >> k) & 1u);
    {
        const double alpha = sqrt(10.0 / 11.0), beta = sqrt(1.0 / 11.0);
        for (k = 0; k < 512; k++) {
            o_cosA[k] = (int)lround(alpha * (double)o_cos[k]);
            o_sinA[k] = (int)lround(alpha * (double)o_sin[k]);
            o_cosB[k] = (int)lround(beta * (double)o_cos[k]);
            o_sinB[k] = (int)lround(beta * (double)o_sin[k]);
        }
    }
    o_tables_ready = 1;
}

/* hex_to_binary_converter + sboc(…,1,1): logic level 0 -> +1, 1 -> -1; each chip c becomes [-c, +c]
 * (src/gal-sig.cpp:9-191, 198-213, 219-233). */
static void o_codegen(short *ca, const uint32_t *packed)
{
    int i;
    for (i = 0; i < GAL_CODE_LEN; i++) {
        short c = ((packed[i >> 5] >> (i & 31)) & 1u) ? -1 : 1;
        ca[2 * i] = (short)-c;
        ca[2 * i + 1] = c;
    }
}

static void o_unpack_page(int *page, const uint32_t *w)
{
    int i;
    for (i = 0; i < GAL_N_SYM_PAGE; i++) page[i] = (int)((w[i >> 5] >> (i & 31)) & 1u);
}

static void o_pack_page(uint32_t *w, const int *page)
{
    int i;
    memset(w, 0, GAL_PAGE_WORDS * sizeof(uint32_t));
    for (i = 0; i < GAL_N_SYM_PAGE; i++)
        if (page[i] > 0) w[i >> 5] |= 1u << (i & 31);
}

static long o_get_nanos(void)
{
    struct timespec ts;
    timespec_get(&ts, TIME_UTC);
    return (long)ts.tv_sec * 1000000000L + ts.tv_nsec;
}

/*
 * Returns 0, or -1 on a malformed batch.  clock_read != 0 reproduces the reference's per-sample
 * get_nanos() (src/galileo-sdr.cpp:485) for the timed CPU baseline.
 * signal: 0 = BOC(1,1), the reference's loop, statement for statement; 1 = CBOC(6,1,1/11), see o_cosA above.  Always
 * called with a literal, so that each mode is compiled as its own loop.
 */
static inline __attribute__((always_inline)) int o_run(const gal_chan_epoch_t *params, int n_epochs, int n_slots,
                                                       int samples_per_epoch, double sample_rate,
                                                       const gal_chan_state_t *state_in, int16_t *iq_out,
                                                       gal_chan_state_t *state_out, int clock_read, const int signal)
{
    ochan_t *chan;
    int e, i, isamp;
    double delt = 1.0 / sample_rate; /* src/galileo-sdr.cpp:162 */
    volatile double sink = 0;

    o_init_tables();
    if (n_slots < 1 || n_slots > GAL_ENGINE_MAX_CHAN) return -1;
    chan = (ochan_t *)calloc((size_t)n_slots, sizeof(ochan_t));
    if (!chan) return -1;

    for (i = 0; i < n_slots; i++) {
        chan[i].prn = 0;
        if (state_in && state_in[i].prn > 0) {
            chan[i].prn = state_in[i].prn;
            chan[i].carr_phase = state_in[i].carr_phase;
            o_unpack_page(chan[i].page, state_in[i].page);
            o_codegen(chan[i].ca_E1B, kE1B[chan[i].prn - 1]);
            o_codegen(chan[i].ca_E1C, kE1C[chan[i].prn - 1]);
        }
    }

    for (e = 0; e < n_epochs; e++) {
        const gal_chan_epoch_t *row = params + (size_t)e * n_slots;
        int16_t *iq_buff = iq_out + (size_t)e * samples_per_epoch * 2;

        /* epoch start: what allocateChannel / computeCodePhase left in chan[] */
        for (i = 0; i < n_slots; i++) {
            const gal_chan_epoch_t *r = &row[i];
            if (r->prn <= 0) { chan[i].prn = 0; continue; }
            if (r->prn > GAL_NUM_PRN) { free(chan); return -1; }
            if (r->flags & GAL_CH_RESTART) {
                chan[i].prn = r->prn;
                o_codegen(chan[i].ca_E1B, kE1B[r->prn - 1]);
                o_codegen(chan[i].ca_E1C, kE1C[r->prn - 1]);
                o_unpack_page(chan[i].page, r->page_init);
                chan[i].carr_phase = r->carr_phase0;
            } else if (chan[i].prn != r->prn) {
                free(chan);
                return -1; /* continuing channel without state */
            }
            chan[i].f_carr = r->f_carr;
            chan[i].f_code = r->f_code;
            chan[i].code_phase = r->code_phase0;
            chan[i].ibit = r->ibit0;
        }

        for (isamp = 0; isamp < samples_per_epoch; isamp++) {
            int i_acc = 0;
            int q_acc = 0;
            if (clock_read) sink = (double)o_get_nanos();
            for (i = 0; i < n_slots; i++) {
                if (chan[i].prn > 0) {
                    int cosPh, sinPh, icode, E1B_chip, E1C_chip, databit, secCode, ip, qp;
                    if (chan[i].code_phase >= GAL_CODE_LEN) {
                        chan[i].code_phase -= GAL_CODE_LEN;
                        chan[i].ibit++;
                        if (chan[i].ibit >= GAL_N_SYM_PAGE) {
                            chan[i].ibit = 0;
                            o_unpack_page(chan[i].page, row[i].page_next);
                        }
                    }
                    cosPh = o_cos[((int)(511 * chan[i].carr_phase)) & 511];
                    sinPh = o_sin[((int)(511 * chan[i].carr_phase)) & 511];

                    icode = (int)(chan[i].code_phase * 2);

                    E1B_chip = chan[i].ca_E1B[icode];
                    E1C_chip = chan[i].ca_E1C[icode];

                    databit = chan[i].page[chan[i].ibit] > 0 ? -1 : 1;
                    secCode = o_cs25[chan[i].ibit % 25] > 0 ? -1 : 1;

                    if (signal == 0) {
                    ip = (E1B_chip * databit - E1C_chip * secCode) * cosPh;
                    qp = (E1B_chip * databit - E1C_chip * secCode) * sinPh;
                    } else {
                        /* E1 OS ICD: e_B (alpha sc_A + beta sc_B) - e_C (alpha sc_A - beta sc_B).  The expanded code
                         * arrays already carry sc_A = (icode odd ? +1 : -1) (sboc); sc_B is discretised the same way
                         * from (int)(12 x): first half of a BOC(6,1) period negative.  sc_A sc_B turns a value that
                         * carries sc_A into one that carries sc_B. */
                        const int k = ((int)(511 * chan[i].carr_phase)) & 511;
                        const int i12 = (int)(chan[i].code_phase * 12.0);
                        const int ab = ((icode ^ i12) & 1) ? -1 : 1;            /* sc_A * sc_B */
                        const int Ba = E1B_chip * databit, Ca = E1C_chip * secCode; /* with sc_A */
                        ip = (Ba - Ca) * o_cosA[k] + ab * (Ba + Ca) * o_cosB[k];
                        qp = (Ba - Ca) * o_sinA[k] + ab * (Ba + Ca) * o_sinB[k];
                    }

                    i_acc += ip;
                    q_acc += qp;

                    chan[i].code_phase += chan[i].f_code * delt;

                    chan[i].carr_phase += (chan[i].f_carr) * delt;
                    chan[i].carr_phase -= (long)chan[i].carr_phase;
                }
            }
            iq_buff[isamp * 2] = (short)i_acc;
            iq_buff[isamp * 2 + 1] = (short)q_acc;
        }
    }

    if (state_out) {
        for (i = 0; i < n_slots; i++) {
            memset(&state_out[i], 0, sizeof(state_out[i]));
            state_out[i].prn = chan[i].prn;
            if (chan[i].prn > 0) {
                state_out[i].carr_phase = chan[i].carr_phase;
                o_pack_page(state_out[i].page, chan[i].page);
            }
        }
    }
    (void)sink;
    free(chan);
    return 0;
}

int gal_oracle_run(const gal_chan_epoch_t *params, int n_epochs, int n_slots, int samples_per_epoch,
                   double sample_rate, const gal_chan_state_t *state_in, int16_t *iq_out,
                   gal_chan_state_t *state_out, int clock_read)
{
    return o_run(params, n_epochs, n_slots, samples_per_epoch, sample_rate, state_in, iq_out, state_out, clock_read, 0);
}

/* The CBOC(6,1,1/11) opt-in mode of the engine (GAL_CFG_CBOC): defined HERE, not by the reference. */
int gal_oracle_run_cboc(const gal_chan_epoch_t *params, int n_epochs, int n_slots, int samples_per_epoch,
                        double sample_rate, const gal_chan_state_t *state_in, int16_t *iq_out,
                        gal_chan_state_t *state_out)
{
    return o_run(params, n_epochs, n_slots, samples_per_epoch, sample_rate, state_in, iq_out, state_out, 0, 1);
}

void gal_oracle_cboc_tables(int *cosA, int *sinA, int *cosB, int *sinB)
{
    o_init_tables();
    memcpy(cosA, o_cosA, sizeof(o_cosA));
    memcpy(sinA, o_sinA, sizeof(o_sinA));
    memcpy(cosB, o_cosB, sizeof(o_cosB));
    memcpy(sinB, o_sinB, sizeof(o_sinB));
}

/* Expanded tables, for checking against oracle/_ref's dump of the reference header. */
void gal_oracle_tables(int *cos512, int *sin512, char *cs25)
{
    o_init_tables();
    memcpy(cos512, o_cos, sizeof(o_cos));
    memcpy(sin512, o_sin, sizeof(o_sin));
    memcpy(cs25, o_cs25, sizeof(o_cs25));
}

void gal_oracle_codegen(int prn, int e1c, short *ca /* 8184 */)
{
    o_codegen(ca, e1c ? kE1C[prn - 1] : kE1B[prn - 1]);
}
