/*
 * ref_loop_harness.cpp -- TEST INFRASTRUCTURE: compiles the reference's per-sample loop FROM ITS OWN TEXT.
 *
 * Nothing of the reference is copied into this repository.  oracle/Makefile (target _ref/libref_loop.so, only when
 * /root/reference is present) cuts, at build time and from where they lie, into git-ignored files under oracle/_ref/:
 *   ref_loop_body.inc    src/galileo-sdr.cpp:481-539     the `for (isamp ...)` statement, verbatim
 *   ref_loop_types.inc   include/structures.h:43-162     galtime_t, gtime_t, datetime_t, ephem_t, ionoutc_t, range_t,
 *                                                        channel_t, verbatim (the header itself needs uhd/boost: :1-2)
 *   ref_loop_codegen.inc src/gal-sig.cpp:9-233           hex_to_binary_converter, sboc, codegen_E1B / codegen_E1C
 * and this file supplies ONLY the locals those fragments name, with the declarations galileo_task() gives them
 * (src/galileo-sdr.cpp:32,95-114,160-162), plus a generateINavMsg that installs the page the caller provides
 * (the real one, src/inav-msg.cpp:28-54, is a host function outside the loop; what it produces is an input here).
 * include/constants.h is the reference's header itself (-I/root/reference/include); it compiles stand-alone.
 * Compiled with the reference's own flags (CMakeLists.txt:18-22: -std=c++11 -g -DDEBUG, no -O).
 *
 * Interface (ctypes, tests/ref_loop_binding.py): the same records as oracle/galsyn_oracle.c, so the restatement and the
 * reference's text run on identical inputs and are compared int16 by int16 and state bit by state bit.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <ctime>

#include "constants.h" /* the reference's: tables, MAX_CHAN, CA_SEQ_LEN_E1, N_SYM_PAGE, TX_SAMPLERATE */

#include "_ref/ref_loop_types.inc"   /* galtime_t ... channel_t: reference text       */
#include "_ref/ref_loop_codegen.inc" /* code expansion: reference text                */

#include "../include/galsynth.h" /* the record layout the tests use (ours) */

namespace {
/* What the fragment names besides chan/iq_buff: the ephemeris store and the page producer, with the reference's own
 * ephem_t / ionoutc_t (`eph = eph_vector[sv][current_eph[sv]]; generateINavMsg(grx, &chan[i], &eph, &iono);`). */
std::vector<ephem_t> eph_vector[GAL_NUM_PRN + 1];
std::vector<int> current_eph;

const gal_chan_epoch_t *g_row; /* the epoch's records: page_next per slot */
channel_t *g_chan0;

/* Installs the page the reference's generator would have produced for this epoch (src/galileo-sdr.cpp:505 calls it with
 * the epoch's grx; chan->page is overwritten in place, src/inav-msg.cpp:44-52). */
void generateINavMsg(galtime_t, channel_t *c, ephem_t *, ionoutc_t *)
{
    const uint32_t *w = g_row[c - g_chan0].page_next;
    for (int i = 0; i < N_SYM_PAGE; i++) c->page[i] = (int)((w[i >> 5] >> (i & 31)) & 1u);
}

long get_nanos() { return 0; } /* src/galileo-sdr.cpp:485 reads the clock into an unused local */
} // namespace

extern "C" int ref_loop_sample_rate(void) { return (int)TX_SAMPLERATE; }
extern "C" int ref_loop_samples_per_epoch(void) { return (int)NUM_IQ_SAMPLES; }

/* codegen_E1B / codegen_E1C of the reference for one PRN (8184 shorts). */
extern "C" void ref_loop_codegen(int prn, int e1c, short *ca)
{
    if (e1c) codegen_E1C(ca, prn); else codegen_E1B(ca, prn);
}

/*
 * n_epochs x n_slots records (n_slots <= MAX_CHAN), state_in/state_out n_slots entries, iq_out n_epochs * iq_buff_size * 2.
 * samples_per_epoch overrides NUM_IQ_SAMPLES for short batches (the loop bound is the local iq_buff_size either way).
 */
extern "C" int ref_loop_run(const gal_chan_epoch_t *params, int n_epochs, int n_slots, int samples_per_epoch,
                            const gal_chan_state_t *state_in, short *iq_out, gal_chan_state_t *state_out)
{
    if (n_slots > MAX_CHAN || n_slots < 1) return -1;
    /* locals of galileo_task(), declared as there (src/galileo-sdr.cpp:95-114) */
    channel_t chan[MAX_CHAN];
    int ip, qp;
    short *iq_buff = NULL;
    double delt;
    int isamp;
    int iq_buff_size;
    int i, sv;
    ephem_t eph;
    ionoutc_t iono;
    galtime_t grx;
    memset(&grx, 0, sizeof(grx));
    memset(chan, 0, sizeof(chan));
    (void)ip; (void)qp;

    iq_buff_size = samples_per_epoch > 0 ? samples_per_epoch : NUM_IQ_SAMPLES; /* :160 */
    delt = 1.0 / (double)TX_SAMPLERATE;                                       /* :162 */

    current_eph.assign(GAL_NUM_PRN + 1, 0);
    for (i = 0; i <= GAL_NUM_PRN; i++) eph_vector[i].assign(1, ephem_t());
    g_chan0 = chan;

    for (i = 0; i < MAX_CHAN; i++) { /* src/channel.cpp:11-14 allocates these per channel */
        chan[i].ca_E1B = (short *)calloc(2 * CA_SEQ_LEN_E1, sizeof(short));
        chan[i].ca_E1C = (short *)calloc(2 * CA_SEQ_LEN_E1, sizeof(short));
        chan[i].page = (int *)calloc(N_SYM_PAGE, sizeof(int));
        chan[i].prn = 0;
    }
    for (i = 0; i < n_slots; i++)
        if (state_in && state_in[i].prn > 0) {
            chan[i].prn = state_in[i].prn;
            chan[i].carr_phase = state_in[i].carr_phase;
            for (int b = 0; b < N_SYM_PAGE; b++) chan[i].page[b] = (int)((state_in[i].page[b >> 5] >> (b & 31)) & 1u);
            codegen_E1B(chan[i].ca_E1B, chan[i].prn);
            codegen_E1C(chan[i].ca_E1C, chan[i].prn);
        }

    for (int e = 0; e < n_epochs; e++) {
        const gal_chan_epoch_t *row = params + (size_t)e * n_slots;
        g_row = row;
        /* what computeCodePhase / allocateChannel leave in channel_t before the loop (src/gal-sig.cpp:308-347,
         * src/channel.cpp:81-99): inputs of the fragment */
        for (i = 0; i < n_slots; i++) {
            const gal_chan_epoch_t *r = row + i;
            if (r->prn <= 0) { chan[i].prn = 0; continue; }
            if ((r->flags & GAL_CH_RESTART) || chan[i].prn != r->prn) {
                if (!(r->flags & GAL_CH_RESTART)) return -2;
                chan[i].prn = r->prn;
                codegen_E1B(chan[i].ca_E1B, r->prn);
                codegen_E1C(chan[i].ca_E1C, r->prn);
                chan[i].carr_phase = r->carr_phase0;
                for (int b = 0; b < N_SYM_PAGE; b++) chan[i].page[b] = (int)((r->page_init[b >> 5] >> (b & 31)) & 1u);
            }
            chan[i].f_carr = r->f_carr;
            chan[i].f_code = r->f_code;
            chan[i].code_phase = r->code_phase0;
            chan[i].ibit = r->ibit0;
        }
        iq_buff = iq_out + (size_t)e * iq_buff_size * 2;

#include "_ref/ref_loop_body.inc" /* src/galileo-sdr.cpp:481-539, verbatim */
    }

    for (i = 0; i < n_slots; i++) {
        memset(&state_out[i], 0, sizeof(state_out[i]));
        state_out[i].prn = chan[i].prn;
        if (chan[i].prn > 0) {
            state_out[i].carr_phase = chan[i].carr_phase;
            for (int b = 0; b < N_SYM_PAGE; b++)
                if (chan[i].page[b] > 0) state_out[i].page[b >> 5] |= 1u << (b & 31);
        }
    }
    for (i = 0; i < MAX_CHAN; i++) { free(chan[i].ca_E1B); free(chan[i].ca_E1C); free(chan[i].page); }
    return 0;
}
