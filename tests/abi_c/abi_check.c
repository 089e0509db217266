/* Plain C99 consumer of the two C-ABI headers: they must compile as C, the record layouts must be what the
 * Python mirror (and INTEGRATION.md) say, and the entry points that need no GPU must work from C. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "galscen.h"
#include "galsynth.h"

_Static_assert(sizeof(gal_chan_epoch_t) == 176, "gal_chan_epoch_t");
_Static_assert(sizeof(gal_chan_state_t) == 80, "gal_chan_state_t");
_Static_assert(offsetof(gal_chan_epoch_t, f_carr) == 16, "f_carr");
_Static_assert(offsetof(gal_chan_epoch_t, page_next) == 48, "page_next");
_Static_assert(offsetof(gal_chan_epoch_t, page_init) == 112, "page_init");
_Static_assert(offsetof(gal_chan_state_t, page) == 8, "state.page");
_Static_assert(offsetof(gal_synth_cfg_t, flags) == 28, "cfg.flags");
_Static_assert(sizeof(gal_synth_cfg_t) == 40, "gal_synth_cfg_t");
_Static_assert(sizeof(gal_synth_stats_t) == 64, "gal_synth_stats_t");
_Static_assert(offsetof(gal_synth_stats_t, ms_plan) == 56, "stats.ms_plan");
_Static_assert(offsetof(gal_synth_stats_t, ms_h2d) == 60, "stats.ms_h2d");
_Static_assert(offsetof(gal_synth_stats_t, ms_walk) == 24, "stats.ms_walk");
_Static_assert(offsetof(gal_synth_stats_t, window_mode) == 32, "stats.window_mode");
_Static_assert(offsetof(gal_synth_stats_t, synth_runs) == 36, "stats.synth_runs");
_Static_assert(offsetof(gal_synth_stats_t, kernel_family) == 40, "stats.kernel_family");
_Static_assert(offsetof(gal_synth_stats_t, repaired_groups) == 44, "stats.repaired_groups");
_Static_assert(offsetof(gal_synth_stats_t, ms_repair) == 48, "stats.ms_repair");

int main(int argc, char **argv)
{
    gal_scen_cfg_t sc;
    gal_scen_t *scen = NULL;
    gal_synth_t *eng = NULL;
    gal_synth_cfg_t cfg;
    int rc;

    printf("%s\n", gal_synth_version());
    if (gal_tables_cs25() == 0 || gal_tables_cos512()[0] != 250) return 2;

    memset(&sc, 0, sizeof(sc));
    sc.nav_file = argc > 1 ? argv[1] : "/nonexistent.rnx";
    sc.llh[0] = -6; sc.llh[1] = 51; sc.llh[2] = 100;
    sc.have_start = 1;
    sc.start[0] = 2022; sc.start[1] = 2; sc.start[2] = 20; sc.start[3] = 12; sc.start[4] = 0;
    sc.duration_s = 2.0;
    sc.iono_enable = 1;
    sc.n_slots = GAL_MAX_CHAN;
    rc = gal_scen_open(&sc, &scen);
    if (argc > 1) {
        gal_chan_epoch_t rows[3 * GAL_MAX_CHAN];
        int n, i, act = 0;
        if (rc != GAL_OK) { printf("open failed: %s\n", gal_scen_last_error()); return 3; }
        if (gal_scen_total_epochs(scen) != 19) return 4;
        n = gal_scen_next(scen, 3, rows);
        if (n != 3) return 5;
        for (i = 0; i < GAL_MAX_CHAN; i++) act += rows[i].prn > 0;
        printf("epochs %d, active channels %d\n", n, act);
        gal_scen_close(scen);
    } else if (rc != GAL_E_IO) {
        return 6;
    }
    /* the synthesis engine has no CPU fallback: without a GPU create() must fail with GAL_E_DEVICE and a message */
    memset(&cfg, 0, sizeof(cfg));
    cfg.sample_rate = 2.6e6; cfg.samples_per_epoch = 260000; cfg.n_slots = GAL_MAX_CHAN; cfg.device = -1;
    rc = gal_synth_create(&cfg, &eng);
    printf("create: %d %s\n", rc, rc ? gal_synth_last_error() : "ok");
    if (rc == GAL_OK) gal_synth_destroy(eng);
    return 0;
}
