// scenario.cpp -- host scenario driver: everything the reference's galileo_task() computes around its
// sample loop, emitted as gal_chan_epoch_t rows for the synthesis engine (C-ABI of include/galscen.h).
//
//   start-time selection ............ src/galileo-sdr.cpp:230-274, src/gnss-time.cpp:101-165
//   initial allocation, dt, grx ..... src/galileo-sdr.cpp:347-352, :436
//   per-epoch range / code phase .... src/galileo-sdr.cpp:438-479, src/gal-sig.cpp:308-347
//   30 s ephemeris / channel refresh  src/galileo-sdr.cpp:545-562
//   allocateChannel ................. src/channel.cpp:21-123
// The page a channel would install if its symbol counter wraps inside an epoch depends only on that
// epoch's receive time, the channel's ephemeris and the iono/UTC block (src/galileo-sdr.cpp:502-505),
// so it is generated here per epoch and shipped in the record (page_next).
#include <arpa/inet.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "scen_internal.h"

using namespace galscen;

namespace {

thread_local char g_scen_err[512] = "";

int scen_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_scen_err, sizeof(g_scen_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Channel {  // the part of channel_t that outlives an epoch
    int prn = 0;
    Range rho0;
    double carr_phase0 = 0.0;
    uint32_t page_init[GAL_PAGE_WORDS];
    bool fresh = false;  // allocated since the last emitted epoch
    // page cache: the page depends on the receive time only through (int)sec (TOW and the word schedule,
    // src/inav-msg.cpp:39-40,186), so it is regenerated once per second, not once per 0.1 s epoch
    int page_key_sec = -1, page_key_week = -1, page_key_eph = -1;
    uint32_t page_cached[GAL_PAGE_WORDS];
};

struct Vec3 {
    double v[3];
};

}  // namespace

struct gal_scen {
    gal_scen_cfg_t cfg;
    NavData nav;
    GalTime g0, grx;
    double xyz0[3];
    std::vector<Vec3> motion;  // -u positions, one per 0.1 s
    int numd = 0;
    int iumd = 1;  // next epoch (the loop starts at 1, src/galileo-sdr.cpp:438)
    int current_eph[kMaxSat];
    int allocated[kMaxSat];
    std::vector<Channel> chan;
    int udp_fd = -1;         // run-time position updates (cfg.udp_port)
    bool have_live = false;
    double xyz_live[3];
    int live_updates = 0, live_rejected = 0;
    bool gap_warned = false;  // ephemeris-gap policy, see gal_scen_next
    int eph_gaps = 0;         // (satellite, refresh) pairs that kept a stale record

    ~gal_scen()
    {
        if (udp_fd >= 0) close(udp_fd);
    }
};

namespace {

// src/galileo-sdr.cpp:443-448: the position of an epoch is whatever the location listener received last
// (include/socket.h:165-180: 3 doubles lat, lon [deg], height [m] per datagram); polled, never blocking
void poll_live_position(gal_scen *s)
{
    if (s->udp_fd < 0) return;
    double llh[3];
    bool got = false;
    for (;;) {  // drain the queue: the last well-formed datagram counts
        double buf[3];
        const ssize_t n = recv(s->udp_fd, buf, sizeof(buf), MSG_DONTWAIT | MSG_TRUNC);  // MSG_TRUNC: the real length
        if (n < 0) break;                          // nothing (more) waiting
        if (n != (ssize_t)sizeof(buf)) continue;   // not a position update: ignored
        // (the reference takes whatever arrives; a NaN or an out-of-range coordinate would go straight into the ranges
        // and from there into the NCO steps, so such datagrams are dropped here)
        if (!(std::isfinite(buf[0]) && std::isfinite(buf[1]) && std::isfinite(buf[2])) || std::fabs(buf[0]) > 90.0 ||
            std::fabs(buf[1]) > 360.0 || buf[2] < -1.0e4 || buf[2] > 1.0e8) {
            s->live_rejected++;
            continue;
        }
        memcpy(llh, buf, sizeof(llh));
        got = true;
    }
    if (!got) return;
    // stderr, not the reference's stdout (include/socket.h:178): stdout may be the IQ sink (-o -)
    if (s->cfg.verbose) fprintf(stderr, "Location Update: %f,%f,%f\n", llh[0], llh[1], llh[2]);
    llh[0] = llh[0] / kR2D;
    llh[1] = llh[1] / kR2D;
    llh_to_ecef(llh, s->xyz_live);
    s->have_live = true;
    s->live_updates++;
}

const double *position(const gal_scen *s, int iumd)
{
    if (s->have_live) return s->xyz_live;
    if (!s->motion.empty()) return s->motion[iumd < (int)s->motion.size() ? iumd : (int)s->motion.size() - 1].v;
    return s->xyz0;
}

// src/channel.cpp:21-123
void allocate_channels(gal_scen *s, const GalTime &grx, const double xyz[3])
{
    const int S = s->cfg.n_slots;
    for (int sv = 0; sv < kMaxSat; ++sv) {
        const std::vector<Ephemeris> &list = s->nav.sv[sv];
        if (list.empty()) continue;
        if (!list[0].valid) continue;
        const int k = match_ephemeris(grx, list);
        if (k < 0) continue;
        const Ephemeris &eph = list[k];
        double azel[2];
        if (sat_visible(eph, grx, xyz, 10, azel) == 1) {
            if (s->allocated[sv] == -1) {
                int i;
                for (i = 0; i < S; ++i) {
                    Channel &c = s->chan[i];
                    if (c.prn != 0) continue;
                    c.prn = sv + 1;
                    c.page_key_sec = -1;
                    int sym[kSymPerPage];
                    inav_page_symbols(grx, eph, s->nav.iono, sym);
                    pack_symbols(sym, c.page_init);
                    Range rho;
                    compute_range(&rho, eph, s->nav.iono, grx, xyz);
                    c.rho0 = rho;
                    const double r_xyz = rho.range;
                    const double origin[3] = {0.0, 0.0, 0.0};
                    compute_range(&rho, eph, s->nav.iono, grx, origin);
                    const double r_ref = rho.range;
                    const double phase_ini = (2.0 * r_ref - r_xyz) / kLambdaL1;
                    c.carr_phase0 = phase_ini - floor(phase_ini);
                    c.fresh = true;
                    if (s->cfg.verbose)
                        fprintf(stderr, "%02d %6.1f %5.1f %11.1f %5.5f\n", c.prn, azel[0] * kR2D, azel[1] * kR2D,
                                c.rho0.range, grx.sec);
                    break;
                }
                if (i < S) s->allocated[sv] = i;
            }
        } else if (s->allocated[sv] >= 0) {
            s->chan[s->allocated[sv]].prn = 0;
            s->allocated[sv] = -1;
        }
    }
}

int load_motion(const char *path, std::vector<Vec3> *out)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    char line[256];
    while (fgets(line, sizeof(line), fp)) {
        double t;
        Vec3 p;
        if (sscanf(line, "%lf,%lf,%lf,%lf", &t, &p.v[0], &p.v[1], &p.v[2]) == 4) out->push_back(p);
    }
    fclose(fp);
    return (int)out->size();
}

}  // namespace

extern "C" {

const char *gal_scen_last_error(void) { return g_scen_err; }

int gal_scen_open(const gal_scen_cfg_t *cfg, gal_scen_t **out)
{
    if (!cfg || !out || !cfg->nav_file) return scen_fail(GAL_E_INVAL, "gal_scen_open: null argument");
    *out = nullptr;
    if (cfg->n_slots < 1 || cfg->n_slots > GAL_ENGINE_MAX_CHAN) return scen_fail(GAL_E_INVAL, "bad n_slots");
    gal_scen *s = new (std::nothrow) gal_scen();
    if (!s) return scen_fail(GAL_E_NOMEM, "out of memory");
    s->cfg = *cfg;
    s->chan.resize(cfg->n_slots);
    std::string err;
    if (load_rinex3(cfg->nav_file, &s->nav, &err) < 0) {
        delete s;
        return scen_fail(GAL_E_IO, "%s", err.c_str());
    }
    if (s->nav.count == 0) {
        delete s;
        return scen_fail(GAL_E_IO, "no usable Galileo I/NAV ephemeris in %s", cfg->nav_file);
    }
    s->nav.iono.enable = cfg->iono_enable ? 1 : 0;
    s->nav.iono.nequick = 0;

    // receiver position: degrees -> radians with the reference's R2D, then ECEF
    double llh[3] = {cfg->llh[0], cfg->llh[1], cfg->llh[2]};
    llh[0] = llh[0] / kR2D;
    llh[1] = llh[1] / kR2D;
    llh_to_ecef(llh, s->xyz0);

    s->numd = (int)(((int)(cfg->duration_s * 10.0 + 0.5)) / 10.0 * 10.0 + 0.5);
    if (cfg->motion_file) {
        const int n = load_motion(cfg->motion_file, &s->motion);
        if (n <= 0) {
            delete s;
            return scen_fail(GAL_E_IO, "cannot read user motion file %s", cfg->motion_file);
        }
        if (n < s->numd) s->numd = n;
        memcpy(s->xyz0, s->motion[0].v, sizeof(s->xyz0));
    }
    if (s->numd < 2) {
        delete s;
        return scen_fail(GAL_E_EMPTY, "duration too short: nothing to generate");
    }

    // earliest / latest usable start (src/galileo-sdr.cpp:230-270)
    GalTime gmin, gmax;
    for (int sv = 0; sv < kMaxSat; ++sv) {
        if (s->nav.sv[sv].empty()) continue;
        if (s->nav.sv[sv][0].valid == 1) {
            gmin = s->nav.sv[sv][0].toc;
            break;
        }
    }
    for (int sv = 0; sv < kMaxSat; ++sv) {
        const size_t n = s->nav.sv[sv].size();
        if (n < 2) continue;
        const Ephemeris &e = s->nav.sv[sv][n - 2];
        if (e.valid == 1 && e.toc.sec > gmax.sec) gmax = e.toc;
    }
    if (cfg->time_overwrite && !cfg->have_start) {
        delete s;
        return scen_fail(GAL_E_INVAL, "time_overwrite (-T) needs a start time");
    }
    if (cfg->have_start) {
        CalTime t0;
        t0.y = cfg->start[0]; t0.m = cfg->start[1]; t0.d = cfg->start[2];
        t0.hh = cfg->start[3]; t0.mm = cfg->start[4];
        t0.sec = cfg->start_sec;
        if (t0.y <= 1980 || t0.m < 1 || t0.m > 12 || t0.d < 1 || t0.d > 31 || t0.hh < 0 || t0.hh > 23 ||
            t0.mm < 0 || t0.mm > 59 || t0.sec < 0.0 || t0.sec >= 60.0) {
            delete s;
            return scen_fail(GAL_E_INVAL, "ERROR: Invalid date and time.");
        }
        t0.sec = floor(t0.sec);
        cal_to_gal(t0, &s->g0);
        if (cfg->time_overwrite) {
            // src/gnss-time.cpp:105-137.  What the reference DOES with -T, built with its own flags: the UTC reference time of
            // the iono / UTC record is overwritten and the range check of -t is skipped -- and no ephemeris record is touched,
            // because the loop that should shift them runs `for (i = 0; i < neph; i++)` with galileo_task's `neph`
            // (src/galileo-sdr.cpp:85) never assigned: the first local frame of a fresh thread, zero.  (Inside the loop the
            // per-satellite vectors are walked with the two indices swapped besides.)  time_overwrite == 1 is that, checked
            // against the reference program itself (tools/ref_task_fuzz.py); == 2 does what the option sets out to do -- and
            // what gps-sdr-sim, its ancestor, does: shift EVERY record.
            GalTime gt;
            gt.week = s->g0.week;
            gt.sec = (double)(((int)(s->g0.sec)) / 7200) * 7200.0;
            const double dsec = gal_diff(gt, gmin);
            s->nav.iono.wnt = gt.week;
            s->nav.iono.tot = (int)gt.sec;
            if (cfg->time_overwrite == 2)
                for (int sv = 0; sv < kMaxSat; ++sv)
                    for (Ephemeris &e : s->nav.sv[sv])
                        if (e.valid == 1) {
                            e.toc.sec = e.toc.sec + dsec;  // incGalTime: seconds only, the week is left alone
                            e.toe.sec = e.toe.sec + dsec;
                        }
        } else if (gal_diff(s->g0, gmin) < 0.0 || gal_diff(gmax, s->g0) < 0.0) {
            CalTime a, b;
            gal_to_cal(gmin, &a);
            gal_to_cal(gmax, &b);
            const int wk = s->g0.week;
            const double sc = s->g0.sec;
            delete s;
            return scen_fail(GAL_E_INVAL,
                             "ERROR: Invalid start time (%d:%.0f). tmin = %4d/%02d/%02d,%02d:%02d:%02.0f "
                             "tmax = %4d/%02d/%02d,%02d:%02d:%02.0f",
                             wk, sc, a.y, a.m, a.d, a.hh, a.mm, a.sec, b.y, b.m, b.d, b.hh, b.mm, b.sec);
        }
    } else {
        s->g0 = gmin;
    }

    // src/galileo-sdr.cpp:297-317, 347-352, 436
    s->grx = s->g0;
    for (int sv = 0; sv < kMaxSat; ++sv) {
        s->current_eph[sv] = match_ephemeris(s->grx, s->nav.sv[sv]);
        s->allocated[sv] = -1;
    }
    s->grx.sec = s->grx.sec + kEpochDt;
    allocate_channels(s, s->grx, s->xyz0);
    if (cfg->time_overwrite) {
        // -T skips the range check of -t: say so when the start it let through leaves the sky empty
        int in_view = 0;
        for (int i = 0; i < cfg->n_slots; ++i) in_view += s->chan[i].prn > 0;
        if (in_view == 0)
            fprintf(stderr, "WARNING: no satellite in view at the start time -T let through%s: the IQ will be all zero\n",
                    cfg->time_overwrite == 1 ? " (time_overwrite 1 = the reference as built: no ephemeris record is shifted; "
                                               "2 / --shift-toe shifts TOC and TOE to the start time)" : "");
    }
    s->grx.sec = s->grx.sec + kEpochDt;
    s->iumd = 1;
    if (cfg->udp_port > 0) {
        s->udp_fd = socket(AF_INET, SOCK_DGRAM, 0);
        struct sockaddr_in addr;
        memset(&addr, 0, sizeof(addr));
        addr.sin_family = AF_INET;
        addr.sin_port = htons((uint16_t)cfg->udp_port);
        addr.sin_addr.s_addr = cfg->udp_loopback ? htonl(INADDR_LOOPBACK) : INADDR_ANY;
        if (s->udp_fd < 0 || bind(s->udp_fd, (struct sockaddr *)&addr, sizeof(addr)) < 0) {
            const int port = cfg->udp_port;
            delete s;
            return scen_fail(GAL_E_BUSY, "cannot listen for position updates on UDP port %d (the reference exits here too: "
                                         "one instance per port)", port);
        }
    }
    *out = s;
    return GAL_OK;
}

int32_t gal_scen_total_epochs(const gal_scen_t *s) { return s ? s->numd - 1 : 0; }

int gal_scen_start_time(const gal_scen_t *s, int32_t *week, double *sec)
{
    if (!s) return scen_fail(GAL_E_INVAL, "null handle");
    if (week) *week = s->g0.week;
    if (sec) *sec = s->g0.sec;
    return GAL_OK;
}

int32_t gal_scen_next(gal_scen_t *s, int32_t max_epochs, gal_chan_epoch_t *rows)
{
    if (!s || !rows || max_epochs < 0) return scen_fail(GAL_E_INVAL, "gal_scen_next: bad argument");
    const int S = s->cfg.n_slots;
    int produced = 0;
    while (produced < max_epochs && s->iumd < s->numd) {
        gal_chan_epoch_t *row = rows + (size_t)produced * S;
        memset(row, 0, sizeof(gal_chan_epoch_t) * S);
        poll_live_position(s);
        const double *xyz = position(s, s->iumd);
        for (int i = 0; i < S; ++i) {
            Channel &c = s->chan[i];
            if (c.prn <= 0) continue;
            const int sv = c.prn - 1;
            const int k = s->current_eph[sv];
            if (k < 0 || k >= (int)s->nav.sv[sv].size())
                return scen_fail(GAL_E_STATE, "PRN %d is allocated but has no current ephemeris (the reference "
                                              "indexes out of bounds here; strict_eph is set)", c.prn);
            const Ephemeris &eph = s->nav.sv[sv][k];
            Range rho;
            compute_range(&rho, eph, s->nav.iono, s->grx, xyz);

            // computeCodePhase, src/gal-sig.cpp:308-347
            const double rhorate = (rho.range - c.rho0.range) / kEpochDt;
            const double f_carr = (-rhorate / kLambdaE1);
            const double f_code = kCodeFreqE1 + f_carr * kCarrToCodeE1;
            double ms = (s->grx.sec - rho.range / kC) * 1000.0;
            const int ipage = ms / 2000.0;
            ms -= ipage * 2000;
            int ibit = (unsigned int)ms / 4;
            ms -= ibit * 4;
            const double code_phase = ms / 4 * GAL_CODE_LEN;
            ibit = (ibit + (kSymPerPage / 2)) % kSymPerPage;
            c.rho0 = rho;

            gal_chan_epoch_t &r = row[i];
            r.prn = c.prn;
            r.ibit0 = ibit;
            r.f_carr = f_carr;
            r.f_code = f_code;
            r.code_phase0 = code_phase;
            if (c.fresh) {
                r.flags |= GAL_CH_RESTART;
                r.carr_phase0 = c.carr_phase0;
                memcpy(r.page_init, c.page_init, sizeof(r.page_init));
                c.fresh = false;
            }
            if (c.page_key_sec != (int)s->grx.sec || c.page_key_week != s->grx.week || c.page_key_eph != k) {
                int sym[kSymPerPage];
                inav_page_symbols(s->grx, eph, s->nav.iono, sym);
                pack_symbols(sym, c.page_cached);
                c.page_key_sec = (int)s->grx.sec;
                c.page_key_week = s->grx.week;
                c.page_key_eph = k;
            }
            memcpy(r.page_next, c.page_cached, sizeof(r.page_next));
        }
        // 30 s refresh, src/galileo-sdr.cpp:545-562
        const int igrx = (int)(s->grx.sec * 10.0 + 0.5);
        if ((int)fmodf((float)igrx, 300) == 0) {
            for (int sv = 0; sv < kMaxSat; ++sv) {
                const int k = match_ephemeris(s->grx, s->nav.sv[sv]);
                // Ephemeris gap: no record of this satellite is within an hour of its TOC any more, but the satellite
                // still occupies a channel (allocateChannel skips satellites without a match, so it neither frees nor
                // re-checks it, src/channel.cpp:38-44).  The reference stores the -1 and then reads
                // eph_vector[sv][-1] (src/galileo-sdr.cpp:458,555-558): undefined behaviour, no parity is definable.
                // Policy here: the channel keeps its last valid record until a later refresh finds
                // a match again or the satellite sets; gal_scen_cfg_t.strict_eph turns the gap into an error instead.
                if (k < 0 && s->allocated[sv] >= 0 && s->current_eph[sv] >= 0 && !s->cfg.strict_eph) {
                    s->eph_gaps++;
                    if (!s->gap_warned) {
                        fprintf(stderr, "WARNING: PRN %d has no ephemeris within an hour of %d:%.1f; its channel keeps the "
                                        "last valid record (the reference indexes out of bounds here; --strict aborts "
                                        "instead)\n", sv + 1, s->grx.week, s->grx.sec);
                        s->gap_warned = true;
                    }
                    continue;
                }
                s->current_eph[sv] = k;
            }
            allocate_channels(s, s->grx, xyz);
        }
        s->grx.sec = s->grx.sec + kEpochDt;
        s->iumd++;
        produced++;
    }
    return produced;
}

int gal_scen_inav_page(gal_scen_t *s, int32_t svid, int32_t eph_index, int32_t week, double sec,
                       uint32_t page_words[GAL_PAGE_WORDS])
{
    if (!s || svid < 1 || svid > kMaxSat) return scen_fail(GAL_E_INVAL, "gal_scen_inav_page: bad argument");
    const std::vector<Ephemeris> &list = s->nav.sv[svid - 1];
    if (eph_index < 0 || eph_index >= (int)list.size()) return scen_fail(GAL_E_INVAL, "no such ephemeris record");
    GalTime g;
    g.week = week;
    g.sec = sec;
    int sym[kSymPerPage];
    inav_page_symbols(g, list[eph_index], s->nav.iono, sym);
    pack_symbols(sym, page_words);
    return GAL_OK;
}

int gal_scen_inav_raw(gal_scen_t *s, int32_t svid, int32_t eph_index, int32_t week, double sec, uint8_t bits[240])
{
    if (!s || !bits || svid < 1 || svid > kMaxSat) return scen_fail(GAL_E_INVAL, "gal_scen_inav_raw: bad argument");
    const std::vector<Ephemeris> &list = s->nav.sv[svid - 1];
    if (eph_index < 0 || eph_index >= (int)list.size()) return scen_fail(GAL_E_INVAL, "no such ephemeris record");
    GalTime g;
    g.week = week;
    g.sec = sec;
    int even_half[120], odd_half[120];
    inav_page_bits(g, list[eph_index], s->nav.iono, even_half, odd_half);
    for (int i = 0; i < 120; ++i) {
        bits[i] = (uint8_t)even_half[i];
        bits[120 + i] = (uint8_t)odd_half[i];
    }
    return GAL_OK;
}

uint32_t gal_scen_crc24q(const uint8_t *bits, int32_t len)
{
    if (!bits || len < 1 || len > 4096) return 0;
    std::vector<int> b(bits, bits + len);
    return inav_crc24q(b.data(), len);
}

int32_t gal_scen_eph_count(const gal_scen_t *s, int32_t svid)
{
    if (!s || svid < 1 || svid > kMaxSat) return 0;
    return (int32_t)s->nav.sv[svid - 1].size();
}

int gal_scen_eph_info(const gal_scen_t *s, int32_t svid, int32_t eph_index, int32_t *iodnav, int32_t *toe_week,
                      double *toe_sec, double *toc_sec)
{
    if (!s || svid < 1 || svid > kMaxSat) return scen_fail(GAL_E_INVAL, "gal_scen_eph_info: bad argument");
    const std::vector<Ephemeris> &list = s->nav.sv[svid - 1];
    if (eph_index < 0 || eph_index >= (int)list.size()) return scen_fail(GAL_E_INVAL, "no such ephemeris record");
    const Ephemeris &e = list[eph_index];
    if (iodnav) *iodnav = e.iodnav;
    if (toe_week) *toe_week = e.toe.week;
    if (toe_sec) *toe_sec = e.toe.sec;
    if (toc_sec) *toc_sec = e.toc.sec;
    return GAL_OK;
}

int32_t gal_scen_eph_gaps(const gal_scen_t *s) { return s ? s->eph_gaps : 0; }
int32_t gal_scen_live_rejected(const gal_scen_t *s) { return s ? s->live_rejected : 0; }

int gal_scen_close(gal_scen_t *s)
{
    delete s;
    return GAL_OK;
}

}  // extern "C"
