/*
 * galscen.h -- C ABI of the host-side scenario front-end: RINEX navigation file + receiver position +
 * start time  ->  per-epoch channel parameters (gal_chan_epoch_t rows) for the synthesis engine.
 *
 * It stands where the reference computes everything its sample loop reads, i.e. the part of
 * galileo_task() around the loop: src/galileo-sdr.cpp:202-352 (setup), :438-479 (per-epoch range /
 * code phase), :545-564 (30 s re-allocation), with src/rinex.cpp, src/gnss-time.cpp, src/geodesy.cpp,
 * src/gal-sig.cpp:242-347, src/channel.cpp and src/inav-msg.cpp behind it.  Host C++, double precision,
 * same operation order as the reference so that the doubles -- and therefore the int16 IQ -- match.
 *
 * Usage:  gal_scen_open() -> repeat gal_scen_next() until it returns 0 -> gal_scen_close().
 */
#ifndef GALSCEN_H_
#define GALSCEN_H_

#include <stdint.h>

#include "galsynth.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gal_scen_cfg {
    const char *nav_file;     /* -e : RINEX 3 Galileo navigation file (src/main.cpp:220)                    */
    const char *motion_file;  /* -u : user motion, CSV `t,x,y,z` ECEF metres at 10 Hz; NULL = static (-l)   */
    double llh[3];            /* -l : lat [deg], lon [deg], height [m]; reference default 42.3601,-71.0589,2 */
    int32_t have_start;       /* -t given                                                                   */
    int32_t start[5];         /* Y, M, D, h, m                                                              */
    double start_sec;         /* s (floored like src/main.cpp:269)                                          */
    double duration_s;        /* -d : seconds; epochs produced = (int)(d*10+0.5) - 1 (src/galileo-sdr.cpp:438) */
    int32_t iono_enable;      /* 0 with -I (src/main.cpp:300); 1 = the obliquity model the reference build runs */
    int32_t n_slots;          /* channel slots per row; reference MAX_CHAN = 16                              */
    int32_t verbose;          /* print the reference's allocation lines to stderr (src/channel.cpp:101)      */
    int32_t time_overwrite;   /* -T (src/main.cpp:237-257, src/gnss-time.cpp:105-137).  The CLI's plain -T and the Python mirror's
                                 time_overwrite=True mean 1 (the reference's bytes for the reference's command line); 2 is opt-in
                                 (--shift-toe / "shift").
                                 1 = what the reference, built with its
                                 own flags, does: the range check of -t is skipped and the UTC reference time (wnt, tot) is
                                 overwritten; NO ephemeris record is shifted (its loop bound `neph`, src/galileo-sdr.cpp:85,
                                 is never assigned: zero), so a start outside the file's span yields an empty sky.
                                 2 = what the option sets out to do: TOC / TOE of every record shifted by the start (floored
                                 to 2 h) minus the file's first TOC, so that the file becomes valid at the requested start */
    int32_t udp_port;         /* > 0: listen on this UDP port for run-time position updates, 3 doubles lat [deg],
                                 lon [deg], height [m] per datagram -- the reference's locations_thread on port 7533
                                 (include/socket.h:165-180), read once per epoch (src/galileo-sdr.cpp:443-448)      */
    int32_t strict_eph;       /* ephemeris-gap policy.  At a 30 s refresh the reference stores epoch_matcher's -1 for a
                                 satellite that still holds a channel and then reads eph_vector[sv][-1]
                                 (src/galileo-sdr.cpp:458,555-558; src/rinex.cpp:27-44): undefined behaviour.  0 (default):
                                 the channel keeps its last valid record until a later refresh matches again or frees it,
                                 one warning on stderr, gal_scen_eph_gaps() counts; 1: gal_scen_next fails with GAL_E_STATE */
    int32_t udp_loopback;     /* 1: bind the position listener to 127.0.0.1 only (the reference binds INADDR_ANY)      */
} gal_scen_cfg_t;

#define GAL_SCEN_UDP_PORT 7533 /* the reference's port (include/socket.h:169) */

typedef struct gal_scen gal_scen_t;

/* Returns GAL_OK or a negative gal_status_t; text via gal_scen_last_error(). */
int gal_scen_open(const gal_scen_cfg_t *cfg, gal_scen_t **out);
const char *gal_scen_last_error(void);
/* Total epochs the scenario will produce. */
int32_t gal_scen_total_epochs(const gal_scen_t *s);
/* Scenario start as Galileo week / seconds (after start-time selection, src/gnss-time.cpp:101-165). */
int gal_scen_start_time(const gal_scen_t *s, int32_t *week, double *sec);
/* Produce up to max_epochs further rows of n_slots records into `rows`; returns the number produced
 * (0 at the end) or a negative gal_status_t. */
int32_t gal_scen_next(gal_scen_t *s, int32_t max_epochs, gal_chan_epoch_t *rows);
int gal_scen_close(gal_scen_t *s);
/* (satellite, refresh) pairs so far at which a channel kept a stale ephemeris record (strict_eph == 0), and position
 * datagrams dropped because a coordinate was not finite or out of range. */
int32_t gal_scen_eph_gaps(const gal_scen_t *s);
int32_t gal_scen_live_rejected(const gal_scen_t *s);

/* I/NAV page generator on its own (reference src/inav-msg.cpp:28-54): 500 symbols of the page that
 * starts at Galileo time (week, sec) for the ephemeris record `eph_index` of `svid` in the opened file. */
int gal_scen_inav_page(gal_scen_t *s, int32_t svid, int32_t eph_index, int32_t week, double sec,
                       uint32_t page_words[GAL_PAGE_WORDS]);


/* The same page BEFORE channel coding (src/inav-msg.cpp:42-44 hands these two halves to generateFrame): bits[0..119]
 * = even half (114 bits + 6 tail zeros), bits[120..239] = odd half, one bit per byte.  This is the layout of the
 * recorded broadcast pages the reference holds under tv/<date>/<svid>.csv (tests/test_inav_kat.py). */
int gal_scen_inav_raw(gal_scen_t *s, int32_t svid, int32_t eph_index, int32_t week, double sec, uint8_t bits[240]);
/* CRC-24Q as the page generator computes it (src/inav-msg.cpp:141-167) over `len` bits, one bit per byte. */
uint32_t gal_scen_crc24q(const uint8_t *bits, int32_t len);
/* Ephemeris records of `svid` in the opened file, in file order (the order epoch_matcher, src/rinex.cpp:4-44, walks). */
int32_t gal_scen_eph_count(const gal_scen_t *s, int32_t svid);
int gal_scen_eph_info(const gal_scen_t *s, int32_t svid, int32_t eph_index, int32_t *iodnav, int32_t *toe_week,
                      double *toe_sec, double *toc_sec);

#ifdef __cplusplus
}
#endif
#endif /* GALSCEN_H_ */
