// scen_internal.h -- types shared by the host scenario front-end (navdata.cpp, orbit.cpp, inav.cpp,
// scenario.cpp).  Host C++ only; compiled with -ffp-contract=off -fno-builtin-{sin,cos} so that every
// double is produced by the same sequence of IEEE operations and libm calls as the reference build.
#ifndef GAL_SCEN_INTERNAL_H_
#define GAL_SCEN_INTERNAL_H_

#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/galscen.h"

namespace galscen {

// ---- constants (reference include/constants.h; values must match digit for digit)
constexpr double kR2D = 57.2957795131;               // :178
constexpr double kPi = 3.141592653589793;            // :103 (PI)
constexpr double kOmegaEarth = 7.2921151467e-5;      // :102, :164
constexpr double kSqrtGM = 19964981.8432173887;      // :100
constexpr double kC = 2.99792458e8;                  // :55
constexpr double kLambdaL1 = 0.190293672798365;      // :56
constexpr double kLambdaE1 = 0.1902936727983649;     // :122
constexpr double kCodeFreqE1 = 1.023e6;              // :126
constexpr double kCarrToCodeE1 = 0.0006493506493506494;  // :128
constexpr double kWgs84A = 6378137.0;                // :52
constexpr double kWgs84E = 0.0818191908426;          // :53
constexpr double kSecWeek = 604800.0, kSecHalfWeek = 302400.0, kSecDay = 86400.0, kSecHour = 3600.0,
                 kSecMinute = 60.0;
constexpr int kMaxSat = 36;                          // :108
constexpr double kEpochDt = 0.10000002314200000;     // src/galileo-sdr.cpp:347
constexpr int kSymPerPage = 500;

struct GalTime {  // galtime_t
    int week = 0;
    double sec = 0.0;
};

struct CalTime {  // datetime_t
    int y = 0, m = 0, d = 0, hh = 0, mm = 0;
    double sec = 0.0;
};

// One broadcast ephemeris record (ephem_t subset that the path reads)
struct Ephemeris {
    int valid = 0;
    int svid = 0;
    GalTime toc, toe;
    int iodnav = 0;
    double deltan = 0, cuc = 0, cus = 0, cic = 0, cis = 0, crc = 0, crs = 0;
    double ecc = 0, sqrta = 0, m0 = 0, omg0 = 0, inc0 = 0, aop = 0, omgdot = 0, idot = 0;
    double af0 = 0, af1 = 0, af2 = 0;
    double bgd_e5a = 0, bgd_e5b = 0;
    int svhealth = 0;
    // derived (src/rinex.cpp:225-229)
    double n = 0, sq1e2 = 0, A = 0, omgkdot = 0;
};

struct IonoUtc {  // ionoutc_t subset
    int enable = 1;
    int nequick = 0;  // ionoutc_t.vflg: never written by the reference (UB); its build reads 0
    double ai0 = 0, ai1 = 0, ai2 = 0, ai3 = 0;
    double A0 = 0, A1 = 0;
    int dtls = 0, tot = 0, wnt = 0, dtlsf = 0, dn = 0, wnlsf = 0;
};

struct NavData {
    std::vector<Ephemeris> sv[kMaxSat];
    IonoUtc iono;
    int count = 0;
};

struct Range {  // range_t
    GalTime g;
    double range = 0;  // pseudorange incl. clock and iono terms
    double d = 0;      // geometric distance
    double azel[2] = {0, 0};
    double iono_delay = 0;
};

// navdata.cpp
int load_rinex3(const char *path, NavData *out, std::string *err);
int match_ephemeris(const GalTime &t, const std::vector<Ephemeris> &list);
void cal_to_gal(const CalTime &t, GalTime *g);
void gal_to_cal(const GalTime &g, CalTime *t);
double gal_diff(const GalTime &a, const GalTime &b);

// orbit.cpp
void llh_to_ecef(const double llh[3], double xyz[3]);
void ecef_to_llh(const double xyz[3], double llh[3]);
void sat_state(const Ephemeris &eph, const GalTime &g, double pos[3], double vel[3], double clk[2]);
int sat_visible(const Ephemeris &eph, const GalTime &g, const double xyz[3], double elv_mask_deg, double azel[2]);
void compute_range(Range *rho, const Ephemeris &eph, const IonoUtc &iono, const GalTime &g, const double xyz[3]);

// inav.cpp
void inav_page_bits(const GalTime &g, const Ephemeris &eph, const IonoUtc &iono, int even_half[120], int odd_half[120]);
unsigned int inav_crc24q(const int *bits, int len);
void inav_page_symbols(const GalTime &g, const Ephemeris &eph, const IonoUtc &iono, int symbols[kSymPerPage]);
void pack_symbols(const int symbols[kSymPerPage], uint32_t words[GAL_PAGE_WORDS]);

}  // namespace galscen
#endif
