// navdata.cpp -- RINEX 3 Galileo navigation reader, ephemeris selection and time conversions of the
// host scenario front-end.  Behaviour follows the reference so that the same records are kept and the
// same doubles come out:
//   readRinexV3 / readContentsData ... src/rinex.cpp:72-249   (fixed-column fields, D->E exponents, only
//                                      records whose data-source word is 517 are kept, :218)
//   epoch_matcher .................... src/rinex.cpp:4-44      (first record with toc within [-1 h, +1 h))
//   date2gal / gal2date / subGalTime . src/gnss-time.cpp:7-88
#include <cmath>
#include <cstdio>
#include <cstring>

#include "scen_internal.h"

namespace galscen {

namespace {

constexpr int kLineMax = 120;  // MAX_CHAR, include/constants.h:105

void exponent_d_to_e(char *s)
{
    for (; *s; ++s)
        if (*s == 'D') *s = 'E';
}

// A 19-column numeric field starting at column `col`; blank (second character is a space) reads as 0.
double field(const char *line, int len, int col)
{
    double v = 0.0;
    if (len > col + 1 && line[col + 1] != ' ') sscanf(line + col, "%lf", &v);
    return v;
}

// URA index -> not used by the signal path; kept out.

}  // namespace

void cal_to_gal(const CalTime &t, GalTime *g)
{
    static const int doy[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    const int ye = t.y - 1980;
    int lpdays = ye / 4 + 1;
    if ((ye % 4) == 0 && t.m <= 2) lpdays--;
    const int de = ye * 365 + doy[t.m - 1] + t.d + lpdays - 6;
    g->week = de / 7;
    g->sec = (double)(de % 7) * kSecDay + t.hh * kSecHour + t.mm * kSecMinute + t.sec;
}

void gal_to_cal(const GalTime &g, CalTime *t)
{
    const int c = (int)(7 * g.week + floor(g.sec / 86400.0) + 2444245.0) + 1537;
    const int d = (int)((c - 122.1) / 365.25);
    const int e = 365 * d + d / 4;
    const int f = (int)((c - e) / 30.6001);
    t->d = c - e - (int)(30.6001 * f);
    t->m = f - 1 - 12 * (f / 14);
    t->y = d - 4715 - ((7 + t->m) / 10);
    t->hh = ((int)(g.sec / 3600.0)) % 24;
    t->mm = ((int)(g.sec / 60.0)) % 60;
    t->sec = g.sec - 60.0 * floor(g.sec / 60.0);
}

double gal_diff(const GalTime &a, const GalTime &b)
{
    double dt = a.sec - b.sec;
    dt += (double)(a.week - b.week) * kSecWeek;
    return dt;
}

int match_ephemeris(const GalTime &t, const std::vector<Ephemeris> &list)
{
    for (size_t i = 0; i < list.size(); ++i) {
        if (list[i].valid != 1) continue;
        const double dt = gal_diff(t, list[i].toc);
        if (dt >= -kSecHour && dt < kSecHour) return (int)i;
    }
    return -1;
}

int load_rinex3(const char *path, NavData *out, std::string *err)
{
    FILE *fp = fopen(path, "r");
    if (!fp) {
        *err = std::string("cannot open navigation file ") + path;
        return -1;
    }
    char line[kLineMax];
    IonoUtc &io = out->iono;

    // ---- header
    while (fgets(line, kLineMax, fp)) {
        if (strncmp(line + 60, "END OF HEADER", 13) == 0) break;
        if (strncmp(line + 60, "IONOSPHERIC CORR", 16) == 0) {
            exponent_d_to_e(line);
            sscanf(line + 4, "%lf %lf %lf %lf", &io.ai0, &io.ai1, &io.ai2, &io.ai3);
        }
        if (strncmp(line + 60, "TIME SYSTEM CORR", 16) == 0 && strncmp(line, "GAUT", 4) == 0) {
            int t_ref = 0, w_ref = 0;
            exponent_d_to_e(line);
            const char keep = line[22];
            line[22] = 0;
            sscanf(line + 4, "%lf", &io.A0);
            line[22] = keep;
            sscanf(line + 22, "%lf %d %d", &io.A1, &t_ref, &w_ref);
            io.tot = (unsigned char)(t_ref >> 12);
            io.wnt = (short)w_ref >> 4;
            io.wnlsf = (short)w_ref;
            io.dtls = 18;
            io.dtlsf = 18;
            io.dn = 7;
        }
    }

    // ---- records: 1 epoch line + 7 orbit lines
    while (fgets(line, kLineMax, fp)) {
        if (line[0] != 'E') continue;
        double v[39];
        memset(v, 0, sizeof(v));
        CalTime toc_cal;
        int sec_int = 0, svid = 0;
        exponent_d_to_e(line);
        int len = (int)strlen(line);
        sscanf(line + 4, "%d %d %d %d %d %d", &toc_cal.y, &toc_cal.m, &toc_cal.d, &toc_cal.hh, &toc_cal.mm, &sec_int);
        toc_cal.sec = (double)sec_int;
        if (line[1] != ' ') sscanf(line + 1, "%2d", &svid);
        v[0] = field(line, len, 23);
        v[1] = field(line, len, 42);
        v[2] = field(line, len, 61);
        for (int k = 0; k < 7; ++k) {
            if (!fgets(line, kLineMax, fp)) break;
            exponent_d_to_e(line);
            len = (int)strlen(line);
            double *d = &v[k * 4 + 3];
            d[0] = field(line, len, 4);
            d[1] = field(line, len, 23);
            d[2] = field(line, len, 42);
            d[3] = field(line, len, 61);
        }
        const unsigned short source = (unsigned short)v[20];
        if (source != 517) continue;  // I/NAV E1-B, E5b-I clock; everything else is dropped
        if (svid < 1 || svid > kMaxSat) continue;

        Ephemeris e;
        e.svid = svid;
        cal_to_gal(toc_cal, &e.toc);
        e.af0 = v[0];
        e.af1 = v[1];
        e.af2 = v[2];
        e.iodnav = (unsigned char)v[3];
        e.crs = v[4];
        e.deltan = v[5];
        e.m0 = v[6];
        e.cuc = v[7];
        e.ecc = v[8];
        e.cus = v[9];
        e.sqrta = v[10];
        e.toe.sec = (int)(v[11] + 0.5);
        e.cic = v[12];
        e.omg0 = v[13];
        e.cis = v[14];
        e.inc0 = v[15];
        e.crc = v[16];
        e.aop = v[17];
        e.omgdot = v[18];
        e.idot = v[19];
        e.toe.week = (int)v[21];
        e.svhealth = (unsigned short)v[24];
        e.bgd_e5a = v[25];
        e.bgd_e5b = (source & 0x2) ? v[25] : v[26];
        e.A = e.sqrta * e.sqrta;
        e.n = kSqrtGM / (e.sqrta * e.A) + e.deltan;
        e.sq1e2 = sqrt(1.0 - e.ecc * e.ecc);
        e.omgkdot = e.omgdot - kOmegaEarth;
        e.valid = 1;
        out->sv[svid - 1].push_back(e);
        out->count++;
    }
    fclose(fp);
    return out->count;
}

}  // namespace galscen
