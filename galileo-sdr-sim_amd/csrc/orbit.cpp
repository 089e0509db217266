// orbit.cpp -- satellite state from broadcast ephemeris, geodesy and pseudorange of the host scenario
// front-end.  Operation order follows the reference expression by expression (the doubles feed the NCO
// parameters, so they must be the same):
//   satpos ................. src/geodesy.cpp:161-273
//   xyz2llh / llh2xyz ...... src/geodesy.cpp:7-91
//   ltcmat/ecef2neu/neu2azel src/geodesy.cpp:97-153
//   checkSatVisibility ..... src/geodesy.cpp:316-344
//   computeRange ........... src/gal-sig.cpp:242-301
//   ionosphericDelay ....... src/iono.cpp:9-20,32-41 (obliquity branch; see DESIGN.md on ionoutc_t.vflg)
#include <cmath>

#include "scen_internal.h"

namespace galscen {

namespace {

double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

void sub3(double y[3], const double a[3], const double b[3])
{
    y[0] = a[0] - b[0];
    y[1] = a[1] - b[1];
    y[2] = a[2] - b[2];
}

// rows: north, east, up
void local_frame(const double llh[3], double t[3][3])
{
    const double slat = sin(llh[0]);
    const double clat = cos(llh[0]);
    const double slon = sin(llh[1]);
    const double clon = cos(llh[1]);
    t[0][0] = -slat * clon;
    t[0][1] = -slat * slon;
    t[0][2] = clat;
    t[1][0] = -slon;
    t[1][1] = clon;
    t[1][2] = 0.0;
    t[2][0] = clat * clon;
    t[2][1] = clat * slon;
    t[2][2] = slat;
}

void to_neu(const double v[3], double t[3][3], double neu[3])
{
    neu[0] = t[0][0] * v[0] + t[0][1] * v[1] + t[0][2] * v[2];
    neu[1] = t[1][0] * v[0] + t[1][1] * v[1] + t[1][2] * v[2];
    neu[2] = t[2][0] * v[0] + t[2][1] * v[1] + t[2][2] * v[2];
}

void to_azel(double azel[2], const double neu[3])
{
    azel[0] = atan2(neu[1], neu[0]);
    if (azel[0] < 0.0) azel[0] += (2.0 * kPi);
    const double ne = sqrt(neu[0] * neu[0] + neu[1] * neu[1]);
    azel[1] = atan2(neu[2], ne);
}

// src/iono.cpp:9-20
double obliquity_delay(const double azel[2])
{
    const double E = azel[1] / kPi;
    const double F = 1.0 + 16.0 * pow((0.53 - E), 3.0);
    return F * 5.0e-9 * kC;
}

}  // namespace

void ecef_to_llh(const double xyz[3], double llh[3])
{
    const double a = kWgs84A, e = kWgs84E;
    const double eps = 1.0e-3;
    const double e2 = e * e;
    if (norm3(xyz) < eps) {
        llh[0] = 0.0;
        llh[1] = 0.0;
        llh[2] = -a;
        return;
    }
    const double x = xyz[0], y = xyz[1], z = xyz[2];
    const double rho2 = x * x + y * y;
    double dz = e2 * z;
    double zdz, nh, slat, n, dz_new;
    while (true) {
        zdz = z + dz;
        nh = sqrt(rho2 + zdz * zdz);
        slat = zdz / nh;
        n = a / sqrt(1.0 - e2 * slat * slat);
        dz_new = n * e2 * slat;
        if (fabs(dz - dz_new) < eps) break;
        dz = dz_new;
    }
    llh[0] = atan2(zdz, sqrt(rho2));
    llh[1] = atan2(y, x);
    llh[2] = nh - n;
}

void llh_to_ecef(const double llh[3], double xyz[3])
{
    const double a = kWgs84A, e = kWgs84E;
    const double e2 = e * e;
    const double clat = cos(llh[0]);
    const double slat = sin(llh[0]);
    const double clon = cos(llh[1]);
    const double slon = sin(llh[1]);
    const double d = e * slat;
    const double n = a / sqrt(1.0 - d * d);
    const double nph = n + llh[2];
    const double tmp = nph * clat;
    xyz[0] = tmp * clon;
    xyz[1] = tmp * slon;
    xyz[2] = ((1.0 - e2) * n + llh[2]) * slat;
}

void sat_state(const Ephemeris &eph, const GalTime &g, double pos[3], double vel[3], double clk[2])
{
    double tk = g.sec - eph.toe.sec;
    if (tk > kSecHalfWeek) tk -= kSecWeek;
    else if (tk < -kSecHalfWeek) tk += kSecWeek;

    const double mk = eph.m0 + eph.n * tk;
    double ek = mk;
    double ekold = ek + 1.0;
    double one_m_ecos = 0;
    int iter = 0;
    while ((fabs(ek - ekold) > 1.0E-14) && iter < 500) {
        iter++;
        ekold = ek;
        one_m_ecos = 1.0 - eph.ecc * cos(ekold);
        ek = ek + (mk - ekold + eph.ecc * sin(ekold)) / one_m_ecos;
    }
    const double sek = sin(ek);
    const double cek = cos(ek);
    const double ekdot = eph.n / one_m_ecos;
    const double relativistic = -4.442807633E-10 * eph.ecc * eph.sqrta * sek;

    const double pk = atan2(eph.sq1e2 * sek, cek - eph.ecc) + eph.aop;
    const double pkdot = eph.sq1e2 * ekdot / one_m_ecos;
    const double s2pk = sin(2.0 * pk);
    const double c2pk = cos(2.0 * pk);

    const double uk = pk + eph.cus * s2pk + eph.cuc * c2pk;
    const double suk = sin(uk);
    const double cuk = cos(uk);
    const double ukdot = pkdot * (1.0 + 2.0 * (eph.cus * c2pk - eph.cuc * s2pk));

    const double rk = eph.A * one_m_ecos + eph.crc * c2pk + eph.crs * s2pk;
    const double rkdot = eph.A * eph.ecc * sek * ekdot + 2.0 * pkdot * (eph.crs * c2pk - eph.crc * s2pk);

    const double ik = eph.inc0 + eph.idot * tk + eph.cic * c2pk + eph.cis * s2pk;
    const double sik = sin(ik);
    const double cik = cos(ik);
    const double ikdot = eph.idot + 2.0 * pkdot * (eph.cis * c2pk - eph.cic * s2pk);

    const double xpk = rk * cuk;
    const double ypk = rk * suk;
    const double xpkdot = rkdot * cuk - ypk * ukdot;
    const double ypkdot = rkdot * suk + xpk * ukdot;

    const double ok = eph.omg0 + tk * eph.omgkdot - kOmegaEarth * eph.toe.sec;
    const double sok = sin(ok);
    const double cok = cos(ok);

    pos[0] = xpk * cok - ypk * cik * sok;
    pos[1] = xpk * sok + ypk * cik * cok;
    pos[2] = ypk * sik;

    const double tmp = ypkdot * cik - ypk * sik * ikdot;
    vel[0] = -eph.omgkdot * pos[1] + xpkdot * cok - tmp * sok;
    vel[1] = eph.omgkdot * pos[0] + xpkdot * sok + tmp * cok;
    vel[2] = ypk * cik * ikdot + ypkdot * sik;

    // clock, referenced to toc
    tk = g.sec - eph.toc.sec;
    if (tk > kSecHalfWeek) tk -= kSecWeek;
    else if (tk < -kSecHalfWeek) tk += kSecWeek;
    clk[0] = eph.af0 + tk * (eph.af1 + tk * eph.af2) + relativistic - eph.bgd_e5b;
    clk[1] = eph.af1 + 2.0 * tk * eph.af2;
}

int sat_visible(const Ephemeris &eph, const GalTime &g, const double xyz[3], double elv_mask_deg, double azel[2])
{
    if (eph.valid != 1) return -1;
    double llh[3], neu[3], pos[3], vel[3], clk[2], los[3], t[3][3];
    ecef_to_llh(xyz, llh);
    local_frame(llh, t);
    sat_state(eph, g, pos, vel, clk);
    sub3(los, pos, xyz);
    to_neu(los, t, neu);
    to_azel(azel, neu);
    return (azel[1] * kR2D > elv_mask_deg) ? 1 : 0;
}

void compute_range(Range *rho, const Ephemeris &eph, const IonoUtc &iono, const GalTime &g, const double xyz[3])
{
    double pos[3] = {0.0}, vel[3] = {0.0}, clk[2] = {0.0}, los[3] = {0.0};
    double llh[3] = {0.0}, neu[3] = {0.0}, t[3][3];

    sat_state(eph, g, pos, vel, clk);

    // light time, then back-propagate the satellite and rotate the frame (Sagnac)
    sub3(los, pos, xyz);
    const double tau = norm3(los) / kC;
    pos[0] -= vel[0] * tau;
    pos[1] -= vel[1] * tau;
    pos[2] -= vel[2] * tau;
    const double xrot = pos[0] + pos[1] * kOmegaEarth * tau;
    const double yrot = pos[1] - pos[0] * kOmegaEarth * tau;
    pos[0] = xrot;
    pos[1] = yrot;

    sub3(los, pos, xyz);
    const double range = norm3(los);
    rho->d = range;
    rho->range = range - kC * clk[0];

    ecef_to_llh(xyz, llh);
    local_frame(llh, t);
    to_neu(los, t, neu);
    to_azel(rho->azel, neu);

    // src/iono.cpp:32-41.  The NeQuick-G branch (ionoutc_t.vflg != 0) is a floating-point no-op in the
    // reference (a delay in seconds of order 1e-24 added to a range in metres) and is not modelled.
    double delay = 0.0;
    if (iono.enable) delay = iono.nequick ? 0.0 : obliquity_delay(rho->azel);
    rho->iono_delay = delay;
    rho->range += rho->iono_delay;
    rho->g = g;
}

}  // namespace galscen
