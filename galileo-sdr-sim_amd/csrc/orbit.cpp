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

namespace {

// seconds from `ref` to `t`, folded into half a week either side
double folded_interval(double t, double ref)
{
    double dt = t - ref;
    if (dt > kSecHalfWeek) dt -= kSecWeek;
    else if (dt < -kSecHalfWeek) dt += kSecWeek;
    return dt;
}

// Eccentric anomaly by the fixed-point form of Newton's iteration, stopped at 1e-14 or 500 rounds.  `denom` is
// 1 - e cos(E) evaluated at the iterate BEFORE the last update -- the value the range-rate terms are built on.
struct Anomaly {
    double E;
    double denom;
};

Anomaly eccentric_anomaly(double mean, double ecc)
{
    Anomaly a;
    a.E = mean;
    a.denom = 0;
    double previous = a.E + 1.0;
    for (int round = 0; fabs(a.E - previous) > 1.0E-14 && round < 500; ++round) {
        previous = a.E;
        a.denom = 1.0 - ecc * cos(previous);
        a.E = a.E + (mean - previous + ecc * sin(previous)) / a.denom;
    }
    return a;
}

// (base + first) + second: the harmonic corrections are added one after the other, in this order
inline double plus2(double base, double first, double second) { return base + first + second; }

// position and velocity in the orbital plane
struct InPlane {
    double x, y, xdot, ydot;
};

// rotate by inclination and (Earth-fixed) longitude of the ascending node; node_rate turns the frame
void to_earth_fixed(const InPlane &q, double incl, double incl_rate, double node, double node_rate, double pos[3],
                    double vel[3])
{
    const double si = sin(incl), ci = cos(incl);
    const double sn = sin(node), cn = cos(node);
    pos[0] = q.x * cn - q.y * ci * sn;
    pos[1] = q.x * sn + q.y * ci * cn;
    pos[2] = q.y * si;
    const double cross = q.ydot * ci - q.y * si * incl_rate;
    vel[0] = -node_rate * pos[1] + q.xdot * cn - cross * sn;
    vel[1] = node_rate * pos[0] + q.xdot * sn + cross * cn;
    vel[2] = q.y * ci * incl_rate + q.ydot * si;
}

}  // namespace

// Broadcast-ephemeris satellite state (OS SIS ICD 5.1.1 user algorithm) with the rates needed for the range rate,
// evaluated in the operation order of the reference's satpos() (src/geodesy.cpp:161-273): the doubles feed the NCO
// parameters and through them the int16 IQ, so the expression trees are not free.
void sat_state(const Ephemeris &eph, const GalTime &g, double pos[3], double vel[3], double clk[2])
{
    const double t_orbit = folded_interval(g.sec, eph.toe.sec);
    const double mean = eph.m0 + eph.n * t_orbit;
    const Anomaly an = eccentric_anomaly(mean, eph.ecc);
    const double sinE = sin(an.E), cosE = cos(an.E);
    const double E_rate = eph.n / an.denom;
    const double clock_rel = -4.442807633E-10 * eph.ecc * eph.sqrta * sinE;

    // argument of latitude before corrections, and the second harmonics all corrections are expressed in
    const double phi = atan2(eph.sq1e2 * sinE, cosE - eph.ecc) + eph.aop;
    const double phi_rate = eph.sq1e2 * E_rate / an.denom;
    const double s2 = sin(2.0 * phi), c2 = cos(2.0 * phi);

    const double lat_arg = plus2(phi, eph.cus * s2, eph.cuc * c2);
    const double s_lat = sin(lat_arg), c_lat = cos(lat_arg);
    const double lat_rate = phi_rate * (1.0 + 2.0 * (eph.cus * c2 - eph.cuc * s2));

    const double radius = plus2(eph.A * an.denom, eph.crc * c2, eph.crs * s2);
    const double radius_rate = eph.A * eph.ecc * sinE * E_rate + 2.0 * phi_rate * (eph.crs * c2 - eph.crc * s2);

    const double incl = plus2(eph.inc0 + eph.idot * t_orbit, eph.cic * c2, eph.cis * s2);
    const double incl_rate = eph.idot + 2.0 * phi_rate * (eph.cis * c2 - eph.cic * s2);

    InPlane q;
    q.x = radius * c_lat;
    q.y = radius * s_lat;
    q.xdot = radius_rate * c_lat - q.y * lat_rate;
    q.ydot = radius_rate * s_lat + q.x * lat_rate;

    const double node = eph.omg0 + t_orbit * eph.omgkdot - kOmegaEarth * eph.toe.sec;
    to_earth_fixed(q, incl, incl_rate, node, eph.omgkdot, pos, vel);

    // satellite clock, referenced to toc
    const double t_clock = folded_interval(g.sec, eph.toc.sec);
    clk[0] = eph.af0 + t_clock * (eph.af1 + t_clock * eph.af2) + clock_rel - eph.bgd_e5b;
    clk[1] = eph.af1 + 2.0 * t_clock * eph.af2;
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
