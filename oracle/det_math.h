/* oracle/det_math.h -- TEST INFRASTRUCTURE (part of the CPU oracle; never linked into the product).
 *
 * Deterministic float trig used by the oracle's eigen33 restatement (see hfpf_oracle.cpp).
 *
 * Why this exists: the reference reaches atan2f/cosf/sinf through pcl::eigen33 -> computeRoots
 * (call site OccupancyGrid.hpp:289; PCL is a third-party dependency that is NOT under
 * /root/reference and is version-unpinned: CMakeLists.txt:213 `find_package(PCL 1.7 REQUIRED)`).
 * libm results differ by ulps between glibc versions and from any GPU math library, and one ulp in
 * theta can flip a discrete voxel-registration decision downstream.  The oracle therefore DEFINES
 * these three leaves: evaluate in IEEE f64 with a fixed operation order (no FMA contraction; build
 * with -ffp-contract=off), then round once to f32.  The result equals the correctly rounded f32
 * value except when the f64 value lies within ~1e-16 relative of an f32 rounding boundary; the test
 * tests/test_oracle_leaves.py measures the agreement with this container's glibc.
 *
 * parity unpinned: the reference ships no test or golden vector for these leaves.
 */
#ifndef HFPF_ORACLE_DET_MATH_H
#define HFPF_ORACLE_DET_MATH_H

#include <math.h>

#define ODM_PI      3.14159265358979323846264338327950288
#define ODM_PI_2    1.57079632679489661923132169163975144
#define ODM_PI_4    0.78539816339744830961566084581987572
#define ODM_TAN_PI8 0.41421356237309504880168872420969808

/* atan(t) for t in [0,1], f64.  Range reduction at tan(pi/8): atan(t) = pi/4 + atan((t-1)/(t+1)).
 * Core: odd Taylor series in u, |u| <= 0.41421357, terms through u^43 (|tail| < 1e-18). */
static inline double odm_atan_unit(double t)
{
    double base = 0.0;
    double u = t;
    if (t > ODM_TAN_PI8) {
        u = (t - 1.0) / (t + 1.0);
        base = ODM_PI_4;
    }
    const double z = u * u;
    double p = 1.0 / 43.0;
    p = -1.0 / 41.0 + z * p;
    p = 1.0 / 39.0 + z * p;
    p = -1.0 / 37.0 + z * p;
    p = 1.0 / 35.0 + z * p;
    p = -1.0 / 33.0 + z * p;
    p = 1.0 / 31.0 + z * p;
    p = -1.0 / 29.0 + z * p;
    p = 1.0 / 27.0 + z * p;
    p = -1.0 / 25.0 + z * p;
    p = 1.0 / 23.0 + z * p;
    p = -1.0 / 21.0 + z * p;
    p = 1.0 / 19.0 + z * p;
    p = -1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = -1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = -1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = -1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    /* atan(u) = u - u*z*p */
    const double a = u - (u * z) * p;
    return base + a;
}

/* atan2f(y, x): all quadrants; NaN in -> NaN out; (0,0) -> 0. */
static inline float odm_atan2f(float yf, float xf)
{
    const double y = (double)yf, x = (double)xf;
    if (!(y == y) || !(x == x)) return (float)(y + x);
    const double ay = y < 0.0 ? -y : y;
    const double ax = x < 0.0 ? -x : x;
    if (ax == 0.0 && ay == 0.0) return 0.0f;
    double r;
    if (ay <= ax) r = odm_atan_unit(ay / ax);
    else          r = ODM_PI_2 - odm_atan_unit(ax / ay);
    if (x < 0.0) r = ODM_PI - r;
    if (y < 0.0) r = -r;
    return (float)r;
}

/* cos/sin Taylor cores for |x| <= pi/2 in f64 (terms through x^28 / x^29: |tail| < 1e-22). */
static inline double odm_cos_core(double x)
{
    const double z = x * x;
    double p = 1.0 / 304888344611713860501504000000.0;       /* 1/28! */
    p = -1.0 / 403291461126605635584000000.0 + z * p;        /* 1/26! */
    p = 1.0 / 620448401733239439360000.0 + z * p;            /* 1/24! */
    p = -1.0 / 1124000727777607680000.0 + z * p;             /* 1/22! */
    p = 1.0 / 2432902008176640000.0 + z * p;                 /* 1/20! */
    p = -1.0 / 6402373705728000.0 + z * p;                   /* 1/18! */
    p = 1.0 / 20922789888000.0 + z * p;                      /* 1/16! */
    p = -1.0 / 87178291200.0 + z * p;                        /* 1/14! */
    p = 1.0 / 479001600.0 + z * p;                           /* 1/12! */
    p = -1.0 / 3628800.0 + z * p;                            /* 1/10! */
    p = 1.0 / 40320.0 + z * p;                               /* 1/8!  */
    p = -1.0 / 720.0 + z * p;                                /* 1/6!  */
    p = 1.0 / 24.0 + z * p;                                  /* 1/4!  */
    p = -1.0 / 2.0 + z * p;                                  /* 1/2!  */
    return 1.0 + z * p;
}

static inline double odm_sin_core(double x)
{
    const double z = x * x;
    double p = 1.0 / 8841761993739701954543616000000.0;      /* 1/29! */
    p = -1.0 / 10888869450418352160768000000.0 + z * p;      /* 1/27! */
    p = 1.0 / 15511210043330985984000000.0 + z * p;          /* 1/25! */
    p = -1.0 / 25852016738884976640000.0 + z * p;            /* 1/23! */
    p = 1.0 / 51090942171709440000.0 + z * p;                /* 1/21! */
    p = -1.0 / 121645100408832000.0 + z * p;                 /* 1/19! */
    p = 1.0 / 355687428096000.0 + z * p;                     /* 1/17! */
    p = -1.0 / 1307674368000.0 + z * p;                      /* 1/15! */
    p = 1.0 / 6227020800.0 + z * p;                          /* 1/13! */
    p = -1.0 / 39916800.0 + z * p;                           /* 1/11! */
    p = 1.0 / 362880.0 + z * p;                              /* 1/9!  */
    p = -1.0 / 5040.0 + z * p;                               /* 1/7!  */
    p = 1.0 / 120.0 + z * p;                                 /* 1/5!  */
    p = -1.0 / 6.0 + z * p;                                  /* 1/3!  */
    return x + (x * z) * p;
}

/* cosf/sinf on [-pi, pi] (eigen33 only ever passes theta in [0, pi/3]); outside -> NaN. */
static inline float odm_cosf(float xf)
{
    double x = (double)xf;
    if (!(x == x)) return xf;
    if (x < 0.0) x = -x;
    if (x > ODM_PI) return (float)NAN;
    if (x > ODM_PI_2) return (float)(-odm_cos_core(ODM_PI - x));
    return (float)odm_cos_core(x);
}

static inline float odm_sinf(float xf)
{
    double x = (double)xf;
    if (!(x == x)) return xf;
    double s = 1.0;
    if (x < 0.0) { x = -x; s = -1.0; }
    if (x > ODM_PI) return (float)NAN;
    if (x > ODM_PI_2) x = ODM_PI - x;
    return (float)(s * odm_sin_core(x));
}

#endif
