// csrc/det_math.hpp -- deterministic f32 atan2/cos/sin for the engine's plane-fit kernel.
//
// The reference reaches atan2f/cosf/sinf through pcl::eigen33 (call site OccupancyGrid.hpp:289).
// libm/ocml results differ in the last ulp between platforms, and one ulp in theta can move a voxel
// registration across a cell boundary, so the engine evaluates them in IEEE f64 with a fixed
// operation order (this translation unit is built with -ffp-contract=off) and rounds once to f32.
// Same algorithm on host and device; written for this engine (the CPU oracle carries its own copy
// of the definition and the two are compared bit-for-bit by tests/test_gpu_leaves.py).
#pragma once
#include <hip/hip_runtime.h>

namespace hfpf {

#define HFPF_HD __host__ __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kPi2 = 1.57079632679489661923132169163975144;
constexpr double kPi4 = 0.78539816339744830961566084581987572;
constexpr double kTanPi8 = 0.41421356237309504880168872420969808;

// atan on [0,1]: fold at tan(pi/8), then the odd Maclaurin series through u^43.
HFPF_HD double atan_unit(double t)
{
    double base = 0.0, u = t;
    if (t > kTanPi8) {
        u = (t - 1.0) / (t + 1.0);
        base = kPi4;
    }
    const double z = u * u;
    double p = 1.0 / 43.0;
#pragma unroll
    for (int m = 41; m >= 3; m -= 2) {
        const double c = (((m - 3) / 2) & 1) ? (-1.0 / (double)m) : (1.0 / (double)m);
        p = c + z * p;
    }
    // p now = 1/3 - z/5 + ... ; atan(u) = u - u*z*p
    return base + (u - (u * z) * p);
}

HFPF_HD float det_atan2f(float yf, float xf)
{
    const double y = (double)yf, x = (double)xf;
    if (!(y == y) || !(x == x)) return (float)(y + x);
    const double ay = y < 0.0 ? -y : y;
    const double ax = x < 0.0 ? -x : x;
    if (ax == 0.0 && ay == 0.0) return 0.0f;
    double r;
    if (ay <= ax) r = atan_unit(ay / ax);
    else r = kPi2 - atan_unit(ax / ay);
    if (x < 0.0) r = kPi - r;
    if (y < 0.0) r = -r;
    return (float)r;
}

// Maclaurin cores on |x| <= pi/2 (through x^28 and x^29).
HFPF_HD double cos_core(double x)
{
    const double z = x * x;
    const double inv[14] = {-1.0 / 2.0,
                            1.0 / 24.0,
                            -1.0 / 720.0,
                            1.0 / 40320.0,
                            -1.0 / 3628800.0,
                            1.0 / 479001600.0,
                            -1.0 / 87178291200.0,
                            1.0 / 20922789888000.0,
                            -1.0 / 6402373705728000.0,
                            1.0 / 2432902008176640000.0,
                            -1.0 / 1124000727777607680000.0,
                            1.0 / 620448401733239439360000.0,
                            -1.0 / 403291461126605635584000000.0,
                            1.0 / 304888344611713860501504000000.0};
    double p = inv[13];
#pragma unroll
    for (int i = 12; i >= 0; --i) p = inv[i] + z * p;
    return 1.0 + z * p;
}

HFPF_HD double sin_core(double x)
{
    const double z = x * x;
    const double inv[14] = {-1.0 / 6.0,
                            1.0 / 120.0,
                            -1.0 / 5040.0,
                            1.0 / 362880.0,
                            -1.0 / 39916800.0,
                            1.0 / 6227020800.0,
                            -1.0 / 1307674368000.0,
                            1.0 / 355687428096000.0,
                            -1.0 / 121645100408832000.0,
                            1.0 / 51090942171709440000.0,
                            -1.0 / 25852016738884976640000.0,
                            1.0 / 15511210043330985984000000.0,
                            -1.0 / 10888869450418352160768000000.0,
                            1.0 / 8841761993739701954543616000000.0};
    double p = inv[13];
#pragma unroll
    for (int i = 12; i >= 0; --i) p = inv[i] + z * p;
    return x + (x * z) * p;
}

HFPF_HD float det_cosf(float xf)
{
    double x = (double)xf;
    if (!(x == x)) return xf;
    if (x < 0.0) x = -x;
    if (x > kPi) return __builtin_nanf("");
    if (x > kPi2) return (float)(-cos_core(kPi - x));
    return (float)cos_core(x);
}

HFPF_HD float det_sinf(float xf)
{
    double x = (double)xf;
    if (!(x == x)) return xf;
    double s = 1.0;
    if (x < 0.0) {
        x = -x;
        s = -1.0;
    }
    if (x > kPi) return __builtin_nanf("");
    if (x > kPi2) x = kPi - x;
    return (float)(s * sin_core(x));
}

}  // namespace hfpf
