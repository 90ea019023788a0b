// csrc/geometry.hpp -- per-point / per-voxel arithmetic of the fusion path, host+device.
//
// Every function states the reference expression it must agree with bit-for-bit
// (grid.hpp = .../include/utilities/OccupancyGrid.hpp, node.cpp = .../src/pointcloud_fusion_and_filter.cpp).
// The translation unit is compiled with -ffp-contract=off: the reference build has no -march flag
// (CMakeLists.txt:5), i.e. x86-64 SSE2 without FMA, so no product-sum may be fused here either.
// f32/f64 divide and sqrt are the correctly rounded forms (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt; never build this file with -ffast-math).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include <stdint.h>

#include "det_math.hpp"

namespace hfpf {

struct F3 {
    float x, y, z;
};

// Grid constants, passed by value to every kernel (lives in SGPRs / kernarg).
struct GridParams {
    double min[3];  // xmin_,ymin_,zmin_   grid.hpp:104
    double max[3];  // xmax_,ymax_,zmax_
    double res;     // xres_ = (double)(float)resolution  grid.hpp:614-619 (all three axes equal, node.cpp:161)
    double inv_res; // 1.0 / res, for the fast path of voxel_axis() only
    int32_t dim[3];   // xdim_,ydim_,zdim_ truncated  grid.hpp:623-625
    int32_t bdim[3];  // bricks per axis covering the (dim+1) storage extent  grid.hpp:626
    double zclip_min, zclip_max;  // node.cpp:92-93
    double cyl_r;                 // grid.hpp:36
    float ball_r;                 // (float)kBballRadius, grid.hpp:35,42
    int32_t K;                    // node.cpp:311
    int32_t gate;                 // grid.hpp:352
    int32_t cov_shifted;          // 0: PCL <= 1.10 single-pass moments; 1: PCL >= 1.11 moments of (p - first point)
    // fixed-point scales (powers of two) of the order-free statistic sums, see stats.hpp
    float fs_scale, fss_scale, fd_scale, fdd_scale;
    float d2_max;  // largest f32 u with (double)sqrtf(u) < cyl_r: membership as one compare on the squared distance
    // The reference's double compares of a float against the bbox / z-clip constants (grid.hpp:639-645, node.cpp:251-255), as float
    // compares against the neighbouring float of each constant (found by the host at create): identical decisions for every float
    // incl. NaN, six f64 compares and three conversions a point less, and 12 scalar registers less in k_integrate.
    //   bb_hi[a] = smallest float >= max[a]:  (double)x >= max[a]  <=>  x >= bb_hi[a]        (same for zc_hi and "<")
    //   bb_lo[a] = largest float  <= min[a]:  (double)x <= min[a]  <=>  x <= bb_lo[a]        (same for zc_lo and ">")
    float bb_lo[3], bb_hi[3], zc_lo, zc_hi;
    // cell key = x << key_sx | y << key_sy | z with just enough bits per axis for 0..dim: ascending keys = the reference's
    // lexicographic (x,y,z) scan order, and the radix sorts run over key_bits bits (30 at 999^3) instead of 64
    uint32_t key_sy, key_sx, key_bits;
};

HFPF_HD float sum3(float a, float b, float c) { return a + (b + c); }  // Eigen fixed-size-3 redux: c0 + (c1 + c2)
HFPF_HD float dot3(F3 a, F3 b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
HFPF_HD F3 sub3(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
HFPF_HD F3 add3(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
HFPF_HD F3 mul3(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
HFPF_HD F3 div3(F3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

// `int v = floor(double)` as the reference's x86 build performs it (cvttsd2si: NaN / overflow -> INT_MIN).
HFPF_HD int32_t to_int_x86(double v)
{
    if (!(v == v) || v >= 2147483648.0 || v < -2147483648.0) return INT_MIN;
    return (int32_t)v;
}

// node.cpp:289 pcl::transformPointCloud<PointXYZRGB,double>: f64 products summed left to right, one rounding to f32.
HFPF_HD F3 transform_point(const double* T, float x, float y, float z)
{
    const double dx = (double)x, dy = (double)y, dz = (double)z;
    F3 q;
    q.x = (float)(((T[0] * dx + T[1] * dy) + T[2] * dz) + T[3]);
    q.y = (float)(((T[4] * dx + T[5] * dy) + T[6] * dz) + T[7]);
    q.z = (float)(((T[8] * dx + T[9] * dy) + T[10] * dz) + T[11]);
    return q;
}

// node.cpp:251-255 z-clip in the camera frame: strict on both sides, NaN rejected.
HFPF_HD bool zclip_pass(const GridParams& g, float z) { return (z < g.zc_hi) && (z > g.zc_lo); }

// grid.hpp:639-645 validPoints: float promoted to double in each compare; strict interior.
HFPF_HD bool valid_point(const GridParams& g, F3 p)
{
    return !(p.x >= g.bb_hi[0] || p.y >= g.bb_hi[1] || p.z >= g.bb_hi[2] || p.x <= g.bb_lo[0] || p.y <= g.bb_lo[1] || p.z <= g.bb_lo[2]);
}

// grid.hpp:630-637 getVoxelCoords(Vector3f): floor((double(p) - min) / res) truncated to int.
// One axis of grid.hpp:630-637, i.e. to_int_x86(floor(a / res)) with the IEEE quotient, without paying for an f64 division
// per axis and point: x = a * (1/res) differs from RN(a / res) by < 4e-16 * |x| (two roundings in x, one in the quotient),
// i.e. by < 6e-9 for |x| < 2^24, so whenever x is at least 1e-6 away from both neighbouring integers the two have the same
// floor.  Anything closer than that (or huge) takes the exact division; NaN gives INT_MIN on either path.
HFPF_HD int32_t voxel_axis(double a, double res, double inv_res)
{
    const double x = a * inv_res;
    double f = floor(x);
    if (fabs(x) >= 16777216.0 || x - f < 1e-6 || (f + 1.0) - x < 1e-6) f = floor(a / res);
    return to_int_x86(f);
}

HFPF_HD void voxel_coords(const GridParams& g, F3 p, int32_t& ix, int32_t& iy, int32_t& iz)
{
    ix = voxel_axis((double)p.x - g.min[0], g.res, g.inv_res);
    iy = voxel_axis((double)p.y - g.min[1], g.res, g.inv_res);
    iz = voxel_axis((double)p.z - g.min[2], g.res, g.inv_res);
}

// grid.hpp:647-650 validCoord: cells with index == dim exist in storage but are never scanned.
HFPF_HD bool valid_coord(const GridParams& g, int32_t x, int32_t y, int32_t z)
{
    return x >= 0 && y >= 0 && z >= 0 && x < g.dim[0] && y < g.dim[1] && z < g.dim[2];
}

// grid.hpp:131-135 getVoxelCenter: f64 `min + res*i + res/2.0`, narrowed to f32.
HFPF_HD F3 voxel_center(const GridParams& g, int32_t x, int32_t y, int32_t z)
{
    const double h = g.res / 2.0;
    F3 c;
    c.x = (float)((g.min[0] + g.res * (double)x) + h);
    c.y = (float)((g.min[1] + g.res * (double)y) + h);
    c.z = (float)((g.min[2] + g.res * (double)z) + h);
    return c;
}

// grid.hpp:40-49 projectPointToVector + grid.hpp:261-262 (norm widened to double, compared with kCylinderRadius), split in
// two: what depends only on the voxel whose line it is (segment end a = centre - r*n, ab = a - b, |ab|^2: computed once per
// normal record and kept in its dependant entries) and what depends on the point.  The f32 operations and their order are the
// reference's; only the place where the per-voxel half is evaluated differs.
HFPF_HD void line_of(const GridParams& g, F3 centre, F3 n, F3& a, F3& ab, float& dd)
{
    const F3 d_xyz = mul3(g.ball_r, n);
    a = sub3(centre, d_xyz);
    const F3 b = add3(centre, d_xyz);
    ab = sub3(a, b);
    dd = dot3(ab, ab);
}

// s = the projection parameter (proj = a - s*ab, so s = 0.5 at the cell centre); d2 = ||pt - proj||^2 in f32.
HFPF_HD void line_project(F3 pt, F3 a, F3 ab, float dd, float& s, F3& proj, float& d2)
{
    const F3 ap = sub3(a, pt);
    s = dot3(ap, ab) / dd;
    proj = sub3(a, mul3(s, ab));
    const F3 df = sub3(pt, proj);
    d2 = dot3(df, df);
}

// The reference's form (used by the leaf probes): returns membership; proj and dist are outputs.
HFPF_HD bool cylinder_member(const GridParams& g, F3 pt, F3 centre, F3 n, F3& proj, double& dist)
{
    F3 a, ab;
    float dd, s, d2;
    line_of(g, centre, n, a, ab, dd);
    line_project(pt, a, ab, dd, s, proj, d2);
    dist = (double)sqrtf(d2);  // Eigen norm(): correctly rounded f32 sqrt, widened (grid.hpp:261)
    return dist < g.cyl_r;
}

// dist = ||p - proj|| as the statistics see it (grid.hpp:261: Eigen's norm(), a correctly rounded f32 sqrt).
// HFPF_EXACT_SQRT=1: the IEEE square root (hardware estimate + the compiler's fix-up, ~13 more instructions per pair);
// 0: the 1-ulp hardware estimate -- membership never depends on it (decided on d2), only the mean_dist / sd_dist sums do.
#ifndef HFPF_EXACT_SQRT
#define HFPF_EXACT_SQRT 0
#endif
__device__ __forceinline__ float dist_sqrt(float d2)
{
#if HFPF_EXACT_SQRT
    return sqrtf(d2);
#else
    return __builtin_amdgcn_sqrtf(d2);
#endif
}

// The kernels' form.  Membership is decided on the squared distance: a correctly rounded sqrt is monotonic, so
// (double)sqrtf(d2) < cyl_r  <=>  d2 <= d2_max with d2_max the largest f32 that passes (found by the host at create and
// checked against the form above by the leaf tests) -- identical decisions without the IEEE sqrt refinement and the f64 compare.
// distf, which only feeds the mean_dist / sd_dist sums, comes from the 1-ulp hardware sqrt.
__device__ __forceinline__ bool line_member(const GridParams& g, F3 pt, F3 a, F3 ab, float dd, float& s, float& distf)
{
    F3 proj;
    float d2;
    line_project(pt, a, ab, dd, s, proj, d2);
    distf = dist_sqrt(d2);
    return d2 <= g.d2_max;
}

// The same with the divisor's share of the division hoisted out of a loop over points.  The compiler expands the IEEE f32
// division n / dd into v_div_scale x2, v_rcp, two refinement steps of the reciprocal, a quotient with two residual corrections
// (the last one v_div_fmas) and v_div_fixup; when neither operand needs v_div_scale's rescaling (both magnitudes within
// 2^+-40 here, far inside the hardware's limits) that is exactly: r = refine(rcp(dd)); q = n*r; q = fma(fma(-dd,q,n), r, q) twice.
// r depends on the line alone, so a loop over the points of a cell keeps it in a register and the quotient costs 5 operations
// instead of 11, bit for bit the same (checked against the plain form on random and boundary inputs by the leaf tests; operands
// outside the window take the plain division).
struct LineDiv {
    float dd, r;
    bool in_window;
};
__device__ __forceinline__ LineDiv line_div_of(float dd)
{
    LineDiv d;
    d.dd = dd;
    const float r0 = __builtin_amdgcn_rcpf(dd);
    const float e = __builtin_fmaf(-dd, r0, 1.0f);
    d.r = __builtin_fmaf(e, r0, r0);
    d.in_window = dd >= 0x1p-40f && dd <= 0x1p40f;
    return d;
}
__device__ __forceinline__ float line_div(float n, const LineDiv& d)
{
    const float an = __builtin_fabsf(n);
    if (d.in_window && an >= 0x1p-40f && an <= 0x1p40f) {
        float q = n * d.r;
        float rem = __builtin_fmaf(-d.dd, q, n);
        q = __builtin_fmaf(rem, d.r, q);
        rem = __builtin_fmaf(-d.dd, q, n);
        return __builtin_fmaf(rem, d.r, q);
    }
    return n / d.dd;
}
__device__ __forceinline__ bool line_member_hoisted(const GridParams& g, F3 pt, F3 a, F3 ab, const LineDiv& dv, float& s, float& distf)
{
    const F3 ap = sub3(a, pt);
    s = line_div(dot3(ap, ab), dv);
    const F3 proj = sub3(a, mul3(s, ab));
    const F3 df = sub3(pt, proj);
    const float d2 = dot3(df, df);
    distf = dist_sqrt(d2);
    return d2 <= g.d2_max;
}

// ---- plane fit: pcl::computeMeanAndCovarianceMatrix + pcl::eigen33 (call sites grid.hpp:302,289) ----

HFPF_HD void swapf(float& a, float& b)
{
    const float t = a;
    a = b;
    b = t;
}

// pcl::computeRoots2 (the literal 4.0 is a double in PCL: the discriminant is formed in f64).
HFPF_HD void compute_roots2(float b, float c, float* roots)
{
    roots[0] = 0.0f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if ((double)d < 0.0) d = 0.0f;
    const float sd = sqrtf(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}

// pcl::computeRoots for a symmetric 3x3 (m00,m01,m02,m11,m12,m22).
HFPF_HD void compute_roots(float m00, float m01, float m02, float m11, float m12, float m22, float* roots)
{
    const float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
    const float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
    const float c2 = m00 + m11 + m22;
    if (fabsf(c0) < FLT_EPSILON) {
        compute_roots2(c2, c1, roots);
        return;
    }
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = 1.7320508075688772f;  // sqrtf(3.0f) = 0x3FDDB3D7
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = sqrtf(-a_over_3);
    const float theta = det_atan2f(sqrtf(-q), half_b) * s_inv3;
    const float cos_theta = det_cosf(theta);
    const float sin_theta = det_sinf(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    if (roots[0] >= roots[1]) swapf(roots[0], roots[1]);
    if (roots[1] >= roots[2]) {
        swapf(roots[1], roots[2]);
        if (roots[0] >= roots[1]) swapf(roots[0], roots[1]);
    }
    if (roots[0] <= 0.0f) compute_roots2(c2, c1, roots);
}

HFPF_HD F3 cross3(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// pcl::eigen33(mat, eigenvalue, eigenvector): unit eigenvector of the smallest eigenvalue of the
// symmetric matrix (c00,c01,c02,c11,c12,c22).
HFPF_HD F3 eigen33_smallest(float c00, float c01, float c02, float c11, float c12, float c22)
{
    float scale = fmaxf(fmaxf(fmaxf(fabsf(c00), fabsf(c01)), fmaxf(fabsf(c02), fabsf(c11))), fmaxf(fabsf(c12), fabsf(c22)));
    if (scale <= FLT_MIN) scale = 1.0f;
    float s00 = c00 / scale, s01 = c01 / scale, s02 = c02 / scale, s11 = c11 / scale, s12 = c12 / scale, s22 = c22 / scale;
    float ev[3];
    compute_roots(s00, s01, s02, s11, s12, s22, ev);
    s00 -= ev[0];
    s11 -= ev[0];
    s22 -= ev[0];
    const F3 r0 = {s00, s01, s02}, r1 = {s01, s11, s12}, r2 = {s02, s12, s22};
    const F3 v1 = cross3(r0, r1), v2 = cross3(r0, r2), v3 = cross3(r1, r2);
    const float l1 = dot3(v1, v1), l2 = dot3(v2, v2), l3 = dot3(v3, v3);
    if (l1 >= l2 && l1 >= l3) return div3(v1, sqrtf(l1));
    if (l2 >= l1 && l2 >= l3) return div3(v2, sqrtf(l2));
    return div3(v3, sqrtf(l3));
}

// Running single-pass f32 moments in neighbour-table order (pcl::computeMeanAndCovarianceMatrix, dense branch).
struct Moments {
    float a[9];
    F3 k;  // shift (PCL >= 1.11: the first point; otherwise 0)
    bool have_k;
    HFPF_HD void clear()
    {
#pragma unroll
        for (int i = 0; i < 9; i++) a[i] = 0.0f;
        k = F3{0.f, 0.f, 0.f};
        have_k = false;
    }
    HFPF_HD void add(F3 q, bool shifted = false)
    {
        if (shifted && !have_k) {
            k = q;
            have_k = true;
        }
        const F3 p = {q.x - k.x, q.y - k.y, q.z - k.z};
        a[0] += p.x * p.x;
        a[1] += p.x * p.y;
        a[2] += p.x * p.z;
        a[3] += p.y * p.y;
        a[4] += p.y * p.z;
        a[5] += p.z * p.z;
        a[6] += p.x;
        a[7] += p.y;
        a[8] += p.z;
    }
    HFPF_HD F3 normal(int n) const
    {
        const float fn = (float)n;
        float m[9];
#pragma unroll
        for (int i = 0; i < 9; i++) m[i] = a[i] / fn;
        const float c00 = m[0] - m[6] * m[6];
        const float c01 = m[1] - m[6] * m[7];
        const float c02 = m[2] - m[6] * m[8];
        const float c11 = m[3] - m[7] * m[7];
        const float c12 = m[4] - m[7] * m[8];
        const float c22 = m[5] - m[8] * m[8];
        return eigen33_smallest(c00, c01, c02, c11, c12, c22);
    }
};

// grid.hpp:393-396: flip the normal toward the viewpoint latched at first occupancy.
HFPF_HD F3 orient_normal(F3 normal, F3 vp, F3 centre)
{
    F3 dir = sub3(vp, centre);
    const float z = dot3(dir, dir);
    if (z > 0.0f) dir = div3(dir, sqrtf(z));  // Eigen 3.3 normalized()
    if (dot3(dir, normal) < 0.0f) normal = {normal.x * -1.0f, normal.y * -1.0f, normal.z * -1.0f};
    return normal;
}

// grid.hpp:405: centre + (float)(i*xres_) * normal
HFPF_HD F3 line_step(const GridParams& g, F3 centre, F3 normal, int i)
{
    const float step = (float)((double)i * g.res);
    return add3(centre, mul3(step, normal));
}

}  // namespace hfpf
