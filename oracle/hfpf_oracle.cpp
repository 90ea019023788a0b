// oracle/hfpf_oracle.cpp -- TEST INFRASTRUCTURE.  CPU restatement of the reference's
// capture -> integrate -> clean -> extract path.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this; the product (libhfpf.so) never links or calls it.
//
// PARITY UNPINNED: the reference (REXJJ/high-fidelity-pointcloud-fusion) ships no tests, fixtures
// or golden vectors, and cannot be built in this image (needs Eigen, PCL, ROS, Boost and the
// un-vendored pointcloud_ros_utilities package; none present, no network).  This file is therefore
// a restatement pinned only by reading the source.  Third-party leaves are restated from the
// published algorithms of PCL 1.8 / Eigen 3.3 (the reference needs Eigen >= 3.3: the expression
// `i*xres_*data->normal`, OccupancyGrid.hpp:405, mixes double and float scalars, which 3.2 rejects):
//   pcl::transformPointCloud<PointT,double>      (call site node.cpp:289)
//   pcl::computeMeanAndCovarianceMatrix (float)  (call site OccupancyGrid.hpp:302)
//   pcl::eigen33 / computeRoots / computeRoots2  (call site OccupancyGrid.hpp:289)
//   Eigen fixed-size-3 reductions: dot/squaredNorm evaluate c0 + (c1 + c2)
//   atan2f/cosf/sinf: defined by det_math.h (see that header)
//
// File:line citations use these aliases:
//   grid.hpp = pointcloud_fusion/pointcloud_fusion/include/utilities/OccupancyGrid.hpp
//   node.cpp = pointcloud_fusion/pointcloud_fusion/src/pointcloud_fusion_and_filter.cpp
//
// Decisions where the reference has undefined behaviour (documented in DESIGN.md):
//   * VoxelInfo::mean_dist, ::normal, ::viewpoint are uninitialised in the reference ctor
//     (grid.hpp:74-81); defined as 0 here.
//   * A point whose transformed coordinates are NaN passes validPoints (all compares false,
//     grid.hpp:644) and then indexes voxels_ with INT_MIN in the reference (crash); dropped here.
//   * Clean iterates `unprocessed_data_` in libstdc++ bucket order (grid.hpp:315); the canonical
//     order here is ascending (x,y,z).  order_mode=1 keeps a real std::unordered_set with the
//     reference's own key values to reproduce the bucket order (only valid while y < 2048,
//     because the reference computes y<<20 in int: grid.hpp:154).
//   * The reference is single-session (clearVoxels leaves stale keys/blocks, grid.hpp:167-183);
//     clear() here is a full reset.
//
// Build: see oracle/Makefile (-ffp-contract=off is required).

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <chrono>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "det_math.h"

// -DORACLE_USE_LIBM swaps the three deterministic trig leaves for this machine's libm (sensitivity studies only:
// tests/test_oracle_build_variants.py measures how many output rows depend on the last ulp of atan2f/cosf/sinf).
#ifdef ORACLE_USE_LIBM
#define ORACLE_ATAN2F(y, x) atan2f((y), (x))
#define ORACLE_COSF(x) cosf((x))
#define ORACLE_SINF(x) sinf((x))
#else
#define ORACLE_ATAN2F(y, x) odm_atan2f((y), (x))
#define ORACLE_COSF(x) odm_cosf((x))
#define ORACLE_SINF(x) odm_sinf((x))
#endif

namespace {

struct V3 {
    float x, y, z;
};

// Eigen fixed-size-3 reduction order: c0 + (c1 + c2).
#ifdef ORACLE_SUM3_LEFT  // sensitivity study only: plain left-to-right order instead of Eigen's redux tree
inline float sum3(float a, float b, float c) { return (a + b) + c; }
#else
inline float sum3(float a, float b, float c) { return a + (b + c); }
#endif
inline float dot3(const V3& a, const V3& b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 sub3(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add3(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mul3(float s, const V3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 div3(const V3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float norm3(const V3& a) { return sqrtf(dot3(a, a)); }
// Eigen 3.3 MatrixBase::normalized(): z = squaredNorm(); z > 0 ? v / sqrt(z) : v.
inline V3 normalized3(const V3& a)
{
    float z = dot3(a, a);
    if (z > 0.0f) return div3(a, sqrtf(z));
    return a;
}

// grid.hpp:40-49  projectPointToVector(pt, norm_pt, n); kBballRadius (double 0.015, grid.hpp:35)
// is converted to the vector's scalar type (float) by Eigen's scalar promotion.
inline V3 project_point_to_vector(const V3& pt, const V3& norm_pt, const V3& n, float ball_r)
{
    V3 d_xyz = mul3(ball_r, n);
    V3 a = sub3(norm_pt, d_xyz);
    V3 b = add3(norm_pt, d_xyz);
    V3 ap = sub3(a, pt);
    V3 ab = sub3(a, b);
    float s = dot3(ap, ab) / dot3(ab, ab);
    V3 p = sub3(a, mul3(s, ab));
    return p;
}

// x86 cvttsd2si semantics for double -> int (what `int xv = floor(...)` does in the reference
// build, grid.hpp:633): NaN and out-of-range give INT_MIN.
inline int to_int_x86(double v)
{
    if (!(v == v) || v >= 2147483648.0 || v < -2147483648.0) return INT_MIN;
    return (int)v;
}

// ---- PCL leaves (restated; see header) -------------------------------------------------------

// pcl::computeRoots2 (float)
inline void compute_roots2(float b, float c, float roots[3])
{
    roots[0] = 0.0f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if ((double)d < 0.0) d = 0.0f;
    float sd = sqrtf(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}

// pcl::computeRoots (float, symmetric 3x3 m)
inline void compute_roots(const float m[3][3], float roots[3])
{
    float c0 = m[0][0] * m[1][1] * m[2][2] + 2.0f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] -
               m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] -
               m[1][2] * m[1][2];
    float c2 = m[0][0] + m[1][1] + m[2][2];

    if (fabsf(c0) < FLT_EPSILON) {
        compute_roots2(c2, c1, roots);
        return;
    }
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;

    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));

    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;

    float rho = sqrtf(-a_over_3);
    float theta = ORACLE_ATAN2F(sqrtf(-q), half_b) * s_inv3;
    float cos_theta = ORACLE_COSF(theta);
    float sin_theta = ORACLE_SINF(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);

    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    if (roots[1] >= roots[2]) {
        std::swap(roots[1], roots[2]);
        if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    }
    if (roots[0] <= 0.0f) compute_roots2(c2, c1, roots);
}

inline V3 cross3(const float a[3], const float b[3])
{
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}

// pcl::eigen33(mat, eigenvalue, eigenvector): eigenvector of the smallest eigenvalue.
inline V3 eigen33_smallest(const float mat[3][3])
{
    float scale = 0.0f;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) scale = std::max(scale, fabsf(mat[i][j]));
    if (scale <= FLT_MIN) scale = 1.0f;
    float s[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) s[i][j] = mat[i][j] / scale;
    float ev[3];
    compute_roots(s, ev);
    s[0][0] -= ev[0];
    s[1][1] -= ev[0];
    s[2][2] -= ev[0];
    V3 vec1 = cross3(s[0], s[1]);
    V3 vec2 = cross3(s[0], s[2]);
    V3 vec3 = cross3(s[1], s[2]);
    float len1 = dot3(vec1, vec1);
    float len2 = dot3(vec2, vec2);
    float len3 = dot3(vec3, vec3);
    if (len1 >= len2 && len1 >= len3) return div3(vec1, sqrtf(len1));
    if (len2 >= len1 && len2 >= len3) return div3(vec2, sqrtf(len2));
    return div3(vec3, sqrtf(len3));
}

// grid.hpp:295-309 getNormal(cloud, normal) -> computeMeanAndCovarianceMatrix (single-pass float
// moments, accumulated in cloud order) + solvePlaneParameters (grid.hpp:282-293) -> eigen33.
inline bool get_normal(const V3* pts, int n, V3& normal, bool shifted = false)
{
    if (n < 3) return false;
    float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef ORACLE_PCL_SHIFTED
    shifted = true;
#endif
    // PCL >= 1.11 accumulates the moments of (p - K), K = first point; PCL <= 1.10 (assumed for the reference) uses K = 0
    const V3 K = shifted ? pts[0] : V3{0.f, 0.f, 0.f};
    for (int i = 0; i < n; i++) {
        const float px = pts[i].x - K.x, py = pts[i].y - K.y, pz = pts[i].z - K.z;
        accu[0] += px * px;
        accu[1] += px * py;
        accu[2] += px * pz;
        accu[3] += py * py;
        accu[4] += py * pz;
        accu[5] += pz * pz;
        accu[6] += px;
        accu[7] += py;
        accu[8] += pz;
    }
    const float fn = (float)n;
    for (int j = 0; j < 9; j++) accu[j] /= fn;
    float c[3][3];
    c[0][0] = accu[0] - accu[6] * accu[6];
    c[0][1] = accu[1] - accu[6] * accu[7];
    c[0][2] = accu[2] - accu[6] * accu[8];
    c[1][1] = accu[3] - accu[7] * accu[7];
    c[1][2] = accu[4] - accu[7] * accu[8];
    c[2][2] = accu[5] - accu[8] * accu[8];
    c[1][0] = c[0][1];
    c[2][0] = c[0][2];
    c[2][1] = c[1][2];
    normal = eigen33_smallest(c);
    return true;
}

// ---- voxel records (grid.hpp:51-82) ------------------------------------------------------------

struct VoxelInfo {
    V3 centroid{0, 0, 0};
    V3 normal{0, 0, 0};  // uninitialised in the reference; defined 0
    V3 sd{0, 0, 0};
    float sd_dist = 0;
    float mean_dist = 0;    // uninitialised in the reference; defined 0
    V3 viewpoint{0, 0, 0};  // uninitialised in the reference; defined 0
    std::vector<V3> buffer;  // reference stores (pt, viewpoint) pairs; the viewpoint half only feeds dead code
    std::vector<uint64_t> dependants;
    bool normal_found = false;
    int count = 0;
    // EXTENSION (not in the reference, which drops colour at grid.hpp:196-197): with Config::fuse_color the
    // buffered points keep their packed rgb and every cylinder member adds its r, g, b to the voxel it updates.
    std::vector<uint32_t> buffer_rgb;
    uint64_t csum[3] = {0, 0, 0};
};

struct Voxel {
    bool occupied = false;
    VoxelInfo* data = nullptr;
};

struct Row {  // 64 bytes; mirrored by oracle.py
    int32_t ix, iy, iz;
    uint32_t count;
    float x, y, z;
    float nx, ny, nz;
    float sdx, sdy, sdz;
    float mean_dist, sd_dist;
    uint32_t rgb;
};

struct Config {  // mirrored by oracle.py
    float resolution;      // passed through float exactly like setResolution(float,...) grid.hpp:614
    double bbox[6];        // xmin,xmax,ymin,ymax,zmin,zmax (setDimensions, grid.hpp:604)
    int32_t k;             // neighbourhood half-width, setK (node.cpp:163) -> 2
    int32_t K;             // line half-length in steps, template arg (node.cpp:311) -> 3
    int32_t gate;          // total > gate (grid.hpp:352) -> 20
    double cylinder_radius;  // kCylinderRadius grid.hpp:36
    double ball_radius;      // kBballRadius grid.hpp:35
    double z_clip_min;       // kZmin node.cpp:92
    double z_clip_max;       // kZmax node.cpp:93
    int32_t order_mode;      // 0 canonical ascending (x,y,z); 1 libstdc++ unordered_set order
    int32_t reserve;         // buffer.reserve(n) on first touch (reference: 1000, grid.hpp:228); 0 = off
    int32_t pcl_shifted_cov; // 0 = PCL <= 1.10 computeMeanAndCovarianceMatrix (default), 1 = PCL >= 1.11 shifted form
    int32_t fuse_color;      // EXTENSION: 1 = per-voxel mean colour of the cylinder members (definition in extract())
    int32_t dense;           // 1 = the reference's storage: one 16-byte Voxel per cell of the (dim+1)^3 box (grid.hpp:108,626);
                             //     0 = hash map of touched cells (needed for the 10^10-cell configs).  Same results either way.
};

class Oracle {
public:
    explicit Oracle(const Config& c) : cfg(c)
    {
        // setResolution(float...) into double members, grid.hpp:614-619
        xres_ = yres_ = zres_ = (double)c.resolution;
        xmin_ = c.bbox[0];
        xmax_ = c.bbox[1];
        ymin_ = c.bbox[2];
        ymax_ = c.bbox[3];
        zmin_ = c.bbox[4];
        zmax_ = c.bbox[5];
        // construct(), grid.hpp:621-628: int truncation
        xdim_ = (int)((xmax_ - xmin_) / xres_);
        ydim_ = (int)((ymax_ - ymin_) / yres_);
        zdim_ = (int)((zmax_ - zmin_) / zres_);
        // setK, grid.hpp:138-149: x outermost, z innermost
        for (int i = -c.k; i <= c.k; i++)
            for (int j = -c.k; j <= c.k; j++)
                for (int kk = -c.k; kk <= c.k; kk++) {
                    dx.push_back(i);
                    dy.push_back(j);
                    dz.push_back(kk);
                }
        ball_r_f = (float)c.ball_radius;
        if (c.dense) {  // construct(), grid.hpp:626: (xdim+1)(ydim+1)(zdim+1) voxels of {bool occupied; void* data}
            dense_n_ = (size_t)(xdim_ + 1) * (size_t)(ydim_ + 1) * (size_t)(zdim_ + 1);
            dense_ = (Voxel*)calloc(dense_n_, sizeof(Voxel));  // all-zero = {false, nullptr}; pages are touched on first use
        }
    }
    ~Oracle()
    {
        clear();
        free(dense_);
    }
    Oracle(const Oracle&) = delete;
    Oracle& operator=(const Oracle&) = delete;

    // ---- voxel storage: dense array as the reference, or a hash map of touched cells ----
    Voxel* dense_ = nullptr;
    size_t dense_n_ = 0;
    std::vector<uint64_t> dense_live_;  // keys of cells whose Voxel::data was ever set (what the map would hold)
    size_t dense_index(int x, int y, int z) const { return ((size_t)x * (size_t)(ydim_ + 1) + (size_t)y) * (size_t)(zdim_ + 1) + (size_t)z; }
    Voxel* find_voxel(int x, int y, int z)  // nullptr when the cell was never touched (sparse mode only)
    {
        if (dense_) return &dense_[dense_index(x, y, z)];
        auto it = voxels_.find(own_key(x, y, z));
        return it == voxels_.end() ? nullptr : &it->second;
    }
    Voxel& get_voxel(int x, int y, int z) { return dense_ ? dense_[dense_index(x, y, z)] : voxels_[own_key(x, y, z)]; }
    void note_live(int x, int y, int z)  // call when a cell's data pointer goes from null to set
    {
        if (dense_) dense_live_.push_back(own_key(x, y, z));
    }
    template <class F>
    void for_each_voxel(F f) const  // f(key, const Voxel&) over every cell that has (or had) a record
    {
        if (dense_) {
            for (uint64_t k : dense_live_) {
                int x, y, z;
                own_coords(k, x, y, z);
                f(k, dense_[dense_index(x, y, z)]);
            }
        } else {
            for (auto& kv : voxels_) f(kv.first, kv.second);
        }
    }

    Config cfg;
    double xmin_, xmax_, ymin_, ymax_, zmin_, zmax_;
    double xres_, yres_, zres_;
    int xdim_, ydim_, zdim_;
    std::vector<int> dx, dy, dz;
    float ball_r_f;
    bool state_changed = false;
    uint64_t n_presented = 0, n_zclip_pass = 0, n_inserted = 0;

    std::unordered_map<uint64_t, Voxel> voxels_;  // sparse stand-in for the dense 3-level vector (grid.hpp:108)
    std::unordered_set<unsigned long long> unprocessed_;  // grid.hpp:129

    static uint64_t own_key(int x, int y, int z) { return ((uint64_t)x << 42) | ((uint64_t)y << 21) | (uint64_t)z; }
    static void own_coords(uint64_t k, int& x, int& y, int& z)
    {
        x = (int)(k >> 42);
        y = (int)((k >> 21) & 0x1FFFFF);
        z = (int)(k & 0x1FFFFF);
    }
    // grid.hpp:151-156 getHashId, with the int shift emulated as 32-bit wraparound
    static unsigned long long ref_key(int x, int y, int z)
    {
        unsigned long long hash = (unsigned long long)x;
        int ys = (int)((uint32_t)y << 20);
        hash = (hash << 40) ^ (unsigned long long)(long long)ys ^ (unsigned long long)(long long)z;
        return hash;
    }
    // grid.hpp:158-165
    static void ref_coords(unsigned long long id, int& x, int& y, int& z)
    {
        const unsigned long long mask = (1 << 20) - 1;
        x = (int)(id >> 40);
        y = (int)((id >> 20) & mask);
        z = (int)(id & mask);
    }
    unsigned long long set_key(int x, int y, int z) const { return cfg.order_mode == 1 ? ref_key(x, y, z) : own_key(x, y, z); }

    // grid.hpp:131-135
    V3 voxel_center(int x, int y, int z) const
    {
        return {(float)(xmin_ + xres_ * (x) + xres_ / 2.0), (float)(ymin_ + yres_ * (y) + yres_ / 2.0),
                (float)(zmin_ + zres_ * (z) + zres_ / 2.0)};
    }
    // grid.hpp:630-637
    void voxel_coords(const V3& p, int& xv, int& yv, int& zv) const
    {
        xv = to_int_x86(floor(((double)p.x - xmin_) / xres_));
        yv = to_int_x86(floor(((double)p.y - ymin_) / yres_));
        zv = to_int_x86(floor(((double)p.z - zmin_) / zres_));
    }
    // grid.hpp:639-645 (float promoted to double in each compare)
    bool valid_point(const V3& p) const
    {
        double x = p.x, y = p.y, z = p.z;
        return !(x >= xmax_ || y >= ymax_ || z >= zmax_ || x <= xmin_ || y <= ymin_ || z <= zmin_);
    }
    // grid.hpp:647-650
    bool valid_coord(int x, int y, int z) const { return (x >= 0 && y >= 0 && z >= 0 && x < xdim_ && y < ydim_ && z < zdim_); }

    Voxel lookup(int x, int y, int z) const
    {
        if (dense_) {
            if (x < 0 || y < 0 || z < 0 || x > xdim_ || y > ydim_ || z > zdim_) return Voxel();
            return dense_[dense_index(x, y, z)];
        }
        auto it = voxels_.find(own_key(x, y, z));
        if (it == voxels_.end()) return Voxel();
        return it->second;
    }

    // Welford update shared by grid.hpp:264-273 and grid.hpp:428-438
    static void welford(VoxelInfo* d, const V3& proj, double distance_to_normal)
    {
        d->count++;
        V3 old_mean = d->centroid;
        const float fc = (float)d->count;
        d->centroid = add3(d->centroid, div3(sub3(proj, d->centroid), fc));
        d->sd.x = d->sd.x + ((proj.x - d->centroid.x) * ((proj.x - old_mean.x)) - d->sd.x) / fc;
        d->sd.y = d->sd.y + ((proj.y - d->centroid.y) * ((proj.y - old_mean.y)) - d->sd.y) / fc;
        d->sd.z = d->sd.z + ((proj.z - d->centroid.z) * ((proj.z - old_mean.z)) - d->sd.z) / fc;
        float old_mean_dist = d->mean_dist;
        d->mean_dist = (float)((double)d->mean_dist + (distance_to_normal - (double)d->mean_dist) / (double)d->count);
        d->sd_dist = (float)((double)d->sd_dist + ((distance_to_normal - (double)d->mean_dist) *
                                                       (distance_to_normal - (double)old_mean_dist) -
                                                   (double)d->sd_dist) /
                                                      (double)d->count);
    }

    static void add_color(VoxelInfo* d, uint32_t rgb)  // packed 0x00RRGGBB; blue with shift 0 (not the reference's shift 1, node.cpp:174)
    {
        d->csum[0] += (rgb >> 16) & 255u;
        d->csum[1] += (rgb >> 8) & 255u;
        d->csum[2] += rgb & 255u;
    }

    // grid.hpp:185-280 addPoints<N>(cloud, viewpoint).  Points are already in the fusion frame.
    // rgb (optional, colour extension): the points' packed colours.
    void add_points(const V3* pts, size_t n, const V3& viewpoint, const uint32_t* rgb = nullptr)
    {
        const bool color = cfg.fuse_color != 0 && rgb != nullptr;
        state_changed = true;
        for (size_t p = 0; p < n; p++) {
            const V3 point = pts[p];
            int x, y, z;
            voxel_coords(point, x, y, z);
            if (!valid_point(point)) continue;
            if (x == INT_MIN || y == INT_MIN || z == INT_MIN) continue;  // NaN coordinate: reference would crash
            n_inserted++;
            const unsigned long long hash = set_key(x, y, z);
            Voxel& voxel = get_voxel(x, y, z);
            const V3 ptv = point;
            if (voxel.occupied) {
                VoxelInfo* data = voxel.data;
                if (!data->normal_found) {
                    data->buffer.push_back(ptv);
                    if (color) data->buffer_rgb.push_back(rgb[p]);
                } else {
                    if (unprocessed_.find(hash) != unprocessed_.end()) unprocessed_.erase(hash);
                }
            } else {
                voxel.occupied = true;
                unprocessed_.insert(hash);
                if (voxel.data == nullptr) {
                    VoxelInfo* data = new VoxelInfo();
                    if (cfg.reserve > 0) data->buffer.reserve((size_t)cfg.reserve);
                    data->viewpoint = viewpoint;
                    data->buffer.push_back(ptv);
                    if (color) data->buffer_rgb.push_back(rgb[p]);
                    voxel.data = data;
                    note_live(x, y, z);
                } else {
                    VoxelInfo* data = voxel.data;  // block attached by a clean pass, grid.hpp:234-241
                    if (cfg.reserve > 0) data->buffer.reserve((size_t)cfg.reserve);
                    data->viewpoint = viewpoint;
                    data->buffer.push_back(ptv);
                    if (color) data->buffer_rgb.push_back(rgb[p]);
                }
            }
            // grid.hpp:244-277 dependant updates
            VoxelInfo* data = voxel.data;
            const int d_size = (int)data->dependants.size();
            for (int i = 0; i < d_size; i++) {
                int xx, yy, zz;
                own_coords(data->dependants[i], xx, yy, zz);
                VoxelInfo* dep = find_voxel(xx, yy, zz)->data;
                V3 dep_centre = voxel_center(xx, yy, zz);
                V3 proj = project_point_to_vector(ptv, dep_centre, dep->normal, ball_r_f);
                double distance_to_normal = (double)norm3(sub3(ptv, proj));
                if (distance_to_normal < cfg.cylinder_radius) {
                    welford(dep, proj, distance_to_normal);
                    if (color) add_color(dep, rgb[p]);
                }
            }
        }
    }

    // node.cpp:182-216 decode (x,y,z via field offsets), node.cpp:251-255 z-clip,
    // node.cpp:289 transformPointCloud(Affine3d), node.cpp:290 viewpoint, then addPoints.
    // `n` must already be row_step/point_step (first-row rule, node.cpp:185,190).
    void capture(const uint8_t* base, size_t n, uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                 const double* T /* 3x4 row-major */, uint32_t off_rgb = 0xFFFFFFFFu)
    {
        std::vector<V3> out;
        std::vector<uint32_t> out_rgb;
        const bool color = cfg.fuse_color != 0 && off_rgb != 0xFFFFFFFFu;
        out.reserve(n);
        for (size_t i = 0; i < n; i++) {
            const uint8_t* rec = base + i * (size_t)point_step;
            float x, y, z;
            memcpy(&x, rec + off_x, 4);
            memcpy(&y, rec + off_y, 4);
            memcpy(&z, rec + off_z, 4);
            n_presented++;
            if (!((double)z < cfg.z_clip_max && (double)z > cfg.z_clip_min)) continue;
            n_zclip_pass++;
            V3 q;
            q.x = (float)(T[0] * (double)x + T[1] * (double)y + T[2] * (double)z + T[3]);
            q.y = (float)(T[4] * (double)x + T[5] * (double)y + T[6] * (double)z + T[7]);
            q.z = (float)(T[8] * (double)x + T[9] * (double)y + T[10] * (double)z + T[11]);
            out.push_back(q);
            if (color) {
                uint32_t c;
                memcpy(&c, rec + off_rgb, 4);
                out_rgb.push_back(c);
            }
        }
        V3 vp = {(float)T[3], (float)T[7], (float)T[11]};
        add_points(out.data(), out.size(), vp, color ? out_rgb.data() : nullptr);
    }

    // grid.hpp:311-454 updateThicknessVectors<N,K>
    void clean()
    {
        state_changed = false;
        std::vector<uint64_t> keys;  // own keys, in processing order
        keys.reserve(unprocessed_.size());
        for (auto key : unprocessed_) {
            int x, y, z;
            if (cfg.order_mode == 1)
                ref_coords(key, x, y, z);
            else
                own_coords(key, x, y, z);
            keys.push_back(own_key(x, y, z));
        }
        if (cfg.order_mode == 0) std::sort(keys.begin(), keys.end());
        const int K = cfg.K;
        std::vector<int> available;
        std::vector<V3> cloud;
        for (size_t ki = 0; ki < keys.size(); ki++) {
            int x, y, z;
            own_coords(keys[ki], x, y, z);
            Voxel* vit = find_voxel(x, y, z);
            if (vit == nullptr || !vit->occupied) continue;
            VoxelInfo* data = vit->data;
            int total = 0;
            available.clear();
            const int nd = (int)dx.size();
            for (int d = 0; d < nd; d++) {
                int i = dx[d], j = dy[d], kk = dz[d];
                if (valid_coord(x + i, y + j, z + kk)) {
                    Voxel nb = lookup(x + i, y + j, z + kk);
                    if (nb.occupied) {
                        available.push_back(d);
                        total += 1;
                    }
                }
            }
            if (total > cfg.gate && !data->normal_found) {
                cloud.resize(total);
                int counter = 0;
                for (int d : available) {
                    int i = dx[d], j = dy[d], kk = dz[d];
                    cloud[counter++] = voxel_center(x + i, y + j, z + kk);
                }
                V3 normal{0, 0, 0};
                V3 centroid = voxel_center(x, y, z);
                get_normal(cloud.data(), total, normal, cfg.pcl_shifted_cov != 0);
                V3 vp = data->viewpoint;
                V3 dir = normalized3(sub3(vp, centroid));
                if (dot3(dir, normal) < 0.0f) normal = {normal.x * -1.0f, normal.y * -1.0f, normal.z * -1.0f};
                data->normal = normal;
                data->normal_found = true;
                const uint64_t hash = keys[ki];
                for (int i = -K; i <= K; i++) {
                    const float step = (float)((double)i * xres_);
                    V3 nb = add3(centroid, mul3(step, data->normal));
                    if (!valid_point(nb)) continue;
                    int xx, yy, zz;
                    voxel_coords(nb, xx, yy, zz);
                    if (xx == INT_MIN || yy == INT_MIN || zz == INT_MIN) continue;
                    if (!valid_coord(xx, yy, zz)) continue;
                    Voxel& nv = get_voxel(xx, yy, zz);
                    if (nv.occupied) {
                        VoxelInfo* nd_ = nv.data;
                        nd_->dependants.push_back(hash);
                        // iterate a snapshot length: when nv is this voxel (i==0) the buffer is not modified here
                        const size_t bl = nd_->buffer.size();
                        for (size_t b = 0; b < bl; b++) {
                            const V3 pt = nd_->buffer[b];
                            V3 proj = project_point_to_vector(pt, centroid, data->normal, ball_r_f);
                            double distance_to_normal = (double)norm3(sub3(pt, proj));
                            if (distance_to_normal < cfg.cylinder_radius) {
                                welford(data, proj, distance_to_normal);
                                if (cfg.fuse_color && b < nd_->buffer_rgb.size()) add_color(data, nd_->buffer_rgb[b]);
                            }
                        }
                    } else {
                        // grid.hpp:443-449: overwrites any previous dependants-only block (leak in the reference)
                        if (nv.data == nullptr) note_live(xx, yy, zz);
                        delete nv.data;
                        VoxelInfo* fresh = new VoxelInfo();
                        fresh->dependants.push_back(hash);
                        nv.data = fresh;
                    }
                }
            }
        }
    }

    // grid.hpp:456-488 downloadData: lexicographic (x,y,z) scan of cells < dim, occupied && normal_found
    void extract(std::vector<Row>& rows) const
    {
        std::vector<uint64_t> keys;
        for_each_voxel([&](uint64_t key, const Voxel& v) {
            if (!v.occupied || !v.data->normal_found) return;
            int x, y, z;
            own_coords(key, x, y, z);
            if (!valid_coord(x, y, z)) return;
            keys.push_back(key);
        });
        std::sort(keys.begin(), keys.end());
        rows.resize(keys.size());
        for (size_t i = 0; i < keys.size(); i++) {
            Row& r = rows[i];
            own_coords(keys[i], r.ix, r.iy, r.iz);
            const VoxelInfo* d = const_cast<Oracle*>(this)->find_voxel(r.ix, r.iy, r.iz)->data;
            r.count = (uint32_t)d->count;
            r.x = d->centroid.x;
            r.y = d->centroid.y;
            r.z = d->centroid.z;
            r.nx = d->normal.x;
            r.ny = d->normal.y;
            r.nz = d->normal.z;
            r.sdx = d->sd.x;
            r.sdy = d->sd.y;
            r.sdz = d->sd.z;
            r.mean_dist = d->mean_dist;
            r.sd_dist = d->sd_dist;
            r.rgb = 0;  // never written by the reference (grid.hpp:471-479)
            if (cfg.fuse_color && d->count > 0) {
                // EXTENSION: mean colour of the cylinder members per channel, round half up: floor(sum / count + 1/2)
                const uint64_t n2 = 2ull * (uint64_t)d->count;
                const uint32_t cr = (uint32_t)((2 * d->csum[0] + (uint64_t)d->count) / n2);
                const uint32_t cg = (uint32_t)((2 * d->csum[1] + (uint64_t)d->count) / n2);
                const uint32_t cb = (uint32_t)((2 * d->csum[2] + (uint64_t)d->count) / n2);
                r.rgb = (std::min(cr, 255u) << 16) | (std::min(cg, 255u) << 8) | std::min(cb, 255u);
            }
        }
    }


    // ---- all-cores variant (TIMING BASELINE ONLY; never used for parity) -----------------------------------------
    // The reference's engine is serial (its OpenMP pragmas are commented out, grid.hpp:190-193,318-321).  This is the
    // "fair multi-core comparison" of BASELINE.md: the same per-point work spread over OpenMP threads with the voxel
    // store sharded 256 ways (one mutex per shard).  Welford updates happen in arrival order, so floats differ in the
    // last bits from run to run; counts, occupancy and normals do not.  Clean stays sequential except the plane fits.
    static constexpr int kShards = 256;
    struct Shard {
        std::mutex m;
        std::unordered_map<uint64_t, Voxel> map;
    };
    std::unique_ptr<Shard[]> shards_;
    static int shard_of(uint64_t key) { return (int)((key * 0x9E3779B97F4A7C15ull) >> 56); }

    void capture_mt(const uint8_t* base, size_t n, uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                    const double* T)
    {
        if (!shards_) shards_.reset(new Shard[kShards]);
        state_changed = true;
        const V3 viewpoint = {(float)T[3], (float)T[7], (float)T[11]};
        uint64_t presented = 0, zpass = 0, inserted = 0;
#pragma omp parallel for schedule(static) reduction(+ : presented, zpass, inserted)
        for (long long i = 0; i < (long long)n; i++) {
            const uint8_t* rec = base + (size_t)i * point_step;
            float x, y, z;
            memcpy(&x, rec + off_x, 4);
            memcpy(&y, rec + off_y, 4);
            memcpy(&z, rec + off_z, 4);
            presented++;
            if (!((double)z < cfg.z_clip_max && (double)z > cfg.z_clip_min)) continue;
            zpass++;
            V3 q;
            q.x = (float)(T[0] * (double)x + T[1] * (double)y + T[2] * (double)z + T[3]);
            q.y = (float)(T[4] * (double)x + T[5] * (double)y + T[6] * (double)z + T[7]);
            q.z = (float)(T[8] * (double)x + T[9] * (double)y + T[10] * (double)z + T[11]);
            int vx, vy, vz;
            voxel_coords(q, vx, vy, vz);
            if (!valid_point(q) || vx == INT_MIN || vy == INT_MIN || vz == INT_MIN) continue;
            inserted++;
            const uint64_t key = own_key(vx, vy, vz);
            uint64_t deps_inline[24];
            std::vector<uint64_t> deps_spill;
            const uint64_t* deps = deps_inline;
            size_t n_deps = 0;
            {
                Shard& sh = shards_[shard_of(key)];
                std::lock_guard<std::mutex> lk(sh.m);
                Voxel& voxel = sh.map[key];
                if (!voxel.data) voxel.data = new VoxelInfo();
                if (voxel.occupied) {
                    if (!voxel.data->normal_found) voxel.data->buffer.push_back(q);
                } else {
                    voxel.occupied = true;
                    voxel.data->viewpoint = viewpoint;
                    voxel.data->buffer.push_back(q);
                }
                // copied so that no two shard locks are ever held together
                const std::vector<uint64_t>& dv = voxel.data->dependants;
                n_deps = dv.size();
                if (n_deps <= 24) {
                    for (size_t d = 0; d < n_deps; d++) deps_inline[d] = dv[d];
                } else {
                    deps_spill = dv;
                    deps = deps_spill.data();
                }
            }
            for (size_t d = 0; d < n_deps; d++) {
                const uint64_t dk = deps[d];
                int xx, yy, zz;
                own_coords(dk, xx, yy, zz);
                const V3 centre = voxel_center(xx, yy, zz);
                Shard& ds = shards_[shard_of(dk)];
                std::lock_guard<std::mutex> lk(ds.m);
                VoxelInfo* dep = ds.map.find(dk)->second.data;
                V3 proj = project_point_to_vector(q, centre, dep->normal, ball_r_f);
                double distance_to_normal = (double)norm3(sub3(q, proj));
                if (distance_to_normal < cfg.cylinder_radius) welford(dep, proj, distance_to_normal);
            }
        }
        n_presented += presented;
        n_zclip_pass += zpass;
        n_inserted += inserted;
    }

    Voxel lookup_mt(int x, int y, int z) const
    {
        const uint64_t k = own_key(x, y, z);
        const Shard& sh = shards_[shard_of(k)];
        auto it = sh.map.find(k);
        if (it == sh.map.end()) return Voxel();
        return it->second;
    }

    void clean_mt()
    {
        state_changed = false;
        if (!shards_) return;
        const bool trace = getenv("HORACLE_TRACE") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t0 = now();
        std::vector<uint64_t> keys;  // sorted for locality only: the one order-dependent rule is resolved by key below
        {
            std::vector<std::vector<uint64_t>> part(kShards);
#pragma omp parallel for schedule(dynamic, 4)
            for (int s = 0; s < kShards; s++)
                for (auto& kv : shards_[s].map)
                    if (kv.second.occupied && !kv.second.data->normal_found) part[s].push_back(kv.first);
            for (int s = 0; s < kShards; s++) keys.insert(keys.end(), part[s].begin(), part[s].end());
            std::sort(keys.begin(), keys.end());
        }
        const int nd = (int)dx.size();
        double t1 = now();
        std::vector<V3> normals(keys.size());
        std::vector<uint8_t> gated(keys.size(), 0);
#pragma omp parallel for schedule(dynamic, 256)
        for (long long ki = 0; ki < (long long)keys.size(); ki++) {  // occupancy is frozen during a pass: fits are independent
            int x, y, z;
            own_coords(keys[ki], x, y, z);
            std::vector<V3> cloud;
            for (int d = 0; d < nd; d++) {
                const int xx = x + dx[d], yy = y + dy[d], zz = z + dz[d];
                if (valid_coord(xx, yy, zz) && lookup_mt(xx, yy, zz).occupied) cloud.push_back(voxel_center(xx, yy, zz));
            }
            if ((int)cloud.size() > cfg.gate) {
                V3 nrm{0, 0, 0};
                get_normal(cloud.data(), (int)cloud.size(), nrm, cfg.pcl_shifted_cov != 0);
                const V3 centroid = voxel_center(x, y, z);
                const V3 vp = lookup_mt(x, y, z).data->viewpoint;
                const V3 dir = normalized3(sub3(vp, centroid));
                if (dot3(dir, nrm) < 0.0f) nrm = {nrm.x * -1.0f, nrm.y * -1.0f, nrm.z * -1.0f};
                normals[ki] = nrm;
                gated[ki] = 1;
            }
        }
        double t2 = now();
        // Registration + replay, parallel over candidates.  A candidate only writes its own statistics, point buffers are
        // frozen during a pass, appends to an occupied target's dependant list take the target's shard lock, and the
        // "last registrant in canonical order wins" overwrite of unoccupied targets (grid.hpp:443-449) is resolved
        // afterwards from a sorted list.
        struct Pre {
            uint64_t target, owner;
        };
        std::vector<Pre> pre_all;
#pragma omp parallel
        {
            std::vector<Pre> pre;
#pragma omp for schedule(dynamic, 64) nowait
            for (long long ki = 0; ki < (long long)keys.size(); ki++) {
                if (!gated[ki]) continue;
                int x, y, z;
                own_coords(keys[ki], x, y, z);
                VoxelInfo* data;
                {
                    Shard& sh = shards_[shard_of(keys[ki])];
                    std::lock_guard<std::mutex> lk(sh.m);
                    data = sh.map.find(keys[ki])->second.data;
                }
                data->normal = normals[ki];
                data->normal_found = true;
                const V3 centroid = voxel_center(x, y, z);
                for (int i = -cfg.K; i <= cfg.K; i++) {
                    const float step = (float)((double)i * xres_);
                    V3 nb = add3(centroid, mul3(step, data->normal));
                    if (!valid_point(nb)) continue;
                    int xx, yy, zz;
                    voxel_coords(nb, xx, yy, zz);
                    if (xx == INT_MIN || yy == INT_MIN || zz == INT_MIN || !valid_coord(xx, yy, zz)) continue;
                    const uint64_t nk = own_key(xx, yy, zz);
                    VoxelInfo* target = nullptr;
                    {
                        Shard& sh = shards_[shard_of(nk)];
                        std::lock_guard<std::mutex> lk(sh.m);
                        auto it = sh.map.find(nk);
                        if (it != sh.map.end() && it->second.occupied) {
                            target = it->second.data;
                            target->dependants.push_back(keys[ki]);
                        }
                    }
                    if (!target) {
                        pre.push_back({nk, keys[ki]});
                        continue;
                    }
                    const size_t bl = target->buffer.size();
                    for (size_t b = 0; b < bl; b++) {
                        const V3 pt = target->buffer[b];
                        V3 proj = project_point_to_vector(pt, centroid, data->normal, ball_r_f);
                        double distance_to_normal = (double)norm3(sub3(pt, proj));
                        if (distance_to_normal < cfg.cylinder_radius) welford(data, proj, distance_to_normal);
                    }
                }
            }
#pragma omp critical
            pre_all.insert(pre_all.end(), pre.begin(), pre.end());
        }
        double t3 = now();
        {
            std::vector<std::vector<Pre>> bucket(kShards);
            for (const Pre& pr : pre_all) bucket[shard_of(pr.target)].push_back(pr);
#pragma omp parallel for schedule(dynamic, 4)
            for (int sh = 0; sh < kShards; sh++) {
                std::vector<Pre>& b = bucket[sh];
                std::sort(b.begin(), b.end(),
                          [](const Pre& a, const Pre& c) { return a.target != c.target ? a.target < c.target : a.owner < c.owner; });
                for (size_t i = 0; i < b.size(); i++) {
                    if (i + 1 < b.size() && b[i + 1].target == b[i].target) continue;  // a later registrant overwrites
                    Voxel& nv = shards_[sh].map[b[i].target];
                    delete nv.data;
                    nv.data = new VoxelInfo();
                    nv.data->dependants.push_back(b[i].owner);
                }
            }
        }
        if (trace)
            fprintf(stderr, "clean_mt: %zu keys gather %.3f fit %.3f register+replay %.3f overwrite %.3f s\n", keys.size(), t1 - t0,
                    t2 - t1, t3 - t2, now() - t3);
    }

    void extract_mt(std::vector<Row>& rows) const
    {
        rows.clear();
        if (!shards_) return;
        std::vector<std::pair<uint64_t, const VoxelInfo*>> found;
        for (int s = 0; s < kShards; s++)
            for (auto& kv : shards_[s].map) {
                if (!kv.second.occupied || !kv.second.data->normal_found) continue;
                int x, y, z;
                own_coords(kv.first, x, y, z);
                if (valid_coord(x, y, z)) found.push_back({kv.first, kv.second.data});
            }
        std::sort(found.begin(), found.end());
        rows.resize(found.size());
        for (size_t i = 0; i < found.size(); i++) {
            const VoxelInfo* d = found[i].second;
            Row& r = rows[i];
            own_coords(found[i].first, r.ix, r.iy, r.iz);
            r.count = (uint32_t)d->count;
            r.x = d->centroid.x, r.y = d->centroid.y, r.z = d->centroid.z;
            r.nx = d->normal.x, r.ny = d->normal.y, r.nz = d->normal.z;
            r.sdx = d->sd.x, r.sdy = d->sd.y, r.sdz = d->sd.z;
            r.mean_dist = d->mean_dist;
            r.sd_dist = d->sd_dist;
            r.rgb = 0;
        }
    }

    uint64_t count_normals_mt() const
    {
        uint64_t n = 0;
        if (shards_)
            for (int s = 0; s < kShards; s++)
                for (auto& kv : shards_[s].map)
                    if (kv.second.occupied && kv.second.data->normal_found) n++;
        return n;
    }

    void clear()
    {
        if (shards_) {
            for (int sh = 0; sh < kShards; sh++) {
                for (auto& kv : shards_[sh].map) delete kv.second.data;
                shards_[sh].map.clear();
            }
        }
        for_each_voxel([&](uint64_t, const Voxel& v) { delete v.data; });
        if (dense_) {
            for (uint64_t k : dense_live_) {
                int x, y, z;
                own_coords(k, x, y, z);
                dense_[dense_index(x, y, z)] = Voxel();
            }
            dense_live_.clear();
        }
        voxels_.clear();
        unprocessed_.clear();
        state_changed = true;  // grid.hpp:169
    }
};

}  // namespace

// ---- C API (ctypes) ------------------------------------------------------------------------------

extern "C" {

void* horacle_create(const Config* cfg) { return new Oracle(*cfg); }
void horacle_destroy(void* h) { delete (Oracle*)h; }
void horacle_dims(void* h, int32_t out[3], double* res_out)
{
    Oracle* o = (Oracle*)h;
    out[0] = o->xdim_;
    out[1] = o->ydim_;
    out[2] = o->zdim_;
    *res_out = o->xres_;
}
void horacle_capture(void* h, const void* base, uint64_t n, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                     uint32_t off_z, const double* pose)
{
    ((Oracle*)h)->capture((const uint8_t*)base, n, point_step, off_x, off_y, off_z, pose);
}
int32_t horacle_is_dense(void* h) { return ((Oracle*)h)->dense_ != nullptr ? 1 : 0; }
void horacle_capture_rgb(void* h, const void* base, uint64_t n, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                         uint32_t off_z, uint32_t off_rgb, const double* pose)
{
    ((Oracle*)h)->capture((const uint8_t*)base, n, point_step, off_x, off_y, off_z, pose, off_rgb);
}
void horacle_add_points(void* h, const float* xyz, uint64_t n, const float* vp)
{
    ((Oracle*)h)->add_points((const V3*)xyz, n, V3{vp[0], vp[1], vp[2]});
}
void horacle_clean(void* h) { ((Oracle*)h)->clean(); }
// all-cores timing baseline (separate voxel store; do not mix with the sequential calls on one handle)
void horacle_capture_mt(void* h, const void* base, uint64_t n, uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                        const double* pose)
{
    ((Oracle*)h)->capture_mt((const uint8_t*)base, n, point_step, off_x, off_y, off_z, pose);
}
void horacle_clean_mt(void* h) { ((Oracle*)h)->clean_mt(); }
// thread count of the all-cores variant; returns what OpenMP will use
int32_t horacle_set_threads(int32_t n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;  // built without -fopenmp: the pragmas are ignored and the variant runs on one thread
#endif
}
uint64_t horacle_normals_mt(void* h) { return ((Oracle*)h)->count_normals_mt(); }
int32_t horacle_is_dirty(void* h) { return ((Oracle*)h)->state_changed ? 1 : 0; }
void horacle_clear(void* h) { ((Oracle*)h)->clear(); }
uint64_t horacle_extract(void* h, Row* out, uint64_t cap)
{
    std::vector<Row> rows;
    ((Oracle*)h)->extract(rows);
    if (out) {
        uint64_t n = std::min<uint64_t>(cap, rows.size());
        memcpy(out, rows.data(), n * sizeof(Row));
    }
    return rows.size();
}
uint64_t horacle_extract_mt(void* h, Row* out, uint64_t cap)
{
    std::vector<Row> rows;
    ((Oracle*)h)->extract_mt(rows);
    if (out) {
        uint64_t n = std::min<uint64_t>(cap, rows.size());
        memcpy(out, rows.data(), n * sizeof(Row));
    }
    return rows.size();
}
void horacle_counters(void* h, uint64_t out[6])
{
    Oracle* o = (Oracle*)h;
    out[0] = o->n_presented;
    out[1] = o->n_zclip_pass;
    out[2] = o->n_inserted;
    uint64_t occ = 0, nf = 0, buffered = 0;
    o->for_each_voxel([&](uint64_t, const Voxel& v) {
        if (v.occupied) occ++;
        if (v.data) {
            if (v.data->normal_found) nf++;
            buffered += v.data->buffer.size();
        }
    });
    out[3] = occ;
    out[4] = nf;
    out[5] = buffered;
}
// occupied voxel triplets, ascending key; returns count
uint64_t horacle_occupied(void* h, int32_t* xyz, uint64_t cap)
{
    Oracle* o = (Oracle*)h;
    std::vector<uint64_t> keys;
    o->for_each_voxel([&](uint64_t key, const Voxel& v) {
        if (v.occupied) keys.push_back(key);
    });
    std::sort(keys.begin(), keys.end());
    if (xyz) {
        uint64_t n = std::min<uint64_t>(cap, keys.size());
        for (uint64_t i = 0; i < n; i++) Oracle::own_coords(keys[i], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    }
    return keys.size();
}
// dependants of a cell (own keys decoded to triplets); returns count
uint64_t horacle_dependants(void* h, int32_t x, int32_t y, int32_t z, int32_t* xyz, uint64_t cap)
{
    Oracle* o = (Oracle*)h;
    Voxel v = o->lookup(x, y, z);
    if (!v.data) return 0;
    uint64_t n = v.data->dependants.size();
    if (xyz)
        for (uint64_t i = 0; i < std::min(n, cap); i++)
            Oracle::own_coords(v.data->dependants[i], xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return n;
}

// ---- leaf probes (unit tests and GPU leaf-parity tests) ----

// a3: node.cpp:289 transform, one pose, n points (xyz packed) -> out xyz
void horacle_probe_transform(const double* T, const float* xyz, uint64_t n, float* out)
{
    for (uint64_t i = 0; i < n; i++) {
        double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        out[3 * i + 0] = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
        out[3 * i + 1] = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
        out[3 * i + 2] = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
    }
}
// a4/a5: index + bbox validity. out_idx: 3 ints per point, out_valid: 1 byte per point
void horacle_probe_index(void* h, const float* xyz, uint64_t n, int32_t* out_idx, uint8_t* out_valid)
{
    Oracle* o = (Oracle*)h;
    for (uint64_t i = 0; i < n; i++) {
        V3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        o->voxel_coords(p, out_idx[3 * i], out_idx[3 * i + 1], out_idx[3 * i + 2]);
        out_valid[i] = o->valid_point(p) ? 1 : 0;
    }
}
void horacle_probe_center(void* h, const int32_t* idx, uint64_t n, float* out)
{
    Oracle* o = (Oracle*)h;
    for (uint64_t i = 0; i < n; i++) {
        V3 c = o->voxel_center(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]);
        out[3 * i] = c.x;
        out[3 * i + 1] = c.y;
        out[3 * i + 2] = c.z;
    }
}
// a11: normal from an occupancy stencil around cell (x,y,z); occ[d] in setK order (size (2k+1)^3).
// vp != NULL applies the orientation of grid.hpp:393-396.  Returns total occupied (valid) count;
// normal written only if total >= 3.
int32_t horacle_probe_normal(void* h, int32_t x, int32_t y, int32_t z, const uint8_t* occ, const float* vp, float* normal_out)
{
    Oracle* o = (Oracle*)h;
    std::vector<V3> cloud;
    for (size_t d = 0; d < o->dx.size(); d++) {
        int xx = x + o->dx[d], yy = y + o->dy[d], zz = z + o->dz[d];
        if (o->valid_coord(xx, yy, zz) && occ[d]) cloud.push_back(o->voxel_center(xx, yy, zz));
    }
    V3 nrm{0, 0, 0};
    if (get_normal(cloud.data(), (int)cloud.size(), nrm, o->cfg.pcl_shifted_cov != 0)) {
        if (vp) {
            V3 dir = normalized3(sub3(V3{vp[0], vp[1], vp[2]}, o->voxel_center(x, y, z)));
            if (dot3(dir, nrm) < 0.0f) nrm = {nrm.x * -1.0f, nrm.y * -1.0f, nrm.z * -1.0f};
        }
        normal_out[0] = nrm.x;
        normal_out[1] = nrm.y;
        normal_out[2] = nrm.z;
    }
    return (int32_t)cloud.size();
}
// a9: projection + distance for n (point, centre, normal) triples.
void horacle_probe_project(float ball_r, const float* pts, const float* centres, const float* normals, uint64_t n,
                           float* proj_out, double* dist_out)
{
    for (uint64_t i = 0; i < n; i++) {
        V3 p{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
        V3 c{centres[3 * i], centres[3 * i + 1], centres[3 * i + 2]};
        V3 nn{normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
        V3 pr = project_point_to_vector(p, c, nn, ball_r);
        proj_out[3 * i] = pr.x;
        proj_out[3 * i + 1] = pr.y;
        proj_out[3 * i + 2] = pr.z;
        dist_out[i] = (double)norm3(sub3(p, pr));
    }
}
void horacle_probe_trig(const float* y, const float* x, uint64_t n, float* atan2_out, float* cos_out, float* sin_out)
{
    for (uint64_t i = 0; i < n; i++) {
        atan2_out[i] = odm_atan2f(y[i], x[i]);
        cos_out[i] = odm_cosf(x[i]);
        sin_out[i] = odm_sinf(x[i]);
    }
}
void horacle_probe_eigen33(const float* m9, float* vec_out)
{
    float m[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) m[i][j] = m9[i * 3 + j];
    V3 v = eigen33_smallest(m);
    vec_out[0] = v.x;
    vec_out[1] = v.y;
    vec_out[2] = v.z;
}
uint64_t horacle_sizeof_row(void) { return sizeof(Row); }
uint64_t horacle_sizeof_config(void) { return sizeof(Config); }

}  // extern "C"
