// csrc/stats.hpp -- order-free per-voxel statistics.
//
// The reference keeps float Welford recurrences per voxel (grid.hpp:264-273, 428-438): count, centroid,
// per-axis sd, mean_dist, sd_dist.  Those depend on the order in which points arrive.  The engine
// instead accumulates exact integer sums, so any interleaving of threads, frames or GPUs gives the
// same bits, and merging two GPUs' partial records is a plain integer add.
//
// One record = 8 int64 words = 64 bytes = ONE memory-side atomic segment (the chip retires ~20 G such
// segments per second however few bytes each carries, so the record is sized to be exactly one):
//
//   word 0      count
//   words 1-3   sum of o_i * 2^e1,  o = proj - c,  c = cell centre of the record's voxel (f32, exact in f64)
//   word 4      sum of |o|^2 * 2^e2
//   word 5      sum of dist * 2^ed                    dist = (double)||p - proj||_f32  (grid.hpp:261)
//   word 6      sum of dist^2 * 2^edd
//   word 7      unused
//
// Every projected point lies on the voxel's line (proj = a - s*ab with ab || n, grid.hpp:40-49), i.e.
// o = t*n up to f32 rounding of the projection itself (~6e-8 m), so the reference's per-axis variance is
// sd_i = n_i^2 * var(t) with var(t) = E|o|^2 - |E o|^2; one second-moment word replaces three.  The
// deviation from three exact per-axis sums is below the rounding noise of the reference's own f32
// recurrence.
//
// Optional colour record (HFPF_FLAG_FUSE_COLOR; the reference never fuses colour): 4 words, sum r,g,b.
//
// Exponents are chosen at create time so that 2^25 samples per voxel cannot overflow an int64.
#pragma once
#include "tables.hpp"

namespace hfpf {

enum StatWord : int { SW_COUNT = 0, SW_S1 = 1, SW_S2 = 4, SW_D = 5, SW_DD = 6, SW_USED = 7 };

// COLOR = false drops the colour accumulators at compile time (the reference's behaviour and the default path).
template <bool COLOR>
struct StatDeltaT {
    long long v[SW_USED];
    long long rgb[COLOR ? 3 : 1];
};

template <bool COLOR>
__device__ __forceinline__ void stat_delta_zero(StatDeltaT<COLOR>& d)
{
#pragma unroll
    for (int i = 0; i < SW_USED; i++) d.v[i] = 0;
    if constexpr (COLOR) d.rgb[0] = d.rgb[1] = d.rgb[2] = 0;
}

// Round-to-nearest-even double -> int64, bit-identical to rn_ll(x) for |x| < 2^51: adding 1.5 * 2^52 leaves the
// rounded integer in the low mantissa bits (the sum's ulp is 1).  Every value converted here is <= 2^38 in magnitude (the
// fixed-point scales are chosen that way at create, hfpf.hip setup_params).  Two cheap instructions instead of the six
// f64 ones (rndne, ldexp, floor, fma, two cvt) the generic conversion expands to -- the pair loops are VALU-bound.
__device__ __forceinline__ long long rn_ll(double x)
{
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    return __double_as_longlong(x + magic) - __double_as_longlong(magic);
}

// Contribution of one cylinder member.
template <bool COLOR>
__device__ __forceinline__ void stat_delta_add(StatDeltaT<COLOR>& d, const GridParams& g, F3 proj, F3 c, double dist, uint32_t rgb)
{
    const double ox = (double)proj.x - (double)c.x;
    const double oy = (double)proj.y - (double)c.y;
    const double oz = (double)proj.z - (double)c.z;
    d.v[SW_COUNT] += 1;
    d.v[SW_S1 + 0] += rn_ll(ox * g.s1_scale);
    d.v[SW_S1 + 1] += rn_ll(oy * g.s1_scale);
    d.v[SW_S1 + 2] += rn_ll(oz * g.s1_scale);
    d.v[SW_S2] += rn_ll(((ox * ox + oy * oy) + oz * oz) * g.s2_scale);
    d.v[SW_D] += rn_ll(dist * g.sd_scale);
    d.v[SW_DD] += rn_ll((dist * dist) * g.sdd_scale);
    if constexpr (COLOR) {
        d.rgb[0] += (long long)((rgb >> 16) & 255u);
        d.rgb[1] += (long long)((rgb >> 8) & 255u);
        d.rgb[2] += (long long)(rgb & 255u);
    }
}

}  // namespace hfpf
