// csrc/stats.hpp -- order-free per-voxel statistics.
//
// The reference keeps float Welford recurrences per voxel (grid.hpp:264-273, 428-438): count, centroid,
// per-axis sd, mean_dist, sd_dist.  Those depend on the order in which points arrive.  The engine
// instead accumulates exact integer sums, so any interleaving of threads, frames or GPUs gives the
// same bits, and merging two GPUs' partial records is a plain integer add.
//
// Every projected point lies on its voxel's line: proj = a - s*ab (grid.hpp:40-49) with a, ab fixed per voxel and
// s the one scalar that depends on the point.  So the voxel's statistics are functions of the moments of s:
//     centroid_i = a_i - E[s]*ab_i                (the reference averages the f32-rounded proj_i; the two differ
//                                                   by the mean of those roundings, <= 6e-8 m, far inside 1e-5)
//     sd_i       = ab_i^2 * (E[s^2] - E[s]^2)      (population variance, as the reference's recurrence)
// and mean_dist / sd_dist are the moments of dist = ||p - proj||_f32 (grid.hpp:261).
//
// One record = 8 int64 words = 64 bytes = ONE memory-side atomic segment (the chip retires ~20 G such
// segments per second however few bytes each carries, so the record is sized to be exactly one):
//
//   word 0      count
//   word 1      sum of rint(s * 2^es)
//   word 2      sum of rint(s*s * 2^ess)           (s*s in f32)
//   word 3      sum of rint(dist * 2^ed)
//   word 4      sum of rint(dist*dist * 2^edd)     (dist*dist in f32)
//   words 5-7   sum of r, g, b of the members      (HFPF_FLAG_FUSE_COLOR only; the reference never fuses colour)
//
// Exponents are chosen at create time (hfpf.hip setup_params) so that ONE contribution is below 2^27 in magnitude:
// 16 of them add up in 32 bits (k_update sums the lanes of a row with DPP adds before it touches the LDS table), and
// 2^25 samples per voxel cannot overflow an int64.  Resolution: 2^-27 of the line segment (30 mm) ~ 2e-10 m.
#pragma once
#include "tables.hpp"

// 1: the moments are those of u = s - 0.5 (the offset from the cell centre in units of the 30 mm segment) instead of s: the
// variance E[u^2] - E[u]^2 then cancels ~0.03 against ~1e-4 instead of 0.25 against ~1e-4, and one fixed-point step is 9x (u) and
// 80x (u^2) finer.  hfpf.hip picks the scales from the same switch, k_extract_rows adds the 0.5 back in f64.
#ifndef HFPF_CENTERED_MOMENTS
#define HFPF_CENTERED_MOMENTS 1
#endif

namespace hfpf {

enum StatWord : int { SW_COUNT = 0, SW_S = 1, SW_SS = 2, SW_D = 3, SW_DD = 4, SW_R = 5, SW_G = 6, SW_B = 7 };
constexpr int kStatUsed = 5;  // words every record uses (colour adds three)

// Contribution of one (point, dependant) pair as four 32-bit integers.  The scalings are by powers of two (exact);
// rintf is round-to-nearest-even (v_rndne_f32).  Every kernel that updates statistics calls this one function.
struct PairDelta {
    int32_t s, ss, d, dd;
};
__device__ __forceinline__ PairDelta pair_delta(const GridParams& g, float s, float distf)
{
    PairDelta q;
#if HFPF_CENTERED_MOMENTS
    const float u = s - 0.5f;  // exact for s in [0.25, 1] (Sterbenz); u = 0 at the cell centre
    q.s = (int32_t)rintf(u * g.fs_scale);
    q.ss = (int32_t)rintf((u * u) * g.fss_scale);
#else
    q.s = (int32_t)rintf(s * g.fs_scale);
    q.ss = (int32_t)rintf((s * s) * g.fss_scale);
#endif
    q.d = (int32_t)rintf(distf * g.fd_scale);
    q.dd = (int32_t)rintf((distf * distf) * g.fdd_scale);
    return q;
}

// Per-thread running sums of member pairs (k_replay, direct form of k_integrate).  COLOR = false drops the colour
// accumulators at compile time (the reference's behaviour and the default path).
template <bool COLOR>
struct StatDeltaT {
    long long v[kStatUsed];
    long long rgb[COLOR ? 3 : 1];
};

template <bool COLOR>
__device__ __forceinline__ void stat_delta_zero(StatDeltaT<COLOR>& d)
{
#pragma unroll
    for (int i = 0; i < kStatUsed; i++) d.v[i] = 0;
    if constexpr (COLOR) d.rgb[0] = d.rgb[1] = d.rgb[2] = 0;
}

template <bool COLOR>
__device__ __forceinline__ void stat_delta_add(StatDeltaT<COLOR>& d, const PairDelta& q, uint32_t rgb)
{
    d.v[SW_COUNT] += 1;
    d.v[SW_S] += (long long)q.s;
    d.v[SW_SS] += (long long)q.ss;
    d.v[SW_D] += (long long)q.d;
    d.v[SW_DD] += (long long)q.dd;
    if constexpr (COLOR) {
        d.rgb[0] += (long long)((rgb >> 16) & 255u);
        d.rgb[1] += (long long)((rgb >> 8) & 255u);
        d.rgb[2] += (long long)(rgb & 255u);
    }
}

}  // namespace hfpf
