// csrc/stats.hpp -- order-free per-voxel statistics.
//
// The reference keeps float Welford recurrences per voxel (grid.hpp:264-273, 428-438): count, centroid,
// per-axis sd, mean_dist, sd_dist.  Those depend on the order in which points arrive.  The engine
// instead accumulates exact integer sums, so any interleaving of threads, frames or GPUs gives the
// same bits, and merging two GPUs' partial records is a plain integer add:
//
//   word 0      count
//   words 1-3   sum of (proj_i - c_i) * 2^e1          c = cell centre of the record's voxel (f32)
//   words 4-6   sum of (proj_i - c_i)^2 * 2^e2
//   word 7      sum of dist * 2^ed                    dist = (double)||p - proj||_f32  (grid.hpp:261)
//   word 8      sum of dist^2 * 2^edd
//   words 9-11  sum of r, g, b of the member points   (extension: the reference never fuses colour)
//   words 12-15 unused (record = 128 bytes = two 64-byte atomic segments)
//
// proj_i - c_i is exact in f64 (both are f32 within a few voxels of each other); each sample is
// rounded once to the fixed-point grid (<= 2^-e1 / 2 absolute).  The exponents are chosen at create
// time so that 2^25 samples per voxel cannot overflow an int64.  hfpf_extract turns the sums into
// the reference's quantities: mean = c + S1/n, sd = S2/n - (S1/n)^2 (the Welford recurrence of
// grid.hpp:267 is the population variance), mean_dist = Sd/n, sd_dist = Sdd/n - mean_dist^2.
#pragma once
#include "tables.hpp"

namespace hfpf {

enum StatWord : int { SW_COUNT = 0, SW_S1 = 1, SW_S2 = 4, SW_D = 7, SW_DD = 8, SW_RGB = 9, SW_USED = 12 };

struct StatDelta {
    long long v[SW_USED];
};

__device__ __forceinline__ void stat_delta_zero(StatDelta& d)
{
#pragma unroll
    for (int i = 0; i < SW_USED; i++) d.v[i] = 0;
}

// Contribution of one cylinder member.
__device__ __forceinline__ void stat_delta_add(StatDelta& d, const GridParams& g, F3 proj, F3 c, double dist, uint32_t rgb)
{
    const double ox = (double)proj.x - (double)c.x;
    const double oy = (double)proj.y - (double)c.y;
    const double oz = (double)proj.z - (double)c.z;
    d.v[SW_COUNT] += 1;
    d.v[SW_S1 + 0] += __double2ll_rn(ox * g.s1_scale);
    d.v[SW_S1 + 1] += __double2ll_rn(oy * g.s1_scale);
    d.v[SW_S1 + 2] += __double2ll_rn(oz * g.s1_scale);
    d.v[SW_S2 + 0] += __double2ll_rn((ox * ox) * g.s2_scale);
    d.v[SW_S2 + 1] += __double2ll_rn((oy * oy) * g.s2_scale);
    d.v[SW_S2 + 2] += __double2ll_rn((oz * oz) * g.s2_scale);
    d.v[SW_D] += __double2ll_rn(dist * g.sd_scale);
    d.v[SW_DD] += __double2ll_rn((dist * dist) * g.sdd_scale);
    d.v[SW_RGB + 0] += (long long)((rgb >> 16) & 255u);
    d.v[SW_RGB + 1] += (long long)((rgb >> 8) & 255u);
    d.v[SW_RGB + 2] += (long long)(rgb & 255u);
}

__device__ __forceinline__ void stat_flush(unsigned long long* rec, const StatDelta& d)
{
#pragma unroll
    for (int i = 0; i < SW_USED; i++)
        if (d.v[i] != 0) atomicAdd(rec + i, (unsigned long long)d.v[i]);
}

}  // namespace hfpf
