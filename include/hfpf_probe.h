/* include/hfpf_probe.h -- leaf probes of libhfpf.so (TEST HOOKS, not part of the drop-in surface).
 *
 * Each probe runs, on the device, exactly the device function the fusion kernels use for one leaf of the
 * reference arithmetic, so tests/ can compare it bit-for-bit with the CPU oracle on large random inputs:
 *   hfpf_probe_points   a3 transform (node.cpp:289), a2 z-clip (node.cpp:251-255), a4 voxel index
 *                       (OccupancyGrid.hpp:630-637), a5 bbox test (OccupancyGrid.hpp:639-645)
 *   hfpf_probe_normals  a11 plane fit over a 5x5x5 occupancy stencil + orientation (OccupancyGrid.hpp:282-309,356-396)
 *   hfpf_probe_project  a9 projection + cylinder membership (OccupancyGrid.hpp:40-49,261-262); member_out bit 0 = the
 *                       reference's form, bit 1 = the form the kernels use (squared distance against the largest passing value),
 *                       bit 2 = the form with the hoisted division gives the same membership, parameter and distance bit for bit
 *   hfpf_probe_trig     the deterministic atan2/cos/sin used inside the plane fit
 * All pointers are HOST pointers; the probes copy in, launch, copy out and synchronise.
 */
#ifndef HFPF_PROBE_H
#define HFPF_PROBE_H
#include "hfpf.h"
#ifdef __cplusplus
extern "C" {
#endif

/* flags_out[i]: bit0 = z-clip pass (on the camera-frame z), bit1 = validPoints(transformed point). */
int hfpf_probe_points(hfpf_handle* h, const double pose_3x4[12], const float* xyz, uint64_t n, float* q_out,
                      int32_t* idx_out, uint8_t* flags_out);
/* cells: n*3 voxel indices; occ: n*125 bytes in setK order (x outermost); vps: n*3 viewpoints. */
int hfpf_probe_normals(hfpf_handle* h, uint64_t n, const int32_t* cells, const uint8_t* occ, const float* vps,
                       float* normals_out, int32_t* totals_out);
int hfpf_probe_project(hfpf_handle* h, uint64_t n, const float* pts, const float* centres, const float* normals,
                       float* proj_out, double* dist_out, uint8_t* member_out);
int hfpf_probe_trig(hfpf_handle* h, uint64_t n, const float* y, const float* x, float* atan2_out, float* cos_out,
                    float* sin_out);

#ifdef __cplusplus
}
#endif
#endif
