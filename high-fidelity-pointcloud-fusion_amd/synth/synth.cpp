// synth.cpp -- ROS-free stand-in for the sensor driver: renders synthetic organised RGB-D clouds
//  into PointCloud2-shaped byte buffers (fields x,y,z,rgb as f32 at caller-given
// offsets, caller-given point_step), published height=1,width=W*H, row-major, so the reference's
// first-row-only decoder (node.cpp:185,190) would consume every point.
//
// Pure integer/IEEE arithmetic (no libm in the per-pixel path), so a (seed, frame, pose) triple
// yields the same bytes on every machine.  Pose generation uses libm sin/cos; poses are *inputs*
// to both the oracle and the engine, so that does not affect parity.
//
// Scene (fusion frame, metres): tilted back plane z = 0.56 + 0.05*x, sphere c=(0.05,0,0.45) r=0.10,
// box [-0.20,-0.08]x[-0.10,0.10]x[0.40,0.60]; nominal camera at the origin looking along +z,
// pinhole fx=fy=615*(W/640) (or fx_override: a W x H crop of a finer sensor), cx=W/2, cy=H/2.  Depth noise ~ N(0, sigma^2) (Irwin-Hall of 4),
// nan_permille of pixels are NaN (all three coordinates), rgb = hash of (frame, pixel).
#include <cmath>
#include <cstdint>
#include <cstring>

#include <omp.h>

namespace {
inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

inline double ray_scene(const double o[3], const double d[3])
{
    double best = 1e30;
    // plane: z - 0.05 x = 0.56
    {
        double den = d[2] - 0.05 * d[0];
        if (den > 1e-12 || den < -1e-12) {
            double t = (0.56 - (o[2] - 0.05 * o[0])) / den;
            if (t > 0 && t < best) best = t;
        }
    }
    // sphere
    {
        const double c[3] = {0.05, 0.0, 0.45};
        const double r = 0.10;
        double oc[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
        double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        double b = 2.0 * (oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2]);
        double cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - r * r;
        double disc = b * b - 4 * a * cc;
        if (disc > 0) {
            double t = (-b - sqrt(disc)) / (2 * a);
            if (t > 0 && t < best) best = t;
        }
    }
    // box (slabs)
    {
        const double lo[3] = {-0.20, -0.10, 0.40}, hi[3] = {-0.08, 0.10, 0.60};
        double t0 = 0, t1 = 1e30;
        bool hit = true;
        for (int k = 0; k < 3 && hit; k++) {
            if (d[k] > -1e-12 && d[k] < 1e-12) {
                if (o[k] < lo[k] || o[k] > hi[k]) hit = false;
            } else {
                double ta = (lo[k] - o[k]) / d[k], tb = (hi[k] - o[k]) / d[k];
                if (ta > tb) { double s = ta; ta = tb; tb = s; }
                if (ta > t0) t0 = ta;
                if (tb < t1) t1 = tb;
                if (t0 > t1) hit = false;
            }
        }
        if (hit && t0 > 0 && t0 < best) best = t0;
    }
    return best;
}
}  // namespace

extern "C" {

// pose_out: 3x4 row-major fusion_frame <- camera.  max_angle_deg = 0 and jitter = 0 give identity.
void hfpf_synth_pose(uint64_t seed, uint32_t frame_idx, double max_angle_deg, double jitter, double pose_out[12])
{
    uint64_t h = splitmix64(seed ^ (0xA5A5A5A5ull + (uint64_t)frame_idx * 0x100000001B3ull));
    double ax[3];
    double nrm = 0;
    do {
        for (int k = 0; k < 3; k++) {
            h = splitmix64(h);
            ax[k] = 2.0 * u01(h) - 1.0;
        }
        nrm = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
    } while (nrm > 1.0 || nrm < 1e-6);
    nrm = sqrt(nrm);
    for (int k = 0; k < 3; k++) ax[k] /= nrm;
    h = splitmix64(h);
    double ang = (2.0 * u01(h) - 1.0) * max_angle_deg * 3.14159265358979323846 / 180.0;
    double c = cos(ang), s = sin(ang), C = 1 - c;
    double R[9] = {c + ax[0] * ax[0] * C,         ax[0] * ax[1] * C - ax[2] * s, ax[0] * ax[2] * C + ax[1] * s,
                   ax[1] * ax[0] * C + ax[2] * s, c + ax[1] * ax[1] * C,         ax[1] * ax[2] * C - ax[0] * s,
                   ax[2] * ax[0] * C - ax[1] * s, ax[2] * ax[1] * C + ax[0] * s, c + ax[2] * ax[2] * C};
    // orbit about the scene centre (0,0,0.45): camera position = ctr + R*(0 - ctr) + jitter
    const double ctr[3] = {0, 0, 0.45};
    double t[3];
    for (int k = 0; k < 3; k++) {
        h = splitmix64(h);
        double j = (2.0 * u01(h) - 1.0) * jitter;
        t[k] = ctr[k] - (R[3 * k] * ctr[0] + R[3 * k + 1] * ctr[1] + R[3 * k + 2] * ctr[2]) + j;
    }
    for (int r = 0; r < 3; r++) {
        pose_out[4 * r + 0] = R[3 * r + 0];
        pose_out[4 * r + 1] = R[3 * r + 1];
        pose_out[4 * r + 2] = R[3 * r + 2];
        pose_out[4 * r + 3] = t[r];
    }
}

// Renders one frame in the CAMERA frame (what the sensor publishes) into `out`
// (W*H records of point_step bytes; bytes outside the four fields are zeroed).
void hfpf_synth_frame(uint64_t seed, uint32_t frame_idx, uint32_t W, uint32_t H, double fx_override,
                      const double pose[12], double noise_sigma, uint32_t nan_permille, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                      uint32_t off_z, uint32_t off_rgb, void* out)
{
    const double fx = fx_override > 0 ? fx_override : 615.0 * ((double)W / 640.0), fy = fx, cx = W / 2.0, cy = H / 2.0;
    const double o[3] = {pose[3], pose[7], pose[11]};
    uint8_t* base = (uint8_t*)out;
    // GPU boxes expose every host CPU but grant a small share: cap the team instead of one thread per visible CPU
    const int n_threads = omp_get_max_threads() < 16 ? omp_get_max_threads() : 16;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t v = 0; v < (int64_t)H; v++) {
        for (uint32_t u = 0; u < W; u++) {
            const uint64_t pix = (uint64_t)v * W + u;
            uint8_t* rec = base + pix * (uint64_t)point_step;
            memset(rec, 0, point_step);
            uint64_t h = splitmix64(seed + 0x51ED27ull * (uint64_t)frame_idx + pix * 0x9E3779B1ull);
            const double dc[3] = {((double)u + 0.5 - cx) / fx, ((double)v + 0.5 - cy) / fy, 1.0};
            double dw[3];
            for (int k = 0; k < 3; k++) dw[k] = pose[4 * k] * dc[0] + pose[4 * k + 1] * dc[1] + pose[4 * k + 2] * dc[2];
            double t = ray_scene(o, dw);
            h = splitmix64(h);
            const bool is_nan = (uint32_t)(h % 1000u) < nan_permille;
            double g = 0;
            for (int k = 0; k < 4; k++) {
                h = splitmix64(h);
                g += u01(h);
            }
            g = (g - 2.0) * 1.7320508075688772;  // unit variance
            float x, y, z;
            if (is_nan || t > 1e29) {
                uint32_t q = 0x7FC00000u;
                memcpy(&x, &q, 4);
                y = x;
                z = x;
            } else {
                double depth = t + g * noise_sigma;
                x = (float)(dc[0] * depth);
                y = (float)(dc[1] * depth);
                z = (float)depth;
            }
            h = splitmix64(h);
            uint32_t rgb = (uint32_t)(h & 0x00FFFFFFu);
            memcpy(rec + off_x, &x, 4);
            memcpy(rec + off_y, &y, 4);
            memcpy(rec + off_z, &z, 4);
            memcpy(rec + off_rgb, &rgb, 4);
        }
    }
}

}  // extern "C"
