// examples/hfpf_demo.cpp -- a C++-only run of the whole path through the node shell (no Python, no ROS):
// synthetic sensor -> ~start -> N x PointCloud2 callbacks with tf poses -> periodic clean -> ~process ->
// <dir>/test_cloud.pcd + <dir>/meta.csv.  Prints per-stage wall times.
//
//   hfpf_demo <out_dir> [frames=30] [W=640] [H=480] [resolution=0.001] [clean_every=10] [seed=0xF051] [pose_seed=0x5E3]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hfpf_node.h"

extern "C" {
void hfpf_synth_pose(uint64_t seed, uint32_t frame_idx, double max_angle_deg, double jitter, double pose_out[12]);
void hfpf_synth_frame(uint64_t seed, uint32_t frame_idx, uint32_t W, uint32_t H, double fx_override, const double pose[12], double noise_sigma,
                      uint32_t nan_permille, uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_rgb, void* out);
}

namespace {
struct Tf {
    uint64_t pose_seed;
};
int lookup(void* user, const char*, const char* source, double pose[12], char* err, uint32_t cap)
{
    const Tf* tf = static_cast<const Tf*>(user);
    if (strncmp(source, "camera_", 7) != 0) {
        snprintf(err, cap, "unknown frame %s", source);
        return 1;
    }
    hfpf_synth_pose(tf->pose_seed, (uint32_t)atoi(source + 7), 30.0, 0.05, pose);
    return 0;
}
double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s <out_dir> [frames] [W] [H] [resolution] [clean_every] [seed] [pose_seed]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const uint32_t frames = argc > 2 ? (uint32_t)atoi(argv[2]) : 30;
    const uint32_t W = argc > 3 ? (uint32_t)atoi(argv[3]) : 640, H = argc > 4 ? (uint32_t)atoi(argv[4]) : 480;
    const float res = argc > 5 ? (float)atof(argv[5]) : 0.001f;
    const uint32_t clean_every = argc > 6 ? (uint32_t)atoi(argv[6]) : 10;
    const uint64_t seed = argc > 7 ? strtoull(argv[7], nullptr, 0) : 0xF051;
    Tf tf{argc > 8 ? strtoull(argv[8], nullptr, 0) : 0x5E3};
    const double fx = W == 640 ? 0.0 : 615.0;  // smaller images are crops of the 640x480 sensor

    const double box[6] = {-0.5, 0.5, -0.5, 0.5, 0.0, 1.0};
    hfpf_node_params p;
    hfpf_node_default_params(&p);
    p.fusion_frame = "base_link";
    p.directory_name = dir.c_str();
    p.bounding_box = box;
    p.bounding_box_len = 6;
    p.engine.resolution = res;
    p.clean_period_s = 0;  // explicit schedule instead of the 5 s thread
    p.final_clean_on_process = 1;
    hfpf_node* node = nullptr;
    if (hfpf_node_create(&p, lookup, &tf, &node) != HFPF_OK) {
        fprintf(stderr, "create: %s\n", hfpf_node_last_error(nullptr));
        return 1;
    }
    hfpf_trigger_response r;
    hfpf_node_start(node, &r);
    std::vector<uint8_t> buf((size_t)W * H * 16);
    double t_cb = 0, t_clean = 0;
    for (uint32_t f = 0; f < frames; f++) {
        double pose[12];
        hfpf_synth_pose(tf.pose_seed, f, 30.0, 0.05, pose);
        hfpf_synth_frame(seed, f, W, H, fx, pose, 0.0005, 20, 16, 0, 4, 8, 12, buf.data());
        const std::string frame_id = "camera_" + std::to_string(f);
        hfpf_cloud_msg m{buf.data(), 1, W * H, 16, W * H * 16, 0, 4, 8, 12, frame_id.c_str()};  // published height=1 (first-row rule)
        double t0 = now();
        if (hfpf_node_on_point_cloud(node, &m) != 1) {
            fprintf(stderr, "frame %u not integrated: %s\n", f, hfpf_node_last_error(node));
            return 1;
        }
        t_cb += now() - t0;
        if (clean_every && (f + 1) % clean_every == 0 && f + 1 < frames) {
            t0 = now();
            hfpf_node_clean_now(node);
            t_clean += now() - t0;
        }
    }
    hfpf_sync(hfpf_node_grid(node));
    double t0 = now();
    if (hfpf_node_process(node, &r) != HFPF_OK || !r.success) {
        fprintf(stderr, "process: %s\n", r.message);
        return 1;
    }
    const double t_proc = now() - t0;
    hfpf_node_stats st;
    hfpf_node_get_stats(node, &st);
    printf("%s\nframes %llu integrated %llu  callbacks %.3f s  cleans %.3f s (%llu passes)  process %.3f s\n", r.message,
           (unsigned long long)st.received, (unsigned long long)st.integrated, t_cb, t_clean, (unsigned long long)st.clean_passes, t_proc);
    hfpf_node_destroy(node);
    return 0;
}
