// host/fusion_node.cpp -- ROS-free shell of pointcloud_fusion_and_filter over the C ABI (include/hfpf_node.h).
//
// What changed relative to the reference's threading (node.cpp:166-168): the decode/clip thread (addPoints,
// node.cpp:218-263) and the transform/insert thread (updateStates, node.cpp:265-299) existed to overlap CPU work;
// both stages are one GPU launch now, so the subscriber callback hands the message straight to hfpf_integrate
// (which copies it to pinned staging and returns).  The clean thread (cleanGrid, node.cpp:301-325) is kept.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hfpf_node.h"

struct hfpf_node {
    hfpf_handle* grid = nullptr;
    std::string fusion_frame, directory_name, err;
    hfpf_tf_lookup_fn tf = nullptr;
    void* tf_user = nullptr;
    std::atomic<bool> start_{false};                       // node.cpp:135 (a plain bool shared across threads there)
    std::atomic<bool> cloud_subscription_started_{false};  // node.cpp:136
    std::string pointcloud_frame_;                         // node.cpp:128
    std::mutex frame_mtx;
    double clean_period_s = 5.0;
    bool final_clean = false;
    bool write_variants = false;
    hfpf_publish_fn publish = nullptr;  // ~pcl_fusion_node/processed_cloud_normals (node.cpp:158)
    void* publish_user = nullptr;
    std::thread clean_thread;
    std::mutex cv_mtx;
    std::condition_variable cv;
    bool quit = false;
    std::atomic<uint64_t> received{0}, integrated{0}, dropped_not_started{0}, dropped_tf{0}, clean_passes{0}, process_calls{0};
};

namespace {
thread_local std::string g_err;  // last error of hfpf_node_create

int nfail(hfpf_node* n, int code, const std::string& msg)
{
    if (n) n->err = msg;
    else g_err = msg;
    return code;
}

void set_res(hfpf_trigger_response* res, bool ok, const std::string& msg)
{
    if (!res) return;
    res->success = ok ? 1 : 0;
    snprintf(res->message, sizeof res->message, "%s", msg.c_str());
}

void clean_loop(hfpf_node* n)  // cleanGrid, node.cpp:301-325
{
    std::unique_lock<std::mutex> lk(n->cv_mtx);
    while (!n->quit) {
        lk.unlock();
        if (hfpf_is_dirty(n->grid) > 0) {  // if(grid_.state_changed)
            if (hfpf_clean(n->grid) == HFPF_OK) n->clean_passes++;
            else fprintf(stderr, "[hfpf_node] clean failed: %s\n", hfpf_last_error(n->grid));
        }
        lk.lock();
        n->cv.wait_for(lk, std::chrono::duration<double>(n->clean_period_s), [n] { return n->quit; });  // sleep(5)
    }
}
}  // namespace

extern "C" {

void hfpf_node_default_params(hfpf_node_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->fusion_frame = "fusion_frame";  // node.cpp:447
    p->directory_name = "./";          // node.cpp:449
    p->bounding_box = nullptr;         // node.cpp:451 default: empty vector
    p->bounding_box_len = 0;
    hfpf_default_config(&p->engine);
    p->clean_period_s = 5.0;  // node.cpp:323
    p->final_clean_on_process = 0;
    p->write_variants = 0;
}

const char* hfpf_node_last_error(const hfpf_node* n) { return n ? n->err.c_str() : g_err.c_str(); }

int hfpf_node_create(const hfpf_node_params* p, hfpf_tf_lookup_fn tf, void* tf_user, hfpf_node** out)
{
    if (!p || !out) return nfail(nullptr, HFPF_ERR_BAD_ARG, "hfpf_node_create: null argument");
    *out = nullptr;
    if (p->struct_size != sizeof(hfpf_node_params)) return nfail(nullptr, HFPF_ERR_BAD_CONFIG, "hfpf_node_params.struct_size mismatch");
    // the reference reads box[0..5] of whatever the parameter server returned (node.cpp:451,162): out-of-bounds when empty
    if (!p->bounding_box || p->bounding_box_len != 6)
        return nfail(nullptr, HFPF_ERR_BAD_CONFIG, "param bounding_box must hold 6 values (xmin,xmax,ymin,ymax,zmin,zmax)");
    hfpf_node* n = new hfpf_node();
    n->fusion_frame = p->fusion_frame ? p->fusion_frame : "fusion_frame";
    n->directory_name = p->directory_name ? p->directory_name : "./";
    n->tf = tf;
    n->tf_user = tf_user;
    n->clean_period_s = p->clean_period_s;
    n->final_clean = p->final_clean_on_process != 0;
    n->write_variants = p->write_variants != 0;
    hfpf_config cfg = p->engine;
    cfg.struct_size = sizeof cfg;
    memcpy(cfg.bbox, p->bounding_box, 6 * sizeof(double));
    int rc = hfpf_create(&cfg, &n->grid);
    if (rc != HFPF_OK) {
        g_err = hfpf_last_error(nullptr);
        delete n;
        return rc;
    }
    if (n->clean_period_s > 0) n->clean_thread = std::thread(clean_loop, n);
    *out = n;
    return HFPF_OK;
}

int hfpf_node_destroy(hfpf_node* n)
{
    if (!n) return HFPF_OK;
    {
        std::lock_guard<std::mutex> lk(n->cv_mtx);
        n->quit = true;
    }
    n->cv.notify_all();
    if (n->clean_thread.joinable()) n->clean_thread.join();
    hfpf_destroy(n->grid);
    delete n;
    return HFPF_OK;
}

int hfpf_node_on_point_cloud(hfpf_node* n, const hfpf_cloud_msg* msg)
{
    if (!n || !msg) return HFPF_ERR_BAD_ARG;
    n->received++;
    {
        std::lock_guard<std::mutex> lk(n->frame_mtx);
        n->pointcloud_frame_ = msg->frame_id ? msg->frame_id : "";  // node.cpp:329
    }
    n->cloud_subscription_started_ = true;  // node.cpp:330
    if (!n->start_) {                       // node.cpp:331
        n->dropped_not_started++;
        return 0;
    }
    double pose[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // Affine3d::Identity(), node.cpp:333
    if (n->tf) {
        char err[256] = {0};
        if (n->tf(n->tf_user, n->fusion_frame.c_str(), msg->frame_id ? msg->frame_id : "", pose, err, sizeof err) != 0) {
            fprintf(stderr, "[hfpf_node] WARN %s\n", err);  // ROS_WARN + drop, node.cpp:340-344
            n->dropped_tf++;
            return 0;
        }
    }
    if (!msg->data || msg->point_step == 0) return nfail(n, HFPF_ERR_BAD_ARG, "empty PointCloud2");
    const uint32_t n_points = msg->row_step / msg->point_step;  // first row only, node.cpp:185,190
    int rc = hfpf_integrate(n->grid, msg->data, n_points, msg->point_step, msg->off_x, msg->off_y, msg->off_z, msg->off_rgb, pose);
    if (rc != HFPF_OK) return nfail(n, rc, hfpf_last_error(n->grid));
    n->integrated++;
    return 1;
}

int hfpf_node_start(hfpf_node* n, hfpf_trigger_response* res)
{
    if (!n) return HFPF_ERR_BAD_ARG;
    n->start_ = true;  // node.cpp:364
    set_res(res, true, "");
    return HFPF_OK;
}

int hfpf_node_stop(hfpf_node* n, hfpf_trigger_response* res)
{
    if (!n) return HFPF_ERR_BAD_ARG;
    n->start_ = false;  // node.cpp:372
    set_res(res, true, "");
    return HFPF_OK;
}

int hfpf_node_reset(hfpf_node* n, hfpf_trigger_response* res)
{
    if (!n) return HFPF_ERR_BAD_ARG;
    n->start_ = false;                       // node.cpp:354
    n->cloud_subscription_started_ = false;  // node.cpp:357 (clouds_.clear(): no queue here; the grid is NOT touched)
    set_res(res, true, "");
    return HFPF_OK;
}

int hfpf_node_process(hfpf_node* n, hfpf_trigger_response* res)
{
    if (!n) return HFPF_ERR_BAD_ARG;
    n->process_calls++;
    // The reference polls until both queues are empty (node.cpp:380-394); integrate calls are already in stream order here.
    if (n->final_clean && hfpf_is_dirty(n->grid) > 0) {
        int rc = hfpf_clean(n->grid);
        if (rc != HFPF_OK) {
            set_res(res, false, hfpf_last_error(n->grid));
            return nfail(n, rc, hfpf_last_error(n->grid));
        }
        n->clean_passes++;
    }
    const std::string cloud_location = n->directory_name + "/test_cloud.pcd";  // node.cpp:395
    const std::string meta_location = n->directory_name + "/meta.csv";         // node.cpp:396
    hfpf_row* rows = nullptr;
    uint64_t nr = 0;
    int rc = hfpf_extract(n->grid, &rows, &nr);  // grid_.downloadData, node.cpp:398
    if (rc == HFPF_OK) rc = hfpf_write_pcd(rows, nr, cloud_location.c_str());
    if (rc == HFPF_OK) rc = hfpf_write_meta_csv(rows, nr, meta_location.c_str());
    if (rc == HFPF_OK && n->publish) n->publish(n->publish_user, rows, nr, n->fusion_frame.c_str());  // processed_cloud_, node.cpp:158
    hfpf_free_rows(rows);
    if (rc == HFPF_OK && n->write_variants) {  // the reference's `#if 0` block, node.cpp:399-437
        struct Variant {
            const char* file;
            double min_count;
            int classify;
            int white;
        };
        const Variant vs[] = {{"test_cloud_50.pcd", 50, -1, 1},   {"test_cloud_100.pcd", 100, -1, 1}, {"test_cloud_150.pcd", 150, -1, 1},
                              {"test_cloud_200.pcd", 200, -1, 1}, {"test_cloud_250.pcd", 250, -1, 1}, {"test_cloud_300.pcd", 300, -1, 1},
                              {"test_cloud_classified.pcd", 0, 100 /* kGoodPointsThreshold, grid.hpp:34 */, 0}};
        for (const Variant& v : vs) {
            hfpf_extract_opts o;
            memset(&o, 0, sizeof o);
            o.struct_size = sizeof o;
            o.min_count = v.min_count;
            o.classify_threshold = v.classify;
            o.paint_white = v.white;
            hfpf_row* vr = nullptr;
            uint64_t vn = 0;
            rc = hfpf_extract_filtered(n->grid, &o, &vr, &vn);  // filtered and colour-coded on the device
            if (rc == HFPF_OK) rc = hfpf_write_pcd_xyzrgb(vr, vn, (n->directory_name + "/" + v.file).c_str(), 0, -1, 0);
            hfpf_free_rows(vr);
            if (rc != HFPF_OK) break;
        }
        if (rc == HFPF_OK) {  // download(PointXYZRGBNormal), grid.hpp:577-601
            hfpf_row* vr = nullptr;
            uint64_t vn = 0;
            rc = hfpf_extract(n->grid, &vr, &vn);
            if (rc == HFPF_OK) rc = hfpf_write_pcd(vr, vn, (n->directory_name + "/test_cloud_normals.pcd").c_str());
            hfpf_free_rows(vr);
        }
    }
    if (rc != HFPF_OK) {
        std::string m = rc == HFPF_ERR_IO ? "cannot write " + cloud_location + " / " + meta_location : std::string(hfpf_last_error(n->grid));
        // The reference ends getFusedCloud with grid_.clearVoxels() whatever happened before (node.cpp:438).  An engine failure
        // (capacity overflow, poisoned handle) leaves nothing worth keeping and hfpf_clear is the only way out of that state, so
        // the grid is cleared here too and the node can capture again without a restart; after an I/O error the fused data is
        // intact and kept, so that ~process can be repeated once the directory is writable.
        if (rc != HFPF_ERR_IO) {
            if (hfpf_clear(n->grid) == HFPF_OK) m += " (grid cleared)";
        }
        set_res(res, false, m);
        return nfail(n, rc, m);
    }
    rc = hfpf_clear(n->grid);  // grid_.clearVoxels(), node.cpp:438
    if (rc != HFPF_OK) {
        set_res(res, false, hfpf_last_error(n->grid));
        return nfail(n, rc, hfpf_last_error(n->grid));
    }
    char m[200];
    snprintf(m, sizeof m, "saved %llu points", (unsigned long long)nr);
    set_res(res, true, std::string(m) + " to " + cloud_location);
    return HFPF_OK;
}

int hfpf_node_set_publisher(hfpf_node* n, hfpf_publish_fn fn, void* user)
{
    if (!n) return HFPF_ERR_BAD_ARG;
    n->publish = fn;
    n->publish_user = user;
    return HFPF_OK;
}

int hfpf_node_clean_now(hfpf_node* n)
{
    if (!n) return HFPF_ERR_BAD_ARG;
    if (hfpf_is_dirty(n->grid) <= 0) return 0;
    int rc = hfpf_clean(n->grid);
    if (rc != HFPF_OK) return nfail(n, rc, hfpf_last_error(n->grid));
    n->clean_passes++;
    return 1;
}

hfpf_handle* hfpf_node_grid(hfpf_node* n) { return n ? n->grid : nullptr; }

int hfpf_node_get_stats(hfpf_node* n, hfpf_node_stats* out)
{
    if (!n || !out) return HFPF_ERR_BAD_ARG;
    out->received = n->received;
    out->integrated = n->integrated;
    out->dropped_not_started = n->dropped_not_started;
    out->dropped_tf = n->dropped_tf;
    out->clean_passes = n->clean_passes;
    out->process_calls = n->process_calls;
    out->started = n->start_ ? 1 : 0;
    out->cloud_subscription_started = n->cloud_subscription_started_ ? 1 : 0;
    return HFPF_OK;
}

}  // extern "C"
