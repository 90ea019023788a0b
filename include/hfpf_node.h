/* include/hfpf_node.h -- ROS-free shell of the reference node `pointcloud_fusion_and_filter`.
 *
 * Reproduces the control surface of class PointcloudFusion
 * (pointcloud_fusion/pointcloud_fusion/src/pointcloud_fusion_and_filter.cpp:99-169,327-440, "node.cpp") on top of
 * libhfpf.so: the four std_srvs/Trigger services, the PointCloud2 subscriber callback with its tf lookup, the
 * periodic clean thread and the two output files.  ROS itself is absent from this image, so the shell takes plain
 * structs; host/ros_shell.cpp (built only where catkin/roscpp exist) maps the ROS types onto it 1:1.
 */
#ifndef HFPF_NODE_H
#define HFPF_NODE_H
#include "hfpf.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hfpf_node hfpf_node;

typedef struct hfpf_node_params {
    uint32_t struct_size;
    const char* fusion_frame;        /* private param `fusion_frame`, default "fusion_frame" (node.cpp:447) */
    const char* directory_name;      /* private param `directory_name`, default "./" (node.cpp:449) */
    const double* bounding_box;      /* private param `bounding_box` (node.cpp:451): xmin,xmax,ymin,ymax,zmin,zmax */
    uint32_t bounding_box_len;       /* must be 6; the reference indexes box[0..5] unchecked (node.cpp:162) */
    hfpf_config engine;              /* every other engine knob; bbox is overwritten from bounding_box */
    double clean_period_s;           /* 5.0 = sleep(5) of cleanGrid (node.cpp:323); <= 0: no thread, use hfpf_node_clean_now */
    int32_t final_clean_on_process;  /* 0 = reference behaviour (process does not clean first, node.cpp:377-398) */
    int32_t write_variants;          /* 1 = also write the files of the reference's `#if 0` block (node.cpp:399-437):
                                        test_cloud_{50,100,150,200,250,300}.pcd (downloadHQ), test_cloud_classified.pcd,
                                        test_cloud_normals.pcd; each a device-side filtered extract.  0 = reference behaviour */
} hfpf_node_params;

/* sensor_msgs/PointCloud2 as the decoder uses it (node.cpp:182-216): fields[0..3] = x,y,z,rgb. */
typedef struct hfpf_cloud_msg {
    const void* data;
    uint32_t height, width, point_step, row_step;
    uint32_t off_x, off_y, off_z, off_rgb; /* fields[0..3].offset */
    const char* frame_id;                  /* header.frame_id */
} hfpf_cloud_msg;

/* std_srvs/TriggerResponse */
typedef struct hfpf_trigger_response {
    int32_t success;
    char message[256];
} hfpf_trigger_response;

/* tf_buffer_.lookupTransform(target, source, ros::Time(0)) (node.cpp:336): return 0 and fill the row-major 3x4
 * pose target<-source, or non-zero for a tf2::TransformException (text in err). */
typedef int (*hfpf_tf_lookup_fn)(void* user, const char* target_frame, const char* source_frame, double pose_3x4[12],
                                 char* err, uint32_t err_cap);

void hfpf_node_default_params(hfpf_node_params* p);
/* PointcloudFusion::PointcloudFusion (node.cpp:146-169): builds the grid, starts the clean thread. */
int hfpf_node_create(const hfpf_node_params* p, hfpf_tf_lookup_fn tf, void* tf_user, hfpf_node** out);
int hfpf_node_destroy(hfpf_node* n);
const char* hfpf_node_last_error(const hfpf_node* n);

/* onReceivedPointCloud (node.cpp:327-349) + the two capture threads (node.cpp:218-299).
 * Returns 1 = integrated, 0 = dropped (not started, or tf failure: warn + drop, node.cpp:340-344), < 0 = error.
 * Only the first row is consumed: n = row_step / point_step (node.cpp:185,190). */
int hfpf_node_on_point_cloud(hfpf_node* n, const hfpf_cloud_msg* msg);

/* ~start ~stop ~reset ~process (node.cpp:154-157, 351-440). start/stop/reset set success=true like the reference;
 * process writes <directory_name>/test_cloud.pcd and /meta.csv (node.cpp:395-396), clears the grid (node.cpp:438)
 * and -- documented deviation -- reports success=true with a message (the reference never sets the response). */
int hfpf_node_start(hfpf_node* n, hfpf_trigger_response* res);
int hfpf_node_stop(hfpf_node* n, hfpf_trigger_response* res);
int hfpf_node_reset(hfpf_node* n, hfpf_trigger_response* res);
int hfpf_node_process(hfpf_node* n, hfpf_trigger_response* res);

/* The latent publisher of the reference: it advertises `~pcl_fusion_node/processed_cloud_normals` (sensor_msgs/PointCloud2,
 * node.cpp:138,158) and never publishes.  Here ~process hands the extracted PointXYZRGBNormal rows (the cloud it is about to
 * save, in the fusion frame) to whoever registered; the ROS adapter (host/ros_shell.cpp) publishes them on that topic.
 * The rows are only valid during the call. */
typedef void (*hfpf_publish_fn)(void* user, const hfpf_row* rows, uint64_t n_rows, const char* frame_id);
int hfpf_node_set_publisher(hfpf_node* n, hfpf_publish_fn fn, void* user);

/* One iteration of cleanGrid (node.cpp:301-325): clean iff state_changed.  Returns 1 if a pass ran. */
int hfpf_node_clean_now(hfpf_node* n);
hfpf_handle* hfpf_node_grid(hfpf_node* n);

typedef struct hfpf_node_stats {
    uint64_t received, integrated, dropped_not_started, dropped_tf, clean_passes, process_calls;
    int32_t started, cloud_subscription_started;
} hfpf_node_stats;
int hfpf_node_get_stats(hfpf_node* n, hfpf_node_stats* out);

#ifdef __cplusplus
}
#endif
#endif
