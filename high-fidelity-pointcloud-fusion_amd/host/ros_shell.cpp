// host/ros_shell.cpp -- the ROS1 adapter of the node shell.  Built ONLY where catkin/roscpp/tf2 exist
// (add_executable guarded by find_package(catkin QUIET) in host/CMakeLists.txt); this image has no ROS, so the
// file is excluded from every build here and exercised through the ROS-free harness (tests/test_gpu_node.py).
//
// Keeps the reference's graph surface (node.cpp:146-169,442-460): node name pointcloud_fusion_and_filter
// (launch file), private services ~reset ~start ~stop ~process (std_srvs/Trigger), subscriber ~input_point_cloud
// (queue 100), latent publisher ~pcl_fusion_node/processed_cloud_normals, private params fusion_frame,
// directory_name, bounding_box (flange_frame is set by the launch file and never read, as in the reference).
#ifdef HFPF_WITH_ROS
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <std_srvs/Trigger.h>
#include <tf2_eigen/tf2_eigen.h>
#include <tf2_ros/transform_listener.h>

#include "../../include/hfpf_node.h"

namespace {
struct Shell {
    tf2_ros::Buffer tf_buffer;
    tf2_ros::TransformListener tf_listener{tf_buffer};
    hfpf_node* node = nullptr;

    static int lookup(void* user, const char* target, const char* source, double pose[12], char* err, uint32_t cap)
    {
        Shell* s = static_cast<Shell*>(user);
        try {
            const Eigen::Affine3d T = tf2::transformToEigen(s->tf_buffer.lookupTransform(target, source, ros::Time(0)));  // node.cpp:336-338
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) pose[4 * r + c] = T(r, c);
            return 0;
        } catch (tf2::TransformException& ex) {
            snprintf(err, cap, "%s", ex.what());
            return 1;
        }
    }
    void on_cloud(const sensor_msgs::PointCloud2ConstPtr& m)
    {
        if (m->fields.size() < 4) return;
        hfpf_cloud_msg msg{m->data.data(), m->height, m->width, m->point_step, m->row_step, m->fields[0].offset, m->fields[1].offset,
                           m->fields[2].offset, m->fields[3].offset, m->header.frame_id.c_str()};
        if (hfpf_node_on_point_cloud(node, &msg) < 0) ROS_ERROR("%s", hfpf_node_last_error(node));
    }
    // ~pcl_fusion_node/processed_cloud_normals: the cloud ~process is about to save, as x y z rgb normal_x normal_y normal_z
    // curvature (8 x f32 per point, the field list savePCDFileASCII writes for PointXYZRGBNormal, grid.hpp:485)
    ros::Publisher* pub = nullptr;
    static void publish(void* user, const hfpf_row* rows, uint64_t n, const char* frame_id)
    {
        Shell* s = static_cast<Shell*>(user);
        if (!s->pub) return;
        sensor_msgs::PointCloud2 m;
        m.header.stamp = ros::Time::now();
        m.header.frame_id = frame_id;
        m.height = 1;
        m.width = (uint32_t)n;
        m.is_bigendian = false;
        m.is_dense = true;
        m.point_step = 32;
        m.row_step = m.point_step * m.width;
        const char* names[8] = {"x", "y", "z", "rgb", "normal_x", "normal_y", "normal_z", "curvature"};
        for (int k = 0; k < 8; ++k) {
            sensor_msgs::PointField f;
            f.name = names[k];
            f.offset = 4 * k;
            f.datatype = sensor_msgs::PointField::FLOAT32;
            f.count = 1;
            m.fields.push_back(f);
        }
        m.data.resize((size_t)m.row_step);
        for (uint64_t i = 0; i < n; ++i) {
            float rec[8] = {rows[i].x, rows[i].y, rows[i].z, 0.f, rows[i].nx, rows[i].ny, rows[i].nz, 0.f};
            const uint32_t rgba = 0xFF000000u | rows[i].rgb;
            memcpy(&rec[3], &rgba, 4);
            memcpy(&m.data[(size_t)i * 32], rec, 32);
        }
        s->pub->publish(m);
    }
    template <int (*F)(hfpf_node*, hfpf_trigger_response*)>
    bool srv(std_srvs::TriggerRequest&, std_srvs::TriggerResponse& res)
    {
        hfpf_trigger_response r;
        F(node, &r);
        res.success = r.success;
        res.message = r.message;
        return true;
    }
};
}  // namespace

int main(int argc, char** argv)
{
    ros::init(argc, argv, "fusion_node");  // node.cpp:444 (renamed by the launch file)
    ros::NodeHandle pnh("~");
    std::string fusion_frame, directory_name;
    std::vector<double> bounding_box;
    pnh.param<std::string>("fusion_frame", fusion_frame, "fusion_frame");
    pnh.param<std::string>("directory_name", directory_name, "./");
    pnh.param("bounding_box", bounding_box, std::vector<double>());
    hfpf_node_params p;
    hfpf_node_default_params(&p);
    p.fusion_frame = fusion_frame.c_str();
    p.directory_name = directory_name.c_str();
    p.bounding_box = bounding_box.data();
    p.bounding_box_len = (uint32_t)bounding_box.size();
    double v;
    int iv;
    if (pnh.getParam("resolution", v)) p.engine.resolution = (float)v;
    if (pnh.getParam("z_clip_min", v)) p.engine.z_clip_min = v;
    if (pnh.getParam("z_clip_max", v)) p.engine.z_clip_max = v;
    if (pnh.getParam("cylinder_radius", v)) p.engine.cylinder_radius = v;
    if (pnh.getParam("ball_radius", v)) p.engine.ball_radius = v;
    if (pnh.getParam("gate", iv)) p.engine.gate = iv;
    if (pnh.getParam("line_half_length", iv)) p.engine.K = iv;
    if (pnh.getParam("device", iv)) p.engine.device = iv;
    if (pnh.getParam("frame_width", iv) && iv > 0) p.engine.frame_width = (uint32_t)iv;  // image width of the sensor: 16x16-pixel tiles (speed only)
    if (pnh.getParam("clean_period", v)) p.clean_period_s = v;
    bool flag;
    if (pnh.getParam("fuse_color", flag) && flag) p.engine.flags |= HFPF_FLAG_FUSE_COLOR;
    if (pnh.getParam("pcl_shifted_covariance", flag) && flag) p.engine.flags |= HFPF_FLAG_PCL_SHIFTED_COV;
    if (pnh.getParam("write_variants", flag) && flag) p.write_variants = 1;  // the files of the reference's `#if 0` block, node.cpp:399-437
    Shell shell;
    if (hfpf_node_create(&p, &Shell::lookup, &shell, &shell.node) != HFPF_OK) {
        ROS_FATAL("%s", hfpf_node_last_error(nullptr));
        return 1;
    }
    ros::Subscriber sub = pnh.subscribe("input_point_cloud", 100, &Shell::on_cloud, &shell);                    // node.cpp:152
    ros::ServiceServer s1 = pnh.advertiseService("reset", &Shell::srv<hfpf_node_reset>, &shell);                // node.cpp:154
    ros::ServiceServer s2 = pnh.advertiseService("start", &Shell::srv<hfpf_node_start>, &shell);                // node.cpp:155
    ros::ServiceServer s3 = pnh.advertiseService("stop", &Shell::srv<hfpf_node_stop>, &shell);                  // node.cpp:156
    ros::ServiceServer s4 = pnh.advertiseService("process", &Shell::srv<hfpf_node_process>, &shell);            // node.cpp:157
    ros::Publisher pub = pnh.advertise<sensor_msgs::PointCloud2>("pcl_fusion_node/processed_cloud_normals", 1);  // node.cpp:158
    shell.pub = &pub;  // latent in the reference (advertised, never published); here ~process publishes the cloud it saves
    hfpf_node_set_publisher(shell.node, &Shell::publish, &shell);
    ros::Rate loop_rate(31);  // node.cpp:453
    while (ros::ok()) {
        ros::spinOnce();
        loop_rate.sleep();
    }
    hfpf_node_destroy(shell.node);
    return 0;
}
#endif  // HFPF_WITH_ROS
