/* include/hfpf.h -- C ABI of the MI355X-native occupancy-grid point-cloud fusion engine (libhfpf.so).
 *
 * This is the drop-in boundary for the reference's `class OccupancyGrid`
 * (pointcloud_fusion/pointcloud_fusion/include/utilities/OccupancyGrid.hpp:99-136, "grid.hpp" below),
 * which the reference node owns by value as `PointcloudFusion::grid_`
 * (pointcloud_fusion/pointcloud_fusion/src/pointcloud_fusion_and_filter.cpp:132, "node.cpp" below).
 * The reference has no FFI layer of its own; each entry point cites the member function / call site
 * it replaces.  Plain pointers and sizes only; no C++ or torch types cross this boundary; nothing
 * throws across it.  All functions return an hfpf_status (0 = ok, negative = error) unless noted,
 * and may be called from any thread (the handle serialises mutating calls internally; the reference
 * serialises with grid_mtx_, node.cpp:142,291,305).
 *
 * The engine has no CPU fallback: every entry point that computes runs HIP kernels on the
 * configured device and fails with HFPF_ERR_HIP when no device is usable.
 */
#ifndef HFPF_H
#define HFPF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HFPF_ABI_VERSION 6

/* hfpf_config.flags */
#define HFPF_FLAG_FUSE_COLOR 1u /* EXTENSION: also average the member points' RGB per voxel (reference: never, grid.hpp:471-479) */
#define HFPF_FLAG_DIRECT_UPDATE 4u /* dependant updates with one memory-side atomic per (point, dependant) pair.  Default (flag
                                      clear) is the two-pass form: points are binned per 8x8x8 brick, one workgroup per brick
                                      accumulates its statistic records in LDS and flushes each once per launch.  Same results
                                      either way (the sums are integers); the binned form is ~1.5x faster on the bench. */
#define HFPF_FLAG_PCL_SHIFTED_COV 2u /* plane fit with pcl::computeMeanAndCovarianceMatrix as PCL >= 1.11 computes it (moments of
                                        p - first point).  Default (0) is the single-pass form of PCL <= 1.10; the two differ
                                        visibly in f32 ~1 m from the origin (call site grid.hpp:302) */

typedef struct hfpf_handle hfpf_handle;

typedef enum hfpf_status {
    HFPF_OK = 0,
    HFPF_ERR_BAD_CONFIG = -1, /* bbox max<=min, resolution<=0, k!=2, K out of range ... */
    HFPF_ERR_BAD_ARG = -2,
    HFPF_ERR_CAPACITY = -3, /* a device pool (bricks, point log, normals, registrations) overflowed */
    HFPF_ERR_HIP = -4,      /* HIP runtime error; see hfpf_last_error */
    HFPF_ERR_STATE = -5,
    HFPF_ERR_IO = -6,
    HFPF_ERR_DIST = -7 /* RCCL / rank bootstrap error */
} hfpf_status;

/* Replaces setResolution/setDimensions/setK/construct (grid.hpp:614,604,138,621; called once from
 * the node ctor, node.cpp:161-164) plus the compile-time constants of the capture stage.
 * Defaults (hfpf_default_config) are the reference's values. */
typedef struct hfpf_config {
    uint32_t struct_size;     /* = sizeof(hfpf_config) */
    float resolution;         /* voxel edge, metres, AS FLOAT: the reference stores a float into a double
                                 member (grid.hpp:614-619), so 0.005f -> 0.004999999888241291. node.cpp:91 */
    double bbox[6];           /* xmin,xmax,ymin,ymax,zmin,zmax; launch param `bounding_box` (node.cpp:451,162) */
    int32_t k;                /* occupancy stencil half-width; must be 2 (grid.hpp:334 hard-codes 125 probes) */
    int32_t K;                /* line half-length in voxel steps (template arg, node.cpp:311,317) = 3 */
    int32_t gate;             /* a voxel gets a normal when occupied neighbours > gate (grid.hpp:352) = 20 */
    double cylinder_radius;   /* kCylinderRadius, grid.hpp:36 = 0.001 */
    double ball_radius;       /* kBballRadius, grid.hpp:35 = 0.015 */
    double z_clip_min;        /* kZmin, node.cpp:92 = 0.28 (camera frame, strict) */
    double z_clip_max;        /* kZmax, node.cpp:93 = 0.6 */
    int32_t device;           /* HIP device ordinal */
    uint32_t flags;           /* HFPF_FLAG_*; 0 = the reference's behaviour */
    /* Device pool capacities; 0 = engine default.  The reference grows without bound (README:12). */
    uint64_t max_bricks;      /* 8x8x8-voxel bricks that may be touched */
    uint64_t max_log_points;  /* points buffered while their voxel has no normal (grid.hpp:211,230) */
    uint64_t max_normals;     /* voxels that may receive a normal */
    uint64_t max_frames;      /* frame ids (viewpoint table) */
    /* Scheduling hint, no reference counterpart: pixels per image row when every frame is a row-major organised image
       (sensor_msgs/PointCloud2 width of an organised cloud, or the sensor's width for clouds republished as height=1);
       0 = unknown.  The integrate kernel then walks a frame in 16x16-pixel patches instead of 256-point runs, which
       touches about half as many voxel-table lines per point.  Used only when width and rows are multiples of 16 and
       n_points is a multiple of width; results never depend on it. */
    uint32_t frame_width;
    uint32_t reserved0;       /* 0 */
    /* Largest n_points * n_frames one hfpf_integrate_device call will carry; sizes the per-call brick bins at create like
       the other pools.  0 = grown on demand (the first call of a new size then pays for the allocation). */
    uint64_t max_call_points;
} hfpf_config;

/* One emitted voxel = one line of test_cloud.pcd + one line of meta.csv (grid.hpp:466-480). 64 bytes. */
typedef struct hfpf_row {
    int32_t ix, iy, iz;  /* voxel index triplet */
    uint32_t count;      /* points in cylinder (VoxelInfo::count) */
    float x, y, z;       /* cylinder-filtered mean of projected points (VoxelInfo::centroid); (0,0,0) when count==0 */
    float nx, ny, nz;    /* VoxelInfo::normal */
    float sdx, sdy, sdz; /* VoxelInfo::sd (population variance per axis, as the reference's Welford recurrence) */
    float mean_dist;     /* VoxelInfo::mean_dist */
    float sd_dist;       /* VoxelInfo::sd_dist */
    uint32_t rgb;        /* 0 (the reference never writes rgb); with HFPF_FLAG_FUSE_COLOR: mean colour of the cylinder members, 0x00RRGGBB */
} hfpf_row;

typedef struct hfpf_counters {
    uint64_t points_presented;  /* input points seen by integrate (before any clip) */
    uint64_t points_zclip_pass; /* survived the camera-frame z-clip */
    uint64_t points_in_bbox;    /* survived the bbox clip (= voxel touches) */
    uint64_t points_buffered;   /* appended to the point log */
    uint64_t dep_pairs_tested;  /* (point, dependant) cylinder tests in integrate */
    uint64_t dep_pairs_member;  /* ... that were inside the cylinder */
    uint64_t voxels_occupied;
    uint64_t voxels_with_normal;
    uint64_t bricks_allocated;
    uint64_t registrations;     /* dependant registrations on occupied cells (grid.hpp:417) */
    uint64_t dep_entries;       /* entries in the current dependant table */
    uint64_t frames_integrated;
    uint64_t clean_passes;
    uint64_t device_bytes;      /* HBM allocated by this handle */
    uint64_t replay_members;    /* buffered points found inside a cylinder when replayed by a clean pass (grid.hpp:418-440) */
    uint64_t points_direct;     /* of points_buffered: appended one by one through the overflow list (no bin region or a full one), outside the bricks' runs */
    uint64_t table_misses;      /* work items of the dependant update that found no slot in the LDS record table (updated HBM directly) */
    uint64_t update_extra_rounds; /* (ABI 5) sort rounds beyond the first that bricks of the dependant update / streaming replay took (a brick with more parked points than one LDS round holds) */
} hfpf_counters;

void hfpf_default_config(hfpf_config* cfg);
int hfpf_abi_version(void);

/* OccupancyGrid(), setResolution, setDimensions, setK, construct  (grid.hpp:111,614,604,138,621). */
int hfpf_create(const hfpf_config* cfg, hfpf_handle** out);
/* The reference never destroys its grid (no destructor; leaks). */
int hfpf_destroy(hfpf_handle* h);
/* Thread-local-free: the last error text of this handle (or of create when h == NULL). */
const char* hfpf_last_error(const hfpf_handle* h);
/* xdim,ydim,zdim as construct() truncates them (grid.hpp:623-625) and the resolution as a double. */
int hfpf_get_dims(const hfpf_handle* h, int32_t dims[3], double* resolution);

/* Integrate one frame given as PointCloud2-style records in HOST memory.
 * Replaces, in one call: pointCloud2ToPclXYZRGBOMP (node.cpp:182-216), the z-clip (node.cpp:251-255),
 * pcl::transformPointCloud (node.cpp:289), the viewpoint (node.cpp:290) and
 * OccupancyGrid::addPoints<N> (grid.hpp:185-280; call sites node.cpp:293,295).
 *   base        first record; n_points records of point_step bytes.  The caller applies the
 *               reference's first-row rule, i.e. n_points = row_step / point_step (node.cpp:185,190).
 *   off_*       byte offsets of the f32 fields x,y,z,rgb inside a record (fields[0..3].offset).
 *   pose_3x4    fusion_frame <- camera, row-major 3x4 f64 (the Affine3d of node.cpp:338).
 * The buffer is copied before the call returns.  Frame ids count up from 0 per handle.
 * The frame's kernels are launched at once when the engine's stream is idle; while it is busy with earlier frames the frame
 * waits (already uploading) and is launched together with the frames behind it, at most HFPF_HOST_BATCH (default 4) per launch.
 * Any other call on the handle launches what is waiting first; an error of a deferred launch is returned by that call.
 * Deferred errors: HFPF_OK means "accepted".  A pool that overflows on the device while the frame's kernels run (HFPF_ERR_CAPACITY)
 * is noticed at the handle's next counter read-back -- hfpf_clean, hfpf_extract, hfpf_sync, hfpf_get_counters -- and returned by
 * that call; the handle then refuses work (HFPF_ERR_STATE) until hfpf_clear.  Frames accepted after the failure and not yet
 * launched are dropped, and the call that finds them waiting says so with HFPF_ERR_STATE. */
int hfpf_integrate(hfpf_handle* h, const void* base, uint32_t n_points, uint32_t point_step, uint32_t off_x,
                   uint32_t off_y, uint32_t off_z, uint32_t off_rgb, const double pose_3x4[12]);

/* The same for a frame in PAGE-LOCKED host memory (hfpf_host_alloc below, or hipHostRegister by the caller): no bounce copy; the
 * upload runs on the engine's copy stream and overlaps the kernels of earlier frames, the call returns at once.  The buffer must
 * stay untouched until the next hfpf_sync / hfpf_clean / hfpf_extract of this handle (all of which wait for queued work).
 * hfpf_integrate itself uploads the same way after copying the caller's buffer into pinned staging. */
int hfpf_integrate_pinned(hfpf_handle* h, const void* pinned_base, uint32_t n_points, uint32_t point_step, uint32_t off_x,
                          uint32_t off_y, uint32_t off_z, uint32_t off_rgb, const double pose_3x4[12]);
int hfpf_host_alloc(hfpf_handle* h, uint64_t bytes, void** host_ptr); /* page-locked host memory */
int hfpf_host_free(hfpf_handle* h, void* host_ptr);

/* Same path for frames already resident in HBM: n_frames frames, frame f at dev_base + f*frame_stride,
 * poses = n_frames*12 f64 in HOST memory, frame_ids = n_frames ids in HOST memory or NULL (auto).
 * One launch covers the whole batch; asynchronous on the engine's stream. */
int hfpf_integrate_device(hfpf_handle* h, const void* dev_base, uint32_t n_frames, uint64_t frame_stride,
                          uint32_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                          uint32_t off_rgb, const double* poses, const uint32_t* frame_ids);

/* OccupancyGrid::state_changed (grid.hpp:110; read at node.cpp:306). Returns 0/1, or a negative status. */
int hfpf_is_dirty(hfpf_handle* h);
/* OccupancyGrid::updateThicknessVectors<N,K> (grid.hpp:311-454; call sites node.cpp:311,317).
 * The result is that of the reference walking its candidates in canonical ascending (x,y,z) order: "the last registrant wins" on
 * an unoccupied cell (grid.hpp:443-449) goes to the candidate with the largest (x,y,z) key of the pass.  The records themselves
 * are numbered in Z-order of their cells inside a pass (neighbouring records = neighbouring cells); DESIGN.md section 3.
 * Deferred errors: a SMALL pass (fewer than 2^21 / (2K+1) candidate cells) is enqueued to its end and HFPF_OK returned without
 * waiting for it; a pool that overflows inside such a pass (HFPF_ERR_CAPACITY: normal records, registrations, dependant table)
 * is returned by the next call that reads the counters back (this one included, on its next invocation), and the handle then
 * refuses work until hfpf_clear.  Integrate calls enqueued in between run on the tables as the failed pass left them; their
 * results are discarded with the handle's state at hfpf_clear.  HFPF_CLEAN_NOWAIT=0 makes every pass wait and report itself. */
int hfpf_clean(hfpf_handle* h);

/* OccupancyGrid::downloadData (grid.hpp:456-488; call site node.cpp:398) split in two: the ordered
 * extract (rows in lexicographic x,y,z order, engine-owned buffer) and the two file writers. */
int hfpf_extract(hfpf_handle* h, hfpf_row** rows, uint64_t* n_rows);
void hfpf_free_rows(hfpf_row* rows);
/* The alternate extractors the reference keeps behind `#if 0` (node.cpp:399-437) as options of the same device-side scan:
 *   downloadHQ(cloud, threshold)   grid.hpp:546-575   min_count = threshold (rows with count < threshold are dropped ON THE
 *                                                      DEVICE, before the sort's compaction: they never reach the host), paint_white = 1
 *   downloadClassified(cloud)      grid.hpp:512-544   classify_threshold = kGoodPointsThreshold = 100 (grid.hpp:34): count > 100
 *                                                      -> red (g = b = 0), else white
 *   download(XYZRGB / XYZRGBNormal) grid.hpp:491-510,577-601  all defaults (rgb stays 0 as a default-constructed PCL point)
 * Row order is the same lexicographic (x,y,z) order.  opts == NULL is hfpf_extract. */
typedef struct hfpf_extract_opts {
    uint32_t struct_size;       /* = sizeof(hfpf_extract_opts) */
    int32_t classify_threshold; /* < 0: off */
    double min_count;           /* 0: keep all */
    int32_t paint_white;        /* 1: rgb = 0x00FFFFFF */
    int32_t reserved0;
} hfpf_extract_opts;
int hfpf_extract_filtered(hfpf_handle* h, const hfpf_extract_opts* opts, hfpf_row** rows, uint64_t* n_rows);
/* <directory_name>/test_cloud.pcd (node.cpp:395): PCD v0.7 ASCII, FIELDS x y z rgb normal_x normal_y normal_z curvature */
int hfpf_write_pcd(const hfpf_row* rows, uint64_t n_rows, const char* path);
/* <directory_name>/meta.csv (node.cpp:396) with the header string of grid.hpp:462 */
int hfpf_write_meta_csv(const hfpf_row* rows, uint64_t n_rows, const char* path);

/* The alternate extractors the reference keeps behind `#if 0` (node.cpp:399-437): download(XYZRGB), downloadHQ(threshold)
 * and downloadClassified (grid.hpp:491-575), as one PointXYZRGB writer over extracted rows.  min_count: rows with
 * count < min_count are skipped (downloadHQ; 0 keeps all).  white != 0 paints r=g=b=255 as those functions do.
 * classify_threshold >= 0 paints rows with count > threshold red (downloadClassified uses kGoodPointsThreshold = 100). */
int hfpf_write_pcd_xyzrgb(const hfpf_row* rows, uint64_t n_rows, const char* path, uint32_t min_count,
                          int32_t classify_threshold, int32_t white);
/* test_cloud.pcd with DATA binary (same fields; for the 10k-frame configs where ASCII formatting dominates). */
int hfpf_write_pcd_binary(const hfpf_row* rows, uint64_t n_rows, const char* path);

/* OccupancyGrid::clearVoxels (grid.hpp:167-183; call site node.cpp:438).  Full reset (documented
 * deviation: the reference leaves stale keys and dependants-only blocks behind).  Waits for the work enqueued on the handle
 * (it reads how many bricks and records the session used and resets that much of the tables, not their whole capacity);
 * frames of hfpf_integrate still waiting for their launch are dropped. */
int hfpf_clear(hfpf_handle* h);

/* Wait for all queued work of this handle; surfaces deferred capacity/HIP errors. */
int hfpf_sync(hfpf_handle* h);
int hfpf_get_counters(hfpf_handle* h, hfpf_counters* out);

/* Occupied voxel triplets in ascending (x,y,z) order (test/diagnostic: bit-exact occupancy parity).
 * xyz may be NULL to query the count; at most cap triplets are written. */
int hfpf_get_occupied(hfpf_handle* h, int32_t* xyz, uint64_t cap, uint64_t* n_out);

/* ---- harness helpers (device staging without any framework) ---- */
int hfpf_device_alloc(hfpf_handle* h, uint64_t bytes, void** dev_ptr);
int hfpf_device_free(hfpf_handle* h, void* dev_ptr);
int hfpf_device_upload(hfpf_handle* h, void* dev_dst, const void* host_src, uint64_t bytes);


/* ---- multi-GPU (one handle per GPU, one process per GPU) ----------------------------------------------
 * The reference is single-process; sharding follows SURVEY.md 8(e): frames (or cameras) are dealt to ranks, each
 * rank integrates only its own frames with GLOBAL frame ids (hfpf_integrate_device frame_ids), and every rank calls
 * hfpf_clean / hfpf_extract at the same points of the schedule (they become collectives).  At a clean the ranks
 * exchange the cells they occupied since the last clean (key, smallest frame id, its viewpoint); normals and
 * registrations are then computed redundantly and identically on every rank; buffers and statistic sums stay
 * private and are added (exact int64) at extract.  The result is bit-identical to one GPU fusing all frames.
 *
 * Transport 1: RCCL.  Rank 0 calls hfpf_dist_unique_id, the launcher broadcasts the 128 bytes (bench.py uses
 * torch.distributed/gloo, a C++ host would use MPI or a file), every rank calls hfpf_dist_init. */
int hfpf_dist_unique_id(void* id128);
int hfpf_dist_init(hfpf_handle* h, int rank, int world, const void* id128);
/* Rank and rank count as the RCCL communicator reports them (ncclCommUserRank / ncclCommCount); 0 and 1 without one. */
int hfpf_dist_info(hfpf_handle* h, int32_t* rank, int32_t* world);
/* Drop the communicator again (e.g. when not every rank managed to create one); clean/extract become local calls. */
int hfpf_dist_disable(hfpf_handle* h);
/* Transport 2: bring your own.  The same exchange as explicit steps on device buffers (also how tests run several
 * virtual ranks on one GPU): export -> move the 16-byte records (ABI 6; 32 bytes until ABI 5: one record per newly occupied cell --
 * key, smallest frame id -- and two per frame integrated since the last exchange -- its viewpoint, which used to ride on every
 * cell record) -> import into every other rank -> hfpf_clean;
 * at the end add the ranks' hfpf_stats_export words and hand the totals to hfpf_extract_with_stats.
 * Exported pointers are device memory owned by the handle, valid until its next mutating call. */
int hfpf_epoch_export(hfpf_handle* h, const void** dev_records, uint64_t* n_records);
int hfpf_epoch_import(hfpf_handle* h, const void* dev_records, uint64_t n_records);
/* Since ABI 3 the optional colour sums are words 5-7 of the same 8-word records: *dev_cwords is NULL, *n_cwords 0, and
 * hfpf_extract_with_stats ignores dev_cwords (both parameters are kept so that ABI-2 callers still link). */
/* The receive side of a padded all-gather, as the RCCL path runs it: `world` slices of slice_stride_bytes each, slice r holding
 * counts[r] 32-byte records (padding behind them is never read); every slice but my_rank's is imported.  Unequal and zero
 * counts are the normal case. */
int hfpf_epoch_import_gathered(hfpf_handle* h, const void* dev_buffer, uint64_t slice_stride_bytes, int32_t world, int32_t my_rank,
                               const uint64_t* counts);
int hfpf_stats_export(hfpf_handle* h, const void** dev_words, uint64_t* n_words, const void** dev_cwords, uint64_t* n_cwords);
int hfpf_extract_with_stats(hfpf_handle* h, const void* dev_words, const void* dev_cwords, hfpf_row** rows, uint64_t* n_rows);
int hfpf_device_download(hfpf_handle* h, void* host_dst, const void* dev_src, uint64_t bytes);
int hfpf_device_copy(hfpf_handle* h, void* dev_dst, const void* dev_src, uint64_t bytes); /* device to device, synchronous */

/* ---- measurement: HIP-event timing of the engine's own kernels on the engine's stream ----
 * kernel ids: 0 = integrate calls (bin plan + k_integrate + k_update_cells + k_buffer), 1 = whole clean passes (first to last
 * kernel of hfpf_clean, host read-backs included).  enable = 2 additionally brackets the kernels of every integrate call:
 * 2 = k_integrate, 3 = k_update_cells / k_update, 4 = k_buffer (three more event records per call: use it for a breakdown
 * pass, not for the headline timing).  total_ms / launches accumulate since enable. */
int hfpf_kernel_timing(hfpf_handle* h, int enable);
int hfpf_get_kernel_time(hfpf_handle* h, int kernel_id, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* HFPF_H */
