#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpts/s fused into a 1 mm voxel grid (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one 50-frame slice of the synthetic 640x480 organised XYZRGB stream (15,360,000 points, published
height=1) pushed through the hot path: decode + z-clip + SE(3) + bbox clip + voxel insert/append + dependant updates.
The stream is handed to the engine one clean epoch at a time (3 steps = 150 frames per hfpf_integrate_device call),
with a clean pass after every epoch and a final clean, all inside the timed region -- the cadence of the reference's
cleanGrid thread (sleep(5) at ~30 Hz, node.cpp:323).  The default K = 20 steps is therefore exactly configs[1] of
BASELINE.json: the 1000-frame stream with random SE(3) poses into a 1 m^3 bbox @ 1 mm on one MI355X, 7 clean passes.
W warm-up steps run the same schedule untimed; the grid is cleared before every timed pass.

Timing protocol: frames are pre-staged in HBM before the clock starts; the K-step stream is timed
--repeats times (clear between), each pass bracketed by barrier + device sync, max over ranks; `value` is the MEDIAN
pass, min/max are reported beside it.  Extract is timed separately (`extract_s`), so is the host-buffer entry point
(`host_path_mpts`, PCIe inclusive, never `value`).

For N>1 the driver launches one rank per GPU with torch.distributed.run; every rank fuses its own camera stream into
the shared grid (weak scaling) and the engine merges occupancy / statistics over RCCL.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "high-fidelity-pointcloud-fusion_amd")
sys.path.insert(0, os.path.join(PKG, "python"))

import numpy as np  # noqa: E402

import hfpf  # noqa: E402
import hfpf_synth as S  # noqa: E402

POINT_STEP = 16
# BASELINE.json configs the bench can run on one GPU (or one stream per GPU).  The default is configs[1], the
# configuration the metric is quoted on; the others are selectable for documentation runs, never the driver's line.
# frames_per_step x default --steps = the frame count BASELINE.json names for the config.
WORKLOADS = {
    "c1": dict(name="configs[1]", W=640, H=480, bbox=(-0.5, 0.5, -0.5, 0.5, 0.0, 1.0), res=0.001, frames_per_step=50, steps=20,
               desc="%d-frame synthetic 640x480 stream, random SE(3) poses, 1 m^3 bbox @ 1 mm"),
    "c3": dict(name="configs[2]", W=2048, H=1536, bbox=(-1.0, 1.0, -0.5, 0.5, 0.0, 1.0), res=0.0005, frames_per_step=5, steps=20,
               desc="%d-frame synthetic 2048x1536 stream, random SE(3) poses, 2 m^3 bbox @ 0.5 mm"),
    "c4": dict(name="configs[3]", W=640, H=480, bbox=(-1.0, 1.0, -0.5, 0.5, 0.0, 1.0), res=0.001, frames_per_step=50, steps=20,
               desc="%d-frame synthetic 640x480 stream per camera (one camera per GPU, distinct pose seeds), shared 2 m^3 bbox @ 1 mm (1999x999x999 cells)"),
    "c5": dict(name="configs[4] grid", W=640, H=480, bbox=(-1.25, 1.25, -1.0, 1.0, 0.0, 2.0), res=0.001, frames_per_step=50, steps=20,
               desc="%d-frame synthetic 640x480 stream, random SE(3) poses, 10 m^3 bbox @ 1 mm (2499x1999x1999 cells)"),
}
W = H = NPTS = 0
BBOX = RES = None
ALGO_BYTES_PER_POINT = 32  # 16 B point read + 16 B voxel-record touch
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBPS = 6290.0  # MI355X_MICROARCH.md: float4 copy
# committed `rocprofv3 --pmc` summaries (tools/pmc_summary.py) per workload; attached only when taken on this build's kernel sources
PMC_JSON = {"c1": os.path.join(ROOT, "profiles", "r04_pmc_hot_path.json"), "c3": os.path.join(ROOT, "profiles", "r04_c3_pmc_hot_path.json")}
SIMDS, CLOCK_HZ = 1024, 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz; one VALU wave-instruction holds a SIMD's issue for 4 cycles
KERNEL_SOURCES = ("kernels.hpp", "tables.hpp", "stats.hpp", "geometry.hpp", "det_math.hpp", "hfpf.hip")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha():
    """Identity of the kernels this run executes: hash of the csrc sources libhfpf.so is built from.  The PMC summary under
    profiles/ records the same hash, so stale counters are never attached to a newer build."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(PKG, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def make_stream(pose_seed, n_frames):
    return np.stack([S.pose(pose_seed, f, 30.0, 0.05) for f in range(n_frames)]).reshape(n_frames, 12)


def cpu_baseline(poses, seed, n_sample, variant="faithful"):
    """The CPU oracle (kind "port": the reference itself cannot be built here) timed on the first n_sample frames of the
    same stream: a clean half-way (so the second half exercises the dependant updates of grid.hpp:244-277 like the
    steady state of the full run) and a final clean.  Variants:
      faithful   one thread (the reference's OpenMP pragmas are commented out), the reference's storage: a dense array of 16-byte
                 voxels over the whole box (grid.hpp:626; 16 GB of address space at 1 m^3 @ 1 mm, touched sparsely) and
                 buffer.reserve(1000) per new voxel (grid.hpp:228); falls back to the hash map when the array cannot be allocated
      sparse     one thread, hash map of touched cells, no reserve (what the 10^10-cell configs need)
      all_cores  OpenMP over points and clean candidates, voxel store sharded 256 ways (oracle capture_mt / clean_mt)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # only the cpu_baseline leg touches the oracle

    cores = 1
    all_cores = variant == "all_cores"
    if all_cores:  # the GPU box gives one GPU a share of 16 host cores
        cores = oracle.set_threads(min(len(os.sched_getaffinity(0)), int(os.environ.get("HFPF_CPU_THREADS", "16"))))
    kw = dict(dense=True, reserve=1000) if variant == "faithful" else {}
    og = oracle.OracleGrid(resolution=RES, bbox=BBOX, **kw)
    storage = "hash map of touched cells, no reserve"
    if variant == "faithful":
        storage = ("dense (dim+1)^3 voxel array + reserve(1000) per voxel, as the reference" if og.is_dense else
                   "hash map of touched cells + reserve(1000) (the dense array could not be allocated)")
    capture, clean = (og.capture_mt, og.clean_mt) if all_cores else (og.capture, og.clean)
    frames = [S.frame(seed, f, W, H, poses[f].reshape(3, 4)) for f in range(n_sample)]
    half = max(1, n_sample // 2)
    t0 = time.perf_counter()
    for f in range(n_sample):
        capture(frames[f], poses[f])
        if f + 1 == half and f + 1 < n_sample:
            clean()
    clean()
    dt = time.perf_counter() - t0
    og.close()
    how = ("OpenMP over points and clean candidates, voxel store sharded 256 ways" if all_cores else "single thread")
    return {"value": round(n_sample * NPTS / dt / 1e6, 4), "unit": "Mpts/s", "cores": cores, "kind": "port",
            "sample": "first %d frames of the same stream, clean after frame %d and at the end, %s, %s (%.1f s)" % (n_sample, half, how, storage, dt)}


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Gloo and RCCL print banners on file descriptor 1 from C code, so keep a private
    handle to the real stdout for that line and point fd 1 (and Python's sys.stdout) at stderr for everything else."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    return real


def load_pmc(src_sha, workload):
    """Measured HBM-side traffic of the integrate hot path from the committed `rocprofv3 --pmc` passes (PMC counters cannot
    be read inside this process).  Returns (dict or None, reason): refuses counters taken on different kernel sources."""
    path = PMC_JSON.get(workload)
    if path is None:
        return None, "no PMC passes are kept for workload %s" % workload
    if not os.path.exists(path):
        return None, "no %s (collect with tools/collect_profiles.sh / tools/collect_c3.sh + tools/pmc_summary.py)" % os.path.relpath(path, ROOT)
    with open(path) as f:
        pmc = json.load(f)
    if pmc.get("source_sha") != src_sha:
        return None, "%s was collected on kernel sources %s, this build is %s: not measured for this build" % (
            os.path.relpath(path, ROOT), pmc.get("source_sha"), src_sha)
    return pmc, "%s (FETCH_SIZE with the gfx950 wide-read correction + WRITE_SIZE of the integrate kernels, separate --pmc passes on this build)" % os.path.relpath(path, ROOT)


def main():
    json_out = claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c1")
    ap.add_argument("--steps", type=int, default=None, help="timed steps; one step = --frames-per-step frames (default 20 = the config's whole stream)")
    ap.add_argument("--warmup", type=int, default=5, help="untimed warm-up steps")
    ap.add_argument("--frames-per-step", type=int, default=None, help="frames in one step (default 50 for 640x480 streams)")
    ap.add_argument("--clean-every-steps", type=int, default=3, help="clean pass every this many steps (3 x 50 frames = 150 frames ~ 5 s at 30 Hz) + final clean")
    ap.add_argument("--repeats", type=int, default=10, help="timed passes over the K-step stream (median reported; SURVEY 8(d): n >= 10)")
    ap.add_argument("--frames-per-call", type=int, default=0, help="frames handed to one hfpf_integrate_device call (0 = one clean epoch)")
    ap.add_argument("--cpu-sample", type=int, default=48, help="frames timed on the CPU oracle, sparse and all-cores variants (0 = skip all CPU legs)")
    ap.add_argument("--cpu-sample-faithful", type=int, default=24, help="frames timed on the faithful variant (dense voxel array + reserve(1000); ~0.4 s per frame on the GPU box)")
    ap.add_argument("--write-dir", default=None, help="also time writing test_cloud.pcd (ASCII + binary) and meta.csv there")
    ap.add_argument("--host-path-frames", type=int, default=200, help="frames also pushed through the host-buffer entry point (0 = skip)")
    ap.add_argument("--allow-host-staged", action="store_true", help="rehearsals with several ranks on ONE GPU: fall back to the gloo host-staged transport when RCCL cannot form a communicator")
    args = ap.parse_args()
    global W, H, NPTS, BBOX, RES
    wl = WORKLOADS[args.workload]
    W, H, BBOX, RES = wl["W"], wl["H"], wl["bbox"], wl["res"]
    NPTS = W * H
    K = args.steps if args.steps is not None else wl["steps"]
    Wm = max(0, args.warmup)
    fps = args.frames_per_step or wl["frames_per_step"]
    if K < 1 or fps < 1 or args.repeats < 1:
        raise SystemExit("--steps, --frames-per-step and --repeats must be >= 1")
    n_frames = K * fps
    clean_every = args.clean_every_steps * fps  # frames
    call_frames = args.frames_per_call if args.frames_per_call > 0 else (clean_every if clean_every else n_frames)
    call_frames = max(1, min(call_frames, n_frames, int(os.environ.get("HFPF_BENCH_CALL_POINTS", 1 << 31)) // NPTS))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist  # rank bootstrap + barriers only; the data path is the engine's own RCCL
        dist.init_process_group("gloo", rank=rank, world_size=world)

    seed = 0xF051 + 7919 * rank  # one camera stream per GPU (BASELINE configs[3] style sharding)
    pose_seed = 0x5E3 + 104729 * rank
    n_gen = max(n_frames, Wm * fps)
    poses = make_stream(pose_seed, n_gen)

    device, n_dev = local_rank, 1
    if world > 1:
        import torch
        n_dev = torch.cuda.device_count()  # does not initialise the GPU
        if n_dev > 0:
            device = local_rank % n_dev  # rehearsals with more ranks than GPUs share devices
    grid = hfpf.OccupancyGrid(resolution=RES, bbox=BBOX, device=device, max_bricks=400000,
                              max_log_points=min(max(n_gen, 64) * NPTS, (1 << 31) - 64), max_normals=24 << 20,
                              max_frames=max(n_gen * max(world, 1) + 16, 4096),
                              frame_width=int(os.environ.get("HFPF_FRAME_WIDTH", W)),  # organised W x H frames: 16x16-pixel tiles
                              max_call_points=call_frames * NPTS)  # per-call bins sized at create, like the other pools
    transport = "none"
    if world > 1:
        import hfpf_dist
        ok, why = hfpf_dist.init_rccl(grid, dist, hfpf)  # data path = the engine's own RCCL collectives over xGMI
        if ok:
            transport = "rccl, %d ranks" % grid.dist_world()
        elif args.allow_host_staged or world > n_dev:
            # several ranks share a GPU (a rehearsal on a 1-GPU box): RCCL cannot put two ranks on one device
            log("rank %d: RCCL bootstrap failed (%s); ranks share a device, using the host-staged transport" % (rank, why))
            grid.attach_transport(hfpf_dist.HostStagedTransport(dist))
            transport = "host-staged (gloo), %d ranks" % world
        else:  # one GPU per rank and still no communicator: a scaling number over gloo would be meaningless
            log("rank %d: RCCL bootstrap failed (%s) with %d ranks on %d devices: refusing to benchmark the fallback transport" % (rank, why, world, n_dev))
            sys.exit(3)

    # ---- stage frames in HBM (not timed) ----
    t_gen = time.perf_counter()
    frame_bytes = NPTS * POINT_STEP
    dev = grid.device_alloc(n_gen * frame_bytes)
    buf = np.empty(frame_bytes, dtype=np.uint8)
    host_frames = []
    n_host = min(args.host_path_frames, n_gen) if (rank == 0 and world == 1) else 0
    pinned = grid.host_alloc(n_host * frame_bytes) if n_host else None  # the same frames in page-locked memory (zero-copy entry point)
    for f in range(n_gen):
        S.frame(seed, f, W, H, poses[f].reshape(3, 4), out=buf)
        grid.device_upload(dev + f * frame_bytes, buf)
        if f < n_host:
            host_frames.append(buf.copy())
            pinned[f * frame_bytes:(f + 1) * frame_bytes] = buf
    log("rank %d: staged %d frames (%.2f GB) in %.1f s" % (rank, n_gen, n_gen * frame_bytes / 1e9, time.perf_counter() - t_gen))

    def run_stream(nf, start=0, final_clean=True):
        """frames [start, nf): one integrate call per clean epoch (or --frames-per-call), clean after every epoch, final clean."""
        done = start
        while done < nf:
            nxt = nf
            if clean_every:
                nxt = min(nxt, (done // clean_every + 1) * clean_every)
            b = min(call_frames, nxt - done)
            ids = ((np.arange(done, done + b, dtype=np.int64) * world) + rank).astype(np.uint32)  # global frame ids
            grid.integrate_device(dev + done * frame_bytes, b, frame_bytes, NPTS, poses[done:done + b], frame_ids=ids)
            done += b
            if clean_every and done % clean_every == 0 and done < nf:
                grid.clean()  # synchronises (reads device counters)
        if final_clean:
            grid.clean()

    # ---- warmup (untimed) ----
    if Wm > 0:
        run_stream(Wm * fps)
        grid.sync()

    # ---- timed region: exactly K steps per pass, `repeats` passes ----
    grid.kernel_timing(True)
    elapsed_all = []
    for rep in range(args.repeats):
        grid.clear()
        grid.sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        run_stream(n_frames)
        grid.sync()
        t1 = time.perf_counter()
        if dist is not None:
            dist.barrier()
        e = t1 - t0
        if dist is not None:
            import torch
            tt = torch.tensor([e], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt.item())
        elapsed_all.append(e)
    elapsed = statistics.median(elapsed_all)
    k_ms, k_launches = grid.kernel_time(0)
    clean_ms, clean_passes = grid.kernel_time(1)
    grid.kernel_timing(False)
    ctr = grid.counters()  # of the last pass (the grid is cleared between passes)

    # ---- one more pass, untimed, with an event pair around every kernel of every integrate call: the per-kernel breakdown ----
    per_kernel = None
    if True:  # every rank: the clean passes inside are collectives
        grid.clear()
        grid.sync()
        names = (("k_integrate", 2), ("k_update_cells", 3), ("k_buffer", 4))
        first = min(clean_every, n_frames) if clean_every else n_frames
        grid.kernel_timing(2)
        run_stream(first, final_clean=False)  # the first epoch (everything is buffered, no dependant exists yet) ...
        if first < n_frames:
            grid.clean()
        grid.sync()
        pk_first = {name: grid.kernel_time(kid) for name, kid in names}
        grid.kernel_timing(2)  # (restarts the accumulators)
        if first < n_frames:
            run_stream(n_frames, start=first)  # ... and the steady state behind the first clean pass, apart
        else:
            grid.clean()
        grid.sync()
        per_kernel_steady = {name: grid.kernel_time(kid) for name, kid in names}
        per_kernel = {name: (pk_first[name][0] + per_kernel_steady[name][0], pk_first[name][1] + per_kernel_steady[name][1]) for name, _ in names}
        grid.kernel_timing(False)

    # ---- extract (timed separately) ----
    t2 = time.perf_counter()
    rows = grid.extract()
    extract_s = time.perf_counter() - t2
    write_times = None
    if rank == 0 and args.write_dir:
        os.makedirs(args.write_dir, exist_ok=True)
        tw0 = time.perf_counter()
        hfpf.write_pcd(rows, os.path.join(args.write_dir, "test_cloud.pcd"))
        tw1 = time.perf_counter()
        hfpf.write_meta_csv(rows, os.path.join(args.write_dir, "meta.csv"))
        tw2 = time.perf_counter()
        hfpf.write_pcd_binary(rows, os.path.join(args.write_dir, "test_cloud_binary.pcd"))
        tw3 = time.perf_counter()
        write_times = {"pcd_ascii_s": round(tw1 - tw0, 4), "meta_csv_s": round(tw2 - tw1, 4), "pcd_binary_s": round(tw3 - tw2, 4)}

    # ---- host-buffer entry point (PCIe-inclusive), informational ----
    host_mpts = host_pinned_mpts = host_link_gbps = None
    if host_frames:
        # (a) hfpf_integrate: caller's pageable buffer -> pinned bounce copy -> upload on the copy stream -> kernels (what a ROS
        #     callback with a sensor_msgs buffer gets); (b) hfpf_integrate_pinned: upload straight from page-locked memory
        grid.clear()
        grid.sync()
        th = time.perf_counter()
        for f, hb in enumerate(host_frames):
            grid.integrate(hb, poses[f])
        grid.sync()
        host_mpts = len(host_frames) * NPTS / (time.perf_counter() - th) / 1e6
        grid.clear()
        grid.sync()
        th = time.perf_counter()
        for f in range(len(host_frames)):
            grid.integrate_pinned(pinned[f * frame_bytes:(f + 1) * frame_bytes], poses[f])
        grid.sync()
        host_pinned_mpts = len(host_frames) * NPTS / (time.perf_counter() - th) / 1e6
        # what the link itself gives this process: the same page-locked frames as ONE hipMemcpy (best of three), no kernels
        link_bytes = len(host_frames) * frame_bytes
        dev_scratch = grid.device_alloc(link_bytes)
        best = None
        for _ in range(3):
            grid.sync()
            th = time.perf_counter()
            grid.device_upload(dev_scratch, pinned[:link_bytes])
            grid.sync()
            dt = time.perf_counter() - th
            best = dt if best is None else min(best, dt)
        host_link_gbps = link_bytes / best / 1e9
        grid.device_free(dev_scratch)
        grid.host_free(pinned)

    total_pts = n_frames * NPTS * world
    value = total_pts / elapsed / 1e6
    if rank == 0:
        R = args.repeats
        warnings = []
        pts_per_launch = (R * n_frames * NPTS) / max(k_launches, 1)
        avg_launch_s = (k_ms / 1e3) / max(k_launches, 1)
        achieved = ALGO_BYTES_PER_POINT * pts_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        src_sha = kernel_source_sha()
        pmc, traffic_src = load_pmc(src_sha, args.workload)
        traffic = atomic_req = None
        if pmc is not None:
            launches_per_pass = k_launches / R
            n_first = min(launches_per_pass, max(1, clean_every // call_frames) if clean_every else launches_per_pass)
            w_first = n_first / launches_per_pass
            bpp = (w_first * pmc["first_epoch_buffer_only"]["traffic_bytes_per_point"] +
                   (1 - w_first) * pmc["steady_state_after_first_clean"]["traffic_bytes_per_point"])
            traffic = round(bpp * pts_per_launch)
            atomic_req = (w_first * pmc["first_epoch_buffer_only"]["atomic_requests"] +
                          (1 - w_first) * pmc["steady_state_after_first_clean"]["atomic_requests"]) * (pts_per_launch / pmc["points_per_launch"])
        # Per-kernel view of the integrate call (the breakdown pass above; clean epochs of the same size, so per-call means compare).
        # Algorithmic bytes per kernel: k_integrate = the 32 B/pt of SURVEY 8(d); k_update_cells = SURVEY 8(d)'s extended term,
        # D x (12 B normal read + 2 x 28 B record update) per surviving point with D = measured pairs per surviving point;
        # k_buffer = 16 B read + 16 B append per buffered point.  Traffic-based fractions come from the committed PMC passes.
        kernels = None
        if per_kernel is not None:
            calls = max(per_kernel["k_integrate"][1], 1)
            pairs_per_call = ctr["dep_pairs_tested"] / calls
            d_bar = ctr["dep_pairs_tested"] / max(ctr["points_in_bbox"], 1)
            # bytes a kernel cannot avoid moving through HBM per call it runs in: k_integrate reads the frame and parks the survivors
            # (16 + 16 B/pt: the 32 B of SURVEY 8(d)); k_update_cells reads the parked points back (16 B each; the dependants'
            # records live in LDS and leave once per brick); k_buffer copies the buffered points into the log (16 + 16 B)
            surv = ctr["points_in_bbox"] / max(ctr["points_presented"], 1)
            algo = {"k_integrate": ALGO_BYTES_PER_POINT * (n_frames * NPTS) / calls,
                    "k_update_cells": 16.0 * surv * (n_frames * NPTS) / calls,
                    "k_buffer": 32.0 * ctr["points_buffered"] / max(per_kernel["k_buffer"][1], 1)}
            kernels = {}
            for name, (ms, n) in per_kernel.items():
                avg_s = ms / 1e3 / max(n, 1)
                k = {"avg_ms_per_call": round(avg_s * 1e3, 5), "calls": int(n),
                     "algorithmic_GBps": round(algo[name] / avg_s / 1e9, 2) if avg_s > 0 else None,
                     "algorithmic_frac": round(algo[name] / avg_s / 1e9 / HBM_PEAK_GBPS, 5) if avg_s > 0 else None}
                if pmc is not None and avg_s > 0:
                    st = pmc["steady_state_after_first_clean"].get("k_update" if name == "k_update_cells" else name)
                    if st:
                        tb = st["fetch_bytes_raw"] + st["write_bytes"] + st.get("fetch_correction_bytes", 0.0)
                        k["traffic_bytes_steady_call"] = round(tb)
                        k["traffic_GBps"] = round(tb / avg_s / 1e9, 2)
                        k["traffic_frac"] = round(tb / avg_s / 1e9 / HBM_PEAK_GBPS, 5)
                if per_kernel_steady[name][1]:
                    k["steady_avg_ms_per_call"] = round(per_kernel_steady[name][0] / per_kernel_steady[name][1], 5)  # calls behind the first clean pass
                kernels[name] = k
            upd_s = per_kernel["k_update_cells"][0] / 1e3
            kernels["pairs_per_call"] = round(ctr["dep_pairs_tested"] / max(per_kernel["k_update_cells"][1], 1))
            kernels["k_update_cells"]["Gpairs_per_s"] = round(ctr["dep_pairs_tested"] / upd_s / 1e9, 2) if upd_s > 0 else None
            kernels["pairs_per_surviving_point"] = round(d_bar, 3)
        # compute-side ceiling of the kernel that is not memory-bound (k_update_cells): what one (point, dependant) pair costs in VALU
        # lane-instructions against the pair loop's own minimum (projection, cylinder test, four fixed-point contributions: 48)
        compute = None
        if pmc is not None and pmc.get("sq", {}).get("k_update"):
            sq = pmc["sq"]["k_update"]
            pairs = pmc["sq"].get("pairs_per_launch") or 0
            if pairs and sq.get("SQ_INSTS_VALU"):
                lanes = sq.get("active_lane_fraction")  # None: the passes of this workload did not collect SQ_THREAD_CYCLES_VALU (all 64 lanes counted)
                li = sq["SQ_INSTS_VALU"] * 64.0 * (lanes or 1.0) / pairs
                compute = {"bound": "valu (k_update_cells)", "unit": "VALU lane-instructions per (point, dependant) pair", "achieved": round(li, 1),
                           "minimum": 48, "frac": round(48.0 / li, 4), "active_lane_fraction": round(lanes, 3) if lanes else None,
                           # the same ceiling as a time: what the kernel's VALU wave-instructions alone need to issue against its measured steady time
                           "valu_issue_floor_ms": round(sq["SQ_INSTS_VALU"] * 4.0 / (SIMDS * CLOCK_HZ) * 1e3, 4),
                           "lds_bank_conflict_share": round(sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"], 4) if sq.get("SQ_LDS_IDX_ACTIVE") else None}
        # ... and of k_integrate: the time its VALU wave-instructions alone need to issue (4 cycles each on one of 1024 SIMDs) against
        # the kernel's measured time per steady call
        compute_integrate = None
        if pmc is not None and pmc.get("sq", {}).get("k_integrate", {}).get("SQ_INSTS_VALU") and per_kernel is not None and per_kernel["k_integrate"][1]:
            sqi = pmc["sq"]["k_integrate"]
            floor_ms = sqi["SQ_INSTS_VALU"] * 4.0 / (SIMDS * CLOCK_HZ) * 1e3
            st = per_kernel_steady["k_integrate"]
            meas_ms = st[0] / st[1] if st[1] else per_kernel["k_integrate"][0] / per_kernel["k_integrate"][1]
            compute_integrate = {"bound": "valu issue (k_integrate)", "unit": "ms per steady integrate launch", "valu_issue_floor_ms": round(floor_ms, 4),
                                 "measured_ms": round(meas_ms, 4), "frac": round(floor_ms / meas_ms, 4) if meas_ms > 0 else None,
                                 "valu_wave_instructions": round(sqi["SQ_INSTS_VALU"]), "salu_wave_instructions": round(sqi.get("SQ_INSTS_SALU", 0)),
                                 "active_lane_fraction": round(sqi["active_lane_fraction"], 3) if sqi.get("active_lane_fraction") else None,
                                 "note": "measured_ms = HIP events around k_integrate + k_integrate_overflow of the calls behind the first clean pass (the steady state the counters are means over)"}
        if ctr["dep_pairs_tested"] == 0:
            warnings.append("dep_pairs_tested == 0: the stream never reached the steady state (no dependant updates ran); not the headline configuration")
        out = {
            "metric": "Mpts/s fused into 1 mm voxel grid" if RES == 0.001 else "Mpts/s fused into %g mm voxel grid" % (RES * 1e3),
            "value": round(value, 3),
            "unit": "Mpts/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": round(elapsed * 1e3 / K, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,  # BASELINE.md holds no published number for this metric
            "dtype": "f64",
            "dtype_detail": "f64 SE(3) transform and voxel index, f32 projection/plane-fit geometry, int64 fixed-point statistic sums",
            "data": "synthetic",
            "config": {"workload": "%s: %s; %d steps of %d frames, one integrate call per %d frames, clean every %d frames + final clean" % (
                           wl["name"], wl["desc"] % n_frames, K, fps, call_frames, clean_every),
                       "points_per_step": fps * NPTS, "frames_per_step": fps, "frames_per_call": call_frames, "frames": n_frames,
                       "parallelism": "one camera stream per GPU, %d rank(s), transport %s" % (world, transport)},
            "repeats": R,
            "value_min": round(total_pts / max(elapsed_all) / 1e6, 3),
            "value_max": round(total_pts / min(elapsed_all) / 1e6, 3),
            "pass_s": [round(e, 6) for e in elapsed_all],
            "extract_s": round(extract_s, 5),
            "clean_s": round(clean_ms / 1e3 / R, 5),  # HIP-event time of the clean passes of ONE timed pass (inside the timed region)
            "clean_passes": int(round(clean_passes / R)),
            "rows_extracted": int(len(rows)),
            "write_times": write_times,
            "integrate_kernel_mpts": round(R * n_frames * NPTS / (k_ms / 1e3) / 1e6, 3) if k_ms > 0 else None,
            "host_path_mpts": round(host_mpts, 3) if host_mpts else None,  # PCIe inclusive, one frame per call, never `value`
            "host_path_pinned_mpts": round(host_pinned_mpts, 3) if host_pinned_mpts else None,
            # one hipMemcpy of the same page-locked frames: the link's own rate on this box (the pinned path's ceiling: x / 16 B per point)
            "host_link_GBps": round(host_link_gbps, 2) if host_link_gbps else None,
            "host_path_frames": len(host_frames),
            "counters": {k: int(v) for k, v in ctr.items()},
            "warnings": warnings,
            "roofline": {"bound": "hbm", "kernel": "the integrate call (bin plan + k_integrate + k_update_cells + k_buffer), HIP events on the engine's stream around every call of the timed passes",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "frac_of_measured_copy_peak": round(achieved / HBM_MEASURED_GBPS, 6),
                         # the same 32 B/pt over the driver-timed quantity itself (ms_per_step: integrate calls AND clean passes)
                         "whole_job_frac": round(ALGO_BYTES_PER_POINT * total_pts / world / elapsed / 1e9 / HBM_PEAK_GBPS, 6),
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_source_sha": src_sha,
                         "algorithmic_bytes_per_point": ALGO_BYTES_PER_POINT, "points_per_launch": round(pts_per_launch), "launches": int(k_launches),
                         "avg_launch_ms": round(avg_launch_s * 1e3, 5),
                         # secondary ceiling: memory-side atomic requests, chip-wide 1.3 TB/s / 64 B (MI355X_MICROARCH.md);
                         # request counts from the same PMC passes, null when those are not of this build
                         "kernels": kernels, "compute": compute, "compute_integrate": compute_integrate,
                         "secondary": {"bound": "memory-side atomic requests", "unit": "Greq/s",
                                       "achieved": round(atomic_req / avg_launch_s / 1e9, 3) if atomic_req and avg_launch_s > 0 else None,
                                       "peak": round(1300.0 / 64.0, 3),
                                       "frac": round(atomic_req / avg_launch_s / 1e9 / (1300.0 / 64.0), 4) if atomic_req and avg_launch_s > 0 else None}},
        }
        if args.cpu_sample > 0 and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(poses, seed, max(2, min(args.cpu_sample_faithful, args.cpu_sample, n_gen)), "faithful")
            out["cpu_baseline_sparse"] = cpu_baseline(poses, seed, min(args.cpu_sample, n_gen), "sparse")
            out["cpu_baseline_all_cores"] = cpu_baseline(poses, seed, min(args.cpu_sample, n_gen), "all_cores")
            out["vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)  # informational; vs_baseline stays null (no published number)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    grid.device_free(dev)
    grid.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
