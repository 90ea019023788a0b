#!/usr/bin/env python3
"""GPU box: what decides multi-GPU scaling of the clean pass, measured with VIRTUAL ranks on ONE GPU (a projection -- no xGMI,
no RCCL, the ranks' kernels take turns on one device).

For world sizes 1, 2, 4, 8: one handle per rank, every rank fuses its OWN camera stream (BASELINE configs[3]: distinct frame and
pose seeds, 640x480, shared 2 m^3 box @ 1 mm) in epochs of 150 frames; at the end of an epoch the ranks exchange the cells they
occupied (device-to-device, the protocol of the RCCL path) and every rank runs the clean pass for the UNION of all cameras'
cells -- gate, plane fit, registration, dependant table -- plus the replay of its own buffered points.  Reported per world size and
pass: the time of one rank's clean pass (HIP events on its stream; passes run one after the other, HFPF_CLEAN_NOWAIT=0), the
records a rank receives, the bytes of the statistics all-reduce an extract would move.

usage: python3 tools/virtual_ranks.py [--epochs 2] [--frames 150] [--worlds 1,2,4,8] > gpurun_out/virtual_ranks.md
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "high-fidelity-pointcloud-fusion_amd", "python"))
os.environ.setdefault("HFPF_CLEAN_NOWAIT", "0")  # every pass waits for itself: the event pair around it then times this rank alone

import numpy as np  # noqa: E402

import hfpf  # noqa: E402
import hfpf_dist  # noqa: E402
import hfpf_synth as S  # noqa: E402

W, H = 640, 480
NPTS = W * H
BBOX = (-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)
RES = 0.001


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def run(world, epochs, frames):
    fb = NPTS * 16
    n_frames = epochs * frames
    grids, devs, poses = [], [], []
    buf = np.empty(fb, np.uint8)
    for r in range(world):
        g = hfpf.OccupancyGrid(resolution=RES, bbox=BBOX, max_bricks=300000, max_log_points=min(n_frames * NPTS, (1 << 31) - 64), max_normals=12 << 20,
                               max_frames=n_frames * world + 16, frame_width=W, max_call_points=frames * NPTS)
        seed, pose_seed = 0xF051 + 7919 * r, 0x5E3 + 104729 * r
        p = np.stack([S.pose(pose_seed, f, 30.0, 0.05) for f in range(n_frames)]).reshape(n_frames, 12)
        d = g.device_alloc(n_frames * fb)
        for f in range(n_frames):
            S.frame(seed, f, W, H, p[f].reshape(3, 4), out=buf)
            g.device_upload(d + f * fb, buf)
        grids.append(g), devs.append(d), poses.append(p)
    vr = hfpf_dist.LocalVirtualRanks(grids)
    rows = []
    for e in range(epochs):
        for r, g in enumerate(grids):
            ids = hfpf_dist.shard_frame_ids(frames, r, world, start=e * frames)
            g.integrate_device(devs[r] + e * frames * fb, frames, fb, NPTS, poses[r][e * frames:(e + 1) * frames], frame_ids=ids)
            g.sync()
        # the exchange of LocalVirtualRanks.clean_all, with the counts kept
        exports = [g.epoch_export() for g in grids]
        t0 = time.perf_counter()
        for i, g in enumerate(grids):
            for j, (ptr, n) in enumerate(exports):
                if i != j and n:
                    g.epoch_import(ptr, n)
        t_import = time.perf_counter() - t0
        clean_ms = []
        for g in grids:
            g.kernel_timing(True)
            g.clean()
            g.sync()
            ms, n = g.kernel_time(1)
            g.kernel_timing(False)
            clean_ms.append(ms)
        ctr = [g.counters() for g in grids]
        sent = [n for _, n in exports]
        recv = [sum(sent) - s for s in sent]
        rows.append(dict(world=world, epoch=e, clean_ms_mean=float(np.mean(clean_ms)), clean_ms_max=float(np.max(clean_ms)),
                         sent_mean=float(np.mean(sent)), recv_mean=float(np.mean(recv)), import_ms_per_rank=t_import * 1e3 / world,
                         normals=ctr[0]["voxels_with_normal"], occupied=ctr[0]["voxels_occupied"],
                         registrations=ctr[0]["registrations"], bricks=ctr[0]["bricks_allocated"]))
        assert len({c["voxels_with_normal"] for c in ctr}) == 1, "ranks disagree on the record count"
        log("world %d epoch %d: clean %.3f ms per rank (max %.3f), %.0f records received per rank, %d normals" % (
            world, e, rows[-1]["clean_ms_mean"], rows[-1]["clean_ms_max"], rows[-1]["recv_mean"], rows[-1]["normals"]))
    for g, d in zip(grids, devs):
        g.device_free(d)
        g.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--worlds", default="1,2,4,8")
    a = ap.parse_args()
    S.build()
    allrows = []
    for w in [int(x) for x in a.worlds.split(",")]:
        allrows += run(w, a.epochs, a.frames)
    base = {r["epoch"]: r for r in allrows if r["world"] == 1}
    print("# Clean pass against the number of cameras: virtual ranks on one MI355X (a PROJECTION: no RCCL, no xGMI)\n")
    print("`python3 tools/virtual_ranks.py --epochs %d --frames %d` -- BASELINE configs[3] cameras (640x480, one per rank, distinct frame and pose seeds),"
          " shared 2 m^3 box @ 1 mm, epochs of %d frames per camera; every rank imports the other ranks' newly occupied cells and runs the clean pass"
          " for the union.  `clean ms` = HIP events around one rank's pass on its own stream (HFPF_CLEAN_NOWAIT=0, ranks one after the other);"
          " `exchange` = 16-byte records a rank receives in the all-gather (32-byte until ABI 5); `all-reduce` = (records + 1) x 64 B, what an extract at that point would reduce.\n" % (
              a.epochs, a.frames, a.frames))
    print("| ranks | pass | clean ms per rank (mean / max) | vs 1 rank | records received per rank | exchange MB per rank | import ms | normal records | statistics all-reduce MB | bricks |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in allrows:
        b = base.get(r["epoch"])
        rel = "%.2fx" % (r["clean_ms_mean"] / b["clean_ms_mean"]) if b and b["clean_ms_mean"] > 0 else "-"
        print("| %d | %d | %.3f / %.3f | %s | %.0f | %.2f | %.3f | %d | %.1f | %d |" % (
            r["world"], r["epoch"] + 1, r["clean_ms_mean"], r["clean_ms_max"], rel, r["recv_mean"], r["recv_mean"] * hfpf_dist.EPOCH_REC_BYTES / 1e6, r["import_ms_per_rank"],
            r["normals"], (r["normals"] + 1) * 64 / 1e6, r["bricks"]))


if __name__ == "__main__":
    main()
