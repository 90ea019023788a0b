#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be built or imported here (C++ needing Eigen/PCL/ROS), so these
fixtures are produced by OUR restatement (oracle/hfpf_oracle.cpp): they pin the oracle against regressions and
give the GPU engine a second, oracle-independent-at-runtime target.  "parity unpinned" still applies.
Inputs are seeded; the scene fixtures store a SHA-256 of the generated frames so a drifting generator is caught.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "high-fidelity-pointcloud-fusion_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle  # noqa: E402
import hfpf_synth as S  # noqa: E402
import scenes  # noqa: E402

LAUNCH_BBOX = (-0.80, 1.80, -1.5, 1.5, 0.0, 1.0)

GRIDS = {  # name -> (resolution, bbox)
    "launch_5mm": (0.005, LAUNCH_BBOX),
    "m1_1mm": (0.001, scenes.BBOX_1M),
    "m2_05mm": (0.0005, (-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)),
}

SCENES = {  # name -> Scene kwargs  (small: the CPU suite must stay fast)
    "s5mm": dict(n_frames=6, W=96, H=72, resolution=0.005, clean_every=3),
    "s1mm": dict(n_frames=5, W=96, H=72, resolution=0.001, fx=615.0, clean_every=2),
    "s1mm_every": dict(n_frames=4, W=80, H=60, resolution=0.001, fx=615.0, clean_every=1),
    "s05mm": dict(n_frames=3, W=96, H=72, resolution=0.0005, fx=1968.0, clean_every=0, bbox=(-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)),
    "identity": dict(n_frames=1, W=128, H=96, resolution=0.005, identity=True),
}


def leaves():
    rng = np.random.default_rng(0x601D)
    out = {}
    for name, (res, bbox) in GRIDS.items():
        g = oracle.OracleGrid(resolution=res, bbox=bbox)
        lo = np.array(bbox[0::2]) - 0.05
        hi = np.array(bbox[1::2]) + 0.05
        pts = rng.uniform(lo, hi, size=(4096, 3)).astype(np.float32)
        # exact boundary values and values one ulp either side of cell boundaries
        edge = np.array([[bbox[0], 0, 0.5], [bbox[1], 0, 0.5], [0, bbox[2], 0.5], [0, bbox[3], 0.5], [0, 0, bbox[4]], [0, 0, bbox[5]]], np.float32)
        r = float(np.float32(res))
        k = rng.integers(1, 150, size=(512, 3))
        cb = (np.array(bbox[0::2]) + k * r).astype(np.float32)
        cb_lo = np.nextafter(cb, np.float32(-10))
        cb_hi = np.nextafter(cb, np.float32(10))
        pts = np.vstack([pts, edge, cb, cb_lo, cb_hi]).astype(np.float32)
        T = S.pose(0x5E3, 3)
        q = oracle.probe_transform(T, pts)
        idx, valid = g.probe_index(q)
        out[name + "_pose"] = T
        out[name + "_pts"] = pts
        out[name + "_q"] = q
        out[name + "_idx"] = idx
        out[name + "_valid"] = valid
        # stencils -> normals
        cells = rng.integers(3, 150, size=(256, 3)).astype(np.int32)
        cells[:8] = [[0, 0, 0], [1, 0, 2], [0, 5, 0], [2, 2, 2], [3, 3, 3], [g.dims[0][0] - 1, 4, 4], [4, g.dims[0][1] - 1, 4], [4, 4, g.dims[0][2] - 1]]
        occ = np.zeros((256, 125), np.uint8)
        vps = rng.uniform(-1, 1, size=(256, 3)).astype(np.float32)
        normals = np.zeros((256, 3), np.float32)
        totals = np.zeros(256, np.int32)
        for i in range(256):
            kind = i % 4
            nn = rng.normal(size=3)
            nn /= np.linalg.norm(nn)
            d = 0
            for a in range(-2, 3):
                for b in range(-2, 3):
                    for c in range(-2, 3):
                        dist = a * nn[0] + b * nn[1] + c * nn[2]
                        if kind == 0:
                            occ[i, d] = abs(dist) < 0.6
                        elif kind == 1:
                            occ[i, d] = abs(dist) < 1.1
                        elif kind == 2:
                            occ[i, d] = (abs(dist) < 0.8) or rng.random() < 0.05
                        else:
                            occ[i, d] = rng.random() < 0.4
                        d += 1
            totals[i], normals[i] = g.probe_normal(int(cells[i, 0]), int(cells[i, 1]), int(cells[i, 2]), occ[i], vps[i])
        out[name + "_cells"] = cells
        out[name + "_occ"] = occ
        out[name + "_vps"] = vps
        out[name + "_normals"] = normals
        out[name + "_totals"] = totals
        g.close()
    # projection / membership near the 1 mm boundary
    n = 4096
    c = rng.uniform(-0.4, 0.4, size=(n, 3)).astype(np.float32)
    nn = rng.normal(size=(n, 3))
    nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    perp = np.cross(nn, rng.normal(size=(n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    rad = np.where(rng.random(n) < 0.5, rng.uniform(0.00099, 0.00101, n), rng.uniform(0, 0.002, n))
    p = (c + perp * rad[:, None] + nn * rng.uniform(-0.01, 0.01, n)[:, None]).astype(np.float32)
    proj, dist = oracle.probe_project(p, c, nn)
    out.update(pj_pts=p, pj_centres=c, pj_normals=nn, pj_proj=proj, pj_dist=dist)
    # deterministic trig
    y = np.abs(rng.normal(size=4096)).astype(np.float32)
    x = rng.normal(size=4096).astype(np.float32)
    th = rng.uniform(0, np.pi / 3, size=4096).astype(np.float32)
    a, _, _ = oracle.probe_trig(y, x)
    _, co, si = oracle.probe_trig(y, th)
    out.update(tr_y=y, tr_x=x, tr_th=th, tr_atan2=a, tr_cos=co, tr_sin=si)
    np.savez_compressed(os.path.join(HERE, "leaves.npz"), **out)


def scene_fixture(name, kw):
    sc = scenes.Scene(**kw)
    h = hashlib.sha256()
    for f in range(sc.n_frames):
        h.update(sc.frame(f).tobytes())
    g = oracle.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    rows = scenes.run(g, sc, "capture")
    occ = g.occupied()
    ctr = g.counters()
    g.close()
    return dict(rows=rows, occupied=occ, poses=np.stack(sc.poses), frames_sha256=np.frombuffer(h.digest(), np.uint8),
                counters=np.array([ctr[k] for k in ("presented", "zclip_pass", "inserted", "occupied", "normals", "buffered")], np.uint64))


def main():
    oracle.build()
    leaves()
    out = {}
    for name, kw in SCENES.items():
        fx = scene_fixture(name, kw)
        for k, v in fx.items():
            out[name + "__" + k] = v
        print(name, "rows", len(fx["rows"]), "occupied", len(fx["occupied"]))
    np.savez_compressed(os.path.join(HERE, "scenes.npz"), **out)


if __name__ == "__main__":
    main()
