#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh (one counter set per pass, as MI355X_MICROARCH.md prescribes) for
the integrate hot path: k_integrate (decode / clip / transform / insert / park) + k_update (per-brick LDS accumulation).

usage: python tools/pmc_summary.py gpurun_out/<mem prefix> profiles/r04_pmc_hot_path [gpurun_out/<sq prefix>] [--workload c3]
(the optional third argument adds the SQ counters of k_update_cells and k_integrate -- instruction counts, active lanes, LDS bank
conflicts -- as section "sq", which bench.py turns into roofline.compute and roofline.compute_integrate; --workload c3: the passes
ran `bench.py --workload c3`, i.e. 100 frames of 2048x1536 in calls of 15 -- the same dispatch pattern at another size)

Each pass ran `bench.py --repeats 1 --warmup 0 --cpu-sample 0 --host-path-frames 0`: the 1000-frame configs[1] stream, one
hfpf_integrate_device call per clean epoch.  k_integrate dispatches: #0 = the dry run of the session's first 8 frames (bin
demand), #1 = the first epoch (everything is buffered, no dependant exists yet), #2..#6 = steady state, 150 frames each (#7 is
the 100-frame tail and is left out of the per-launch means); k_update dispatches #0..#4 belong to the steady launches;
k_buffer dispatch #0 to the first epoch, #1..#5 to the steady launches; k_integrate_overflow dispatches (#0 first epoch, #1..#5 steady)
are added to k_integrate's figures.
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide
(16 B/lane) coalesced stream, so 8 B per streamed 16-byte record are added (the frame read in k_integrate, the bin read-back in
k_update); narrower scattered reads are uncalibrated.
Writes <out>.json (read by bench.py, with the hash of the kernel sources it was taken on) and <out>.md.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES_PER_LAUNCH = 150
NPTS = 640 * 480
KERNEL_SOURCES = ("kernels.hpp", "tables.hpp", "stats.hpp", "geometry.hpp", "det_math.hpp", "hfpf.hip")


def source_sha():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "high-fidelity-pointcloud-fusion_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load(prefix):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> counter -> dispatch id -> value
    for f in sorted(glob.glob(prefix + "_*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            which = ("k_overflow" if "k_integrate_overflow" in k else "k_integrate" if "k_integrate" in k else
                     "k_update" if "k_update" in k else "k_buffer" if "k_buffer" in k else None)
            if which:
                per[which][r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    return per


def series(per, kernel, counter):
    d = per[kernel].get(counter, {})
    return [d[i] for i in sorted(d)]


def sq_section(sq_prefix):
    """Means over the steady launches of k_update_cells (all but the last, shorter one) + the pairs one such launch tests, and
    over the steady launches of k_integrate (#2..#6: behind the dry run and the first epoch, in front of the shorter tail)."""
    per = load(sq_prefix)
    out = {}
    for c, d in per["k_update"].items():
        v = [d[i] for i in sorted(d)]
        v = v[:-1] if len(v) > 1 else v
        out[c] = sum(v) / len(v)
    integ = {}
    for c, d in per["k_integrate"].items():
        v = [d[i] for i in sorted(d)][2:7]
        if v:
            integ[c] = sum(v) / len(v)
    if integ.get("SQ_THREAD_CYCLES_VALU") and integ.get("SQ_ACTIVE_INST_VALU"):
        integ["active_lane_fraction"] = integ["SQ_THREAD_CYCLES_VALU"] / (integ["SQ_ACTIVE_INST_VALU"] * 64.0)
    pairs = None
    for f in sorted(glob.glob(sq_prefix + "_*.json")):
        try:
            line = json.loads(open(f).read().strip().splitlines()[-1])
            pairs = line["counters"]["dep_pairs_tested"] * FRAMES_PER_LAUNCH / float(line["config"]["frames"] - FRAMES_PER_LAUNCH)
            break
        except Exception:
            continue
    if out.get("SQ_THREAD_CYCLES_VALU") and out.get("SQ_ACTIVE_INST_VALU"):
        out["active_lane_fraction"] = out["SQ_THREAD_CYCLES_VALU"] / (out["SQ_ACTIVE_INST_VALU"] * 64.0)
    return {"k_update": out, "k_integrate": integ, "pairs_per_launch": pairs}


def main():
    global FRAMES_PER_LAUNCH, NPTS
    argv = list(sys.argv[1:])
    if "--workload" in argv:
        i = argv.index("--workload")
        if argv[i + 1] == "c3":
            FRAMES_PER_LAUNCH, NPTS = 15, 2048 * 1536
        del argv[i:i + 2]
    prefix, out = argv[0], argv[1]
    per = load(prefix)
    mean = lambda v: sum(v) / len(v) if v else 0.0
    in_bbox_frac = 0.826  # points_in_bbox / points_presented of the configs[1] stream; replaced by the passes' own bench line when there is one
    for f in sorted(glob.glob(prefix + "_*.json")):
        try:
            c = json.loads(open(f).read().strip().splitlines()[-1])["counters"]
            in_bbox_frac = c["points_in_bbox"] / float(c["points_presented"])
            break
        except Exception:
            continue
    pts = FRAMES_PER_LAUNCH * NPTS
    res = {"source_sha": source_sha(), "points_per_launch": pts, "frames_per_launch": FRAMES_PER_LAUNCH,
           "source": "rocprofv3 --pmc, separate passes (tools/pmc_passes.sh), bench.py --repeats 1 --warmup 0"}
    for phase, sel_a, sel_b, sel_c in (("first_epoch_buffer_only", lambda v: v[1:2], None, lambda v: v[:1]),
                                       ("steady_state_after_first_clean", lambda v: v[2:7], lambda v: v[:5], lambda v: v[1:6])):
        # k_integrate_overflow (one dispatch behind every k_integrate but the dry run) belongs to the integrate stage
        sel_o = (lambda v: v[0:1]) if sel_b is None else (lambda v: v[1:6])
        ga = lambda c: mean(sel_a(series(per, "k_integrate", c))) + mean(sel_o(series(per, "k_overflow", c)))
        gb = (lambda c: mean(sel_b(series(per, "k_update", c)))) if sel_b else (lambda c: 0.0)
        gc = lambda c: mean(sel_c(series(per, "k_buffer", c)))
        fa, wa, aa = ga("FETCH_SIZE") * 1024, ga("WRITE_SIZE") * 1024, ga("TCC_EA0_ATOMIC_sum")
        fb, wb, ab = gb("FETCH_SIZE") * 1024, gb("WRITE_SIZE") * 1024, gb("TCC_EA0_ATOMIC_sum")
        fc, wc, ac = gc("FETCH_SIZE") * 1024, gc("WRITE_SIZE") * 1024, gc("TCC_EA0_ATOMIC_sum")
        # wide-read correction: the frame stream (all points), the bin read-back of k_update (steady state: the in-bbox points) or
        # of k_buffer (first epoch: the in-bbox points, all of them buffered)
        corr = 0.5 * 16 * pts + 0.5 * 16 * in_bbox_frac * pts
        traffic = fa + fb + fc + corr + wa + wb + wc
        hit = lambda g: g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)
        steady = sel_b is not None
        res[phase] = {
            "k_integrate": {"fetch_bytes_raw": fa, "write_bytes": wa, "atomic_requests": aa, "l2_hit_rate": hit(ga), "fetch_correction_bytes": 0.5 * 16 * pts},
            "k_update": {"fetch_bytes_raw": fb, "write_bytes": wb, "atomic_requests": ab, "l2_hit_rate": hit(gb) if sel_b else None,
                         "fetch_correction_bytes": 0.5 * 16 * in_bbox_frac * pts if steady else 0.0},
            "k_buffer": {"fetch_bytes_raw": fc, "write_bytes": wc, "atomic_requests": ac, "l2_hit_rate": hit(gc),
                         "fetch_correction_bytes": 0.0 if steady else 0.5 * 16 * in_bbox_frac * pts},
            "fetch_correction_bytes": corr,
            "atomic_requests": aa + ab + ac,
            "traffic_bytes_per_launch": traffic,
            "traffic_bytes_per_point": traffic / pts,
            "algorithmic_bytes_per_launch": 32 * pts,
        }
    res["in_bbox_fraction"] = in_bbox_frac
    if len(argv) > 2:
        res["sq"] = sq_section(argv[2])
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        f.write("# Integrate hot path: memory-side counters (rocprofv3 --pmc, one counter set per pass)\n\n")
        f.write("Kernel sources `%s` (bench.py attaches these numbers only to a build with the same hash).\n" % res["source_sha"])
        f.write("Per launch = one `hfpf_integrate_device` call of %d frames = %d points. FETCH_SIZE corrected per MI355X_MICROARCH.md (HBM):\n"
                "+8 B per streamed 16-byte record (gfx950 tallies wide coalesced reads at half); scattered 4-8 B table reads are uncalibrated.\n\n" % (FRAMES_PER_LAUNCH, pts))
        f.write("| phase | kernel | FETCH raw | WRITE | atomic requests | L2 hit rate |\n|---|---|---|---|---|---|\n")
        for phase in ("first_epoch_buffer_only", "steady_state_after_first_clean"):
            for k in ("k_integrate", "k_update", "k_buffer"):
                r = res[phase][k]
                if r["l2_hit_rate"] is None:
                    continue
                f.write("| %s | `%s` | %.3f GB | %.3f GB | %.2f M | %.2f |\n" % (phase, k, r["fetch_bytes_raw"] / 1e9, r["write_bytes"] / 1e9,
                                                                          r["atomic_requests"] / 1e6, r["l2_hit_rate"]))
        f.write("\n| phase | traffic / launch (corrected) | B / point | algorithmic (32 B/pt) | atomic requests / launch |\n|---|---|---|---|---|\n")
        for phase in ("first_epoch_buffer_only", "steady_state_after_first_clean"):
            r = res[phase]
            f.write("| %s | %.3f GB | %.0f | %.3f GB | %.2f M |\n" % (phase, r["traffic_bytes_per_launch"] / 1e9, r["traffic_bytes_per_point"],
                                                                 r["algorithmic_bytes_per_launch"] / 1e9, r["atomic_requests"] / 1e6))
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
