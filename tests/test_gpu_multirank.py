"""GPU: the multi-GPU protocol with several virtual ranks on ONE device, plus the RCCL code path at
world size 1.  Frames are dealt round-robin with global frame ids; the merged result must be bit-identical to one
handle fusing every frame in id order, and match the oracle."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
SMALL = dict(max_bricks=60000, max_log_points=4 << 20, max_normals=1 << 20, max_frames=4096)


def _run_virtual(hfpf_mod, sc, world, fuse_color=False, deal=None, gathered=False):
    import hfpf_dist
    grids = [hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, fuse_color=fuse_color, **SMALL) for _ in range(world)]
    vr = hfpf_dist.LocalVirtualRanks(grids, gathered=gathered)
    deal = deal or (lambda f: f % world)
    fb = sc.W * sc.H * 16
    try:
        for ev in sc.schedule():
            if ev[0] == "integrate":
                f = ev[1]
                g = grids[deal(f)]
                dev = g.device_alloc(fb)
                g.device_upload(dev, sc.frame(f))
                g.integrate_device(dev, 1, fb, sc.W * sc.H, sc.poses[f].reshape(1, 12), frame_ids=np.array([f], np.uint32))
                g.sync()
                g.device_free(dev)
            else:
                vr.clean_all()
        rows = vr.extract(on=world - 1)
        occ = grids[0].occupied()
        ctrs = [g.counters() for g in grids]
    finally:
        for g in grids:
            g.close()
    return rows, occ, ctrs


@pytest.mark.parametrize("world", [2, 3])
def test_virtual_ranks_bit_identical_to_single_handle(oracle_mod, hfpf_mod, synth_mod, world):
    sc = scenes.Scene(7, 160, 120, 0.001, fx=615.0, clean_every=3)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as one:
        single = scenes.run(one, sc, "integrate")
        occ_single = one.occupied()
    rows, occ, ctrs = _run_virtual(hfpf_mod, sc, world)
    assert np.array_equal(occ, occ_single), "replicated occupancy differs from the single-GPU run"
    assert rows.tobytes() == single.tobytes(), "sharded result is not bit-identical to one GPU fusing all frames"
    assert sum(c["points_presented"] for c in ctrs) == sc.n_frames * sc.W * sc.H
    # every rank holds the same normal records (replicated clean)
    assert len({c["voxels_with_normal"] for c in ctrs}) == 1
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ref = scenes.run(og, sc, "capture")
    scenes.compare_rows(ref, rows)


def test_virtual_ranks_viewpoint_latch_is_global_min(hfpf_mod, synth_mod):
    """A cell first seen by rank 1 (frame 1) and later by rank 0 (frame 2) must orient with frame 1's viewpoint on
    both ranks: 5 mm voxels make most cells multi-frame."""
    sc = scenes.Scene(6, 160, 120, 0.005, clean_every=2)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as one:
        single = scenes.run(one, sc, "integrate")
    rows, _, _ = _run_virtual(hfpf_mod, sc, 2)
    assert rows.tobytes() == single.tobytes()


def test_gathered_import_unequal_and_zero_counts(hfpf_mod, synth_mod):
    """The receive side of the RCCL exchange (one padded all-gather buffer, per-rank counts and slice offsets:
    hfpf_epoch_import_gathered, the function dist_exchange_locked calls) with unequal per-rank counts, one of them zero:
    rank 2 gets no frame before the first clean, rank 0 two frames, rank 1 one."""
    sc = scenes.Scene(7, 160, 120, 0.001, fx=615.0, clean_every=3)
    owner = {0: 0, 1: 1, 2: 0, 3: 2, 4: 2, 5: 1, 6: 2}
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as one:
        single = scenes.run(one, sc, "integrate")
    rows, _, ctrs = _run_virtual(hfpf_mod, sc, 3, deal=lambda f: owner[f], gathered=True)
    assert rows.tobytes() == single.tobytes()
    assert [c["frames_integrated"] for c in ctrs] == [2, 2, 3]
    rows2, _, _ = _run_virtual(hfpf_mod, sc, 3, deal=lambda f: owner[f], gathered=False)  # the record-list transport agrees
    assert rows2.tobytes() == single.tobytes()


def test_failed_clean_poisons_the_handle_until_clear(hfpf_mod, synth_mod):
    """A capacity error in the middle of a clean pass leaves the tables half updated: the handle refuses further work
    (HFPF_ERR_STATE) until hfpf_clear instead of silently losing candidates on a retry."""
    sc = scenes.Scene(2, 160, 120, 0.001, fx=615.0)
    caps = dict(SMALL, max_normals=256)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **caps) as g:
        g.integrate(sc.frame(0), sc.poses[0])
        with pytest.raises(hfpf_mod.HfpfError) as e:
            g.clean()
        assert e.value.code == -3
        for call in (lambda: g.integrate(sc.frame(1), sc.poses[1]), g.clean, g.extract):
            with pytest.raises(hfpf_mod.HfpfError) as e2:
                call()
            assert e2.value.code == -5 and "hfpf_clear" in str(e2.value)
        g.clear()
        assert len(g.extract()) == 0
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:  # and a roomy handle is unaffected
        assert len(scenes.run(g, sc, "integrate")) > 1000


def test_virtual_ranks_with_colour(oracle_mod, hfpf_mod, synth_mod):
    sc = scenes.Scene(4, 160, 120, 0.001, fx=615.0, clean_every=2)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, fuse_color=True, **SMALL) as one:
        single = scenes.run(one, sc, "integrate")
    rows, _, _ = _run_virtual(hfpf_mod, sc, 2, fuse_color=True)
    assert rows.tobytes() == single.tobytes()
    assert (rows["rgb"][rows["count"] > 0] != 0).any()
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox, fuse_color=True)
    scenes.compare_rows(scenes.run(og, sc, "capture", color=True), rows)  # colour sums merge exactly across ranks


def test_rccl_path_world_size_one(hfpf_mod, synth_mod):
    """The engine's own RCCL collectives (dlopen'ed librccl: all-gather of counts/records in clean, all-reduce of the
    statistic sums in extract) with a single rank: exercises loading, communicator creation and every call site."""
    sc = scenes.Scene(4, 160, 120, 0.001, fx=615.0, clean_every=2)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as a:
        ref = scenes.run(a, sc, "integrate")
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as b:
        b.dist_init_rccl(0, 1, hfpf_mod.dist_unique_id())
        got = scenes.run(b, sc, "integrate")
    assert ref.tobytes() == got.tobytes()


def _two_proc_worker(rank, world, port, q, scene_kw):
    """One OS process = one rank (both on GPU 0): own camera stream, global frame ids, host-staged gloo transport."""
    try:
        import os
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        root = os.path.dirname(here)
        for p in (os.path.join(root, "high-fidelity-pointcloud-fusion_amd", "python"), here):
            if p not in sys.path:
                sys.path.insert(0, p)
        import numpy as np
        import torch.distributed as dist
        import hfpf
        import hfpf_dist
        import scenes
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sc = scenes.Scene(seed=0xF051 + 7919 * rank, pose_seed=0x5E3 + 104729 * rank, **scene_kw)
        g = hfpf.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL)
        g.attach_transport(hfpf_dist.HostStagedTransport(dist))
        fb = sc.W * sc.H * 16
        dev = g.device_alloc(fb)
        for f in range(sc.n_frames):
            g.device_upload(dev, sc.frame(f))
            g.integrate_device(dev, 1, fb, sc.W * sc.H, sc.poses[f].reshape(1, 12), frame_ids=hfpf_dist.shard_frame_ids(1, rank, world, f))
            g.sync()
            if (f + 1) % sc.clean_every == 0 and f + 1 < sc.n_frames:
                g.clean()  # collective
        g.clean()
        rows = g.extract()  # collective: int64 sums all-reduced over gloo
        q.put((rank, rows.tobytes()))
        g.device_free(dev)
        g.close()
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


def test_two_processes_one_stream_each_equal_single_process(hfpf_mod, synth_mod):
    """bench.py's N>1 shape: one camera stream per rank into one shared grid.  Every rank must end with the rows a single
    process gets when it is fed both streams interleaved in global frame-id order."""
    import multiprocessing as mp
    import socket
    kw = dict(n_frames=4, W=160, H=120, resolution=0.001, fx=615.0, clean_every=2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_proc_worker, args=(r, 2, port, q, kw)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert not (isinstance(res[r], str)), res[r]
    assert res[0] == res[1], "ranks disagree on the merged rows"
    streams = [scenes.Scene(seed=0xF051 + 7919 * r, pose_seed=0x5E3 + 104729 * r, **kw) for r in range(2)]
    fb = 160 * 120 * 16
    with hfpf_mod.OccupancyGrid(resolution=0.001, bbox=streams[0].bbox, **SMALL) as one:
        dev = one.device_alloc(fb)
        for f in range(kw["n_frames"]):
            for r in range(2):
                one.device_upload(dev, streams[r].frame(f))
                one.integrate_device(dev, 1, fb, 160 * 120, streams[r].poses[f].reshape(1, 12), frame_ids=np.array([f * 2 + r], np.uint32))
                one.sync()
            if (f + 1) % kw["clean_every"] == 0 and f + 1 < kw["n_frames"]:
                one.clean()
        one.clean()
        single = one.extract()
        one.device_free(dev)
    assert single.tobytes() == res[0], "two processes differ from one process fusing both streams"


def test_virtual_rank_failure_abandons_the_pass_on_every_rank(hfpf_mod, synth_mod):
    """Failure consensus: rank 1 overflows its private point log; the epoch exchange of the next
    clean must fail on EVERY rank (no rank cleans alone and then waits for the failed one at the following collective), and so
    must the statistics merge of extract."""
    import hfpf_dist
    sc = scenes.Scene(2, 160, 120, 0.001, fx=615.0)
    grids = [hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL),
             hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **dict(SMALL, max_log_points=4096))]
    vr = hfpf_dist.LocalVirtualRanks(grids)
    try:
        for r, g in enumerate(grids):
            g.integrate(sc.frame(r), sc.poses[r])
        with pytest.raises(hfpf_dist.DistError) as e:
            vr.clean_all()
        assert "rank 1 failed" in str(e.value) and "point log" in str(e.value)
        assert [g.counters()["clean_passes"] for g in grids] == [0, 0]  # nobody went ahead
        with pytest.raises(hfpf_dist.DistError) as e2:
            vr.extract()
        assert "rank 1 failed" in str(e2.value)
        grids[1].clear()  # the way out, as for a single handle
        for r, g in enumerate(grids):
            g.clear()
    finally:
        for g in grids:
            g.close()


def test_rccl_world_size_one_poisoned_rank_returns_instead_of_blocking(hfpf_mod, synth_mod):
    """The engine's own collectives with the failure-consensus status gather (dist_status_gather_locked) in front of them, at
    the only world size this box can form: a rank whose clean pass failed (max_normals overflow) must come back from the next
    hfpf_clean and from hfpf_extract with an error -- it enters the status all-gather with its fail bit set, which is what
    keeps its peers from waiting in ncclAllGather / ncclAllReduce -- and hfpf_clear makes it usable again."""
    sc = scenes.Scene(2, 160, 120, 0.001, fx=615.0)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **dict(SMALL, max_normals=256)) as g:
        g.dist_init_rccl(0, 1, hfpf_mod.dist_unique_id())
        g.integrate(sc.frame(0), sc.poses[0])
        with pytest.raises(hfpf_mod.HfpfError) as e:
            g.clean()
        assert e.value.code == -3  # capacity, found in the middle of the pass (after the exchange)
        for call in (g.clean, g.extract):
            with pytest.raises(hfpf_mod.HfpfError) as e2:
                call()
            assert e2.value.code == -5 and "hfpf_clear" in str(e2.value)
        g.clear()
        assert len(g.extract()) == 0  # through the status gather + all-reduce again, healthy
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **dict(SMALL, max_log_points=4096)) as g:
        g.dist_init_rccl(0, 1, hfpf_mod.dist_unique_id())
        g.integrate(sc.frame(0), sc.poses[0])
        with pytest.raises(hfpf_mod.HfpfError) as e:  # sticky overflow bit: found by the export in FRONT of the exchange
            g.clean()
        assert e.value.code == -3 and "point log" in str(e.value)
        with pytest.raises(hfpf_mod.HfpfError):
            g.extract()


def _fail_by_deferred_host_frame(g, hfpf_mod, sc):
    """Leaves the handle (created with max_frames = 50) with a host frame that is WAITING for its launch and whose launch must fail
    (frame id 50 = max_frames).  hfpf_integrate defers a frame's launch while the engine's stream is busy behind an earlier host
    frame (HFPF_HOST_BATCH): host frame 0 sets the staging ring up; 48 device frames without a bin plan (every point through the
    overflow list: milliseconds of kernels, nothing the host waits for) keep the stream busy; host frame 49 is launched behind
    them; host frame 50 finds that launch unfinished and stays pending -- the next call on the handle (the clean pass, as in the
    node) flushes it and gets the error.  Should a very slow host see the stream drained already, the error comes out of
    hfpf_integrate itself and only the frame is dropped; the callers below accept either."""
    big = scenes.Scene(48, 640, 480, sc.resolution, bbox=sc.bbox)
    fb = 640 * 480 * 16
    dev = g.device_alloc(48 * fb)
    for f in range(48):
        g.device_upload(dev + f * fb, big.frame(f))
    small = sc.frame(0)
    g.integrate(small, sc.poses[0])  # frame id 0
    g.sync()
    g.integrate_device(dev, 48, fb, 640 * 480, np.stack(big.poses))  # ids 1..48
    deferred = True
    try:
        g.integrate(small, sc.poses[0])  # id 49: launched behind the batch
        g.integrate(small, sc.poses[0])  # id 50 = max_frames: waits, fails when flushed
    except hfpf_mod.HfpfError as e:
        assert e.code == -3 and "max_frames" in str(e)
        deferred = False
    return dev, deferred


def test_rccl_world_size_one_deferred_host_frame_failure_goes_through_the_consensus(hfpf_mod, synth_mod):
    """Round-3 review: hfpf_clean / hfpf_extract returned the error of a deferred host-frame launch BEFORE the status gather, so a
    rank failing that way would have left its peers in ncclAllGather.  The failure is now folded into the status word.  At the only
    world size RCCL can form here: the failing rank comes back from clean and extract with its error (not a hang), is poisoned,
    and hfpf_clear revives it."""
    sc = scenes.Scene(1, 160, 120, 0.001, fx=615.0)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **dict(SMALL, max_frames=50, max_log_points=24 << 20, max_normals=4 << 20)) as g:
        g.dist_init_rccl(0, 1, hfpf_mod.dist_unique_id())
        dev, deferred = _fail_by_deferred_host_frame(g, hfpf_mod, sc)
        assert deferred, "the host frame was launched at once: the stream was idle (box too slow for this test's timing assumption)"
        codes = []
        for call in (g.clean, g.extract, g.clean):
            with pytest.raises(hfpf_mod.HfpfError) as e:
                call()
            codes.append(e.value.code)
        assert codes == [-3, -5, -5], codes  # capacity where the frame was still waiting, "failed earlier" after that
        g.clear()
        g.device_free(dev)
        g.integrate(sc.frame(0), sc.poses[0])
        g.clean()
        assert len(g.extract()) >= 0


def _two_proc_failing_worker(rank, world, port, q, scene_kw, mode="log"):
    """Like _two_proc_worker, but rank 1 fails in front of the first collective clean: its point log is far too small (mode "log":
    the export finds the overflow), or a host frame whose launch was deferred fails when the clean flushes it (mode "deferred")."""
    try:
        import os
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        root = os.path.dirname(here)
        for p in (os.path.join(root, "high-fidelity-pointcloud-fusion_amd", "python"), here):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        import hfpf
        import hfpf_dist
        import scenes
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sc = scenes.Scene(seed=0xF051 + 7919 * rank, pose_seed=0x5E3 + 104729 * rank, **scene_kw)
        caps = SMALL
        if rank == 1:
            caps = dict(SMALL, max_log_points=4096) if mode == "log" else dict(SMALL, max_frames=50, max_log_points=24 << 20, max_normals=4 << 20)
        g = hfpf.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **caps)
        g.attach_transport(hfpf_dist.HostStagedTransport(dist))
        if rank == 1 and mode == "deferred":
            _, was_deferred = _fail_by_deferred_host_frame(g, hfpf, sc)
            if not was_deferred:
                raise RuntimeError("the host frame was launched at once: the stream was idle (timing assumption of the test)")
        else:
            g.integrate(sc.frame(0), sc.poses[0])
        out = []
        for call in (g.clean, g.extract):
            try:
                call()
                out.append("no error")
            except hfpf_dist.DistError as e:
                out.append("mine" if "rank %d failed" % rank in str(e) else "peer")
        q.put((rank, out))
        g.close()
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("mode", ["log", "deferred"])
def test_two_processes_failure_on_one_rank_stops_both(hfpf_mod, synth_mod, mode):
    """Two OS processes on the GPU over the gloo host-staged transport, rank 1 overflowing its point log (mode "log") or failing
    the deferred launch of a host frame (mode "deferred": what the real node's hfpf_integrate path can do to a clean): both ranks
    must return an error from the collective clean and from the collective extract; a hang fails the test through its timeout."""
    import multiprocessing as mp
    import queue as queue_mod
    import socket
    kw = dict(n_frames=1, W=160, H=120, resolution=0.001, fx=615.0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_proc_failing_worker, args=(r, 2, port, q, kw, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in procs:
            r, v = q.get(timeout=240)
            res[r] = v
    except queue_mod.Empty:
        pass
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert res == {0: ["peer", "peer"], 1: ["mine", "mine"]}, res
