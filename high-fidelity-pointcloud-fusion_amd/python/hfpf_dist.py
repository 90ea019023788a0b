"""Rank plumbing for the multi-GPU path: frame sharding, RCCL bootstrap, and a host-staged
transport over any torch.distributed backend (gloo included) built on the C ABI's transport-free entry points
(hfpf_epoch_export / hfpf_epoch_import / hfpf_stats_export / hfpf_extract_with_stats).

The data path of a production run is the engine's own RCCL collectives (hfpf_dist_init); the host-staged
transport exists so the same protocol runs where RCCL cannot (several virtual ranks on one GPU, gloo-only
boxes) and as a fallback when communicator creation fails.  This module holds host logic only.
"""
import numpy as np

EPOCH_REC_BYTES = 16  # one record per newly occupied cell (key, first frame id), two per frame integrated since the last exchange (its viewpoint)


def shard_frame_ids(n_local, rank, world, start=0):
    """Global frame ids of this rank's local frames [start, start+n_local): interleaved, so that an epoch of E local
    frames per rank covers the global ids [start*world, (start+E)*world) on every rank (aligned clean points)."""
    return ((np.arange(start, start + n_local, dtype=np.int64) * world) + rank).astype(np.uint32)


def broadcast_bytes(dist, payload, nbytes, src=0):
    """Broadcast `nbytes` raw bytes from rank `src` (payload may be None elsewhere)."""
    import torch
    t = torch.zeros(nbytes, dtype=torch.uint8)
    if dist.get_rank() == src:
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    dist.broadcast(t, src=src)
    return bytes(t.numpy().tobytes())


def allgather_bytes(dist, local):
    """Variable-length all-gather of uint8 arrays: returns a list (one per rank, own rank included)."""
    import torch
    world = dist.get_world_size()
    local = np.ascontiguousarray(local, dtype=np.uint8).reshape(-1)
    n = torch.tensor([local.size], dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if mx == 0:
        return [np.zeros(0, np.uint8) for _ in range(world)]
    pad = torch.zeros(mx, dtype=torch.uint8)
    if local.size:
        pad[:local.size] = torch.from_numpy(local.copy())
    outs = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o.numpy()[:s].copy() for o, s in zip(outs, sizes)]


def allreduce_words(dist, words):
    """Sum of uint64 word arrays across ranks with two's-complement wraparound (the engine's sums are int64)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(words, dtype=np.uint64).view(np.int64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().view(np.uint64)


def all_agree(dist, ok):
    """True only if every rank passes ok=True (MIN all-reduce over the launcher's process group)."""
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


class DistError(RuntimeError):
    """Raised on EVERY rank when any rank failed in front of a collective step: the step is abandoned everywhere, so no
    rank is left waiting in an all-gather / all-reduce its peer never enters (the engine's RCCL path does the same with a
    status word, csrc/hfpf.hip dist_status_gather_locked)."""


def init_rccl(grid, dist, hfpf_mod):
    """Bootstrap the engine's RCCL communicator: rank 0 creates the id, the launcher's process group carries it.
    Returns (ok, error text).  Either every rank ends up with a communicator or none does."""
    rank, world = dist.get_rank(), dist.get_world_size()
    err = ""
    uid = None
    if rank == 0:
        try:
            uid = hfpf_mod.dist_unique_id()
        except hfpf_mod.HfpfError as e:
            err = str(e)
            uid = bytes(128)
    uid = broadcast_bytes(dist, uid, 128, src=0)
    ok = False
    if uid != bytes(128):
        try:
            grid.dist_init_rccl(rank, world, uid)
            ok = True
        except hfpf_mod.HfpfError as e:
            err = str(e)
    if not all_agree(dist, ok):
        if ok:
            grid.dist_disable()
            err = "another rank failed to create its communicator"
        return False, err or "rank 0 could not create a unique id"
    return True, ""


class HostStagedTransport:
    """The epoch exchange and the statistics merge staged through host memory over torch.distributed."""

    def __init__(self, dist):
        self.dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def _consensus(self, err, what):
        """Every rank calls this in front of a collective with its local outcome; either all proceed or all raise."""
        if all_agree(self.dist, err is None):
            return
        if err is not None:
            raise DistError("rank %d failed before the %s: %s" % (self.rank, what, err)) from err
        raise DistError("another rank failed before the %s; the call is abandoned on every rank" % what)

    def exchange(self, grid):
        err, mine = None, np.zeros(0, np.uint8)
        try:
            ptr, n = grid.epoch_export()
            if n:
                mine = grid.device_download(ptr, n * EPOCH_REC_BYTES)
        except Exception as e:  # a poisoned / overflowed handle: tell the peers instead of leaving them in the all-gather
            err = e
        self._consensus(err, "epoch exchange")
        parts = allgather_bytes(self.dist, mine)
        foreign = [p for r, p in enumerate(parts) if r != self.rank and p.size]
        if foreign:
            buf = np.concatenate(foreign)
            dev = grid.device_alloc(buf.nbytes)
            grid.device_upload(dev, buf)
            grid.epoch_import(dev, buf.nbytes // EPOCH_REC_BYTES)
            grid.device_free(dev)

    def merged_extract(self, grid):
        err, mine, p, n, pc, nc = None, None, 0, 0, 0, 0
        try:
            p, n, pc, nc = grid.stats_export()
            mine = grid.device_download(p, n * 8, np.uint64)
        except Exception as e:
            err = e
        self._consensus(err, "statistics all-reduce of extract")
        import torch
        sizes = torch.tensor([n, -n], dtype=torch.int64)  # MAX of (n, -n) = (max n, -min n): equal record counts on every rank?
        self.dist.all_reduce(sizes, op=self.dist.ReduceOp.MAX)
        if int(sizes[0]) != -int(sizes[1]):
            raise DistError("the ranks hold between %d and %d statistic words: they did not run the same clean schedule" % (-int(sizes[1]), int(sizes[0])))
        total = allreduce_words(self.dist, mine)
        dev = grid.device_alloc(total.nbytes)
        grid.device_upload(dev, total)
        devc = 0
        if nc:
            totc = allreduce_words(self.dist, grid.device_download(pc, nc * 8, np.uint64))
            devc = grid.device_alloc(totc.nbytes)
            grid.device_upload(devc, totc)
        rows = grid.extract_with_stats(dev, devc)
        grid.device_free(dev)
        if devc:
            grid.device_free(devc)
        return rows


class LocalVirtualRanks:
    """Several handles on one GPU in one process standing in for ranks (tests): same protocol, device-to-device."""

    def __init__(self, grids, gathered=False):
        """gathered=True moves the records the way the RCCL path does: one padded buffer of world slices (what
        ncclAllGather leaves on every rank), imported with hfpf_epoch_import_gathered."""
        self.grids = grids
        self.gathered = gathered

    def clean_all(self):
        exports, failed = [], []
        for r, g in enumerate(self.grids):
            try:
                exports.append(g.epoch_export())
            except Exception as e:  # same consensus as the real transports: one failed rank abandons the pass on all of them
                exports.append((0, 0))
                failed.append((r, e))
        if failed:
            raise DistError("rank %d failed before the epoch exchange: %s; the pass is abandoned on every rank" % failed[0]) from failed[0][1]
        if self.gathered:
            world = len(self.grids)
            counts = np.array([n for _, n in exports], dtype=np.uint64)
            stride = max(int(counts.max()), 1) * EPOCH_REC_BYTES
            g0 = self.grids[0]
            buf = g0.device_alloc(world * stride)
            for r, (ptr, n) in enumerate(exports):
                if n:
                    g0.device_copy(buf + r * stride, ptr, n * EPOCH_REC_BYTES)
            for i, g in enumerate(self.grids):
                g.epoch_import_gathered(buf, stride, world, i, counts)
            g0.device_free(buf)
        else:
            for i, g in enumerate(self.grids):
                for j, (ptr, n) in enumerate(exports):
                    if i != j and n:
                        g.epoch_import(ptr, n)
        for g in self.grids:
            g.clean()

    def extract(self, on=0):
        g0 = self.grids[on]
        tot = totc = None
        for r, g in enumerate(self.grids):
            try:
                p, n, pc, nc = g.stats_export()
            except Exception as e:
                raise DistError("rank %d failed before the statistics merge of extract: %s" % (r, e)) from e
            w = g.device_download(p, n * 8, np.uint64)
            tot = w if tot is None else (tot + w)  # uint64 wraparound add
            if nc:
                c = g.device_download(pc, nc * 8, np.uint64)
                totc = c if totc is None else (totc + c)
        dev = g0.device_alloc(tot.nbytes)
        g0.device_upload(dev, tot)
        devc = 0
        if totc is not None:
            devc = g0.device_alloc(totc.nbytes)
            g0.device_upload(devc, totc)
        rows = g0.extract_with_stats(dev, devc)
        g0.device_free(dev)
        if devc:
            g0.device_free(devc)
        return rows
