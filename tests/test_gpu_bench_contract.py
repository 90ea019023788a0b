"""GPU: bench.py's one-line JSON contract (driver-facing), on a tiny run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "2",
                          "--frames-per-step", "4", "--repeats", "3", "--cpu-sample", "2", "--host-path-frames", "2"], capture_output=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2
    assert d["unit"] == "Mpts/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] * 1e3 - d["config"]["points_per_step"]) < 1e-2 * d["config"]["points_per_step"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "dense" in c["sample"]
    assert d["cpu_baseline_sparse"]["cores"] == 1 and d["cpu_baseline_all_cores"]["cores"] >= 1
    assert d["rows_extracted"] > 0
    # 12 steps of 4 frames, clean every 3 steps: the steady state (dependant updates) is inside the timed region
    assert d["config"]["points_per_step"] == 4 * 640 * 480 and d["config"]["frames"] == 48
    assert d["repeats"] == 3 and len(d["pass_s"]) == 3 and d["value_min"] <= d["value"] <= d["value_max"]
    assert d["clean_passes"] == 4 and d["counters"]["dep_pairs_tested"] > 0 and d["warnings"] == []
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["kernel_source_sha"] and r["traffic_source"]


def test_bench_two_ranks_one_line_on_stdout():
    """Launched the way the driver launches N > 1 (one rank per GPU; here both ranks share the one GPU, so the RCCL
    communicator is refused and the ranks agree to fall back to the host-staged transport).  Gloo and RCCL print banners on
    fd 1; stdout must still carry exactly one JSON line, from rank 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2",
                          "--frames-per-step", "4", "--repeats", "2"],
                         capture_output=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must be exactly one line, got %d: %r" % (len(lines), lines[:3])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["scaling"] == "weak"
    assert d["config"]["points_per_step"] == 4 * 640 * 480
    assert "host-staged (gloo), 2 ranks" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    assert d["value"] > 0 and d["rows_extracted"] > 0
