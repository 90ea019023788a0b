"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hfpf_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(hfpf_mod):
    L = hfpf_mod.lib()
    names = [n for n in _declared("hfpf.h") + _declared("hfpf_probe.h") if not n.endswith("_fn")]
    assert len(names) >= 27
    for n in names:
        assert hasattr(L, n), "libhfpf.so does not export %s" % n
    assert sorted(names) == sorted(hfpf_mod.EXPORTS), "python binding EXPORTS out of date with the headers"
    assert L.hfpf_abi_version() == 6
    import hfpf_node
    NL = hfpf_node.lib()
    node_names = [n for n in _declared("hfpf_node.h") if n.startswith("hfpf_node_")]
    assert sorted(node_names) == sorted(hfpf_node.EXPORTS)
    for n in node_names:
        assert hasattr(NL, n), "libhfpf_node.so does not export %s" % n


def test_default_config_matches_reference_constants(hfpf_mod):
    c = hfpf_mod.default_config()
    assert c.struct_size == C.sizeof(hfpf_mod.Config)
    assert abs(c.resolution - 0.005) < 1e-9            # node.cpp:91
    assert list(c.bbox) == [-0.80, 1.80, -1.5, 1.5, 0.0, 1.0]  # launch file
    assert (c.k, c.K, c.gate) == (2, 3, 20)            # node.cpp:163,311; grid.hpp:352
    assert (c.cylinder_radius, c.ball_radius) == (0.001, 0.015)  # grid.hpp:35-36
    assert (c.z_clip_min, c.z_clip_max) == (0.28, 0.6)  # node.cpp:92-93


def test_create_fails_loudly_without_gpu_or_with_bad_config(hfpf_mod):
    """No silent CPU fallback: without a HIP device create must fail with HFPF_ERR_HIP; bad configs are rejected first."""
    import pytest
    with pytest.raises(hfpf_mod.HfpfError) as e:
        hfpf_mod.OccupancyGrid(bbox=(0, 1, 0, 1, 1, 0))
    assert e.value.code == -1
    with pytest.raises(hfpf_mod.HfpfError) as e:
        hfpf_mod.OccupancyGrid(k=3)
    assert e.value.code == -1
    with pytest.raises(hfpf_mod.HfpfError):
        hfpf_mod.OccupancyGrid(bbox=(0, 1, 0, 1))
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if not has_gpu:
        with pytest.raises(hfpf_mod.HfpfError) as e:
            hfpf_mod.OccupancyGrid()
        assert e.value.code == -4 and "no CPU path" in str(e.value)


def test_row_layout_is_64_bytes(hfpf_mod, oracle_mod):
    assert hfpf_mod.ROW_DTYPE.itemsize == 64
    assert hfpf_mod.ROW_DTYPE == oracle_mod.ROW_DTYPE
