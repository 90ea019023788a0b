import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "high-fidelity-pointcloud-fusion_amd", "python"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU cases")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth_mod():
    import hfpf_synth
    hfpf_synth.build()
    return hfpf_synth


@pytest.fixture(scope="session")
def hfpf_mod():
    """The engine binding.  On the GPU box the .so must already be in-tree (built by __graft_entry__.build())."""
    import hfpf
    hfpf.lib()
    return hfpf
