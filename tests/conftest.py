import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(params=["eigen_tree", "left_to_right"])
def fp64_order(request, oracle):
    """Runs a test under both orders of the 3-term FP64 sums of the gate chain (okvfe_set_fp64_reduction /
    orc_set_reduction): the oracle and -- on a GPU box -- the device flag are switched together and restored
    to the default (Eigen's x0 + (x1 + x2)) afterwards.  Apply with @pytest.mark.usefixtures("fp64_order")."""
    tree = request.param == "eigen_tree"
    oracle.set_reduction(tree)
    fe = None
    try:
        import torch
        if torch.cuda.is_available():
            from okvis2_amd import capi
            fe = capi.Frontend(64, 64, 10.0, 0, 50, 10)
            fe.set_fp64_reduction(tree)
    except ImportError:
        pass
    yield request.param
    oracle.set_reduction(True)
    if fe is not None:
        fe.set_fp64_reduction(True)
