import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "xbf16_contract: x_storage='bf16' case asserted at the 1e-4 contract (tests/test_parity_gpu.py)")
    config.addinivalue_line("markers", "xbf16_same_inputs: x_storage='bf16' case against the oracle on the same rounded bags (TOL)")
    # The SUITE's default arithmetic is the exact fp32 mode (tests that cover bf16x3 select it themselves, and put it back): the
    # library's own default is bf16x3 (advmil_amd/_lib.py), which would run the 2e-6 "same result on two paths" comparisons at
    # 2^-17 per product. Child processes of the suite (two-rank workers, fuzzers) inherit the setting.
    os.environ["ADVMIL_GEMM_MODE"] = "exact"
    # unwritten fp32 tokens of gradients / activations that exist as operand planes only are filled with NaN under the suite: anything
    # that reads one (instead of its planes) poisons a checked result (advmil_amd/ops.py::PlaneHandover)
    os.environ["ADVMIL_POISON_TOKENS"] = "1"


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def golden2():
    """Round-2 fixtures at the BASELINE sizes (tests/golden/gen_golden_r2.py): 32768-patch ESAT eval forward, optimizer steps at
    8192 / 32768 patches through the reference's own handler."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
