import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DTYPES = {"torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16}
SMALL_CASES = ["fp32_plain", "fp32_unit", "fp32_cross", "fp32_aux", "fp32_aux_unit", "fp32_sched",
               "bf16_plain", "bf16_unit", "bf16_aux"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no libmtc.so (built artefacts are not tracked): build it once, in-tree, if a compiler is here
    lib = os.path.join(ROOT, "matryoshka_transcoder_amd", "libmtc.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.run(["make", "-j8"], cwd=ROOT, check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def load_golden(name):
    """Fixtures were written by tools/make_golden.py from the reference itself; tensors only."""
    fx = torch.load(os.path.join(GOLDEN, f"{name}.pt"), weights_only=True)
    cfg = dict(fx["cfg"])
    cfg["dtype"] = DTYPES[cfg["dtype"]]
    return fx, cfg


def tol(dt):
    """Stated tolerances: fp32 rel 1e-4; bf16 rel 1e-2 on losses/FVU."""
    return (1e-4, 1e-6) if dt == torch.float32 else (2e-2, 1e-3)


@pytest.fixture(scope="session")
def gpu_lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from matryoshka_transcoder_amd import _lib
    return _lib.load()
