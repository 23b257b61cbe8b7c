"""pytest configuration: markers + shared paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU compute).
`-m gpu`      : parity tests proper; every one calls through the C-ABI (libnps.so) on cuda:0.
"""
import os
import sys

import pytest

# torch ships its own copy of the HIP runtime: import it before libnps.so pulls in /opt/rocm's, or a
# later torch.cuda initialisation in the same process finds no device (the order bench.py uses too)
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def need_free_hbm(gib):
    """Full-size tests need most of an MI355X's 288 GB.  On a smaller part they are skipped; on an MI355X whose HBM is
    taken by something else they FAIL (a busy GPU must be an error, not an "s" in a green run)."""
    import torch
    free, total = torch.cuda.mem_get_info()
    if free >= gib * (1 << 30):
        return
    assert total < 256 * (1 << 30), ("this device has %.0f GB of HBM but only %.0f GB are free (%d GB needed): "
                                     "another job holds the GPU" % (total / 2 ** 30, free / 2 ** 30, gib))
    pytest.skip("needs %d GB of free HBM (device has %.0f GB)" % (gib, total / 2 ** 30))
