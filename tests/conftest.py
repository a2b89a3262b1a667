import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The A/B knobs of include/s4g_ops.h are honoured only with the master switch: the tests drive them (every alternative
# path is checked against the default / the oracle), a production process never sets it.
os.environ["S4G_TEST_KNOBS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _variants_built():
    """True for a measurement build of the library (-DS4G_VARIANTS: csrc/variants/*.inc compiled in)."""
    try:
        from s4g_release_amd import _cabi
        return bool(_cabi.lib().s4g_build_variants())
    except Exception:
        return False


# the measured-slower kernel variants are exercised only where they exist (tools/build_variant.sh all -DS4G_VARIANTS)
VARIANTS = _variants_built()


def with_variants(default, extra):
    """Parameter list: `default` always, `extra` only when the library carries the variants."""
    return list(default) + (list(extra) if VARIANTS else [])
