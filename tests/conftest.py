import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suite runs the library of THIS checkout: one that is missing or stale is rebuilt first when hipcc is there (a snapshot of
    # the tree may carry a library built before the last source edit); without a compiler `_lib.lib()` fails loudly, as it must
    try:
        from instantavatar_amd import build
        build.ensure_current(verbose=True)
    except Exception as e:      # (reported by the first test that loads the library)
        print("conftest: could not bring libinstantavatar_hip.so up to date:", repr(e)[:200], file=sys.stderr)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc
