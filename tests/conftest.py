import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def built_library():
    """Tests bind the in-tree liba3d.so; build it (hipcc cross-compiles without a GPU) if this is a fresh checkout."""
    from ann3depth_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.LIB_PATH


@pytest.fixture(scope='session')
def lib():
    """The C-ABI library, loaded through the product's own loader (fails loudly if not built)."""
    from ann3depth_amd import _lib
    return _lib.load()
