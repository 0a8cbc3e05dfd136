import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    # the suite needs the in-tree libcc4.so (ABI tests on CPU, everything on the GPU); build it if it is not there yet
    # (hipcc cross-compiles gfx950 without a GPU).  Staleness is build()'s business, not the tests'.
    from cage_challenge_4_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()


def _has_gpu():
    # no torch needed: ask the HIP runtime through libcc4's own create call
    try:
        from cage_challenge_4_amd import CC4VecEnv
        e = CC4VecEnv(1, steps=6)
        e.close()
        return True
    except Exception:
        return False


@pytest.fixture(scope='session')
def has_gpu():
    return _has_gpu()


@pytest.fixture(scope='session')
def oracle_lib():
    import oracle_binding
    oracle_binding.build()
    return oracle_binding.load()
