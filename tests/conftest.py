import os
import sys
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'dcs-net_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_sessionstart(session):
    # the CPU suite checks the C-ABI library's exports: make sure it exists (hipcc cross-compiles
    # without a GPU; on the GPU box the prebuilt in-tree .so travels with the snapshot)
    lib = os.path.join(PKG, 'lib', 'libdcsnet_hip.so')
    if not os.path.exists(lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location('dcsnet_build', os.path.join(PKG, 'build.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
