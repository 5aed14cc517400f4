"""pytest configuration: the `gpu` marker and the import paths of the test infrastructure.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol checks (no device compute).
`-m gpu`       : parity tests proper, through the C ABI of libhipsdp.so on a real MI355X.  They FAIL (not skip) when the
                 library or the device is missing: a silent fallback would void the parity claim."""
import os
import sys
import importlib.util
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load_binding():
    spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def hb():
    """ctypes binding of libhipsdp.so (build it with __graft_entry__.build() if missing)"""
    mod = _load_binding()
    if not os.path.exists(mod.LIBPATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def gpu(hb):
    """binding + a hard requirement that a device is present"""
    n = hb.device_count()
    assert n > 0, "no HIP device visible: the gpu-marked tests must run on an MI355X (no CPU fallback exists)"
    return hb


GOLDEN = os.path.join(ROOT, "tests", "golden")
