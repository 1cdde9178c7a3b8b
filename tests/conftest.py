import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{r.stdout}\n{r.stderr}")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so), built on demand.  Test infrastructure only."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def emul():
    """Host build of the device arithmetic + index logic (tests/native/emul_device.cpp)."""
    import ctypes
    so = os.path.join(ROOT, "tests", "native", "libemul.so")
    csrc = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "native", "emul_device.cpp"), os.path.join(csrc, "imt_params.cpp"),
            os.path.join(csrc, "imt_trace_layout.cpp"), os.path.join(csrc, "imt_gadget_layout.cpp")]
    deps = srcs + [os.path.join(csrc, f) for f in ("imt_device.hpp", "imt_consts.hpp", "imt_sweep.hpp",
                                                    "imt_params.hpp", "imt_fr_host.hpp", "imt_prep_logic.hpp",
                                                    "imt_trace_device.hpp", "imt_ctx.hpp")] + [
        os.path.join(ROOT, "include", "imt.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        # imt_trace_layout.cpp sees the context struct, hence the HIP *headers* (types only; nothing of HIP is linked)
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", csrc, "-I", "/opt/rocm/include",
              "-D__HIP_PLATFORM_AMD__", "-o", so] + srcs, ROOT)
    lib = ctypes.CDLL(so)
    assert lib.emul_init() == 0
    return lib


@pytest.fixture(scope="session")
def imt():
    """The product package.  csrc/libimt_hip.so is (re)built first when it is missing or older than its sources
    (hipcc cross-compiles for gfx950 without a GPU); a failed build fails loudly -- there is nothing to fall back to."""
    csrc = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc")
    so = os.path.join(csrc, "libimt_hip.so")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".hpp"))] + [
        os.path.join(ROOT, "include", "imt.h")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
        _run(["make", "-C", csrc, "libimt_hip.so", "ARCH=gfx950"], ROOT)
    import imt_amd
    return imt_amd


@pytest.fixture(scope="session")
def ctx(imt):
    c = imt.Context(0)
    yield c
    c.close()
