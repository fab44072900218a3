"""Shared fixtures.

Backends:
  * ``emul``  -- tests/emul/libscope_emul.so: the UNMODIFIED csrc/*.hip sources compiled for the
    CPU against a fiber-per-GPU-thread stand-in of the HIP runtime (tests/emul/).  Test
    infrastructure only; lets kernel index arithmetic and the host glue be checked without a GPU.
  * ``hip``   -- obs-color-monitor_amd/libscope_hip.so on a real MI355X (``-m gpu``).

The checker is always oracle/ (C restatement of the reference loops).  Nothing under tests/ reads
/root/reference at run time.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def locked_make(directory, *targets):
    """`make` under a file lock: pytest-xdist workers (and a concurrent build()) must not write one .so at once."""
    import fcntl

    with open(os.path.join(directory, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", directory, *targets])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import scope_oracle

    return scope_oracle.COracle()


@pytest.fixture(scope="session")
def emul_lib():
    d = os.path.join(ROOT, "tests", "emul")
    locked_make(d, "libscope_emul.so")
    return os.path.join(d, "libscope_emul.so")


@pytest.fixture(scope="session")
def emul_ctx(emul_lib):
    import obs_color_monitor_amd as ocm

    ctx = ocm.ScopeContext(0, emul_lib)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def hip_ctx():
    import torch  # first: the library then shares torch's HIP runtime (same libamdhip64 soname)

    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    # normally a no-op (build() ran and the .so travelled with the tree); hipcc exists on the GPU box too
    locked_make(os.path.join(ROOT, "obs-color-monitor_amd", "csrc"))
    locked_make(os.path.join(ROOT, "obs-color-monitor_amd", "host"))
    import obs_color_monitor_amd as ocm

    ctx = ocm.ScopeContext(0)  # raises if libscope_hip.so is missing: no fallback
    yield ctx
    ctx.close()
