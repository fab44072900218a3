"""The C host mirror of the plugin's scope sources (drop-in boundary), CPU and GPU flavours."""
import os
import subprocess

import pytest

import host_cases as hc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_on_emulated_kernels(emul_lib, oracle):
    d = os.path.join(ROOT, "tests", "emul")
    from conftest import locked_make

    locked_make(d, "libscope_host_emul.so")
    host = hc.Host(os.path.join(d, "libscope_host_emul.so"))
    try:
        hc.run_host_checks(host, oracle)
    finally:
        host.close()


@pytest.mark.gpu
def test_host_mirror_on_gpu(hip_ctx, oracle):
    import torch  # noqa: F401

    path = os.path.join(ROOT, "obs-color-monitor_amd", "libscope_host.so")
    assert os.path.exists(path), "build() must produce libscope_host.so"
    host = hc.Host(path)
    try:
        hc.run_host_checks(host, oracle, w=640, h=360)
    finally:
        host.close()
