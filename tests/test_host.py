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


def test_host_mirror_under_thread_sanitizer(emul_lib):
    """The host mirror's own sources under -fsanitize=thread, its three threads flat out (tests/emul/host_tsan.c): the graphics
    thread stages into the ring, every source's color-monitor thread runs its callback and flips the double buffer, a render
    thread draws and asks for sizes all the while.  The reference's bare `volatile int w_tex_buf` (src/vectorscope.c:48,264,410)
    is reported by exactly this driver (data race, vss_surface_cb's flip against vss_render's read); the mirror's ordered flip
    (scope_sources.h) is not.  The driver also checks that nothing staged was lost and that the last tables are reproducible."""
    d = os.path.join(ROOT, "tests", "emul")
    from conftest import locked_make

    locked_make(d, "host_tsan")
    out = subprocess.run([os.path.join(d, "host_tsan")], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-4000:]
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-2000:])
    assert "rc 0" in out.stdout and "processed" in out.stdout


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
