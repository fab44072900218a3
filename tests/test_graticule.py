"""graticule geometry of the host mirror against oracle/graticule_oracle.py (the reference's
vertex-buffer + transform formulation), bit for bit, plus known answers read off the reference's constants."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import graticule_oracle as go

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G():
    d = os.path.join(ROOT, "tests", "emul")
    from conftest import locked_make

    locked_make(d, "libscope_host_emul.so")
    L = C.CDLL(os.path.join(d, "libscope_host_emul.so"))
    f32p = C.POINTER(C.c_float)
    L.scope_graticule_vectorscope.restype = C.c_uint32
    L.scope_graticule_vectorscope.argtypes = [C.c_int, C.c_uint32, C.c_uint32, f32p, f32p]
    L.scope_graticule_waveform.restype = C.c_uint32
    L.scope_graticule_waveform.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, f32p, C.c_uint32]
    L.scope_graticule_histogram.restype = C.c_uint32
    L.scope_graticule_histogram.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                            f32p, C.c_uint32]
    return L


def _buf(n):
    a = np.full((n, 2), np.nan, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("cs", [1, 2])
@pytest.mark.parametrize("flags", [1, 2, 1 | 256])
@pytest.mark.parametrize("skin", [0x0054FF, 0x808080, 0x123456, 0xFFFFFF, 0x0000FF])
def test_vectorscope_graticule(G, cs, flags, skin):
    labels = np.full((6, 8), np.nan, dtype=np.float32)
    lines, lp = _buf(12 * 16 + 4)
    n = G.scope_graticule_vectorscope(cs, flags, skin, labels.ctypes.data_as(C.POINTER(C.c_float)), lp)
    want_labels, want_lines = go.vectorscope(cs, flags, skin)
    assert n == len(want_lines)
    assert same(labels, want_labels) and same(lines[:n], want_lines)
    assert np.isnan(lines[n:]).all()


def test_vectorscope_graticule_known_answers(G):
    lines, lp = _buf(12 * 16 + 4)
    n = G.scope_graticule_vectorscope(1, 1, 0x808080, None, lp)
    assert n == 12 * 16  # a neutral skin tone colour has no direction: no line (src/vectorscope.c:366)
    # BT.601 red target (U, V) = (90, 240): first bracket corner at (90 - 6, 256 - 240 - 6)
    assert tuple(lines[0]) == (84.0, 10.0) and tuple(lines[1]) == (88.0, 10.0) and tuple(lines[3]) == (84.0, 14.0)
    n = G.scope_graticule_vectorscope(2, 1, 0x0054FF, None, lp)
    assert n == 12 * 16 + 2 and tuple(lines[12 * 16]) == (127.5, 127.5)
    u, v = lines[12 * 16 + 1]
    assert abs(np.hypot(u - 128, (255 - v) - 128) - 128) < 1e-3  # the line ends on the circle of radius 128
    n = G.scope_graticule_vectorscope(2, 1 | 256, 0x0054FF, None, lp)
    assert n == 12 * 16 + 4


@pytest.mark.parametrize("lines_", [0, 1, 3, 5, 10])
@pytest.mark.parametrize("display,ncomp", [(0, 3), (1, 3), (2, 3), (1, 2), (2, 1)])
def test_waveform_graticule(G, lines_, display, ncomp):
    want = go.waveform(lines_, display, ncomp, 1920)
    buf, p = _buf(len(want) + 4)
    n = G.scope_graticule_waveform(lines_, display, ncomp, 1920, p, len(buf))
    assert n == len(want) and same(buf[:n], want)
    assert G.scope_graticule_waveform(lines_, display, ncomp, 1920, None, 0) == n  # size query
    if lines_ and display == 1:
        assert n == 2 * ((lines_ + 1) + (ncomp - 1) * lines_)  # lower bands do not repeat their top line


@pytest.mark.parametrize("v,step,log,fixed,ratio", [(5, 0, 0, 0, 0), (0, 50, 0, 200, 0), (5, 50, 0, 200, 0), (4, 2.5, 0, 0, 100),
                                                   (5, 50, 1, 200, 0), (0, 0, 0, 0, 0), (3, 0.7, 0, 0, 33), (5, 1, 0, 100, 0)])
@pytest.mark.parametrize("display,ncomp", [(0, 3), (1, 3), (2, 3), (2, 2)])
def test_histogram_graticule(G, v, step, log, fixed, ratio, display, ncomp):
    want = go.histogram(v, step, log, fixed, ratio, display, ncomp, 200)
    buf, p = _buf(len(want) + 4)
    n = G.scope_graticule_histogram(v, step, log, fixed, ratio, display, ncomp, 200, p, len(buf))
    assert n == len(want) and same(buf[:n], want)
