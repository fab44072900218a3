"""Property-style parity on the CPU-emulated kernels: random geometry (ragged widths, arbitrary 4-byte
aligned pitches, misaligned plane starts), random content and random tunables must never change a table."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import parity_cases as pc

GEOM = st.tuples(st.integers(1, 97), st.integers(1, 33), st.integers(0, 9), st.sampled_from([0, 4, 8, 12]))
OPTS = st.fixed_dictionaries({"vss.merge": st.sampled_from([0, 2, 3]), "vss.hash": st.sampled_from([0, 1]),
                              "vss.groups": st.integers(0, 3), "vss.unroll": st.sampled_from([1, 2, 4]),
                              "slices": st.sampled_from([0, 1, 2, 5]), "nt": st.sampled_from([0, 1])})
SETTINGS = dict(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


def _dev(a):
    buf = np.ascontiguousarray(a).copy()
    return buf, buf.ctypes.data


@settings(**SETTINGS)
@given(geom=GEOM, dist=st.integers(1, 5), opts=OPTS)
def test_vectorscope_any_geometry_any_variant(emul_ctx, oracle, geom, dist, opts):
    w, h, pad_words, off = geom
    ls = 4 * w + 4 * pad_words
    with pc.options(emul_ctx, **opts):
        pc.check_vectorscope(emul_ctx, oracle, w, h, dist, ls)          # host path (re-staged, 16-byte rows)
        pc.check_device_path(emul_ctx, oracle, _dev, lambda b: b, w, h, ls, n=1, offset=off)  # device path as given


@settings(**SETTINGS)
@given(geom=GEOM, dist=st.integers(1, 5), comp=st.sampled_from(pc.COMPONENTS), bands=st.sampled_from([0, 1, 3]))
def test_waveform_histogram_any_geometry(emul_ctx, oracle, geom, dist, comp, bands):
    w, h, pad_words, _ = geom
    ls = 4 * w + 4 * pad_words
    with pc.options(emul_ctx, slices=bands):
        pc.check_waveform(emul_ctx, oracle, w, h, dist, comp, ls)
        pc.check_histogram_counts(emul_ctx, oracle, w, h, dist, comp, ls)


@settings(**SETTINGS)
@given(w=st.integers(1, 80), h=st.integers(1, 40), cs=st.sampled_from([1, 2]), seed=st.integers(0, 2**31))
def test_fused_k0_equals_k0_then_scope(emul_ctx, oracle, w, h, cs, seed):
    """SCOPE_FUSE_K0 on the RGB plane == the scope on the K0 plane (linearity of the pipeline)."""
    rgb = oracle.fill_frame(w, h, 1, seed)
    yuv = emul_ctx.convert_rgb_yuv(rgb, w, cs)
    a, _ = emul_ctx.vectorscope(rgb=rgb, width=w, colorspace=cs, fuse_k0=True)
    b, _ = emul_ctx.vectorscope(yuv=yuv, width=w)
    assert np.array_equal(a, b)
    a, _ = emul_ctx.waveform(rgb=rgb, yuv=None, width=w, components=0x70, colorspace=cs, fuse_k0=True)
    b, _ = emul_ctx.waveform(rgb=rgb, yuv=yuv, width=w, components=0x70)
    assert np.array_equal(a, b)


@settings(**SETTINGS)
@given(w=st.integers(4, 64), h=st.integers(2, 40), cut=st.integers(1, 39), dist=st.integers(1, 5))
def test_row_tiles_add_up(emul_ctx, oracle, w, h, cut, dist):
    """Splitting a frame into two row tiles and saturating-adding the tables == the whole frame
    (what the multi-GPU path relies on); histogram counts add up exactly."""
    cut = min(cut, h - 1)
    rgb, yuv = pc.frames(oracle, w, h, dist)
    whole, _ = emul_ctx.vectorscope(yuv=yuv, width=w)
    top, _ = emul_ctx.vectorscope(yuv=np.ascontiguousarray(yuv[:cut]), width=w)
    bot, _ = emul_ctx.vectorscope(yuv=np.ascontiguousarray(yuv[cut:]), width=w)
    assert np.array_equal(np.minimum(top.astype(np.int32) + bot, 255).astype(np.uint8), whole)
    cw, _ = emul_ctx.histogram_counts(rgb=rgb, yuv=yuv, width=w, components=0x07)
    ct, _ = emul_ctx.histogram_counts(rgb=np.ascontiguousarray(rgb[:cut]), yuv=None, width=w, components=0x07)
    cb, _ = emul_ctx.histogram_counts(rgb=np.ascontiguousarray(rgb[cut:]), yuv=None, width=w, components=0x07)
    assert np.array_equal(ct + cb, cw)
