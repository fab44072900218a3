"""CPU oracle for the scope-accumulation path -- TEST INFRASTRUCTURE ONLY.

Two things live here:

* ``COracle``  -- ctypes binding of ``oracle/libscope_oracle.so`` (``scope_oracle.c``), the C
  restatement that tests, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use
  as the checker / the reported CPU baseline.
* ``np_*``     -- a second, independently written numpy restatement (bincount based, no
  per-pixel loop) used only to cross-check the C one.

PARITY STATUS: "parity unpinned" -- see ``scope_oracle.h``.  Nothing in the product path may
import this module.  Citations are relative to /root/reference (obs-color-monitor v0.9.5).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libscope_oracle.so")


class _Surface(C.Structure):
    _fields_ = [
        ("rgb_data", C.c_void_p),
        ("yuv_data", C.c_void_p),
        ("linesize", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("colorspace", C.c_int),
    ]


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, a second or two)."""
    src = os.path.join(_HERE, "scope_oracle.c")
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libscope_oracle.so"])
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class COracle:
    """ctypes front end of scope_oracle.c.  Frames are numpy uint8 arrays [H, linesize]."""

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.oracle_rgb_to_yuv.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_vectorscope.argtypes = [C.c_void_p, C.POINTER(_Surface)]
        L.oracle_waveform.argtypes = [C.c_void_p, C.POINTER(_Surface), C.c_uint32]
        L.oracle_histogram_counts.argtypes = [C.c_void_p, C.POINTER(_Surface), C.c_uint32]
        L.oracle_histogram_counts.restype = C.c_int
        L.oracle_histogram.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Surface), C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.oracle_histogram_level.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.oracle_display_vectorscope.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_display_vectorscope_zoom.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        L.oracle_display_waveform.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        L.oracle_display_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
        L.oracle_display_size.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_vectorscope_mt.argtypes = [C.c_void_p, C.POINTER(_Surface), C.c_uint32, C.c_uint32]
        L.oracle_waveform_mt.argtypes = [C.c_void_p, C.POINTER(_Surface), C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_histogram_counts_mt.argtypes = [C.c_void_p, C.POINTER(_Surface), C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_stage_region.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.oracle_stage_region.restype = C.c_int
        L.oracle_fill_frame.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64]
        L.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_fnv1a64.restype = C.c_uint64

    # -- helpers -----------------------------------------------------------------------------
    @staticmethod
    def _surface(rgb, yuv, width, colorspace):
        ref = rgb if rgb is not None else yuv
        assert ref is not None and ref.dtype == np.uint8 and ref.ndim == 2 and ref.flags.c_contiguous
        if rgb is not None and yuv is not None:
            assert rgb.shape == yuv.shape
        sd = _Surface(_ptr(rgb), _ptr(yuv), ref.shape[1], width, ref.shape[0], colorspace)
        return sd

    def fill_frame(self, width, height, dist, seed=0, linesize=None):
        linesize = linesize or 4 * width
        f = np.empty((height, linesize), dtype=np.uint8)
        self.lib.oracle_fill_frame(_ptr(f), linesize, width, height, dist, seed)
        return f

    def fnv(self, a) -> int:
        a = np.ascontiguousarray(a)
        return int(self.lib.oracle_fnv1a64(_ptr(a), a.nbytes))

    # -- the path ----------------------------------------------------------------------------
    def rgb_to_yuv(self, rgb, width, colorspace):
        out = np.full_like(rgb, 0xA5)
        self.lib.oracle_rgb_to_yuv(_ptr(rgb), rgb.shape[1], _ptr(out), out.shape[1], width, rgb.shape[0], colorspace)
        return out

    def vectorscope(self, yuv, width):
        out = np.empty((256, 256), dtype=np.uint8)
        sd = self._surface(None, yuv, width, 2)
        self.lib.oracle_vectorscope(_ptr(out), C.byref(sd))
        return out

    def waveform(self, rgb, yuv, width, components):
        out = np.empty((256, width, 4), dtype=np.uint8)
        sd = self._surface(rgb, yuv, width, 2)
        self.lib.oracle_waveform(_ptr(out), C.byref(sd), components)
        return out

    def histogram_counts(self, rgb, yuv, width, components):
        out = np.empty((256, 4), dtype=np.uint32)
        sd = self._surface(rgb, yuv, width, 2)
        self.lib.oracle_histogram_counts(_ptr(out), C.byref(sd), components)
        return out

    def histogram(self, rgb, yuv, width, components, logscale=0, level_fixed=0, level_ratio=0, hi_max_in=(0, 0, 0)):
        tex = np.empty(4096, dtype=np.uint8)
        hi = np.array(hi_max_in, dtype=np.uint32)
        sd = self._surface(rgb, yuv, width, 2)
        self.lib.oracle_histogram(_ptr(tex), _ptr(hi), C.byref(sd), components, logscale, level_fixed, level_ratio)
        return tex.view(np.float32).reshape(256, 4), hi

    def histogram_level(self, counts, width, height, components, logscale=0, level_fixed=0, level_ratio=0):
        tex = np.ascontiguousarray(counts, dtype=np.uint32).copy().view(np.uint8).reshape(-1)
        hi = np.zeros(3, dtype=np.uint32)
        self.lib.oracle_histogram_level(_ptr(tex), _ptr(hi), width, height, components, logscale, level_fixed, level_ratio)
        return tex.view(np.float32).reshape(256, 4), hi


    # -- multi-threaded courtesy baselines (same results) --------------------------------
    def vectorscope_mt(self, yuv, width, threads, reps=1):
        out = np.empty((256, 256), dtype=np.uint8)
        sd = self._surface(None, yuv, width, 2)
        assert self.lib.oracle_vectorscope_mt(_ptr(out), C.byref(sd), threads, reps) == 0
        return out

    def waveform_mt(self, rgb, yuv, width, components, threads, reps=1):
        out = np.empty((256, width, 4), dtype=np.uint8)
        sd = self._surface(rgb, yuv, width, 2)
        assert self.lib.oracle_waveform_mt(_ptr(out), C.byref(sd), components, threads, reps) == 0
        return out

    def histogram_counts_mt(self, rgb, yuv, width, components, threads, reps=1):
        out = np.empty((256, 4), dtype=np.uint32)
        sd = self._surface(rgb, yuv, width, 2)
        assert self.lib.oracle_histogram_counts_mt(_ptr(out), C.byref(sd), components, threads, reps) == 0
        return out

    def stage_region(self, plane, width, scale, x=0, y=0, cx=0, cy=0):
        """Scale + crop of cm_render_target; cx == 0 or cy == 0 selects the whole scaled frame."""
        h = plane.shape[0]
        cx = cx or width // scale
        cy = cy or h // scale
        out = np.empty((cy, cx * 4), dtype=np.uint8)
        rc = self.lib.oracle_stage_region(_ptr(plane), plane.shape[1], width, h, scale, x, y, cx, cy, _ptr(out), cx * 4)
        if rc:
            raise ValueError("bad region")
        return out

    # -- display shaders D1-D3 -----------------------------------------------------------------
    def display_size(self, base_w, base_h, components, display):
        ow, oh = C.c_uint32(), C.c_uint32()
        self.lib.oracle_display_size(base_w, base_h, components, display, C.byref(ow), C.byref(oh))
        return int(ow.value), int(oh.value)

    def display_vectorscope(self, table, intensity=25, color_type=1, colorspace=2, zoom=1.0):
        out = np.empty((256, 256, 4), dtype=np.uint8)
        t = np.ascontiguousarray(table, dtype=np.uint8)
        self.lib.oracle_display_vectorscope_zoom(_ptr(out), _ptr(t), intensity, color_type, colorspace, zoom)
        return out

    def display_waveform(self, table, components, display, intensity=25):
        t = np.ascontiguousarray(table, dtype=np.uint8)
        ow, oh = self.display_size(t.shape[1], 256, components, display)
        out = np.empty((oh, ow, 4), dtype=np.uint8)
        self.lib.oracle_display_waveform(_ptr(out), _ptr(t), t.shape[1], components, display, intensity)
        return out

    def display_histogram(self, tex, hi_max, components, display, level_height=200):
        t = np.ascontiguousarray(tex, dtype=np.float32)
        hi = np.ascontiguousarray(hi_max, dtype=np.uint32)
        ow, oh = self.display_size(256, level_height, components, display)
        out = np.empty((oh, ow, 4), dtype=np.uint8)
        self.lib.oracle_display_histogram(_ptr(out), t.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
                                          components, display, level_height)
        return out


# ------------------------------------------------------------------------------------------------
# Independent numpy restatement (cross-check of the C one; not used as the checker itself)
# ------------------------------------------------------------------------------------------------

_K0 = {
    # data/common.effect:27-29 / :38-40 -- rows U, Y, V; columns r, g, b
    1: np.array([[-0.147643, -0.289855, 0.437500], [0.299, 0.587, 0.114], [0.437500, -0.366351, -0.071147]], np.float32),
    2: np.array([[-0.100643, -0.338571, 0.439216], [0.2126, 0.7152, 0.0722], [0.439216, -0.398941, -0.040273]], np.float32),
}


def _texels(frame, width):
    """[H, linesize] uint8 -> [H, width, 4] view of the visible texels."""
    h = frame.shape[0]
    return frame[:, : 4 * width].reshape(h, width, 4)


def np_rgb_to_yuv(rgb, width, colorspace):
    """data/common.effect:23-43 with the UNORM8 store rule floor(clamp01(x)*255+0.5)."""
    m = _K0[1 if colorspace == 1 else 2]
    t = _texels(rgb, width)
    f255 = np.float32(255.0)
    r = t[..., 2].astype(np.float32) / f255
    g = t[..., 1].astype(np.float32) / f255
    b = t[..., 0].astype(np.float32) / f255

    def lin(row, tail):
        acc = m[row, 0] * r
        acc = acc + m[row, 1] * g
        acc = acc + m[row, 2] * b
        for c in tail:
            acc = acc + np.float32(c)
        return acc

    def q(x):
        x = np.clip(x, np.float32(0), np.float32(1))
        return np.floor(x * f255 + np.float32(0.5)).astype(np.uint8)

    u = q(lin(0, (0.5, -1.0 / 256.0)))
    y = q(lin(1, ()))
    v = q(lin(2, (0.5,)))
    out = np.full_like(rgb, 0xA5)
    o = _texels(out, width)
    o[..., 0], o[..., 1], o[..., 2], o[..., 3] = u, y, v, 255
    return out


def np_vectorscope(yuv, width):
    """src/vectorscope.c:217-238 as min(bincount, 255)."""
    t = _texels(yuv, width)
    idx = t[..., 0].astype(np.int64) + 256 * (255 - t[..., 2].astype(np.int64))
    cnt = np.bincount(idx.ravel(), minlength=65536)
    return np.minimum(cnt, 255).astype(np.uint8).reshape(256, 256)


def _plane(rgb, yuv, components):
    if components & 0x07:
        return rgb
    if components & 0x70:
        return yuv
    return None


def np_waveform(rgb, yuv, width, components):
    """src/waveform.c:220-257 as per-(value, column) bincounts clamped at 255."""
    out = np.zeros((256, width, 4), dtype=np.uint8)
    plane = _plane(rgb, yuv, components)
    if plane is None:
        return out
    t = _texels(plane, width)
    keep = t[..., 3] != 0
    cols = np.broadcast_to(np.arange(width, dtype=np.int64), keep.shape)
    for c in range(3):
        if not components & (0x11 << c):
            continue
        key = (255 - t[..., c].astype(np.int64)) * width + cols
        cnt = np.bincount(key[keep], minlength=256 * width)
        out[..., c] = np.minimum(cnt, 255).reshape(256, width)
    return out


def np_histogram_counts(rgb, yuv, width, components):
    """src/histogram.c:357-395; slots R,G,B,unused."""
    out = np.zeros((256, 4), dtype=np.uint32)
    plane = _plane(rgb, yuv, components)
    if plane is None:
        return out
    t = _texels(plane, width)
    keep = t[..., 3] != 0
    for slot, byte, mask in ((0, 2, 0x44), (1, 1, 0x22), (2, 0, 0x11)):
        if components & mask:
            out[:, slot] = np.bincount(t[..., byte][keep], minlength=256)
    return out


def np_stage_region(plane, width, scale, x=0, y=0, cx=0, cy=0):
    """Scale + ROI crop of cm_render_target (src/common.c:249-250,272-282), point-sampled at cell centres."""
    h = plane.shape[0]
    cx = cx or width // scale
    cy = cy or h // scale
    t = _texels(plane, width)  # [h][width][4]
    rows = (np.arange(cy) + y) * scale + scale // 2
    cols = (np.arange(cx) + x) * scale + scale // 2
    return np.ascontiguousarray(t[rows][:, cols]).reshape(cy, cx * 4)
