"""ctypes front end of libscope_hip.so (C-ABI: include/scope_hip.h).

Used by the tests, bench.py and __graft_entry__.py to drive the product path exactly as a C host
would: every call below is one C-ABI call.  There is no Python or CPU implementation of the path
here -- if the shared library is missing or no HIP device is usable this module raises.

Host buffers are numpy uint8 arrays shaped [height, linesize]; device buffers are anything with a
``data_ptr()`` (torch tensors on the GPU), which PyTorch-ROCm provides together with streams and
``torch.distributed``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libscope_hip.so")

OK, EINVAL, ENOMEM, ENODEV, EIO, ENODATA, EAGAIN = 0, -22, -12, -19, -5, -61, -11
IN_DEVICE, OUT_DEVICE, ASYNC, FUSE_K0, OUT_PACKED16, SIDE_REDUCE = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
COMP_RGB, COMP_Y, COMP_UV, COMP_YUV = 0x07, 0x20, 0x50, 0x70

# every symbol include/scope_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "scope_abi_version", "scope_strerror", "scope_last_error",
    "scope_ctx_create", "scope_ctx_destroy", "scope_ctx_set_stream", "scope_ctx_get_stream", "scope_ctx_sync", "scope_ctx_fence",
    "scope_ctx_device_info", "scope_ctx_set_option", "scope_ctx_get_stat",
    "scope_convert_rgb_yuv",
    "scope_vectorscope", "scope_waveform", "scope_histogram_counts", "scope_histogram", "scope_histogram_level",
    "scope_vectorscope_batch", "scope_waveform_batch", "scope_histogram_counts_batch",
    "scope_waveform_histogram", "scope_waveform_histogram_batch", "scope_three_scopes_batch",
    "scope_table_widen_u8_u32", "scope_table_clamp_u32_u8",
    "scope_table_pack_u8_u16x2", "scope_table_unpack_clamp_u16x2_u8",
    "scope_ctx_profile", "scope_ctx_kernel_ms",
    "scope_pipe_create", "scope_pipe_destroy", "scope_pipe_submit", "scope_pipe_collect", "scope_pipe_size",
    "scope_region_size", "scope_stage_region", "scope_surface_begin", "scope_surface_end",
    "scope_rccl_unique_id", "scope_rccl_comm_create", "scope_rccl_comm_destroy", "scope_table_allreduce", "scope_rccl_comm_info",
    "scope_frame_pool_alloc", "scope_frame_pool_free",
    "scope_display_graticule", "scope_graticule_label_strip",
    "scope_display_vectorscope", "scope_display_vectorscope_zoom", "scope_display_waveform", "scope_display_histogram", "scope_display_size",
)


class SurfaceData(C.Structure):
    """struct scope_surface_data == struct cm_surface_data (src/common.h:24-30)."""

    _fields_ = [
        ("rgb_data", C.c_void_p),
        ("yuv_data", C.c_void_p),
        ("linesize", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("colorspace", C.c_int),
        ("tex", C.c_void_p),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("planes", C.POINTER(C.c_void_p)),
        ("n_frames", C.c_uint32),
        ("linesize", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("planes_are_rgb", C.c_int),
        ("colorspace", C.c_int),
    ]


class Region(C.Structure):
    """struct scope_region: target_scale and the ROI rectangle in scaled texels (cx or cy == 0: whole frame)."""

    _fields_ = [("scale", C.c_uint32), ("x", C.c_uint32), ("y", C.c_uint32), ("cx", C.c_uint32), ("cy", C.c_uint32)]


class PipeConfig(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32), ("colorspace", C.c_int),
                ("wvs_components", C.c_uint32), ("his_components", C.c_uint32), ("want_vectorscope", C.c_int),
                ("region", Region)]


class DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128),
        ("arch", C.c_char * 32),
        ("device", C.c_int),
        ("compute_units", C.c_int),
        ("lds_bytes_per_cu", C.c_int),
        ("clock_khz", C.c_int),
        ("hbm_bytes", C.c_uint64),
    ]


class ScopeError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"scope_hip status {status}: {msg}")
        self.status = status


def load_library(path: str | None = None) -> C.CDLL:
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the scope path)")
    L = C.CDLL(path)
    vp, u32, i32, u64 = C.c_void_p, C.c_uint32, C.c_int, C.c_uint64
    sig = {
        "scope_abi_version": (i32, []),
        "scope_strerror": (C.c_char_p, [i32]),
        "scope_last_error": (C.c_char_p, [vp]),
        "scope_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "scope_ctx_destroy": (None, [vp]),
        "scope_ctx_set_stream": (i32, [vp, vp]),
        "scope_ctx_get_stream": (vp, [vp]),
        "scope_ctx_sync": (i32, [vp]),
        "scope_ctx_fence": (i32, [vp]),
        "scope_ctx_device_info": (i32, [vp, C.POINTER(DeviceInfo)]),
        "scope_ctx_set_option": (i32, [vp, C.c_char_p, i32]),
        "scope_ctx_get_stat": (i32, [vp, C.c_char_p, C.POINTER(u64)]),
        "scope_convert_rgb_yuv": (i32, [vp, vp, u32, vp, u32, u32, u32, i32, u32]),
        "scope_vectorscope": (i32, [vp, C.POINTER(SurfaceData), u32, vp]),
        "scope_waveform": (i32, [vp, C.POINTER(SurfaceData), u32, u32, vp]),
        "scope_histogram_counts": (i32, [vp, C.POINTER(SurfaceData), u32, u32, vp]),
        "scope_histogram": (i32, [vp, C.POINTER(SurfaceData), u32, i32, i32, i32, u32, vp, vp]),
        "scope_histogram_level": (None, [vp, vp, u32, u32, u32, i32, i32, i32]),
        "scope_vectorscope_batch": (i32, [vp, C.POINTER(Batch), u32, vp]),
        "scope_waveform_batch": (i32, [vp, C.POINTER(Batch), u32, u32, vp]),
        "scope_histogram_counts_batch": (i32, [vp, C.POINTER(Batch), u32, u32, vp]),
        "scope_waveform_histogram": (i32, [vp, C.POINTER(SurfaceData), u32, u32, u32, vp, vp]),
        "scope_waveform_histogram_batch": (i32, [vp, C.POINTER(Batch), u32, u32, u32, vp, vp]),
        "scope_three_scopes_batch": (i32, [vp, C.POINTER(Batch), u32, u32, u32, vp, vp, vp]),
        "scope_table_widen_u8_u32": (i32, [vp, vp, vp, u64]),
        "scope_table_clamp_u32_u8": (i32, [vp, vp, vp, u64]),
        "scope_table_pack_u8_u16x2": (i32, [vp, vp, vp, u64]),
        "scope_table_unpack_clamp_u16x2_u8": (i32, [vp, vp, vp, u64]),
        "scope_rccl_unique_id": (i32, [vp]),
        "scope_rccl_comm_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "scope_rccl_comm_destroy": (i32, [vp]),
        "scope_table_allreduce": (i32, [vp, vp, vp, u64]),
        "scope_rccl_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "scope_frame_pool_alloc": (i32, [vp, u64, u32, C.POINTER(vp), C.POINTER(C.c_float), C.POINTER(u32)]),
        "scope_frame_pool_free": (i32, [vp, vp]),
        "scope_ctx_profile": (i32, [vp, i32]),
        "scope_ctx_kernel_ms": (i32, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
        "scope_display_vectorscope": (i32, [vp, vp, i32, i32, i32, u32, vp]),
        "scope_display_vectorscope_zoom": (i32, [vp, vp, i32, i32, i32, C.c_float, u32, vp]),
        "scope_display_waveform": (i32, [vp, vp, u32, u32, i32, i32, u32, vp]),
        "scope_display_histogram": (i32, [vp, vp, vp, u32, i32, u32, u32, vp]),
        "scope_display_size": (None, [u32, u32, u32, i32, C.POINTER(u32), C.POINTER(u32)]),
        "scope_display_graticule": (i32, [vp, vp, u32, u32, vp, u32, vp, u32, u32, u32]),
        "scope_graticule_label_strip": (None, [vp]),
        "scope_pipe_create": (i32, [vp, C.POINTER(PipeConfig), C.POINTER(vp)]),
        "scope_pipe_destroy": (None, [vp]),
        "scope_pipe_submit": (i32, [vp, vp, u32]),
        "scope_pipe_collect": (i32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "scope_pipe_size": (i32, [vp, C.POINTER(u32), C.POINTER(u32)]),
        "scope_surface_begin": (i32, [vp, u32]),
        "scope_surface_end": (i32, [vp, C.POINTER(u32), C.POINTER(u32)]),
        "scope_region_size": (i32, [u32, u32, C.POINTER(Region), C.POINTER(u32), C.POINTER(u32)]),
        "scope_stage_region": (i32, [vp, vp, u32, u32, u32, C.POINTER(Region), vp, u32, u32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    return L


def _hptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags.c_contiguous
    return a.ctypes.data_as(C.c_void_p)


def _dptr(t):
    return C.c_void_p(int(t.data_ptr())) if hasattr(t, "data_ptr") else C.c_void_p(int(t))


class ScopeContext:
    """One scope_ctx of libscope_hip.so (the default, and the only library the product ships: without it, or without a
    HIP device, construction raises -- there is no CPU fallback).  ``lib_path`` exists for tests/ only: they point the same
    binding at tests/emul/libscope_emul.so, the kernel sources compiled against a CPU stand-in of the HIP runtime."""

    def __init__(self, device: int = 0, lib_path: str | None = None):
        self.device, self.lib_path = device, lib_path
        self.L = load_library(lib_path)
        h = C.c_void_p()
        st = self.L.scope_ctx_create(device, C.byref(h))
        if st != OK:
            raise ScopeError(st, self.L.scope_strerror(st).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.scope_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st, allow=()):
        if st != OK and st not in allow:
            raise ScopeError(st, f"{self.L.scope_strerror(st).decode()} ({self.L.scope_last_error(self.h).decode()})")
        return st

    # -- context -------------------------------------------------------------------------------
    def device_info(self) -> dict:
        di = DeviceInfo()
        self._check(self.L.scope_ctx_device_info(self.h, C.byref(di)))
        return {"name": di.name.decode(), "arch": di.arch.decode(), "device": di.device,
                "compute_units": di.compute_units, "lds_bytes_per_cu": di.lds_bytes_per_cu,
                "clock_khz": di.clock_khz, "hbm_bytes": int(di.hbm_bytes)}

    def set_option(self, key: str, value: int):
        self._check(self.L.scope_ctx_set_option(self.h, key.encode(), int(value)))

    def get_stat(self, key: str) -> int:
        """scope_ctx_get_stat: which form this context's launches took ("vss.rows_launches", "vss.fold_launches", ...)."""
        v = C.c_uint64()
        self._check(self.L.scope_ctx_get_stat(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    def set_stream(self, hip_stream: int | None):
        """Run on the caller's hipStream_t.  A NULL/0 handle selects the context's OWN stream, which does
        not synchronise with torch's default stream: pass an explicit torch.cuda.Stream().cuda_stream."""
        self._check(self.L.scope_ctx_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    def get_stream(self) -> int:
        return int(self.L.scope_ctx_get_stream(self.h) or 0)

    def sync(self):
        self._check(self.L.scope_ctx_sync(self.h))

    def fence(self):
        """The context's stream waits (on the device) for the side stream's reductions of side_reduce=True calls."""
        self._check(self.L.scope_ctx_fence(self.h))

    # -- host-buffer calls (the plugin's situation: a mapped staging surface) --------------------
    @staticmethod
    def _sd(rgb, yuv, width, colorspace):
        ref = rgb if rgb is not None else yuv
        h, ls = (ref.shape[0], ref.shape[1]) if ref is not None else (0, 0)
        return SurfaceData(_hptr(rgb), _hptr(yuv), ls, width, h, colorspace, None)

    def convert_rgb_yuv(self, rgb, width, colorspace=2, out_linesize=None):
        h = rgb.shape[0]
        out = np.full((h, out_linesize or rgb.shape[1]), 0xA5, dtype=np.uint8)
        self._check(self.L.scope_convert_rgb_yuv(self.h, _hptr(rgb), rgb.shape[1], _hptr(out), out.shape[1], width, h,
                                                 colorspace, 0))
        return out

    def surface_begin(self, fuse_k0=False):
        """Until surface_end(): host planes handed to the single-surface calls are uploaded once and reused."""
        self._check(self.L.scope_surface_begin(self.h, FUSE_K0 if fuse_k0 else 0))

    def surface_end(self):
        """Returns (planes uploaded, uploads saved)."""
        up, hit = C.c_uint32(), C.c_uint32()
        self._check(self.L.scope_surface_end(self.h, C.byref(up), C.byref(hit)))
        return int(up.value), int(hit.value)

    def region_size(self, width, height, scale, x=0, y=0, cx=0, cy=0):
        r = Region(scale, x, y, cx, cy)
        ow, oh = C.c_uint32(), C.c_uint32()
        st = self.L.scope_region_size(width, height, C.byref(r), C.byref(ow), C.byref(oh))
        if st != OK:
            raise ScopeError(st, "bad region")
        return int(ow.value), int(oh.value)

    def stage_region(self, plane, width, scale, x=0, y=0, cx=0, cy=0, out_linesize=None):
        """target_scale + ROI crop of a host plane (cm_render_target's pre-steps); returns the staged plane."""
        h = plane.shape[0]
        ow, oh = self.region_size(width, h, scale, x, y, cx, cy)
        out = np.full((oh, out_linesize or ow * 4), 0xA5, dtype=np.uint8)
        r = Region(scale, x, y, cx, cy)
        self._check(self.L.scope_stage_region(self.h, _hptr(plane), plane.shape[1], width, h, C.byref(r), _hptr(out),
                                              out.shape[1], 0))
        return out

    def stage_region_dev(self, src_dev, src_linesize, width, height, region, dst_dev, dst_linesize, sync=True):
        r = Region(*region)
        self._check(self.L.scope_stage_region(self.h, _dptr(src_dev), src_linesize, width, height, C.byref(r),
                                              _dptr(dst_dev), dst_linesize, IN_DEVICE | OUT_DEVICE | (0 if sync else ASYNC)))

    def vectorscope(self, yuv=None, rgb=None, width=0, colorspace=2, fuse_k0=False, out=None):
        out = np.full((256, 256), 0xEE, dtype=np.uint8) if out is None else out
        sd = self._sd(rgb, yuv, width, colorspace)
        st = self._check(self.L.scope_vectorscope(self.h, C.byref(sd), FUSE_K0 if fuse_k0 else 0, _hptr(out)),
                         allow=(ENODATA,))
        return out, st

    def waveform(self, rgb=None, yuv=None, width=0, components=COMP_RGB, colorspace=2, fuse_k0=False, out=None):
        out = np.full((256, width, 4), 0xEE, dtype=np.uint8) if out is None else out
        sd = self._sd(rgb, yuv, width, colorspace)
        st = self._check(self.L.scope_waveform(self.h, C.byref(sd), components, FUSE_K0 if fuse_k0 else 0, _hptr(out)),
                         allow=(ENODATA,))
        return out, st

    def histogram_counts(self, rgb=None, yuv=None, width=0, components=COMP_RGB, colorspace=2, fuse_k0=False):
        out = np.full((256, 4), 0xEEEEEEEE, dtype=np.uint32)
        sd = self._sd(rgb, yuv, width, colorspace)
        st = self._check(self.L.scope_histogram_counts(self.h, C.byref(sd), components, FUSE_K0 if fuse_k0 else 0,
                                                       out.ctypes.data_as(C.c_void_p)), allow=(ENODATA,))
        return out, st

    def waveform_histogram(self, rgb=None, yuv=None, width=0, wvs_components=COMP_RGB, his_components=COMP_RGB,
                           colorspace=2, fuse_k0=False):
        """K2 + K3 of one surface in one pass where the masks allow; (waveform table, counts, status)."""
        out = np.full((256, width, 4), 0xEE, dtype=np.uint8)
        cnt = np.full((256, 4), 0xEEEEEEEE, dtype=np.uint32)
        sd = self._sd(rgb, yuv, width, colorspace)
        st = self._check(self.L.scope_waveform_histogram(self.h, C.byref(sd), wvs_components, his_components,
                                                         FUSE_K0 if fuse_k0 else 0, _hptr(out),
                                                         cnt.ctypes.data_as(C.c_void_p)), allow=(ENODATA,))
        return out, cnt, st

    def waveform_histogram_batch_dev(self, batch, wvs_components, his_components, wvs_dev, his_dev, sync=True):
        self._check(self.L.scope_waveform_histogram_batch(self.h, C.byref(batch), wvs_components, his_components,
                                                          IN_DEVICE | OUT_DEVICE | (0 if sync else ASYNC),
                                                          _dptr(wvs_dev), _dptr(his_dev)))

    def three_scopes_batch_dev(self, batch, wvs_components, his_components, vss_dev, wvs_dev, his_dev, sync=True):
        """The dock's three scopes of every frame of a batch of RGB planes; small launches read the planes once."""
        self._check(self.L.scope_three_scopes_batch(self.h, C.byref(batch), wvs_components, his_components,
                                                    IN_DEVICE | OUT_DEVICE | (0 if sync else ASYNC),
                                                    _dptr(vss_dev), _dptr(wvs_dev), _dptr(his_dev)))

    def three_scopes(self, rgb, width, wvs_components=COMP_RGB, his_components=COMP_RGB, colorspace=2):
        """Host planes in, host tables out (one frame): (vectorscope, waveform, histogram counts, status)."""
        rgb = np.ascontiguousarray(rgb)
        h, linesize = rgb.shape
        ptrs = (C.c_void_p * 1)(rgb.ctypes.data)
        b = Batch(C.cast(ptrs, C.POINTER(C.c_void_p)), 1, linesize, width, h, 1, colorspace)
        vss = np.full((256, 256), 0xEE, dtype=np.uint8)
        wvs = np.full((256, width, 4), 0xEE, dtype=np.uint8)
        his = np.full((256, 4), 0xEEEEEEEE, dtype=np.uint32)
        st = self.L.scope_three_scopes_batch(self.h, C.byref(b), wvs_components, his_components, 0,
                                             vss.ctypes.data, wvs.ctypes.data, his.ctypes.data)
        return vss, wvs, his, st

    def histogram(self, rgb=None, yuv=None, width=0, components=COMP_RGB, logscale=0, level_fixed=0, level_ratio=0,
                  colorspace=2, fuse_k0=False, hi_max_in=(0, 0, 0)):
        tex = np.full(4096, 0xEE, dtype=np.uint8)
        hi = np.array(hi_max_in, dtype=np.uint32)
        sd = self._sd(rgb, yuv, width, colorspace)
        st = self._check(self.L.scope_histogram(self.h, C.byref(sd), components, logscale, level_fixed, level_ratio,
                                                FUSE_K0 if fuse_k0 else 0, _hptr(tex), hi.ctypes.data_as(C.c_void_p)),
                         allow=(ENODATA,))
        return tex.view(np.float32).reshape(256, 4), hi, st

    def histogram_level(self, counts, width, height, components, logscale=0, level_fixed=0, level_ratio=0):
        tex = np.ascontiguousarray(counts, dtype=np.uint32).copy().view(np.uint8).reshape(-1)
        hi = np.zeros(3, dtype=np.uint32)
        self.L.scope_histogram_level(_hptr(tex), hi.ctypes.data_as(C.c_void_p), width, height, components, logscale,
                                     level_fixed, level_ratio)
        return tex.view(np.float32).reshape(256, 4), hi

    # -- batched calls, host or device planes ------------------------------------------------------
    def _batch(self, plane_ptrs, linesize, width, height, planes_are_rgb, colorspace):
        arr = (C.c_void_p * len(plane_ptrs))(*plane_ptrs)
        b = Batch(C.cast(arr, C.POINTER(C.c_void_p)), len(plane_ptrs), linesize, width, height, int(planes_are_rgb),
                  colorspace)
        b._keep = arr
        return b

    def vectorscope_batch_host(self, frames, width, planes_are_rgb=False, colorspace=2):
        n, (h, ls) = len(frames), frames[0].shape
        out = np.full((n, 256, 256), 0xEE, dtype=np.uint8)
        b = self._batch([f.ctypes.data for f in frames], ls, width, h, planes_are_rgb, colorspace)
        self._check(self.L.scope_vectorscope_batch(self.h, C.byref(b), 0, _hptr(out)))
        return out

    def waveform_batch_host(self, frames, width, components, planes_are_rgb=True, colorspace=2):
        n, (h, ls) = len(frames), frames[0].shape
        out = np.full((n, 256, width, 4), 0xEE, dtype=np.uint8)
        b = self._batch([f.ctypes.data for f in frames], ls, width, h, planes_are_rgb, colorspace)
        st = self._check(self.L.scope_waveform_batch(self.h, C.byref(b), components, 0, _hptr(out)), allow=(ENODATA,))
        return out, st

    def histogram_counts_batch_host(self, frames, width, components, planes_are_rgb=True, colorspace=2):
        n, (h, ls) = len(frames), frames[0].shape
        out = np.full((n, 256, 4), 0xEEEEEEEE, dtype=np.uint32)
        b = self._batch([f.ctypes.data for f in frames], ls, width, h, planes_are_rgb, colorspace)
        st = self._check(self.L.scope_histogram_counts_batch(self.h, C.byref(b), components, 0,
                                                             out.ctypes.data_as(C.c_void_p)), allow=(ENODATA,))
        return out, st

    def make_device_batch(self, frames_dev, linesize, width, height, planes_are_rgb=False, colorspace=2):
        """frames_dev: device tensor [n, height, linesize] uint8 (or a list of plane tensors)."""
        if hasattr(frames_dev, "data_ptr"):
            base, n = int(frames_dev.data_ptr()), int(frames_dev.shape[0])
            ptrs = [base + i * height * linesize for i in range(n)]
        else:
            ptrs = [int(t.data_ptr()) for t in frames_dev]
        return self._batch(ptrs, linesize, width, height, planes_are_rgb, colorspace)

    def vectorscope_batch_dev(self, batch, out_dev, sync=True, packed16=False, side_reduce=False):
        """packed16: out_dev is int32[n][32768], two bins per word (the wire format of TableAllReduce).
        side_reduce (with sync=False): the partial-table reduction goes to the context's second stream; the table is
        complete after fence() / sync() / the next call without it."""
        self._check(self.L.scope_vectorscope_batch(self.h, C.byref(batch),
                                                   IN_DEVICE | OUT_DEVICE | (0 if sync else ASYNC) | (OUT_PACKED16 if packed16 else 0) |
                                                   (SIDE_REDUCE if side_reduce else 0), _dptr(out_dev)))

    def waveform_batch_dev(self, batch, components, out_dev, sync=True):
        self._check(self.L.scope_waveform_batch(self.h, C.byref(batch), components,
                                                IN_DEVICE | OUT_DEVICE | (0 if sync else ASYNC), _dptr(out_dev)))

    def histogram_counts_batch_dev(self, batch, components, out_dev, sync=True):
        self._check(self.L.scope_histogram_counts_batch(self.h, C.byref(batch), components,
                                                        IN_DEVICE | OUT_DEVICE | (0 if sync else ASYNC), _dptr(out_dev)))

    def convert_rgb_yuv_dev(self, rgb_dev, rgb_linesize, yuv_dev, yuv_linesize, width, height, colorspace=2, sync=True):
        self._check(self.L.scope_convert_rgb_yuv(self.h, _dptr(rgb_dev), rgb_linesize, _dptr(yuv_dev), yuv_linesize,
                                                 width, height, colorspace, IN_DEVICE | OUT_DEVICE | (0 if sync else ASYNC)))

    def widen_dev(self, in_dev, out_dev, n):
        self._check(self.L.scope_table_widen_u8_u32(self.h, _dptr(in_dev), _dptr(out_dev), n))

    def clamp_dev(self, in_dev, out_dev, n):
        self._check(self.L.scope_table_clamp_u32_u8(self.h, _dptr(in_dev), _dptr(out_dev), n))

    # -- display post-passes (host tables in, host RGBA pictures out) --------------------------------
    def display_size(self, base_w, base_h, components, display):
        ow, oh = C.c_uint32(), C.c_uint32()
        self.L.scope_display_size(base_w, base_h, components, display, C.byref(ow), C.byref(oh))
        return int(ow.value), int(oh.value)

    def display_vectorscope(self, table, intensity=25, color_type=1, colorspace=2, zoom=1.0):
        out = np.full((256, 256, 4), 0xEE, dtype=np.uint8)
        t = np.ascontiguousarray(table, dtype=np.uint8)
        self._check(self.L.scope_display_vectorscope_zoom(self.h, _hptr(t), intensity, color_type, colorspace, zoom, 0,
                                                         _hptr(out)))
        return out

    def display_waveform(self, table, components, display, intensity=25):
        t = np.ascontiguousarray(table, dtype=np.uint8)
        w = t.shape[1]
        ow, oh = self.display_size(w, 256, components, display)
        out = np.full((oh, ow, 4), 0xEE, dtype=np.uint8)
        self._check(self.L.scope_display_waveform(self.h, _hptr(t), w, components, display, intensity, 0, _hptr(out)))
        return out

    def display_histogram(self, tex, hi_max, components, display, level_height=200):
        t = np.ascontiguousarray(tex, dtype=np.float32)
        hi = np.ascontiguousarray(hi_max, dtype=np.uint32)
        ow, oh = self.display_size(256, level_height, components, display)
        out = np.full((oh, ow, 4), 0xEE, dtype=np.uint8)
        self._check(self.L.scope_display_histogram(self.h, t.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
                                                   components, display, level_height, 0, _hptr(out)))
        return out

    def display_graticule(self, picture, labels=None, lines=None, argb=0x80FFBF00):
        """Blends label quads (float32 [n, 8]: x, y, w, h, u0, v0, u1, v1) and GS_LINES segments (float32 [2k, 2]) into
        an RGBA8 host picture [h, w, 4]; returns the new picture."""
        pic = np.ascontiguousarray(picture, dtype=np.uint8).copy()
        h, w = pic.shape[0], pic.shape[1]
        lab = np.ascontiguousarray(labels if labels is not None else np.zeros((0, 8)), dtype=np.float32)
        lin = np.ascontiguousarray(lines if lines is not None else np.zeros((0, 2)), dtype=np.float32)
        self._check(self.L.scope_display_graticule(self.h, pic.ctypes.data_as(C.c_void_p), w, h, lab.ctypes.data_as(C.c_void_p),
                                                   lab.shape[0], lin.ctypes.data_as(C.c_void_p), lin.shape[0], argb, 0))
        return pic

    def graticule_label_strip(self):
        strip = np.zeros((16, 96), dtype=np.uint8)
        self.L.scope_graticule_label_strip(strip.ctypes.data_as(C.c_void_p))
        return strip

    def pack16_dev(self, in_dev, out_dev, n_pairs):
        self._check(self.L.scope_table_pack_u8_u16x2(self.h, _dptr(in_dev), _dptr(out_dev), n_pairs))

    def unpack16_clamp_dev(self, in_dev, out_dev, n_pairs):
        self._check(self.L.scope_table_unpack_clamp_u16x2_u8(self.h, _dptr(in_dev), _dptr(out_dev), n_pairs))

    # -- the table sum over RCCL through the C-ABI (no torch.distributed in the data path) ---------------------------
    def rccl_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        st = self.L.scope_rccl_unique_id(C.cast(buf, C.c_void_p))
        if st != OK:
            raise ScopeError(st, "scope_rccl_unique_id")
        return bytes(buf)

    def rccl_comm_create(self, unique_id: bytes, n_ranks: int, rank: int) -> int:
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        comm = C.c_void_p()
        self._check(self.L.scope_rccl_comm_create(self.h, C.cast(buf, C.c_void_p), n_ranks, rank, C.byref(comm)))
        return int(comm.value)

    def rccl_comm_destroy(self, comm: int):
        self.L.scope_rccl_comm_destroy(C.c_void_p(comm))

    def rccl_comm_info(self, comm: int) -> dict:
        """What the collective library reports for `comm`: {"ranks", "rank", "version"} (-1 = not offered)."""
        n, r, v = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        st = self.L.scope_rccl_comm_info(C.c_void_p(comm), C.byref(n), C.byref(r), C.byref(v))
        if st != OK:
            raise ScopeError(st, "scope_rccl_comm_info")
        return {"ranks": n.value, "rank": r.value, "version": v.value}

    def table_allreduce(self, comm: int, words_dev, n_words: int):
        """In-place uint32 sum over the communicator, enqueued on the context's stream."""
        self._check(self.L.scope_table_allreduce(self.h, C.c_void_p(comm), _dptr(words_dev), n_words))

    # -- frame pools placed by the library (scope_frame_pool_alloc) ---------------------------------------------------
    def frame_pool_alloc(self, n_bytes: int, candidates: int = 6):
        """(FramePool, read GB/s per candidate -- the kept one first).  FramePool.tensor(torch, shape) views the memory."""
        ptr, rates, tried = C.c_void_p(), (C.c_float * 16)(), C.c_uint32()
        self._check(self.L.scope_frame_pool_alloc(self.h, n_bytes, candidates, C.byref(ptr), rates, C.byref(tried)))
        return FramePool(self, int(ptr.value), n_bytes), [round(float(rates[i]), 1) for i in range(tried.value)]

    def profile(self, enable: bool):
        self._check(self.L.scope_ctx_profile(self.h, int(enable)))

    def kernel_ms(self):
        """(number of bracketed accumulation launches, their summed duration in ms)."""
        n, ms = C.c_uint32(), C.c_float()
        self._check(self.L.scope_ctx_kernel_ms(self.h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)


class FramePool:
    """Device memory from scope_frame_pool_alloc; exposes the CUDA array interface so that torch can view it."""

    def __init__(self, ctx: ScopeContext, ptr: int, n_bytes: int):
        self.ctx, self.ptr, self.n_bytes = ctx, ptr, n_bytes
        self.__cuda_array_interface__ = {"shape": (n_bytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    def tensor(self, torch, shape):
        """A uint8 torch tensor over the pool (the pool must outlive it: keep this object)."""
        t = torch.as_tensor(self, device="cuda")
        return t[: int(np.prod(shape))].view(*shape)

    def free(self):
        if self.ptr:
            self.ctx._check(self.ctx.L.scope_frame_pool_free(self.ctx.h, C.c_void_p(self.ptr)))
            self.ptr = 0


class ScopePipe:
    """scope_pipe: `depth` frames in flight over upload / compute / download streams."""

    def __init__(self, ctx: ScopeContext, width, height, depth=3, colorspace=2, wvs_components=COMP_RGB,
                 his_components=COMP_RGB, want_vectorscope=True, region=(0, 0, 0, 0, 0)):
        """region = (target_scale, x, y, cx, cy); the scopes then see self.w x self.h = the staged size."""
        self.ctx = ctx
        self.cfg = PipeConfig(width, height, depth, colorspace, wvs_components, his_components, int(want_vectorscope),
                              Region(*region))
        h = C.c_void_p()
        ctx._check(ctx.L.scope_pipe_create(ctx.h, C.byref(self.cfg), C.byref(h)))
        self.h_pipe = h
        ow, oh = C.c_uint32(), C.c_uint32()
        ctx._check(ctx.L.scope_pipe_size(h, C.byref(ow), C.byref(oh)))
        self.w, self.h = int(ow.value), int(oh.value)

    def close(self):
        if getattr(self, "h_pipe", None):
            self.ctx.L.scope_pipe_destroy(self.h_pipe)
            self.h_pipe = None

    def submit(self, rgb_ptr: int, linesize: int) -> bool:
        """False when the ring is full (collect first)."""
        st = self.ctx._check(self.ctx.L.scope_pipe_submit(self.h_pipe, C.c_void_p(rgb_ptr), linesize), allow=(EAGAIN,))
        return st == OK

    def collect(self):
        """(vectorscope, waveform, histogram counts) numpy views of the oldest frame, or None when empty."""
        v, w, hh = C.c_void_p(), C.c_void_p(), C.c_void_p()
        st = self.ctx._check(self.ctx.L.scope_pipe_collect(self.h_pipe, C.byref(v), C.byref(w), C.byref(hh)), allow=(EAGAIN,))
        if st != OK:
            return None

        def view(p, shape, dtype):
            if not p.value:
                return None
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_uint8 * n).from_address(p.value), dtype=dtype).reshape(shape)

        return view(v, (256, 256), np.uint8), view(w, (256, self.w, 4), np.uint8), view(hh, (256, 4), np.uint32)
