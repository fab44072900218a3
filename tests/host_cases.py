"""Drives the C host mirror (obs-color-monitor_amd/host) the way OBS drives the plugin:
obs_module_load -> obs_source_info.create/update/video_tick -> frames staged into the 3-slot ring ->
the "color-monitor" thread calls the scope callbacks -> the render side reads tex_buf[w_tex_buf ^ 1]."""
import ctypes as C

import numpy as np

CB = {
    "create": C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p),
    "destroy": C.CFUNCTYPE(None, C.c_void_p),
    "update": C.CFUNCTYPE(None, C.c_void_p, C.c_void_p),
    "tick": C.CFUNCTYPE(None, C.c_void_p, C.c_float),
    "size": C.CFUNCTYPE(C.c_uint32, C.c_void_p),
    "defaults": C.CFUNCTYPE(None, C.c_void_p),
    "name": C.CFUNCTYPE(C.c_char_p, C.c_void_p),
    "render": C.CFUNCTYPE(None, C.c_void_p, C.c_void_p),
    "props": C.CFUNCTYPE(C.c_void_p, C.c_void_p),
    "enum": C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p),
    "wheel": C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int),
}


class SourceInfo(C.Structure):  # obs_shim.h's struct obs_source_info
    _fields_ = [("id", C.c_char_p), ("version", C.c_uint32), ("type", C.c_int), ("output_flags", C.c_uint32),
                ("get_name", CB["name"]), ("create", CB["create"]), ("destroy", CB["destroy"]),
                ("get_width", CB["size"]), ("get_height", CB["size"]), ("get_defaults", CB["defaults"]),
                ("update", CB["update"]), ("video_tick", CB["tick"]), ("video_render", CB["render"]),
                ("get_properties", CB["props"]), ("enum_active_sources", CB["enum"]), ("mouse_wheel", CB["wheel"])]


class Host:
    def __init__(self, lib_path):
        self.L = L = C.CDLL(lib_path)
        L.obs_module_load.restype = C.c_bool
        L.fake_obs_find_source.restype = C.POINTER(SourceInfo)
        L.fake_obs_find_source.argtypes = [C.c_char_p]
        L.fake_obs_data_create.restype = C.c_void_p
        L.fake_obs_data_set_int.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        L.fake_obs_data_set_double.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.fake_obs_data_release.argtypes = [C.c_void_p]
        L.scope_host_cm.restype = C.c_void_p
        L.scope_host_cm.argtypes = [C.c_void_p]
        L.cm_stage_surface.restype = C.c_bool
        L.cm_stage_surface.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.cm_render_target.restype = C.c_bool
        L.cm_render_target.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.roi_target_render_full.restype = C.c_bool
        L.roi_target_render_full.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.cm_drain.argtypes = [C.c_void_p]
        L.roi_register_source.argtypes = [C.c_void_p, C.c_void_p]
        L.roi_unregister_source.argtypes = [C.c_void_p, C.c_void_p]
        L.roi_target_render.restype = C.c_bool
        L.roi_target_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.scope_host_vss_table.restype = C.POINTER(C.c_uint8)
        L.scope_host_vss_table.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.scope_host_wvs_table.restype = C.POINTER(C.c_uint8)
        L.scope_host_wvs_table.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.scope_host_his_table.restype = C.POINTER(C.c_float)
        L.scope_host_his_table.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        for name, meta in (("scope_host_vss_copy", C.POINTER(C.c_int)), ("scope_host_wvs_copy", C.POINTER(C.c_uint32)),
                           ("scope_host_his_copy", C.POINTER(C.c_uint32))):
            fn = getattr(L, name)
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, meta]
        L.scope_host_picture.restype = C.POINTER(C.c_uint8)
        L.scope_host_picture.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.scope_host_frames_processed.restype = C.c_uint64
        L.scope_host_frames_processed.argtypes = [C.c_void_p]
        L.scope_host_frames_dropped.restype = C.c_uint64
        L.scope_host_frames_dropped.argtypes = [C.c_void_p]
        for f in (L.scope_host_plane_uploads, L.scope_host_plane_reuses):
            f.restype, f.argtypes = C.c_uint64, [C.c_void_p]
        L.fake_obs_reset()
        assert L.obs_module_load(), "module refused to load (no GPU / no HIP library)"

    def close(self):
        self.L.obs_module_unload()

    def create(self, source_id, **settings):
        info = self.L.fake_obs_find_source(source_id.encode()).contents
        d = self.L.fake_obs_data_create()
        info.get_defaults(d)
        for k, v in settings.items():
            if isinstance(v, float):
                self.L.fake_obs_data_set_double(d, k.encode(), v)  # the reference's double-typed settings
            else:
                self.L.fake_obs_data_set_int(d, k.encode(), int(v))
        obj = info.create(d, None)
        self.L.fake_obs_data_release(d)
        return info, obj

    def stage(self, obj, surface, width, height):
        """surface: numpy [sheight, linesize] stacked planes."""
        return bool(self.L.cm_stage_surface(self.L.scope_host_cm(obj), surface.ctypes.data_as(C.c_void_p),
                                            surface.shape[1], width, height))

    def render_target(self, obj, target_rgb, width, height):
        """target_rgb: numpy [height, linesize], the full-size frame OBS would render the target into."""
        return bool(self.L.cm_render_target(self.L.scope_host_cm(obj), target_rgb.ctypes.data_as(C.c_void_p),
                                            target_rgb.shape[1], width, height))

    def drain(self, obj):
        self.L.cm_drain(self.L.scope_host_cm(obj))

    def vss_table(self, obj):
        """The copy-out accessor (host_api.c): taken under the source's tex_mutex, legal with frames in flight."""
        cs = C.c_int()
        out = np.empty((256, 256), np.uint8)
        n = self.L.scope_host_vss_copy(obj, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(cs))
        return (out if n else None), cs.value

    def wvs_table(self, obj):
        w = C.c_uint32()
        self.L.scope_host_wvs_copy(obj, None, 0, C.byref(w))  # the width first
        if not w.value:
            return None
        out = np.empty((256, w.value, 4), np.uint8)
        n = self.L.scope_host_wvs_copy(obj, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(w))
        return out if n == out.nbytes else None

    def picture(self, obj, kind):
        w, h = C.c_uint32(), C.c_uint32()
        p = self.L.scope_host_picture(obj, kind, C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy() if p else None

    def his_table(self, obj):
        hi = (C.c_uint32 * 3)()
        out = np.empty((256, 4), np.float32)
        n = self.L.scope_host_his_copy(obj, out.ctypes.data_as(C.c_void_p), out.nbytes, hi)
        return (out if n else None), list(hi)


def check_source_surface(host):
    """The outer obs_source_info surface: the obsolete v1 table, get_properties, enum_active_sources, mouse_wheel."""
    L = host.L
    L.fake_obs_find_source_version.restype = C.POINTER(SourceInfo)
    L.fake_obs_find_source_version.argtypes = [C.c_char_p, C.c_uint32]
    L.fake_obs_properties_count.argtypes = [C.c_void_p]
    L.fake_obs_property.restype = C.c_char_p
    L.fake_obs_property.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 4
    L.obs_properties_destroy.argtypes = [C.c_void_p]
    L.obs_data_get_int.restype = C.c_longlong
    L.obs_data_get_int.argtypes = [C.c_void_p, C.c_char_p]
    v1 = L.fake_obs_find_source_version(b"vectorscope_source", 0).contents
    v2 = L.fake_obs_find_source_version(b"vectorscope_source", 2).contents
    OBSOLETE, INTERACTION = 1 << 9, 1 << 5
    assert v1.output_flags & OBSOLETE and not v2.output_flags & OBSOLETE and v1.output_flags & INTERACTION
    for info, color_type in ((v1, 0), (v2, 1)):  # settings saved before "color_type" existed keep the white trace
        d = L.fake_obs_data_create()
        info.get_defaults(d)
        assert L.obs_data_get_int(d, b"color_type") == color_type
        assert L.obs_data_get_int(d, b"intensity") == 25 and L.obs_data_get_int(d, b"graticule") == 1 | 256
        assert L.obs_data_get_int(d, b"graticule_skintone_color") == 0x0054FF
        L.fake_obs_data_release(d)
    want = {
        b"vectorscope_source": {"target_scale": (0, 1, 128), "bypass": (1, 0, 0), "intensity": (0, 1, 255), "color_type": (2, 0, 1),
                                "graticule": (2, 0, 2 + 256)},
        b"waveform_source": {"target_scale": (0, 1, 128), "bypass": (1, 0, 0), "display": (2, 0, 2), "components": (2, 0x07, 0x70),
                             "intensity": (0, 1, 255), "graticule_lines": (2, 0, 10)},
        b"histogram_source": {"target_scale": (0, 1, 128), "bypass": (1, 0, 0), "display": (2, 0, 2), "components": (2, 0x07, 0x70),
                              "level_height": (0, 50, 2048), "logscale": (1, 0, 0), "level_mode": (2, 0, 2),
                              "level_fixed_value": (0, 1, 1 << 30), "level_ratio_value": (0, 1, 1000)},
        b"colormonitor_roi": {"target_scale": (0, 1, 128), "interleave": (0, 0, 1)},  # an ROI has no bypass (common.c:125-126)
    }
    seen = []
    ENUM = CB["enum"]
    for sid, props_want in want.items():
        info, obj = host.create(sid.decode())
        assert info.get_properties and info.enum_active_sources
        props = info.get_properties(obj)
        got = {}
        for i in range(L.fake_obs_properties_count(props)):
            kind, lo, hi, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            name = L.fake_obs_property(props, i, C.byref(kind), C.byref(lo), C.byref(hi), C.byref(n))
            got[name.decode()] = (kind.value, lo.value, hi.value)
        L.obs_properties_destroy(props)
        assert got == props_want, (sid, got)
        info.enum_active_sources(obj, ENUM(lambda parent, child, param: seen.append(child)), None)  # no target set: nothing
        info.destroy(obj)
    assert seen == []


def check_histogram_level_modes(host):
    """his_update's level_mode switch as the reference writes it (src/histogram.c:131-156): mode 0 zeroes the two level values and
    leaves the graticule step of the mode before; 1 / 2 set their own value and step and zero the other value; a bad mode only logs."""
    L = host.L
    L.scope_host_his_levels.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]

    def levels(obj):
        f, r, s = C.c_int(), C.c_int(), C.c_float()
        L.scope_host_his_levels(obj, C.byref(f), C.byref(r), C.byref(s))
        return f.value, r.value, s.value

    def update(info, obj, **settings):
        d = L.fake_obs_data_create()
        info.get_defaults(d)
        for k, v in settings.items():
            (L.fake_obs_data_set_double if isinstance(v, float) else L.fake_obs_data_set_int)(d, k.encode(), v)
        info.update(obj, d)
        L.fake_obs_data_release(d)

    info, his = host.create("histogram_source", level_mode=1, level_fixed_value=777, graticule_horizontal_step_fixed=50.0)
    assert levels(his) == (777, 0, 50.0)
    update(info, his, level_mode=0)
    assert levels(his) == (0, 0, 50.0)  # the step stays
    update(info, his, level_mode=2, level_ratio_value=12.5, graticule_horizontal_step_ratio=2.5)
    assert levels(his) == (0, 125, 2.5)
    update(info, his, level_mode=9, level_fixed_value=5, level_ratio_value=1.0)
    assert levels(his) == (0, 125, 2.5)  # an invalid blob changes nothing
    update(info, his, level_mode=1, level_fixed_value=31, graticule_horizontal_step_fixed=4.0)
    assert levels(his) == (31, 0, 4.0)
    info.destroy(his)


def check_profile_zones(host, expect_at_least):
    """The GPU calls sit in the reference's profiler zones (ENABLE_PROFILE builds): every start has its end."""
    L = host.L
    L.fake_obs_profile_zone.restype = C.c_long
    L.fake_obs_profile_zone.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    for name, least in expect_at_least.items():
        bal = C.c_int(0)
        n = L.fake_obs_profile_zone(name.encode(), C.byref(bal))
        assert n >= least and bal.value == 1, (name, n, bal.value)


def run_host_checks(host, oracle, w=96, h=40):
    ids = [b"vectorscope_source", b"waveform_source", b"histogram_source", b"colormonitor_roi"]
    assert host.L.fake_obs_source_count() == 5  # the obsolete version-1 vectorscope stays registered (plugin-main.c:82-85)
    for i in ids:
        assert host.L.fake_obs_find_source(i).contents.id == i
    assert host.L.fake_obs_find_source(b"vectorscope_source").contents.version == 2
    check_source_surface(host)
    check_histogram_level_modes(host)

    frames = []
    for k, dist in enumerate((1, 2, 5)):
        rgb = oracle.fill_frame(w, h, dist, 40 + k)
        frames.append((rgb, oracle.rgb_to_yuv(rgb, w, 2)))

    # -- a stand-alone vectorscope source: its own ring + "color-monitor" thread ------------------------
    info, vss = host.create("vectorscope_source", colorspace=2)
    assert info.get_width(vss) == 256 and info.get_height(vss) == 256
    assert host.vss_table(vss)[0] is None  # nothing drawn yet
    # thread not started yet (no video_tick): two slots fill, the third frame is dropped (common.c:260-268)
    assert [host.stage(vss, frames[0][1], w, h) for _ in range(3)] == [True, True, False]
    assert host.L.scope_host_frames_dropped(vss) == 1
    info.video_tick(vss, 0.0)
    host.drain(vss)
    assert host.L.scope_host_frames_processed(vss) == 2
    for rgb, yuv in frames:
        assert host.stage(vss, yuv, w, h)
        host.drain(vss)
        got, cs = host.vss_table(vss)
        assert cs == 2 and np.array_equal(got, oracle.vectorscope(yuv, w))
    info.destroy(vss)

    # -- three scopes behind one ROI capture: one staged surface, three callbacks on one thread ------------
    roi_info, roi = host.create("colormonitor_roi")
    vinfo, vss = host.create("vectorscope_source")
    winfo, wvs = host.create("waveform_source", components=0x07)
    hinfo, his = host.create("histogram_source", components=0x07, logscale=1)
    for obj in (vss, wvs, his):
        host.L.roi_register_source(roi, host.L.scope_host_cm(obj))
    for info_, obj in ((vinfo, vss), (winfo, wvs), (hinfo, his)):
        info_.video_tick(obj, 0.0)  # attached to an ROI: no thread of their own (common.c:587-588)
    roi_info.video_tick(roi, 0.0)   # merges the scopes' plane flags (roi.c:534-540), starts the ROI thread
    for rgb, yuv in frames:
        stacked = np.ascontiguousarray(np.concatenate([rgb, yuv]))  # RGB rows then YUV rows (common.c:284-291)
        assert host.stage(roi, stacked, w, h)
        host.drain(roi)
        assert np.array_equal(host.vss_table(vss)[0], oracle.vectorscope(yuv, w))
        assert np.array_equal(host.wvs_table(wvs), oracle.waveform(rgb, yuv, w, 0x07))
        want_f, want_hi = oracle.histogram(rgb, yuv, w, 0x07, logscale=1)
        got_f, got_hi = host.his_table(his)
        assert np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32)) and got_hi == list(want_hi)
    assert winfo.get_width(wvs) == w and winfo.get_height(wvs) == 256
    # three scopes read each staged surface, but its two planes crossed PCIe once each (scope_surface_begin/end)
    assert host.L.scope_host_plane_uploads(roi) == 2 * len(frames) and host.L.scope_host_plane_reuses(roi) == len(frames)
    # video_render: the display shaders as GPU post-passes, then the graticule (labels + lines) blended over the picture --
    # the composited picture the plugin would have drawn, against oracle display + oracle geometry + oracle rasteriser
    import graticule_oracle as go

    host.L.scope_graticule_label_strip.argtypes = [C.c_void_p]
    strip = np.zeros((16, 96), dtype=np.uint8)
    host.L.scope_graticule_label_strip(strip.ctypes.data_as(C.c_void_p))
    AMBER = 0x80FFBF00

    def vss_picture(zoom=1.0):
        pic = oracle.display_vectorscope(oracle.vectorscope(yuv, w), 25, 1, 2, zoom)
        labels, lines = go.vectorscope(2, 1 | 256, 0x0054FF)  # the defaults: amber, I/Q cross (src/vectorscope.c:159-160)
        if zoom > 1.01:
            z, ofst = np.float32(zoom), np.float32(127.5) * (np.float32(1.0) - np.float32(zoom))
            labels = labels.copy()
            labels[:, 0:2] = labels[:, 0:2] * z + ofst
            labels[:, 2:4] = labels[:, 2:4] * z
            lines = lines * z + ofst
        return go.rasterise(pic, labels, lines, AMBER, strip)

    rgb, yuv = frames[-1]
    vinfo.video_render(vss, None)
    assert np.array_equal(host.picture(vss, 0), vss_picture())
    # the mouse wheel zooms the vectorscope about its centre (vss_mouse_wheel, vectorscope.c:473-483), graticule included
    assert vinfo.mouse_wheel  # the table carries it, with the reference's signature (src/vectorscope.c:473)
    vinfo.mouse_wheel(vss, None, 0, 1400)
    zoom = float(np.float32(1.0) * np.exp(np.float32(1400 * 5e-4)).astype(np.float32))
    vinfo.video_render(vss, None)
    assert np.array_equal(host.picture(vss, 0), vss_picture(zoom))
    vinfo.mouse_wheel(vss, None, 0, -100000)  # never below 1
    vinfo.video_render(vss, None)
    assert np.array_equal(host.picture(vss, 0), vss_picture())
    winfo.video_render(wvs, None)
    want = go.rasterise(oracle.display_waveform(oracle.waveform(rgb, yuv, w, 0x07), 0x07, 0, 51), [], go.waveform(5, 0, 3, w), AMBER, strip)
    assert np.array_equal(host.picture(wvs, 1), want)
    hinfo.video_render(his, None)
    flt, hi = oracle.histogram(rgb, yuv, w, 0x07, logscale=1)
    want = go.rasterise(oracle.display_histogram(flt, hi, 0x07, 0, 200), [], go.histogram(5, 0.0, 1, 0, 0, 0, 3, 200), AMBER, strip)
    assert np.array_equal(host.picture(his, 2), want)
    assert hinfo.get_width(his) == 256 and hinfo.get_height(his) == 200
    # a waveform switched to luma gets the YUV plane of the same surface
    d = host.L.fake_obs_data_create()
    winfo.get_defaults(d)
    host.L.fake_obs_data_set_int(d, b"components", 0x20)
    winfo.update(wvs, d)
    host.L.fake_obs_data_release(d)
    rgb, yuv = frames[0]
    assert host.stage(roi, np.ascontiguousarray(np.concatenate([rgb, yuv])), w, h)
    host.drain(roi)
    assert np.array_equal(host.wvs_table(wvs), oracle.waveform(rgb, yuv, w, 0x20))
    for obj in (vss, wvs, his):
        host.L.roi_unregister_source(roi, host.L.scope_host_cm(obj))
    for info_, obj in ((vinfo, vss), (winfo, wvs), (hinfo, his), (roi_info, roi)):
        info_.destroy(obj)

    # -- ROI rectangle (crop, common.c:272-282) and interleave (roi.c:266-277,523-532) --------------------
    x0, y0, x1, y1 = 8, 4, 8 + 50, 4 + 23  # an odd-sized sub-rectangle: ragged width, arbitrary pitch
    roi_info, roi = host.create("colormonitor_roi", interleave=1, roi_x0=x0, roi_y0=y0, roi_x1=x1, roi_y1=y1)
    vinfo, vss = host.create("vectorscope_source")
    winfo, wvs = host.create("waveform_source", components=0x07)
    for obj in (vss, wvs):
        host.L.roi_register_source(roi, host.L.scope_host_cm(obj))
        (vinfo if obj == vss else winfo).video_tick(obj, 0.0)
    staged = []
    for t in range(6):  # interleave = 1: every second frame is captured (doc/dock.md:58-65)
        rgb, yuv = frames[t % 3]
        roi_info.video_tick(roi, 0.0)
        stacked = np.ascontiguousarray(np.concatenate([rgb, yuv]))
        skipped = host.L.roi_target_render(roi, stacked.ctypes.data_as(C.c_void_p), stacked.shape[1], w, h)
        staged.append(not skipped)
        if not skipped:
            host.drain(roi)
            crop_rgb = np.ascontiguousarray(rgb[y0:y1, 4 * x0:4 * x1])
            crop_yuv = np.ascontiguousarray(yuv[y0:y1, 4 * x0:4 * x1])
            assert np.array_equal(host.vss_table(vss)[0], oracle.vectorscope(crop_yuv, x1 - x0))
            assert np.array_equal(host.wvs_table(wvs), oracle.waveform(crop_rgb, crop_yuv, x1 - x0, 0x07))
    assert staged == [True, False, True, False, True, False]
    assert host.L.scope_host_frames_processed(roi) == 3
    for obj in (vss, wvs):
        host.L.roi_unregister_source(roi, host.L.scope_host_cm(obj))
    for info_, obj in ((vinfo, vss), (winfo, wvs), (roi_info, roi)):
        info_.destroy(obj)

    # -- the whole producer half on a full-size target: target_scale, ROI, K0 on the GPU (common.c:223-333) ----
    tw, th = 2 * w + 1, 2 * h + 1  # odd: the scaled size is floor(target / scale) (common.c:249-250)
    target = oracle.fill_frame(tw, th, 1, 99, 4 * tw + 8)
    info, vss = host.create("vectorscope_source", colorspace=1)  # target_scale defaults to 2 (vectorscope.c:157)
    info.video_tick(vss, 0.0)
    assert host.render_target(vss, target, tw, th)
    host.drain(vss)
    scaled = oracle.stage_region(target, tw, 2)
    assert scaled.shape == (h, 4 * w)
    got, cs = host.vss_table(vss)
    assert cs == 1 and np.array_equal(got, oracle.vectorscope(oracle.rgb_to_yuv(scaled, w, 1), w))
    assert not host.render_target(vss, target[:1], tw, 1)  # scaled height 0: nothing staged (common.c:251-254)
    info.destroy(vss)

    x0, y0, x1, y1 = 3, 2, 3 + 41, 2 + 17  # rectangle in SCALED pixels (common.c:272-282)
    roi_info, roi = host.create("colormonitor_roi", interleave=0, target_scale=3, roi_x0=x0, roi_y0=y0, roi_x1=x1, roi_y1=y1)
    winfo, wvs = host.create("waveform_source", components=0x07)
    # level settings with the reference's keys and types: level_ratio_value is a double PERCENTAGE that the plugin keeps in
    # tenths, (int)(v * 10 + 0.5) (src/histogram.c:146-147) -- 12.5 % = 125 per mille of the frame's pixels
    hinfo, his = host.create("histogram_source", components=0x70, level_mode=2, level_ratio_value=12.5,
                             graticule_horizontal_step_ratio=2.5)
    for info_, obj in ((winfo, wvs), (hinfo, his)):
        host.L.roi_register_source(roi, host.L.scope_host_cm(obj))
        info_.video_tick(obj, 0.0)
    roi_info.video_tick(roi, 0.0)
    tw, th = 3 * 60 + 2, 3 * 30 + 1
    target = oracle.fill_frame(tw, th, 5, 123)
    host.L.roi_target_render_full(roi, target.ctypes.data_as(C.c_void_p), target.shape[1], tw, th)
    host.drain(roi)
    crop = oracle.stage_region(target, tw, 3, x0, y0, x1 - x0, y1 - y0)
    crop_yuv = oracle.rgb_to_yuv(crop, x1 - x0, 2)
    assert np.array_equal(host.wvs_table(wvs), oracle.waveform(crop, crop_yuv, x1 - x0, 0x07))
    want_f, want_hi = oracle.histogram(crop, crop_yuv, x1 - x0, 0x70, level_ratio=125)
    got_f, got_hi = host.his_table(his)
    assert np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32)) and got_hi == list(want_hi)
    assert got_hi == [(x1 - x0) * (y1 - y0) * 125 // 1000] * 3
    for obj in (wvs, his):
        host.L.roi_unregister_source(roi, host.L.scope_host_cm(obj))
    for info_, obj in ((winfo, wvs), (hinfo, his), (roi_info, roi)):
        info_.destroy(obj)
    # every GPU call above ran inside the profiler zone the reference gives the CPU loop it replaces
    check_profile_zones(host, {"cm_pipeline_thread_loop": 5, "draw_vectorscope": 3, "draw_waveform": 3, "draw_histogram": 2,
                               "render_target": 2, "convert_yuv": 1, "stage_surface": 2})
