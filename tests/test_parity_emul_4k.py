"""BASELINE geometry WITHOUT a GPU: the unmodified kernel sources on an emulated 256-CU chip (tests/emul with
SCOPE_EMUL_CUS=256), so that a 3840x2160 frame takes the very launch shapes it takes on an MI355X -- 160 vectorscope slices
with sparse partial tables and the parallel reduce, 240 sixteen-column waveform strips, the histogram's in-launch combine
over 192 slices, the one-read dock kernel -- each table against the oracle."""
import os

import numpy as np
import pytest

import parity_cases as pc

W, H = 3840, 2160


@pytest.fixture(scope="module")
def chip(emul_lib):
    import obs_color_monitor_amd as ocm

    old = os.environ.get("SCOPE_EMUL_CUS")
    os.environ["SCOPE_EMUL_CUS"] = "256"
    ctx = ocm.ScopeContext(0, emul_lib)
    assert ctx.device_info()["compute_units"] == 256
    yield ctx
    ctx.close()
    if old is None:
        os.environ.pop("SCOPE_EMUL_CUS", None)
    else:
        os.environ["SCOPE_EMUL_CUS"] = old


def _np_dev(a):
    buf = np.ascontiguousarray(a).copy()
    return buf, buf.ctypes.data


@pytest.mark.parametrize("dist", [1, 5])
def test_4k_one_frame_scopes(chip, oracle, dist):
    pc.check_vectorscope(chip, oracle, W, H, dist)
    pc.check_vectorscope(chip, oracle, W, H, dist, None, fuse=True)
    pc.check_waveform(chip, oracle, W, H, dist, 0x07)
    pc.check_histogram_counts(chip, oracle, W, H, dist, 0x07)


def test_4k_three_scopes_one_read(chip, oracle):
    pc.check_three_scopes_dev(chip, oracle, _np_dev, lambda hnd: hnd, W, H, [1], reps=1)
    pc.check_three_scopes_dev(chip, oracle, _np_dev, lambda hnd: hnd, W, H, [2], 0x05, 0x04, reps=1)


def test_4k_ring_of_three(chip, oracle):
    """Three frames per launch (85 vectorscope slices per frame on 512-byte boundaries, 64-column waveform strips)."""
    pc.check_three_scopes_dev(chip, oracle, _np_dev, lambda hnd: hnd, W, H, [1, 5, 2], reps=1)


def test_4k_batch_shapes(chip, oracle):
    """The shapes of a 64-frame launch on a few frames: four vectorscope slices per frame (eight lane-group streams, four
    loads per lane per batch, 42 clamp passes per slice), and more waveform units than workgroups (the persistent walk
    with the deep queue across flushes at 34 slots per wave)."""
    with pc.options(chip, slices=4):
        pc.check_vectorscope(chip, oracle, W, H, 1)
        pc.check_vectorscope(chip, oracle, W, H, 3)  # flat: absorb mode, every clamp pass matters
        pc.check_histogram_counts(chip, oracle, W, H, 5, 0x07)
    rgbs = [oracle.fill_frame(W, H, d, 90 + d) for d in (1, 2, 5, 1, 3)]
    got, st = chip.waveform_batch_host(rgbs, W, 0x07)
    assert st == 0
    for f, r in enumerate(rgbs):
        pc.assert_same(got[f], oracle.waveform(r, None, W, 0x07), f"4K waveform batch frame {f}")


def test_config1_and_config4_geometries(chip, oracle):
    """BASELINE config 1 (720p RGB histogram) and one rank's tile of config 4 (7680 x 540 of a 7680 x 4320 frame on 8 GPUs)."""
    pc.check_histogram_counts(chip, oracle, 1280, 720, 5, 0x07)
    pc.check_vectorscope(chip, oracle, 7680, 540, 1)
    pc.check_waveform(chip, oracle, 7680, 540, 2, 0x20)


def test_randomised_geometries_on_the_emulated_chip(chip, oracle):
    """tools/fuzz_parity.py without a GPU: the same random geometries, masks and launch sizes the -m gpu suite sweeps, on the
    256-CU emulation (narrow strips, the dock kernel, row bands and batches take their real shapes)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(chip, oracle, 10, 21, lambda m: None, _np_dev, lambda hnd: hnd) == 10
