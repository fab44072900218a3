"""The CPU emulation's 64-lane wave model (tests/emul/emul_runtime.cpp, SCOPE_EMUL_WAVE64=1): the threads of a wave run in lock step
between cross-lane operations and every ballot / readlane / DPP move / mbcnt / wave sum / wave barrier is a rendezvous of the
wave's live lanes -- the semantics of the hardware.  By default the emulation's waves are ONE lane wide (cheap; enough for per-lane
arithmetic), which leaves the kernels' wave votes, run-length merges (bin_rle: run heads from a ballot of the breaks, detail texels
packed through the wave's scratch row with v_mbcnt), absorb-mode flushes and K0's redo votes to the GPU suite.  Here the WHOLE
emulation parity file runs once more under the 64-lane model, in a process of its own (the model is chosen when the library
loads), so that those paths are checked against the oracle without a GPU as well."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave64_vocabulary(emul_lib):
    """ballot, all, lane-below inside DPP rows of 16, mbcnt, wave sum, readlane, early exits: closed forms, two workgroups of three waves."""
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); r = L.emul_wave64_selftest(); print('selftest', r); sys.exit(0 if r == 0 else 1)")
    out = subprocess.run([sys.executable, "-c", code, emul_lib], env=dict(os.environ, SCOPE_EMUL_WAVE64="1"), capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and "selftest 0" in out.stdout, (out.stdout, out.stderr[-2000:])
    L = ctypes.CDLL(emul_lib)
    if not os.environ.get("SCOPE_EMUL_WAVE64"):
        assert L.emul_wave64_selftest() == -1  # this process loaded the one-lane model


def test_parity_suite_under_64_lane_waves(emul_lib):
    """tests/test_parity_emul.py, every test of it, with the kernels' waves 64 lanes wide."""
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_parity_emul.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         env=dict(os.environ, SCOPE_EMUL_WAVE64="1"), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = out.stdout[-1500:]
    assert out.returncode == 0 and " passed" in tail and "failed" not in tail, tail + out.stderr[-1500:]
