"""The C-ABI library loads and exports every symbol include/scope_hip.h declares (no compute
calls: there is no GPU here), and it refuses to work without a device instead of falling back."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scope_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scope_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def hip_lib_path():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "obs-color-monitor_amd", "csrc")])
    return os.path.join(ROOT, "obs-color-monitor_amd", "libscope_hip.so")


def test_header_and_binding_agree():
    import obs_color_monitor_amd as ocm

    assert declared_functions() == sorted(ocm.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib_path):
    lib = ctypes.CDLL(hip_lib_path)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/scope_hip.h but not exported"
    lib.scope_abi_version.restype = ctypes.c_int
    assert lib.scope_abi_version() == 2


def test_c_abi_only_c_types(hip_lib_path):
    """No C++ or torch types in the boundary: the header must compile as plain C."""
    src = '#include "scope_hip.h"\nint main(void){return sizeof(struct scope_surface_data)==40?0:1;}\n'
    exe = "/tmp/scope_abi_c_check"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe],
                   input=src, text=True, check=True)
    assert subprocess.run([exe]).returncode == 0  # layout of struct cm_surface_data on LP64


def test_emulation_library_exports_the_same_abi(emul_lib):
    lib = ctypes.CDLL(emul_lib)
    for name in declared_functions():
        assert hasattr(lib, name)


def test_no_device_means_error_not_fallback(hip_lib_path):
    """On a box without a GPU the product library must fail to create a context."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import obs_color_monitor_amd as ocm

    with pytest.raises(ocm.ScopeError) as e:
        ocm.ScopeContext(0, hip_lib_path)
    assert e.value.status == ocm.ENODEV
