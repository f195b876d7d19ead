"""CPU checks of the drop-in boundary: libellc_hip.so builds for gfx950 without a GPU, loads, and exports exactly
the entry points include/ellc_abi.h declares (no compute calls here)."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "ellc_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ellc_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_python_binding_agree():
    from egomotion_with_local_loop_closures_amd import _lib
    assert sorted(_lib.ABI_SYMBOLS) == header_functions()


def test_library_builds_loads_and_exports_every_symbol():
    import __graft_entry__ as g
    g.build()
    from egomotion_with_local_loop_closures_amd import _lib
    lib = _lib.lib()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.ellc_abi_version() == 1


def test_code_object_is_gfx950():
    so = os.path.join(ROOT, "egomotion_with_local_loop_closures_amd", "csrc", "libellc_hip.so")
    data = open(so, "rb").read()
    assert b"gfx950" in data
    assert b"gn_fca_accumulate" in data and b"dm_observe" in data and b"pyr_down_u8" in data


def test_struct_layouts_match_header():
    from egomotion_with_local_loop_closures_amd import _lib
    # ellc_config: 3 ints, 4 floats, 8 ints, 5 ints
    assert ctypes.sizeof(_lib.EllcConfig) == 4 * (3 + 4 + 8 + 5)
    assert ctypes.sizeof(_lib.EllcHypotheses) == 7 * ctypes.sizeof(ctypes.c_void_p)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device the context cannot be created; nothing silently runs on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from egomotion_with_local_loop_closures_amd import api
    with pytest.raises(api.EllcError):
        api.Context(api.default_config(64, 48, 3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "egomotion_with_local_loop_closures_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "ellc_oracle" not in txt and "libellc_oracle" not in txt, f


def test_header_is_plain_c_and_facade_is_cxx11(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 without torch / HIP types; the facade as C++11."""
    import subprocess
    c = tmp_path / "abi.c"
    c.write_text('#include "ellc_abi.h"\nint main(void) { ellc_config c; (void)c; return ELLC_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(c)])
    cc = tmp_path / "facade.cpp"
    cc.write_text('#include "ellc_facade.hpp"\nint main() { return 0; }\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(cc)])
