"""CPU checks of the drop-in boundary: libellc_hip.so builds for gfx950 without a GPU, loads, and exports exactly
the entry points include/ellc_abi.h declares (no compute calls here)."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="ellc_abi.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ellc_[a-z0-9_]+)\s*\(", txt)))


def exported_functions(so):
    """the ellc_* functions a shared library exports (nm -D --defined-only)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    return sorted(set(line.split()[-1] for line in out.splitlines() if line.split()[-2:-1] == ["T"] and line.split()[-1].startswith("ellc_")))


def test_header_and_python_binding_agree():
    from egomotion_with_local_loop_closures_amd import _lib
    assert sorted(_lib.ABI_SYMBOLS) == header_functions()
    assert sorted(_lib.DIAG_SYMBOLS) == header_functions("ellc_abi_diag.h")
    assert not set(_lib.ABI_SYMBOLS) & set(_lib.DIAG_SYMBOLS)
    # nothing of the measurement / self-test / test-hook kind is left in the product interface
    assert not [n for n in _lib.ABI_SYMBOLS if n.startswith(("ellc_profile_", "ellc_selftest_", "ellc_debug_"))]


def test_library_builds_loads_and_exports_every_symbol():
    import __graft_entry__ as g
    g.build()
    from egomotion_with_local_loop_closures_amd import _lib
    lib = _lib.lib()
    for name in header_functions():
        assert hasattr(lib, name), name
    import re
    header = open(os.path.join(ROOT, "include", "ellc_abi.h")).read()
    assert lib.ellc_abi_version() == int(re.search(r"#define ELLC_ABI_VERSION (\d+)", header).group(1))


def test_shipping_library_exports_exactly_the_reference_facing_set():
    """libellc_hip.so exports what include/ellc_abi.h declares and nothing else: the measurement hooks, device self-tests and test
    hooks (include/ellc_abi_diag.h) exist in libellc_hip_diag.so only — the same sources with -DELLC_DIAG_ABI — and neither their
    entry points nor their kernels are in the shipping binary."""
    import __graft_entry__ as g
    g.build()
    from egomotion_with_local_loop_closures_amd import _lib
    assert exported_functions(_lib.SO_PATH) == header_functions()
    assert exported_functions(_lib.DIAG_SO_PATH) == sorted(header_functions() + header_functions("ellc_abi_diag.h"))
    ship = open(_lib.SO_PATH, "rb").read()
    for kernel in (b"calib_read_f32", b"stream_read_f32x4", b"selftest_div_pair", b"selftest_lu", b"maxgrad_vertical", b"maxgrad_horizontal"):
        assert kernel not in ship, kernel
    diag = _lib.diag_lib()
    for name in header_functions("ellc_abi_diag.h"):
        assert hasattr(diag, name), name
    assert diag.ellc_abi_version() == _lib.lib().ellc_abi_version()


def test_code_object_is_gfx950():
    so = os.path.join(ROOT, "egomotion_with_local_loop_closures_amd", "csrc", "libellc_hip.so")
    data = open(so, "rb").read()
    assert b"gfx950" in data
    assert b"gn_fca_accumulate" in data and b"dm_observe" in data and b"pyr_down_chain_u8" in data


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of the ABI structs as the C compiler sees the header == the ctypes mirrors."""
    import subprocess
    from egomotion_with_local_loop_closures_amd import _lib
    fields = {"ellc_config": [f[0] for f in _lib.EllcConfig._fields_], "ellc_hypotheses": [f[0] for f in _lib.EllcHypotheses._fields_]}
    lines = ['#include "ellc_abi.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for st, fs in fields.items():
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fs:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    seen = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for st, cls in (("ellc_config", _lib.EllcConfig), ("ellc_hypotheses", _lib.EllcHypotheses)):
        assert int(seen[st]) == ctypes.sizeof(cls)
        for f in fields[st]:
            assert int(seen["%s.%s" % (st, f)]) == getattr(cls, f).offset, (st, f)


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
    c.write_text('#include "ellc_abi.h"\n#include "ellc_abi_diag.h"\nint main(void) { ellc_config c; (void)c; return ELLC_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(c)])
    cc = tmp_path / "facade.cpp"
    cc.write_text('#include "ellc_facade.hpp"\nint main() { return 0; }\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(cc)])


def test_config_struct_matches_the_python_binding(tmp_path):
    """ellc_config as the C header lays it out against the ctypes mirror in _lib.py: same size, and the defaults
    ellc_default_config writes land in the fields of the same name (no GPU needed)."""
    import ctypes
    import subprocess
    from egomotion_with_local_loop_closures_amd import _lib, api
    c = tmp_path / "size.c"
    c.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ellc_abi.h"\nint main(void) { printf("%zu %zu %zu", sizeof(ellc_config), '
                 'offsetof(ellc_config, arith), offsetof(ellc_config, coalesce)); return 0; }\n')
    exe = tmp_path / "size"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c)])
    size, off_arith, off_coalesce = (int(x) for x in subprocess.check_output([str(exe)]).split())
    assert ctypes.sizeof(_lib.EllcConfig) == size
    assert _lib.EllcConfig.arith.offset == off_arith and _lib.EllcConfig.coalesce.offset == off_coalesce
    cfg = api.default_config(640, 480, 4)
    assert (cfg.width, cfg.height, cfg.levels) == (640, 480, 4)
    assert cfg.concurrent_batches == 1 and cfg.coalesce == 1 and cfg.early_exit == 1
    assert list(cfg.max_iter)[:4] == [4, 7, 9, 12]
