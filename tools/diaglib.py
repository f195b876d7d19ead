"""Diagnostic tools only: `import diaglib` (after the repo root is on sys.path, before any library call) makes the tool create its
contexts in the diagnostic library (csrc/libellc_hip_diag.so: include/ellc_abi_diag.h, the measurement hooks) or, when ELLC_LIB_PATH
names one, in that build — build/libellc_hip_envdiag.so, ..._stamps.so, or an A/B variant (all carry the diagnostic ABI) — through
_lib.use_library(). The shipping loader itself reads nothing from the environment."""
import os

from egomotion_with_local_loop_closures_amd import _lib, api

api.default_diag = True

_p = os.environ.get("ELLC_LIB_PATH")
if _p:
    _lib.use_library(_p)
