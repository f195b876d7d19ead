"""Diagnostic tools only: `import diaglib` (after the repo root is on sys.path, before any library call) makes the tool load
the build named by ELLC_LIB_PATH — build/libellc_hip_diag.so, ..._stamps.so, or an A/B variant — through
_lib.use_library(). The shipping loader itself reads nothing from the environment."""
import os

from egomotion_with_local_loop_closures_amd import _lib

_p = os.environ.get("ELLC_LIB_PATH")
if _p:
    _lib.use_library(_p)
