"""tam-tr_amd: MI355X-native (gfx950) implementation of TAM-TR's text-image attention hot path.

Import name: `tamtr_amd` (see /tamtr_amd.py; the directory name `tam-tr_amd` is not a Python identifier).
Layout: csrc/ = HIP kernels + C ABI (include/tamtr_hip.h); _lib.py = ctypes binding; ops.py = autograd wrappers;
modules.py / head.py / backbone.py / model.py / loss.py = the ultralytics-style plugin surface (same class names,
constructor/forward signatures and state_dict keys as the reference).
"""
from ._lib import LIB_PATH, TamtrHipError  # noqa: F401
