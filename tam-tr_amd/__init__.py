"""tam-tr_amd: MI355X-native (gfx950) implementation of TAM-TR's text-image attention hot path.

Import name: `tamtr_amd` (see /tamtr_amd.py; the directory name `tam-tr_amd` is not a Python identifier).
Layout: csrc/ = HIP kernels + C ABI (include/tamtr_hip.h); _lib.py = ctypes binding; ops.py = autograd wrappers;
modules.py / head.py / backbone.py / model.py / loss.py = the ultralytics-style plugin surface (same class names,
constructor/forward signatures and state_dict keys as the reference).
"""
import os

# The trunk runs NHWC (model.py: TAMTR_CHANNELS_LAST); MIOpen takes channels-last tensors as they are only when asked to, and
# ATen reads this switch on its first convolution.
if os.environ.get('TAMTR_CHANNELS_LAST', '1') != '0':
    os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', '1')

# (HIP graphs, graphs.py: the runtime's AQL packet capture stays at its default - on - since round 4; the recorded part holds no memset node)
# ... except in the deterministic mode: MIOpen's deterministic NCHW solvers zero their outputs with memsets (59 memset nodes in the recorded
# backward), which do not replay in order under packet capture - that mode keeps the node-by-node launches (~20 ms of host time per graph launch
# on a 200 ms step).  Read at HIP initialisation: must be set before the first HIP call of the process.
if os.environ.get('TAMTR_DETERMINISTIC') == '1':
    os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

from ._lib import LIB_PATH, TamtrHipError  # noqa: F401
