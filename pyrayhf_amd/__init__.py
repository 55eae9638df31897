"""MI355X-native vertical-ionogram forward operator (drop-in for PyRayHF's hot path).

``pyrayhf_amd.library.vertical_forward_operator`` keeps the signature of the reference's
``PyRayHF.library.vertical_forward_operator`` (reference ``PyRayHF/library.py:459-460``)
and runs on hand-written HIP kernels for gfx950 through a C-ABI shim (``include/prhf.h``).
"""

import logging

logger = logging.getLogger("pyrayhf_amd")

__version__ = "0.1.0"
__all__ = ["logger", "__version__"]
