"""ctypes binding + host-side launch planning for libesr_hip.so (gfx950 kernels of the RRDB + CEM path).

There is NO CPU fallback here: importing works anywhere (so the module tree, options and checkpoint logic
can be used and tested on a CPU box), but every compute entry point raises if the library cannot be loaded
or the tensors are not on an AMD GPU.
"""
from ._lib import lib, load_library, EsrError, library_path  # noqa: F401
