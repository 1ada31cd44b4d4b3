"""TEST INFRASTRUCTURE ONLY — never imported by the product package.

Shims that let the read-only reference (/root/reference/codes) import and run on
this CPU-only container so that golden vectors can be generated from it
(oracle/gen_golden.py).  Nothing here travels to the GPU box in any useful way:
/root/reference does not exist there, and every consumer of this module is a
generator script that is run by hand in the build container.

What is shimmed:
  * sys.path gets /root/reference/codes in front (the reference imports `CEM.*`,
    `models.*`, `utils.*` as top-level packages).
  * scipy.signal.gaussian (moved to scipy.signal.windows in current SciPy;
    imported at codes/CEM/imresize_CEM.py:4).
  * stub modules for dependencies that are not installed here and are not on
    the RRDB+CEM path (torchvision, GPUtil, lmdb, ...).  Only
    cv2.resize(..., interpolation=INTER_CUBIC) has to really work, because
    codes/CEM/imresize_CEM.py:104-110 derives the bicubic taps from it.  The
    restatement of OpenCV's bicubic resize lives in oracle/cv2_cubic.py.
  * torch.cuda.FloatTensor / .cuda() → CPU (the reference hard-codes CUDA
    tensor types at codes/CEM/CEMnet.py:81,247).
"""
import sys
import types
import importlib

REFERENCE_ROOT = '/root/reference/codes'


def _stub(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)     # torch._dynamo probes find_spec() of well-known module names
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    import os
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError('reference tree not present (expected in the build container only)')
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the repo's own package also exposes top-level `CEM` / `models`; make sure the
    # reference's win inside a generator process
    for k in [k for k in sys.modules if k.split('.')[0] in ('CEM', 'models', 'utils', 'options', 'data')]:
        del sys.modules[k]

    import numpy as np
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, 'gaussian'):
        scipy.signal.gaussian = scipy.signal.windows.gaussian

    import torch
    from oracle import cv2_cubic
    _stub('cv2', resize=cv2_cubic.resize, INTER_CUBIC=cv2_cubic.INTER_CUBIC,
          INTER_LINEAR=1, INTER_NEAREST=0, IMREAD_UNCHANGED=-1, dilate=None, Sobel=None, CV_64F=6)
    tv = _stub('torchvision')
    tv.utils = _stub('torchvision.utils', make_grid=lambda *a, **k: None)
    tv.models = _stub('torchvision.models')
    tv.transforms = _stub('torchvision.transforms')
    _stub('GPUtil', getAvailable=lambda *a, **k: [])
    _stub('lmdb')
    _stub('imagesize')
    _stub('deepdiff', DeepDiff=lambda *a, **k: {})
    sk = _stub('skimage')
    sk.io = _stub('skimage.io')
    sk.transform = _stub('skimage.transform', resize=None)
    sk.color = _stub('skimage.color', rgb2hsv=None, hsv2rgb=None)
    sk.measure = _stub('skimage.measure')
    _stub('imageio')
    _stub('tensorboardX')
    _stub('tensorboard_logger')
    try:
        import tqdm  # noqa: F401  (installed here; stubbed only where it is missing)
    except ImportError:
        _stub('tqdm', tqdm=lambda x, *a, **k: x)
    import scipy.ndimage
    if 'scipy.ndimage.morphology' not in sys.modules:      # Z_optimization.py:7 imports binary_opening from the removed sub-module
        _stub('scipy.ndimage.morphology', binary_opening=scipy.ndimage.binary_opening)

    if not torch.cuda.is_available():
        torch.cuda.FloatTensor = torch.FloatTensor
        torch.cuda.DoubleTensor = torch.DoubleTensor
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        _map_cuda_device_to_cpu(torch)
    return True


def _map_cuda_device_to_cpu(torch):
    """The reference hard-codes torch.device('cuda') in places (Z_optimization.py:369, loss.py): on this CPU-only container every
    `.to(<cuda device>)` of a tensor or module becomes `.to('cpu')`."""
    if getattr(torch.Tensor.to, '_esr_shim', False):
        return

    def fix(a):
        if isinstance(a, torch.device) and a.type == 'cuda':
            return torch.device('cpu')
        if isinstance(a, str) and a.startswith('cuda'):
            return 'cpu'
        return a
    t_to, m_to = torch.Tensor.to, torch.nn.Module.to

    def tensor_to(self, *a, **k):
        return t_to(self, *[fix(x) for x in a], **{kk: fix(v) for kk, v in k.items()})

    def module_to(self, *a, **k):
        return m_to(self, *[fix(x) for x in a], **{kk: fix(v) for kk, v in k.items()})
    tensor_to._esr_shim = True
    torch.Tensor.to, torch.nn.Module.to = tensor_to, module_to


def ref_import(name):
    install()
    return importlib.import_module(name)
