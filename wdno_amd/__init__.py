"""wdno_amd -- MI355X-native (gfx950) implementation of the WDNO hot path.

Layout
  csrc/ + libwdno_hip.so   hand-written HIP kernels behind the C ABI of include/wdno_hip.h
  _lib.py, ops.py          ctypes binding + autograd wrappers (PyTorch = memory, streams, autograd tape only)
  wavelets.py              DWT / IDWT operators (pytorch_wavelets / ptwt call-compatible classes in third_party/)
  burgers/, smoke/         drop-in module trees with the reference's import paths, class names, ctor signatures and
                           state_dict keys (ddpm_burgers.unet.Unet2D, ddpm.diffusion_2d.GaussianDiffusion, ...)
  trainer.py               flat-buffer Adam / clip / EMA step and the RCCL gradient all-reduce
"""
import os

__version__ = '0.1.0'
PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def tree_path(name):
    """Directory to put on sys.path so that the reference's own imports resolve to this package's modules:
    'burgers' (ddpm_burgers.*, wave_trans), 'smoke' (ddpm.*, video_diffusion_pytorch.*, wave_trans_2d) or
    'third_party' (pytorch_wavelets, ptwt, pywt call-compatible operators)."""
    p = os.path.join(PKG_DIR, name)
    if not os.path.isdir(p):
        raise ValueError(name)
    return p
