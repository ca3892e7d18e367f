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


def reference_fallthrough(module_name, this_file):
    """Module-level __getattr__ for a drop-in module: names the MI355X module does not define (classes that are not on
    the WDNO path, e.g. the 2-D `Unet` of smoke/ddpm/diffusion_2d.py) are looked up in the *reference's* module of the
    same name, if the reference tree is on sys.path behind this one. Nothing is loaded until such a name is asked for."""
    import importlib.util
    import sys
    state = {}

    def _load():
        if 'mod' in state:
            return state['mod']
        rel = module_name.replace('.', os.sep) + '.py'
        mod = None
        for d in sys.path:
            cand = os.path.join(d or '.', rel)
            if os.path.isfile(cand) and os.path.abspath(cand) != os.path.abspath(this_file):
                spec = importlib.util.spec_from_file_location('_reference_' + module_name.replace('.', '_'), cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                break
        state['mod'] = mod
        return mod

    def __getattr__(name):
        if name.startswith('__'):
            raise AttributeError(name)
        mod = _load()
        if mod is None or not hasattr(mod, name):
            raise AttributeError(f"module '{module_name}' (wdno_amd drop-in) has no attribute '{name}' and no reference module "
                                 f"providing it was found on sys.path")
        return getattr(mod, name)
    return __getattr__
