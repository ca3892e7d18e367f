"""MI355X drop-in for smoke/ddpm (diffusion_2d, wave_utils).

Modules this package does not provide (datasets, result I/O, solvers, ...) fall through to the reference's package of
the same name when that is also on sys.path: put wdno_amd's tree *before* the reference directory and the drivers run
unchanged, with the hot-path modules resolved here."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
