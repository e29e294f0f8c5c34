"""Import alias: ``import msd_amd`` -> the package in ``music-spectrogram-diffusion_amd/``
(a dash is not a valid identifier, so the package is loaded by name here).
Use attribute access (``msd_amd.config``), not ``import msd_amd.config``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
  sys.path.insert(0, _root)
_pkg = importlib.import_module('music-spectrogram-diffusion_amd')
sys.modules[__name__] = _pkg
