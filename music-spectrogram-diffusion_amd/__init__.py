"""MI355X-native DDPM spectrogram synthesizer: the hot path of
magenta/music-spectrogram-diffusion (``InferenceModel.predict`` ->
``models/diffusion`` denoising loop) as hand-written gfx950 HIP kernels behind a
C-ABI (include/msd_amd.h), mirrored here behind the reference's Python API.

The directory name contains a dash, so import it through the ``msd_amd`` alias
module at the repo root (``import msd_amd``).  Importing this package does NOT
load the HIP library; ``native.load()`` does, and fails loudly if it is missing.
"""
from . import config
from . import synthetic
from . import audio_codecs
from . import gin_lite
from . import native
from . import inference
from . import sharding
from . import frontend
from . import checkpoints
from . import jax_random
from .inference import InferenceModel, parse_training_gin_file

__all__ = ['config', 'synthetic', 'audio_codecs', 'gin_lite', 'native', 'inference',
           'sharding', 'frontend', 'checkpoints', 'jax_random', 'InferenceModel', 'parse_training_gin_file']
