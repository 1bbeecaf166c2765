"""elasticdiffusion_official_amd -- MI355X-native (gfx950) implementation of ONE hot path of ElasticDiffusion: the
patched global/local denoising loop behind ``ElasticDiffusion.generate_image()``.

    from elasticdiffusion_official_amd import ElasticDiffusion      # drop-in for the reference class, GPU only

HIP kernels + C ABI: csrc/elastic_kernels.hip / include/elastic_hip.h (loaded by _hip.py, wrapped by ops.py).
The package never imports ``oracle`` and has no CPU path.
"""
import os as _os

# MIOpen JIT-compiles its convolution kernels on first use (minutes for the SDXL UNet on a fresh box).  Keep its
# user find-db and compiled-kernel cache inside the repo tree so they travel with the source snapshot like the built
# .so does (git-ignored build artefacts); an empty / missing cache only costs compile time.
_MIOPEN_CACHE = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "miopen_cache")
_os.environ.setdefault("MIOPEN_USER_DB_PATH", _MIOPEN_CACHE)
_os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", _MIOPEN_CACHE)
try:
    _os.makedirs(_MIOPEN_CACHE, exist_ok=True)
except OSError:
    pass

from .schedule import ConstScheduler, CosineScheduler, DDIMSchedule, LinearScheduler  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require a GPU (build check, CPU host-logic tests)
    if name in ("ElasticDiffusion", "ElasticDiffusionControlNet"):
        from . import pipeline
        return getattr(pipeline, name)
    raise AttributeError(name)
