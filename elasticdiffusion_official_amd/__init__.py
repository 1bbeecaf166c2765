"""elasticdiffusion_official_amd -- MI355X-native (gfx950) implementation of ONE hot path of ElasticDiffusion: the
patched global/local denoising loop behind ``ElasticDiffusion.generate_image()``.

    from elasticdiffusion_official_amd import ElasticDiffusion      # drop-in for the reference class, GPU only

HIP kernels + C ABI: csrc/elastic_kernels.hip / include/elastic_hip.h (loaded by _hip.py, wrapped by ops.py).
The package never imports ``oracle`` and has no CPU path.
"""
from .schedule import ConstScheduler, CosineScheduler, DDIMSchedule, LinearScheduler  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require a GPU (build check, CPU host-logic tests)
    if name == "ElasticDiffusion":
        from .pipeline import ElasticDiffusion
        return ElasticDiffusion
    raise AttributeError(name)
