"""Import the *real* reference modules in the builder container (test infrastructure only).

/root/reference cannot be imported as shipped: ``diffusers``, ``torchvision`` and ``cv2`` are not installed and there
is no network.  Empty stand-in modules are registered for those names (nothing in them is ever *executed* on the
paths we drive: the UNet / VAE / scheduler are injected objects), the reference directory is put on ``sys.path``
and bytecode writing is disabled so nothing is written into the read-only tree.

This file is used only by tests/golden/make_golden.py and by tests that are skipped when /root/reference does not
exist (it never exists on the GPU box).  Nothing from the reference is copied: only its *outputs* are stored.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

REFERENCE_DIR = "/root/reference"


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "elastic_diffusion.py"))


def _to_pil(pic):
    """What torchvision.transforms.ToPILImage does for a float CHW tensor: mul(255).byte(), HWC, RGB."""
    from PIL import Image

    arr = pic.detach().cpu().mul(255).byte().permute(1, 2, 0).numpy()
    return Image.fromarray(arr.squeeze(-1) if arr.shape[-1] == 1 else arr)


def _install_stubs():
    import transformers  # noqa: F401  (must come first: its availability probes reject spec-less fake modules)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Missing:
        def __init__(self, *a, **k):
            raise RuntimeError("stub class: the golden harness injects models, it never constructs them")

        @classmethod
        def from_pretrained(cls, *a, **k):
            raise RuntimeError("stub class: no pretrained weights in this container")

    if "diffusers" not in sys.modules:
        mod("diffusers", AutoencoderKL=_Missing, UNet2DConditionModel=_Missing, DDIMScheduler=_Missing)
        mod("diffusers.models", ControlNetModel=_Missing)
        mod("diffusers.models.attention_processor", AttnProcessor2_0=_Missing, LoRAAttnProcessor2_0=_Missing,
            LoRAXFormersAttnProcessor=_Missing, XFormersAttnProcessor=_Missing)
        mod("diffusers.image_processor", VaeImageProcessor=_Missing)
    if "torchvision" not in sys.modules:
        tv = mod("torchvision")
        tv.transforms = mod("torchvision.transforms", ToPILImage=lambda: _to_pil)
        tv.utils = mod("torchvision.utils", make_grid=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub")))
    if "cv2" not in sys.modules:
        mod("cv2")


def load_reference(controlnet=False):
    """Returns the imported reference module (elastic_diffusion or elastic_diffusion_w_controlnet)."""
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    _install_stubs()
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    import importlib

    return importlib.import_module("elastic_diffusion_w_controlnet" if controlnet else "elastic_diffusion")


def make_reference_pipeline(unet, vae, scheduler, text_embed_fn, sd_version="1.5", view_batch_size=1,
                            low_vram=False, pooled_dim=16, controlnet=None, verbose=False):
    """Build a reference ``ElasticDiffusion`` without running its model-loading constructor
    (elastic_diffusion.py:111-157) and inject deterministic models."""
    ref = load_reference(controlnet is not None)
    pipe = ref.ElasticDiffusion.__new__(ref.ElasticDiffusion)
    nn.Module.__init__(pipe)
    pipe.device = torch.device("cpu")
    pipe.sd_version = sd_version
    pipe.verbose = verbose
    pipe.torch_dtype = torch.float16 if low_vram else torch.float32
    pipe.view_batch_size = view_batch_size
    pipe.log_freq = 5
    pipe.low_vram = low_vram
    pipe.unet = unet
    pipe.vae = vae
    pipe.scheduler = scheduler
    pipe.text_encoder = [SimpleNamespace(), SimpleNamespace(config=SimpleNamespace(projection_dim=pooled_dim))]
    pipe.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
    pipe.set_view_config()
    pipe.get_text_embeds = text_embed_fn
    if controlnet is not None:
        pipe.controlnet = controlnet
        pipe.controlnet_model = "depth"
    return pipe, ref
