"""-m gpu: the command line (python -m elasticdiffusion_official_amd; reference __main__ blocks ED:1134-1210) and the
``grid=True`` output of generate_image (ED:1123-1124)."""
import glob
import os

import pytest
import torch

from tests.fakes import FakeUNet, FakeVAE
from tests.test_hip_parity import _embed_fn_batch

pytestmark = pytest.mark.gpu


def test_cli_main_saves_images_log_and_args(tmp_path, capsys):
    """SD1.5 512x512, 2 steps, verbose: 0.png, the verbose image-log PNGs (ED:1201-1205), args.txt and the timing lines."""
    from elasticdiffusion_official_amd.__main__ import main
    save_dir = main(["--sd_version", "1.5", "--H", "512", "--W", "512", "--steps", "2", "--resampling_steps", "1",
                     "--verbose", "true", "--log_freq", "1", "--outdir", str(tmp_path), "--exp", "t", "--seed", "3",
                     "--prompt", "a test prompt", "--view_batch_size", "4"])
    files = {os.path.basename(f) for f in glob.glob(os.path.join(save_dir, "*"))}
    assert {"0.png", "args.txt", "global_img.png", "intermediate_x0_imgs.png"} <= files, files
    from PIL import Image
    assert Image.open(os.path.join(save_dir, "0.png")).size == (512, 512)
    assert "seed: 3" in open(os.path.join(save_dir, "args.txt")).read()
    out = capsys.readouterr().out
    assert "Time taken:" in out and "loop_done" in out and "host:picks" in out


def test_grid_output():
    """grid=True: ONE image = torchvision make_grid of the batch with its defaults (ED:1124), here 2 prompts side by side."""
    from elasticdiffusion_official_amd import ElasticDiffusion
    pipe = ElasticDiffusion("cuda:0", "1.5", view_batch_size=4, unet=FakeUNet(64), vae=FakeVAE(), text_encoder=_embed_fn_batch(2))
    kw = dict(height=512, width=512, num_inference_steps=2, resampling_steps=1, output_type="pt", progress=lambda it: it)
    pipe.seed_everything(5)
    single, _ = pipe.generate_image(["a", "b"], "", grid=False, **kw)
    pipe.seed_everything(5)
    grid, _ = pipe.generate_image(["a", "b"], "", grid=True, **kw)
    assert tuple(single.shape) == (2, 3, 512, 512) and tuple(grid.shape) == (1, 3, 512 + 4, 2 * 512 + 6)
    assert torch.equal(grid[0, :, 2:514, 2:514], single[0]) and torch.equal(grid[0, :, 2:514, 516:1028], single[1])
    assert float(grid[0, :, :2].abs().max()) == 0.0 and float(grid[0, :, :, 514:516].abs().max()) == 0.0
    pil, _ = pipe.generate_image(["a", "b"], "", grid=True, **dict(kw, output_type="pil"))
    assert len(pil) == 1 and pil[0].size == (2 * 512 + 6, 512 + 4)
