"""CPU: the torch model modules (plumbing behind the model boundary) and the CLIP text hook.
  * architecture hyper-parameters give exactly the published parameter counts of SD1.5 / SD2 / SDXL UNets, the SD VAE and
    the SD1.5 / SDXL ControlNets (a cheap guard that the HF-layout state dict is complete);
  * HF-layout safetensors round trip through load_weights;
  * the text hook reproduces ED:248-265 (penultimate hidden states + projected pooled output for SDXL)."""
import os
from types import SimpleNamespace

import pytest
import torch

from elasticdiffusion_official_amd import models as M
from elasticdiffusion_official_amd.text import ClipTextEncoder


def n_params(m):
    return sum(p.numel() for p in m.parameters())


def test_published_parameter_counts():
    with torch.device("meta"):
        assert n_params(M.UNet2DConditionModel(**M.UNET_CONFIGS["sd15"])) == 859_520_964
        assert n_params(M.UNet2DConditionModel(**M.UNET_CONFIGS["sd2"])) == 865_910_724
        assert n_params(M.UNet2DConditionModel(**M.UNET_CONFIGS["sdxl"])) == 2_567_463_684
        assert n_params(M.AutoencoderKL()) == 83_653_863
        assert n_params(M.ControlNetModel(M.UNET_CONFIGS["sd15"])) == 361_279_120
        assert n_params(M.ControlNetModel(M.UNET_CONFIGS["sdxl"])) == 1_251_014_160


def test_hf_layout_key_names_and_safetensors_round_trip(tmp_path):
    from safetensors.torch import save_file
    cfg = dict(M.UNET_CONFIGS["sdxl"])
    cfg.update(block_out_channels=(32, 64, 128), heads=(1, 2, 4), transformer_depth=(1, 1, 2), cross_attention_dim=48,
               addition_time_embed_dim=8, pooled_projection_dim=16, sample_size=16)
    a = M.UNet2DConditionModel(**cfg)
    keys = set(a.state_dict())
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "add_embedding.linear_2.bias",
              "down_blocks.0.resnets.0.norm1.weight", "down_blocks.0.downsamplers.0.conv.weight",
              "down_blocks.1.attentions.0.proj_in.weight", "down_blocks.1.attentions.1.transformer_blocks.0.attn1.to_q.weight",
              "down_blocks.2.attentions.0.transformer_blocks.1.attn2.to_out.0.bias",
              "down_blocks.2.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
              "down_blocks.2.attentions.0.transformer_blocks.0.ff.net.2.weight", "mid_block.resnets.1.conv2.weight",
              "mid_block.attentions.0.transformer_blocks.0.norm3.bias", "up_blocks.0.resnets.2.conv_shortcut.weight",
              "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.2.resnets.0.time_emb_proj.weight",
              "conv_norm_out.weight", "conv_out.bias"):
        assert k in keys, k
    f = os.path.join(tmp_path, "diffusion_pytorch_model.safetensors")
    save_file({k: v.contiguous() for k, v in a.state_dict().items()}, f)
    b = M.UNet2DConditionModel(**cfg)
    M.load_weights(b, f)
    x, t, e = torch.randn(2, 4, 16, 16), torch.tensor(10), torch.randn(2, 77, 48)
    kw = {"text_embeds": torch.randn(2, 16), "time_ids": torch.zeros(2, 6)}
    with torch.no_grad():
        assert torch.equal(a.eval()(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"],
                           b.eval()(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"])
    vae_keys = set(M.AutoencoderKL(block_out_channels=(32, 32, 64, 64)).state_dict())
    for k in ("encoder.down_blocks.0.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.to_q.weight",
              "encoder.mid_block.attentions.0.group_norm.weight", "decoder.up_blocks.0.upsamplers.0.conv.bias",
              "decoder.mid_block.attentions.0.to_out.0.weight", "quant_conv.weight", "post_quant_conv.bias",
              "decoder.conv_norm_out.weight"):
        assert k in vae_keys, k
    with pytest.raises(RuntimeError):
        save_file({"conv_in.weight": torch.zeros(1)}, f)
        M.load_weights(b, f)


class _Tok:
    model_max_length = 8

    def __call__(self, prompts, padding, max_length, truncation, return_tensors):
        ids = torch.zeros(len(prompts), max_length, dtype=torch.long)
        for i, p in enumerate(prompts):
            v = [min(ord(c), 99) for c in p][:max_length]
            ids[i, : len(v)] = torch.tensor(v)
        return SimpleNamespace(input_ids=ids)


def test_clip_text_hook_matches_reference_recipe():
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    torch.manual_seed(0)
    c1 = CLIPTextConfig(vocab_size=100, hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                        max_position_embeddings=8)
    c2 = CLIPTextConfig(vocab_size=100, hidden_size=48, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                        max_position_embeddings=8, projection_dim=24)
    e1, e2 = CLIPTextModel(c1).eval(), CLIPTextModelWithProjection(c2).eval()
    enc = ClipTextEncoder([_Tok(), _Tok()], [e1, e2], xl=True)
    emb, pooled = enc(["hello", "a corgi"])
    assert emb.shape == (2, 8, 32 + 48) and pooled.shape == (2, 24)
    ids = _Tok()(["hello", "a corgi"], "max_length", 8, True, "pt").input_ids
    with torch.no_grad():
        o1, o2 = e1(ids, output_hidden_states=True), e2(ids, output_hidden_states=True)
    assert torch.equal(emb, torch.cat([o1.hidden_states[-2], o2.hidden_states[-2]], dim=-1))
    assert torch.equal(pooled, o2[0])
    sd = ClipTextEncoder([_Tok()], [e1], xl=False)
    e, p = sd("hello")
    assert e.shape == (1, 8, 32) and p is e and torch.equal(e, e1(ids[:1])[0])
