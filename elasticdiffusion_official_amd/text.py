"""CLIP text encoding for the pipeline's ``text_encoder=`` hook (elastic_diffusion.py:248-265).

The reference tokenises with ``CLIPTokenizer`` (padding to ``model_max_length``, truncation) and runs
``CLIPTextModel`` (+ ``CLIPTextModelWithProjection`` for SDXL) from ``transformers``:
  * SD 1.x / 2.x : embeddings = encoder(...)[0] (last hidden state), "pooled" = the same tensor      (ED:260-262)
  * SDXL         : embeddings = concat(hidden_states[-2] of both encoders, dim=-1),
                   pooled = text_encoder_2(...)[0] (the projected text embedding)                   (ED:256-259)
No CLIP weights or vocab files exist in the build image, so the pipeline defaults to synthetic embeddings; with a
local HF snapshot ``load_clip(model_dir, xl)`` builds the real thing.
"""
import os

import torch


class ClipTextEncoder:
    """Callable ``prompts -> (text_embeddings, pooled)`` over already constructed tokenizers / encoders."""

    def __init__(self, tokenizers, encoders, xl, device="cpu"):
        assert len(tokenizers) == len(encoders) == (2 if xl else 1)
        self.tokenizers, self.encoders, self.xl, self.device = tokenizers, encoders, xl, device

    def _encode(self, prompts, k):
        tok = self.tokenizers[k]
        ids = tok(prompts, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
        return self.encoders[k](ids.input_ids.to(self.device), output_hidden_states=True)

    @torch.no_grad()
    def __call__(self, prompts):
        if isinstance(prompts, str):
            prompts = [prompts]
        if self.xl:
            a, b = self._encode(prompts, 0), self._encode(prompts, 1)
            return torch.cat([a.hidden_states[-2], b.hidden_states[-2]], dim=-1), b[0]
        e = self._encode(prompts, 0)[0]
        return e, e


def load_clip(model_dir, xl, device="cuda", dtype=torch.float32):
    """HF snapshot layout: tokenizer/, text_encoder/ (+ tokenizer_2/, text_encoder_2/ for SDXL)  (ED:145-151)."""
    from transformers import CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer
    toks = [CLIPTokenizer.from_pretrained(os.path.join(model_dir, "tokenizer"))]
    encs = [CLIPTextModel.from_pretrained(os.path.join(model_dir, "text_encoder"), torch_dtype=dtype).to(device).eval()]
    if xl:
        toks.append(CLIPTokenizer.from_pretrained(os.path.join(model_dir, "tokenizer_2")))
        encs.append(CLIPTextModelWithProjection.from_pretrained(os.path.join(model_dir, "text_encoder_2"),
                                                                torch_dtype=dtype).to(device).eval())
    return ClipTextEncoder(toks, encs, xl, device)
