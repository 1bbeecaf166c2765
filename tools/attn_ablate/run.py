"""Times the ablation builds of the pipelined attention kernel (tools/attn_ablate/build.py) on the MI355X: the SDXL
self-attention shapes at batch 20, v_path 4, interleaved rounds, median.  Output values of the ablated builds are wrong by
construction; the base build is checked against the product library first.

    python tools/attn_ablate/run.py [--rounds 5] [--out file.jsonl]"""
import argparse
import ctypes
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch

from elasticdiffusion_official_amd import _hip, ops


def load(path):
    L = ctypes.CDLL(path)
    L.ed_flash_attention.argtypes = _hip.SIGNATURES["ed_flash_attention"]
    L.ed_flash_attention.restype = ctypes.c_int
    return L


def call(L, q, k, v, out, H, v_path):
    B, Nq, HD = q.shape
    rc = L.ed_flash_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), ops.ED_F16, B, H, Nq, k.shape[1], 64, q.stride(0),
                              q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1), 0.125, v_path,
                              torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    libs = {os.path.basename(p)[len("libattn_"):-3]: load(p) for p in sorted(glob.glob(os.path.join(HERE, "libattn_*.so")))}
    out = open(a.out, "w") if a.out else None
    g = torch.Generator(device="cuda").manual_seed(0)
    for (B, H, N) in [(20, 10, 4096), (20, 20, 1024), (6, 10, 4096), (6, 20, 1024)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(torch.float16)   # column slices of a fused projection, as in the UNet
        q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
        o = torch.empty(B, N, H * 64, device="cuda", dtype=torch.float16)
        call(libs["base"], q, k, v, o, H, 4)
        assert torch.equal(o, ops.flash_attention(q, k, v, H, v_path=4)), "the base build must be the product kernel"
        flops = 4.0 * B * H * N * N * 64
        times = {n: [] for n in libs}
        for _ in range(a.rounds):
            for n, L in libs.items():
                times[n].append(timed(lambda: call(L, q, k, v, o, H, 4)))
        med = {n: sorted(t)[len(t) // 2] for n, t in times.items()}
        rec = {"B": B, "H": H, "N": N, "us": {n: round(1e3 * t, 1) for n, t in med.items()},
               "tflops_equiv": {n: round(flops / t / 1e9, 1) for n, t in med.items()},
               "share_of_base": {n: round(t / med["base"], 3) for n, t in med.items()}}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
