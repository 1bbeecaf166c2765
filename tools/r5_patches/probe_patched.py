"""GPU probe: one build of the library against another, entry point by entry point -- results must be bit-identical (the round-5
patches move instructions, they do not change arithmetic); interleaved timing, median.  Since the patches landed in the product
(round 5) the default pair is: base = the round-4 product library kept as tools/r5_patches/build/libelastic_hip_r4_product.so,
new = the product's libelastic_hip.so.  A case either library rejects is reported with its return codes and skipped.
    python tools/r5_patches/probe_patched.py [--rounds 5] [--base FILE] [--lib FILE]"""
import argparse
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch

from elasticdiffusion_official_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--base", default="libelastic_hip_r4_product.so", help="file under tools/r5_patches/build/, or 'product'")
ap.add_argument("--lib", default="product", help="file under tools/r5_patches/build/ (build_patched.py --only ... --out ...), or 'product'")
a = ap.parse_args()


def load(name):
    if name == "product":
        return _hip.lib()
    L = ctypes.CDLL(os.path.join(HERE, "build", name))
    for fn in ("ed_linear", "ed_geglu_gemm", "ed_conv3x3_nhwc", "ed_flash_attention"):
        getattr(L, fn).argtypes = _hip.SIGNATURES[fn]
        getattr(L, fn).restype = ctypes.c_int
    return L


prod, pat = load(a.base), load(a.lib)
st = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
g = torch.Generator(device="cuda").manual_seed(0)


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


def compare(name, call, outs, flops, rounds=a.rounds):
    """call(lib, out) launches; outs = two output tensors"""
    same = True
    for _ in range(3):
        outs[0].zero_(), outs[1].zero_()
        rc = (call(prod, outs[0]), call(pat, outs[1]))
        if rc != (0, 0):
            print(json.dumps({"case": name, "skipped": True, "rc_base": rc[0], "rc_new": rc[1]}), flush=True)
            return
        same = same and bool(torch.equal(outs[0], outs[1]))
    tp, tq = [], []
    for _ in range(rounds):
        tp.append(timed(lambda: call(prod, outs[0])))
        tq.append(timed(lambda: call(pat, outs[1])))
    mp, mq = sorted(tp)[len(tp) // 2], sorted(tq)[len(tq) // 2]
    print(json.dumps({"case": name, "bit_identical": same, "product_us": round(1e3 * mp, 1), "patched_us": round(1e3 * mq, 1),
                      "product_tflops": round(flops / mp / 1e9, 1), "patched_tflops": round(flops / mq / 1e9, 1), "speedup": round(mp / mq, 4)}), flush=True)


dt = torch.float16
for (M, K, N, res) in [(81920, 640, 640, 0), (81920, 640, 640, 1), (81920, 640, 1920, 0), (20480, 1280, 1280, 0), (20480, 2560, 640, 1), (1000, 320, 200, 0), (700, 128, 304, 1),
                       (327680, 640, 320, 0), (81920, 2560, 640, 0), (300, 256, 104, 1), (600, 64, 384, 0)]:     # N mod 256 in (0, 128]: 0005's half tiles
    x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(dt)
    w = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).to(dt)
    b = (torch.rand(N, device="cuda", generator=g) * 2 - 1).to(dt)
    r = (torch.rand(M, N, device="cuda", generator=g) * 2 - 1).to(dt) if res else None
    outs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(2)]
    compare(f"ed_linear {M}x{K}->{N}{' + residual' if res else ''}",
            lambda L, o: L.ed_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, o.data_ptr(), 1, M, K, N, st()), outs, 2.0 * M * K * N)
for (M, K, I) in [(20480, 1280, 5120), (81920, 640, 2560), (6144, 1280, 5120), (24576, 640, 2560), (300, 192, 256), (5000, 64, 128)]:     # 0003: persistent GEGLU
    x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(dt)
    w = ((torch.rand(2 * I, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).to(dt)
    b = (torch.rand(2 * I, device="cuda", generator=g) * 2 - 1).to(dt)
    outs = [torch.empty(M, I, device="cuda", dtype=dt) for _ in range(2)]
    compare(f"ed_geglu_gemm {M}x{K}->{I}", lambda L, o: L.ed_geglu_gemm(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, I, st()), outs, 4.0 * M * K * I)
for (B, H, W, Cin, N) in [(20, 32, 32, 1280, 1280), (2, 12, 20, 64, 200), (20, 128, 128, 320, 320), (20, 64, 64, 640, 640), (2, 12, 20, 64, 104)]:     # without addends (the upsampler's); 320 / 640 / 104: half tiles
    cl = torch.channels_last
    x = (torch.rand(B, Cin, H, W, device="cuda", generator=g) * 2 - 1).to(dt).contiguous(memory_format=cl)
    w = ((torch.rand(N, Cin, 3, 3, device="cuda", generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dt).contiguous(memory_format=cl)
    b = (torch.rand(N, device="cuda", generator=g) * 2 - 1).to(dt)
    outs = [torch.empty(B, N, H, W, device="cuda", dtype=dt).contiguous(memory_format=cl) for _ in range(2)]
    compare(f"ed_conv3x3_nhwc {B}x{H}x{W} {Cin}->{N} (bias only)",
            lambda L, o: L.ed_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, o.data_ptr(), 1, B, H, W, Cin, N, st()), outs, 2.0 * B * H * W * 9 * Cin * N)
for (B, H, W, Cin, N) in [(20, 128, 128, 320, 320), (6, 64, 64, 640, 640), (2, 9, 20, 128, 104)]:     # the ResnetBlock form: per-sample bias + residual, half tiles
    cl = torch.channels_last
    x = (torch.rand(B, Cin, H, W, device="cuda", generator=g) * 2 - 1).to(dt).contiguous(memory_format=cl)
    w = ((torch.rand(N, Cin, 3, 3, device="cuda", generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dt).contiguous(memory_format=cl)
    b = (torch.rand(N, device="cuda", generator=g) * 2 - 1).to(dt)
    sb = (torch.rand(B, N, device="cuda", generator=g) * 2 - 1).to(dt)
    rs = (torch.rand(B, N, H, W, device="cuda", generator=g) * 2 - 1).to(dt).contiguous(memory_format=cl)
    outs = [torch.empty(B, N, H, W, device="cuda", dtype=dt).contiguous(memory_format=cl) for _ in range(2)]
    compare(f"ed_conv3x3_nhwc {B}x{H}x{W} {Cin}->{N} (+ per-sample bias + residual)",
            lambda L, o: L.ed_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), sb.data_ptr(), rs.data_ptr(), o.data_ptr(), 1, B, H, W, Cin, N, st()), outs,
            2.0 * B * H * W * 9 * Cin * N)
for dtt, code in ((torch.float16, 1), (torch.bfloat16, 2)):
    for (B, H, Nq, Nk, vps) in [(20, 10, 4096, 4096, (5,)), (20, 20, 1024, 1024, (5, 4)), (6, 20, 1024, 1024, (5,)), (2, 3, 200, 333, (4, 5, 7, 9, 10)), (1, 2, 256, 128, (4, 5, 9, 10)),
                                (1, 2, 130, 64, (4, 5))]:
        qkv = torch.randn(B, max(Nq, Nk), 3 * H * 64, device="cuda", generator=g).to(dtt)
        q, k, v = qkv[:, :Nq, :H * 64], qkv[:, :Nk, H * 64:2 * H * 64], qkv[:, :Nk, 2 * H * 64:]
        if Nk == 333:      # late outlier keys: the lazy variants' exact slow path
            k = k.clone()
            k[:, 300:303] = (q[:, 5:8] * 6).to(dtt)
        outs = [torch.empty(B, Nq, H * 64, device="cuda", dtype=dtt) for _ in range(2)]
        for vp in vps:
            if dtt == torch.bfloat16 and Nq == 4096:
                continue
            compare(f"ed_flash_attention v_path {vp} {str(dtt)[6:]} B={B} H={H} Nq={Nq} Nk={Nk}",
                    lambda L, o: L.ed_flash_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), code, B, H, Nq, Nk, 64, q.stride(0), q.stride(1),
                                                      k.stride(0), k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1), 64 ** -0.5, vp, st()),
                    outs, 4.0 * B * H * Nq * Nk * 64, rounds=3 if Nq < 1024 else a.rounds)
