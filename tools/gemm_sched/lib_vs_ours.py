"""Kernel-level comparison for the shape policy (ops.linear_wins): hipBLASLt (F.linear) vs the product's ed_linear vs the experiments, every
call timed as a hipGraph of 10 launches (no host time in the number -- tools/probe_gemm.py times Python wrappers back to back, which
overstates short kernels: 80.7 vs 67 us for [20480, 1280 -> 1280]).
    python tools/gemm_sched/lib_vs_ours.py [--rounds 5]"""
import argparse
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
import torch.nn.functional as F

from elasticdiffusion_official_amd import _hip

_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--patched", action="store_true", help="add an arm for tools/r5_patches/build/libelastic_hip_patched.so (ed_linear)")
a = ap.parse_args()
prod = _hip.lib()
S = ctypes.CDLL(os.path.join(HERE, "libgemm_sched.so"))
S.ed_s_linear.argtypes = [_i, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]
P = ctypes.CDLL(os.path.join(os.path.dirname(HERE), "gemm_persist", "libgemm_persist.so"))
P.ed_p_linear.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]
P.ed_p2_linear.argtypes = P.ed_p_linear.argtypes
PAT = None
if a.patched:
    PAT = ctypes.CDLL(os.path.join(os.path.dirname(HERE), "r5_patches", "build", "libelastic_hip_patched.so"))
    PAT.ed_linear.argtypes = _hip.SIGNATURES["ed_linear"]
st = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
g = torch.Generator().manual_seed(0)
dt = torch.float16


def graphed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()

    def timed():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay()
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    return timed


shapes = [(20480, 1280, 1280, 192), (20480, 1280, 3840, 60), (20480, 5120, 1280, 60), (20480, 2560, 1280, 2), (20480, 640, 1280, 1),
          (6144, 1280, 1280, 192), (6144, 1280, 3840, 60), (6144, 5120, 1280, 60), (24576, 640, 640, 40), (24576, 2560, 640, 10),
          (81920, 640, 640, 40), (81920, 2560, 640, 10)]
for (M, K, N, calls) in shapes:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to("cuda", dt)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
    b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
    o = torch.empty(M, N, device="cuda", dtype=dt)
    ref = F.linear(x, w, b)
    arms = {
        "hipblaslt": lambda: F.linear(x, w, b),
        "product": lambda: prod.ed_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, o.data_ptr(), 1, M, K, N, st()),
        "no_addend_epilogue": lambda: S.ed_s_linear(0, x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, N, st()),
        "four_interval_loop": lambda: S.ed_s_linear(6, x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, N, st()),
        "persistent_v1": lambda: P.ed_p_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, o.data_ptr(), 1, M, K, N, 256, st()),
        "persistent_v2": lambda: P.ed_p2_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, o.data_ptr(), 1, M, K, N, 256, st()),
    }
    if PAT is not None:
        arms["patched_library"] = lambda: PAT.ed_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, o.data_ptr(), 1, M, K, N, st())
    timers = {k: graphed(f) for k, f in arms.items()}
    err = float((o.float() - ref.float()).abs().max())
    ts = {k: [] for k in arms}
    for _ in range(a.rounds):
        for k, t in timers.items():
            ts[k].append(t())
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    flops = 2.0 * M * K * N
    rec = {"shape": [M, K, N], "calls_per_forward": calls, "tiles": -(-M // 256) * -(-N // 256), "max_abs_diff_vs_lib": err}
    for k in arms:
        rec[k] = {"us": round(1e3 * med[k], 1), "tflops": round(flops / med[k] / 1e9, 1)}
    best = min((k for k in arms if k != "hipblaslt"), key=lambda k: med[k])
    rec["best_of_ours"] = best
    rec["lib_over_best"] = round(med["hipblaslt"] / med[best], 3)
    rec["lib_over_product"] = round(med["hipblaslt"] / med["product"], 3)
    print(json.dumps(rec), flush=True)
