"""VALU issue rates on the MI355X (tools/mfma_power/mfma_power.hip: k_valu_loop): time per loop iteration and SIMD, two waves per SIMD.
    python tools/mfma_power/valu.py"""
import ctypes
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libmfma_power.so"))
_vp, _i = ctypes.c_void_p, ctypes.c_int
L.ed_valu_loop.argtypes = [_i, _vp, _vp, _i, _i, _vp]
ITERS = 40000
DESC = {0: "4 v_pk_fma_f32 (8 FMAs)", 1: "8 v_exp_f32 + 4 v_pk_add_f32", 2: "8 x (v_fma + v_exp + v_add) + 4 v_pk_add_f32 = 8 lazy-softmax numerators",
        3: "as 2 + 4 v_mfma_f32_16x16x32_f16", 4: "as 2 + 2 v_mfma_f32_32x32x16_f16 (the same MACs; 8 accumulators in rotation)",
        5: "4 v_mfma_f32_16x16x32_f16 alone", 6: "2 v_mfma_f32_32x32x16_f16 alone"}
for blocks, label in ((256, "2 waves / SIMD"), (512, "4 waves / SIMD")):
    x = (torch.rand(blocks * 512 * 8, device="cuda") - 0.5)
    out = torch.empty(blocks * 512, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for op in (0, 1, 2, 3, 4, 5, 6):
        L.ed_valu_loop(op, x.data_ptr(), out.data_ptr(), blocks, 100, st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert L.ed_valu_loop(op, x.data_ptr(), out.data_ptr(), blocks, ITERS, st) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[2]
        waves_per_simd = blocks * 8 / 1024
        ns_iter = ms * 1e6 / ITERS                      # all waves of a SIMD run concurrently: time per iteration of the set
        print(json.dumps({"occupancy": label, "loop_body": DESC[op], "ms": round(ms, 3), "ns_per_iteration_per_simd": round(ns_iter, 2),
                          "ns_per_iteration_per_wave": round(ns_iter / waves_per_simd, 2),
                          "cycles_per_iteration_per_wave_at_2.4GHz": round(2.4 * ns_iter / waves_per_simd, 1)}), flush=True)
