"""Loader / builder for libelastic_hip.so (the C ABI declared in include/elastic_hip.h).

The library is built in-tree by ``hipcc --offload-arch=gfx950`` (``build_library``; driven by
``__graft_entry__.build()``) and loaded with ctypes *after* ``import torch`` so that the process holds one HIP
runtime (libamdhip64.so.7, the SONAME torch-ROCm bundles).  There is NO fallback: if the shared object is missing or
does not load, every product entry point raises.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must be imported before the .so so libamdhip64 is already resident)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_DIR = os.path.dirname(PKG_DIR)
SOURCES = [os.path.join(PKG_DIR, "csrc", n) for n in ("elastic_kernels.hip", "unet_kernels.hip", "attention_kernels.hip", "gemm_kernels.hip", "vae_kernels.hip")]
SRC = SOURCES[0]
INCLUDE = os.path.join(ROOT_DIR, "include")
SO_PATH = os.path.join(PKG_DIR, "libelastic_hip.so")
ABI_VERSION = 9

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

_i32p = ctypes.POINTER(ctypes.c_int32)
_vp, _i, _f, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64

# name -> argtypes; mirrors include/elastic_hip.h one to one (tests/test_abi.py checks header <-> this table <-> .so)
SIGNATURES = {
    "ed_version": [],
    "ed_error_string": [_i],
    "ed_gather_views": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _f, _vp],
    "ed_scatter_centres": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "ed_pick_assemble": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ed_unpad_direction": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ed_fill_directions": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ed_cfg_ddim_step": [_vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i64, _vp],
    "ed_undo_step": [_vp, _vp, _vp, _vp, _i, _i64, _vp],
    "ed_rrg_update": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _vp],
    "ed_gather2d": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ed_tile_gather_pad": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _vp],
    "ed_tile_accumulate_normalise": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "ed_geglu": [_vp, _vp, _i, _i64, _i, _vp],
    "ed_groupnorm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ed_groupnorm_workspace": [_i, _i, _i, _i],
    "ed_bias_residual_add": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ed_add_layernorm": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp],
    "ed_tokens_add_nchw": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ed_layernorm": [_vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp],
    "ed_layernorm_s32": [_vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp],
    "ed_add_layernorm_s32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _vp],
    "ed_groupnorm_nhwc_s32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ed_groupnorm_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ed_groupnorm_nhwc_workspace": [_i, _i, _i, _i],
    "ed_groupnorm_nhwc_cat": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ed_assemble_rows": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                         _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ed_phase_epilogue": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f,
                          _vp],
    "ed_softmax_rows": [_vp, _i64, _i64, _f, _vp],
    "ed_groupnorm_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "ed_groupnorm_f32_workspace": [_i, _i, _i, _i],
    "ed_flash_attention": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _vp],
    "ed_geglu_gemm": [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp],
    "ed_linear": [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp],
    "ed_conv3x3_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ed_conv3x3_nhwc_up2x": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ed_conv3x3_nhwc_s2": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ed_absmax_f32": [_vp, _i64, _vp, _vp],
    "ed_split_f32_nhwc": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "ed_groupnorm_nhwc_f32_workspace": [_i, _i, _i, _i],
    "ed_groupnorm_nhwc_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ed_conv3x3_nhwc_f32out": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "ed_conv3x3_nhwc_f32out_s2": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp],
}

_LIB = None


def build_library(force=False, verbose=False):
    """hipcc cross-compiles gfx950 without a GPU; the .so is git-ignored but travels to the GPU box."""
    deps = SOURCES + [os.path.join(INCLUDE, "elastic_hip.h")]
    if not force and os.path.isfile(SO_PATH) and os.path.getmtime(SO_PATH) >= max(os.path.getmtime(d) for d in deps):
        return SO_PATH
    # one hipcc per translation unit, side by side (the five files are independent; ~25 s instead of ~60 s), then one link
    import tempfile
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="ed_build_") as tmp:
        objs = [os.path.join(tmp, os.path.basename(src) + ".o") for src in SOURCES]
        cmds = [["hipcc", *flags, "-I", INCLUDE, "-c", src, "-o", obj] for src, obj in zip(SOURCES, objs)]
        if verbose:
            for c in cmds:
                print(" ".join(c))
        procs = [subprocess.Popen(c) for c in cmds]
        codes = [p.wait() for p in procs]
        if any(codes):
            raise subprocess.CalledProcessError(next(c for c in codes if c), cmds[[bool(c) for c in codes].index(True)])
        link = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", SO_PATH + ".tmp"]
        if verbose:
            print(" ".join(link))
        subprocess.run(link, check=True)
        os.replace(SO_PATH + ".tmp", SO_PATH)      # never leave a half-written library where lib() would load it
    return SO_PATH


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.isfile(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: the HIP kernels are the product path and there is no CPU fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        L = ctypes.CDLL(SO_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export what the header declares
            fn.argtypes = argtypes
            fn.restype = (ctypes.c_char_p if name == "ed_error_string" else
                          ctypes.c_int64 if name.endswith("_workspace") else ctypes.c_int)
        if L.ed_version() != ABI_VERSION:
            raise RuntimeError(f"libelastic_hip.so ABI {L.ed_version()} != expected {ABI_VERSION}; rebuild")
        _LIB = L
    return _LIB


def check(err, what):
    if err != 0:
        msg = lib().ed_error_string(err)
        raise RuntimeError(f"{what}: HIP error {err} ({msg.decode() if msg else '?'})")
