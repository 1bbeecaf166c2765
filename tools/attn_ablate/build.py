"""Ablation builds of the pipelined attention kernel (csrc/attention_kernels.hip, v_path 4): textual patches that remove ONE
ingredient each -- results are numerically wrong on purpose, only the timing is read (CDNA guide 5.4: "ablate empirically
before optimising"; values are kept live with empty asm statements so nothing upstream is dead-code-eliminated).
Cross-compiled here (no GPU needed); tools/attn_ablate/run.py times them on the MI355X.

    python tools/attn_ablate/build.py        -> tools/attn_ablate/libattn_<variant>.so
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = open(os.path.join(ROOT, "elasticdiffusion_official_amd", "csrc", "attention_kernels.hip")).read()


def sub(text, old, new, count=1):
    assert text.count(old) >= 1, old[:60]
    return text.replace(old, new, count)


def no_exp(t):          # the 33 v_exp_f32 of the exact softmax slices become plain moves of their (live) arguments
    i = t.index("__device__ __forceinline__ void softmax_slice(")
    j = t.index("// LAZY variant (v_path 5)")
    body = t[i:j].replace("__builtin_amdgcn_exp2f(", "(")
    return t[:i] + body + t[j:]


def no_softmax(t):      # no VALU softmax at all: P = 16-bit pack of S (4 packs per tile stay)
    i = t.index("__device__ __forceinline__ void softmax_slice(")
    j = t.index("// LAZY variant (v_path 5)")
    new = '''__device__ __forceinline__ void softmax_slice(int i, f32x16 (&s)[2], float sl, SoftmaxRun& r, typename T::v8 (&pf)[4]) {
  if ((i & 3) == 3) {
    const int g = i >> 2;
    f32x8 pv;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = s[g >> 1][8 * (g & 1) + j];
    pf[g] = T::pack(pv);
    asm volatile("" : "+v"(pf[g]));
  }
  r.alpha = 1.0f;
}

'''
    return t[:i] + new + t[j:]


def no_lds_reads(t):    # MFMA A operands from registers (an opaque per-lane constant) instead of ds_read_b128 / ds_read_b64_tr_b16
    old = '''  auto fetch = [&](int i) -> Vec16 {
    if (i < NQK) return *reinterpret_cast<const Vec16*>(&k_next[(32 * (i & 1) + ln) * K_LD + 16 * (i >> 1) + 8 * hi]);
    const int j = i - NQK;
    return v_frag_tr(v_prev, lane, hi, j >> 1, j & 1);
  };'''
    new = '''  auto fetch = [&](int i) -> Vec16 {
    Vec16 c = {{0x3c003c00u + (uint32_t)lane, 0x3c003c00u, 0x3c003c00u + (uint32_t)i, 0x3c003c00u}};
    asm volatile("" : "+v"(c.w[0]), "+v"(c.w[1]), "+v"(c.w[2]), "+v"(c.w[3]));
    return c;
  };'''
    return sub(t, old, new)


def no_staging(t):      # no global loads / LDS writes of K and V tiles inside the loop (barrier stays)
    t = sub(t, "    load_k(t + 2);  // unconditional", "    // load_k(t + 2);  // unconditional")
    t = sub(t, "    load_v(t + 1);\n    // the first tile", "    // load_v(t + 1);\n    // the first tile")
    t = sub(t, "    write_k(kb_write);\n    write_v(vb_next);", "    // write_k / write_v removed")
    return t


def no_barrier(t):      # no per-tile workgroup barrier
    old = '''    kb_cur = kb_next, kb_next = kb_write, kb_write = NK == 3 ? ktmp : kb_cur;
    __syncthreads();'''
    return sub(t, old, old.replace("    __syncthreads();", "    // barrier removed"))


def no_mfma(t):         # the 16 MFMAs of the region become register moves (accumulators stay live): VALU + LDS + staging only
    i = t.index("__device__ __forceinline__ void pipe_region(")
    j = t.index("template <typename T, bool LAZY, bool EXP2 = false, int DEPTH = 1>")
    body = t[i:j]
    body = re.sub(r"T::mfma\(as_v8<typename T::v8>\(a_cur\), qf\[0\], negm\)", "negm", body)
    body = re.sub(r"T::mfma\(as_v8<typename T::v8>\(a_cur\), qf\[i >> 1\], s_next\[i & 1\]\)", "s_next[i & 1]", body)
    body = re.sub(r"T::mfma\(as_v8<typename T::v8>\(a_cur\), p_prev\[j >> 1\], o\[j & 1\]\)", "o[j & 1]", body)
    body = body.replace("const Vec16 a_cur = ring[i % (DEPTH + 1)];", "Vec16 a_cur = ring[i % (DEPTH + 1)];\n      asm volatile(\"\" :: \"v\"(a_cur.w[0]), \"v\"(a_cur.w[1]), \"v\"(a_cur.w[2]), \"v\"(a_cur.w[3]));")
    return t[:i] + body + t[j:]


VARIANTS = {
    "base": [],
    "noexp": [no_exp],
    "nosoftmax": [no_softmax],
    "nolds": [no_lds_reads],
    "nostage": [no_staging],
    "nobar": [no_barrier],
    "nomfma": [no_mfma],
    "mfma_only": [no_softmax, no_lds_reads, no_staging, no_barrier],
    "nolds_nostage_nobar": [no_lds_reads, no_staging, no_barrier],
}

if __name__ == "__main__":
    only = sys.argv[1:]
    for name, patches in VARIANTS.items():
        if only and name not in only:
            continue
        text = SRC
        for p in patches:
            text = p(text)
        src = os.path.join("/tmp", f"attn_{name}.hip")
        open(src, "w").write(text)
        so = os.path.join(HERE, f"libattn_{name}.so")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
               src, "-o", so]
        r = subprocess.run(cmd, capture_output=True, text=True)
        print(name, "ok" if r.returncode == 0 else "FAILED\n" + r.stderr[-1500:], flush=True)
