"""Regenerate tools/attn16/attn5_inregion.hip from csrc/attention_kernels.hip: the product's pipelined lazy attention kernel with the next
tiles' global loads / LDS writes issued inside the MFMA region (four textual edits on a copy of `pipe_region` and `k_flash_attn_pipe`).
    python tools/attn16/make_attn5_inregion.py"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
src = open(os.path.join(ROOT, "elasticdiffusion_official_amd", "csrc", "attention_kernels.hip")).read()
cur = open(os.path.join(HERE, "attn5_inregion.hip")).read()
i0 = src.index("template <typename T, bool HAS_PV, bool HAS_NEXT, bool LAZY, bool EXP2, int DEPTH>\n__device__ __forceinline__ void pipe_region(")
i1 = src.index("// WAVES = waves per workgroup (4: 128 query rows")
region = src[i0:i1]
k0 = src.index("template <typename T, bool LAZY, bool EXP2 = false, int DEPTH = 1, int WAVES = 4>\n__global__")
k1 = src.index("// ---------------------------------------------------------------------------------------------------------------------\n// small-KV (cross-attention) kernel")
kern = src[k0:k1]


def sub(text, old, new):
    assert old in text, old[:60]
    return text.replace(old, new)


region = sub(region, "template <typename T, bool HAS_PV, bool HAS_NEXT, bool LAZY, bool EXP2, int DEPTH>\n__device__ __forceinline__ void pipe_region(",
             "template <typename T, bool HAS_PV, bool HAS_NEXT, bool LAZY, bool EXP2, int DEPTH, class Filler>\n__device__ __forceinline__ void pipe_region_ir(Filler&& filler, ")
region = sub(region, "    else softmax_slice<T>(i, s_cur, sl, run, p_cur);\n    __builtin_amdgcn_sched_barrier(0);",
             "    else softmax_slice<T>(i, s_cur, sl, run, p_cur);\n    filler(i);\n    __builtin_amdgcn_sched_barrier(0);")
kern = sub(kern, "k_flash_attn_pipe(const Params p) {", "k_flash_attn_pipe_ir(const Params p) {")
kern = sub(kern, """    load_k(t + 2);  // unconditional: a tile past the end reads as zeros (buffer bounds check) into a buffer nobody reads
    load_v(t + 1);
""", """    // in-region variant: the next tiles' global loads go behind MFMA slots 0..3 of the region, their LDS writes behind slots 11..14
    const uint32_t kbase = k_off + (uint32_t)(t + 2) * 2u * k_half, vbase = v_off + (uint32_t)(t + 1) * 2u * v_half;
    uint16_t* const kdst = sm.k[kb_write];
    uint16_t* const vdst = sm.v[vb_next];
    auto filler = [&](int i) {
      if (i == 0) kreg[0] = buf_load16(k_rs, kbase, 0);
      else if (i == 1) kreg[1] = buf_load16(k_rs, kbase + k_half, 0);
      else if (i == 2) vreg[0] = buf_load16(v_rs, vbase, 0);
      else if (i == 3) vreg[1] = buf_load16(v_rs, vbase + v_half, 0);
      else if (i == 11) *reinterpret_cast<Vec16*>(&kdst[st_row * K_LD + st_col]) = kreg[0];
      else if (i == 12) *reinterpret_cast<Vec16*>(&kdst[(st_row + 32) * K_LD + st_col]) = kreg[1];
      else if (i == 13) *reinterpret_cast<Vec16*>(&vdst[st_row * V_LD_TR + st_col]) = vreg[0];
      else if (i == 14) *reinterpret_cast<Vec16*>(&vdst[(st_row + 32) * V_LD_TR + st_col]) = vreg[1];
    };
""")
kern = sub(kern, "pipe_region<T, decltype(has_pv)::value, decltype(has_next)::value, lazy, EXP2, DEPTH>(sm.k[kb_next],",
           "pipe_region_ir<T, decltype(has_pv)::value, decltype(has_next)::value, lazy, EXP2, DEPTH>(filler, sm.k[kb_next],")
kern = sub(kern, "    write_k(kb_write);\n    write_v(vb_next);\n    const int tmp = vb_prev;", "    const int tmp = vb_prev;")
# edit 5: the prologue's three tile loads (K(0), V(0), K(1)) in flight together instead of K(1) after the first two have been waited for and
# written: one memory round trip less before the first MFMA (tools/attn16's stamps: the prologue is 9-15 % of a wave's cycles)
kern = sub(kern, """  load_k(0);
  load_v(0);
  write_k(0);
  write_v(0);
  if (n_tiles > 1) {
    load_k(1);
    write_k(1);
  }
  __syncthreads();
""", """  load_k(0);
  load_v(0);
  Vec16 k1reg[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) k1reg[i] = buf_load16(k_rs, k_off + 2u * k_half + i * k_half, 0);   // K(1): rows past Nk read as zeros
  write_k(0);
  write_v(0);
#pragma unroll
  for (int i = 0; i < NST; ++i) *reinterpret_cast<Vec16*>(&sm.k[1][(st_row + 32 * i) * K_LD + st_col]) = k1reg[i];
  __syncthreads();
""")
kern = sub(kern, 'static_assert(WAVES == 4 || WAVES == 8, "4 or 8 waves per workgroup");',
           'static_assert(WAVES == 4 && LAZY && !EXP2, "the in-region experiment covers the default variant only");')
head = cur[:cur.index("namespace {\n\n") + len("namespace {\n\n")]
tail = cur[cur.index("\n}  // namespace\n\nextern \"C\" int ed_x_flash_attention5_inregion"):]
new = head + region + "\n" + kern + tail
open(os.path.join(HERE, "attn5_inregion.hip"), "w").write(new)
print("unchanged" if new == cur else "rewritten")
