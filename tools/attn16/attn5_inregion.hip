// attn5_inregion.hip -- EXPERIMENT for round 5 (not in libelastic_hip.so): the PRODUCT's pipelined lazy attention kernel (v_path 5, 32x32x16
// layout) with the next tiles' global loads issued behind MFMA slots 0..3 of the loop's MFMA region and their LDS writes behind slots 11..14,
// instead of in front of / behind the region (tools/attn16 measured +3...5 % for this change on the 16x16x32 rebuild).  The kernel text below
// is generated from csrc/attention_kernels.hip by tools/attn16/make_attn5_inregion.py (region + kernel copied, five edits -- the fifth: the
// prologue's K(0), V(0), K(1) loads in flight together); results must be
// bit-identical to ed_flash_attention(v_path = 5).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include tools/attn16/attn5_inregion.hip -o tools/attn16/libattn5_inregion.so
#include "../../elasticdiffusion_official_amd/csrc/attention_kernels.hip"

namespace {

template <typename T, bool HAS_PV, bool HAS_NEXT, bool LAZY, bool EXP2, int DEPTH, class Filler>
__device__ __forceinline__ void pipe_region_ir(Filler&& filler, const uint16_t* k_next, const uint16_t* v_prev, int lane, int ln, int hi,
                                            const typename T::v8 (&qf)[4], f32x16 (&s_cur)[2], f32x16 (&s_next)[2],
                                            const typename T::v8 (&p_prev)[4], typename T::v8 (&p_cur)[4],
                                            f32x16 (&o)[2], float sl, SoftmaxRun& run, const f32x16& negm) {
  constexpr int NQK = HAS_NEXT ? 8 : 0, N = NQK + (HAS_PV ? 8 : 0);
  auto fetch = [&](int i) -> Vec16 {
    if (i < NQK) return *reinterpret_cast<const Vec16*>(&k_next[(32 * (i & 1) + ln) * K_LD + 16 * (i >> 1) + 8 * hi]);
    const int j = i - NQK;
    return v_frag_tr(v_prev, lane, hi, j >> 1, j & 1);
  };
  if (HAS_NEXT && !EXP2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s_next[0][i] = s_next[1][i] = 0.f;
  }
  Vec16 ring[DEPTH + 1];   // statically indexed (the loop is fully unrolled): fragment i lives in ring[i % (DEPTH + 1)]
#pragma unroll
  for (int r = 0; r <= DEPTH; ++r) ring[r] = Vec16{{0u, 0u, 0u, 0u}};
#pragma unroll
  for (int r = 0; r < DEPTH; ++r)
    if (r < N) ring[r] = fetch(r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < N) {
      if (i + DEPTH < N) ring[(i + DEPTH) % (DEPTH + 1)] = fetch(i + DEPTH);
      const Vec16 a_cur = ring[i % (DEPTH + 1)];
      if (i < NQK) {
        if (EXP2 && i < 2) s_next[i & 1] = T::mfma(as_v8<typename T::v8>(a_cur), qf[0], negm);
        else s_next[i & 1] = T::mfma(as_v8<typename T::v8>(a_cur), qf[i >> 1], s_next[i & 1]);
        asm volatile("" : "+v"(s_next[i & 1]));
      } else {
        const int j = i - NQK;
        o[j & 1] = T::mfma(as_v8<typename T::v8>(a_cur), p_prev[j >> 1], o[j & 1]);
        asm volatile("" : "+v"(o[j & 1]));
      }
    }
    if (LAZY) softmax_slice_lazy<T, EXP2>(i, s_cur, sl, run, p_cur);
    else softmax_slice<T>(i, s_cur, sl, run, p_cur);
    filler(i);
    __builtin_amdgcn_sched_barrier(0);
  }
}


template <typename T, bool LAZY, bool EXP2 = false, int DEPTH = 1, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, 2)
k_flash_attn_pipe_ir(const Params p) {
  static_assert(LAZY || !EXP2, "the exponent-domain variant is built on the lazy-maximum loop");
  static_assert(WAVES == 4 && LAZY && !EXP2, "the in-region experiment covers the default variant only");
  constexpr int NK = LAZY ? 3 : 2;
  constexpr int NST = 8 / WAVES;   // 16-byte chunks of a 64 x 64 tile per thread: 512 chunks over 64 WAVES threads
  __shared__ SmemPipe<NK> sm;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hi = lane >> 5;
  int bh, qblk;
  {
    const int id = blockIdx.x, per = 8 * p.nqb, grp = id / per, r = id - grp * per;
    if ((grp + 1) * 8 <= p.BH) {
      bh = grp * 8 + (r & 7);
      qblk = r >> 3;
    } else {
      bh = grp * 8 + r / p.nqb;
      qblk = r % p.nqb;
    }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const uint16_t* qg = p.q + b * p.q_sb + h * D;
  const uint16_t* kg = p.k + b * p.k_sb + h * D;
  const uint16_t* vg = p.v + b * p.v_sb + h * D;
  uint16_t* og = p.o + b * p.o_sb + h * D;

  const int q_row = qblk * (32 * WAVES) + wave * 32 + ln;
  typename T::v8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = as_v8<typename T::v8>(load_row16(qg, p.q_sn, q_row, p.Nq, 16 * ks + 8 * hi));

  // K / V tiles through buffer loads: one wave-uniform descriptor per tensor whose size ends with the last valid row, a
  // 32-bit per-lane byte offset and a scalar tile offset -- rows past Nk read as zeros (hardware bounds check), no
  // per-lane 64-bit addresses, no predication branches
  const BufRsrc k_rs = make_rsrc(kg, ((int64_t)(p.Nk - 1) * p.k_sn + D) * 2);
  const BufRsrc v_rs = make_rsrc(vg, ((int64_t)(p.Nk - 1) * p.v_sn + D) * 2);
  const int st_row = tid >> 3, st_col = (tid & 7) * 8;
  const uint32_t k_off = (uint32_t)(((int64_t)st_row * p.k_sn + st_col) * 2), k_half = (uint32_t)(32 * p.k_sn * 2);
  const uint32_t v_off = (uint32_t)(((int64_t)st_row * p.v_sn + st_col) * 2), v_half = (uint32_t)(32 * p.v_sn * 2);
  Vec16 kreg[NST], vreg[NST];   // thread's rows: st_row (+ 32 with 4 waves); with 8 waves st_row already spans the 64 rows
  // The whole byte offset goes into the per-lane (VGPR) offset: the hardware range check of a raw buffer load covers
  // VGPR offset + immediate only, a scalar offset is added AFTER it -- a tile offset passed there would let the
  // unconditional loads of tiles past the end, and the rows past Nk of a ragged tile, read whatever follows the tensor.
  auto load_k = [&](int t) {
    const uint32_t base = k_off + (uint32_t)t * 2u * k_half;
#pragma unroll
    for (int i = 0; i < NST; ++i) kreg[i] = buf_load16(k_rs, base + i * k_half, 0);
  };
  auto load_v = [&](int t) {
    const uint32_t base = v_off + (uint32_t)t * 2u * v_half;
#pragma unroll
    for (int i = 0; i < NST; ++i) vreg[i] = buf_load16(v_rs, base + i * v_half, 0);
  };
  auto write_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NST; ++i) *reinterpret_cast<Vec16*>(&sm.k[buf][(st_row + 32 * i) * K_LD + st_col]) = kreg[i];
  };
  auto write_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NST; ++i) *reinterpret_cast<Vec16*>(&sm.v[buf][(st_row + 32 * i) * V_LD_TR + st_col]) = vreg[i];
  };

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) oacc[0][i] = oacc[1][i] = 0.f;
  SoftmaxRun run;
  run.mb = -INFINITY, run.l = 0.f, run.mx = 0.f, run.use = 0.f, run.alpha = 1.f, run.psum = 0.f;
  const float sl = EXP2 ? 1.0f : p.scale_log2e;  // EXP2: q arrives multiplied by scale * log2(e)
  const int n_tiles = (p.Nk + KT - 1) / KT, n_full = p.Nk / KT;
  f32x16 negm;  // EXP2: -mb in every register (the C operand that starts both S chains of a tile)
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;

  load_k(0);
  load_v(0);
  Vec16 k1reg[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) k1reg[i] = buf_load16(k_rs, k_off + 2u * k_half + i * k_half, 0);   // K(1): rows past Nk read as zeros
  write_k(0);
  write_v(0);
#pragma unroll
  for (int i = 0; i < NST; ++i) *reinterpret_cast<Vec16*>(&sm.k[1][(st_row + 32 * i) * K_LD + st_col]) = k1reg[i];
  __syncthreads();

  // ping-pong register sets (named, statically indexed: no copies between iterations)
  f32x16 sA[2], sB[2];
  typename T::v8 pA[4], pB[4];
  int vb_prev = 2, vb_cur = 0, vb_next = 1;  // V buffers of tiles t-1, t, t+1 (mod 3)
  int kb_cur = 0, kb_next = 1, kb_write = NK == 3 ? 2 : 0;  // K buffers of tiles t, t+1 and the one tile t+2 is staged into

  // one full (unmasked) tile t: S_cur holds K(t) Q^T on entry
  auto iter = [&](auto has_pv, auto has_next, int t, f32x16 (&s_cur)[2], f32x16 (&s_next)[2],
                  typename T::v8 (&p_prev)[4], typename T::v8 (&p_cur)[4]) {
    // in-region variant: the next tiles' global loads go behind MFMA slots 0..3 of the region, their LDS writes behind slots 11..14
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
    // the first tile (no PV yet) takes the exact softmax; EXP2 found its maximum before the loop and is lazy throughout
    constexpr bool lazy = LAZY && (EXP2 || decltype(has_pv)::value);
    pipe_region_ir<T, decltype(has_pv)::value, decltype(has_next)::value, lazy, EXP2, DEPTH>(filler, sm.k[kb_next], sm.v[vb_prev], lane, ln,
                                                                                          hi, qf, s_cur, s_next, p_prev, p_cur,
                                                                                          oacc, sl, run, negm);
    if (lazy) {
      run.alpha = 1.0f;
      if (__any(!(run.psum <= RESCALE_SUM_MAX))) {  // (also catches inf / NaN sums)
        const float mb_old = run.mb;
        resoftmax_tile<T>(sm.k[kb_cur], ln, hi, qf, s_cur, sl, run, p_cur);
        if (EXP2) {  // S(t+1) was started from the old reference: move it (and the next chains' start) to the new one
          const float d = run.mb - mb_old;
          if (decltype(has_next)::value) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s_next[0][i] -= d, s_next[1][i] -= d;
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) negm[i] = -run.mb;
          asm volatile("" : "+v"(negm));
        }
      }
      run.l = __builtin_fmaf(run.l, run.alpha, run.psum);
    }
    if (__any(run.alpha != 1.0f)) {  // first tile, or a row's maximum grew by more than 2^RESCALE_LOG2 (rare)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        oacc[0][i] *= run.alpha;
        oacc[1][i] *= run.alpha;
      }
    }
    const int tmp = vb_prev;
    vb_prev = vb_cur, vb_cur = vb_next, vb_next = tmp;
    const int ktmp = kb_cur;  // two buffers: (cur, next, write) = (a, b, a) -> (b, a, b); three: a rotation
    kb_cur = kb_next, kb_next = kb_write, kb_write = NK == 3 ? ktmp : kb_cur;
    __syncthreads();
  };

  if (n_full > 0) {
    qk_tile<T>(sm.k[0], ln, hi, qf, sA);
    if (EXP2) {  // exact row maximum of tile 0 = the first reference; from here on S is produced relative to it
      float mx = sA[0][0];
#pragma unroll
      for (int j = 1; j < 16; ++j) mx = fmaxf(mx, sA[0][j]);
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, sA[1][j]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      run.mb = mx;
#pragma unroll
      for (int i = 0; i < 16; ++i) sA[0][i] -= mx, sA[1][i] -= mx, negm[i] = -mx;
      asm volatile("" : "+v"(negm));
    }
    // tile 0's iteration ends by overwriting sm.k[0] with tile 2: every wave must be done with the reads above first (inside
    // the loop the barrier that closes iteration t-1 plays that role for the buffer iteration t overwrites)
    __syncthreads();
    if (n_full == 1) {
      iter(False{}, False{}, 0, sA, sB, pB, pA);
    } else {
      iter(False{}, True{}, 0, sA, sB, pB, pA);
      int t = 1;  // odd tiles: S in sB, P(t-1) in pA; even tiles: S in sA, P(t-1) in pB
      for (; t + 2 < n_full; t += 2) {
        iter(True{}, True{}, t, sB, sA, pA, pB);
        iter(True{}, True{}, t + 1, sA, sB, pB, pA);
      }
      if (n_full - t == 2) {
        iter(True{}, True{}, t, sB, sA, pA, pB);
        iter(True{}, False{}, t + 1, sA, sB, pB, pA);
      } else {
        iter(True{}, False{}, t, sB, sA, pA, pB);
      }
    }
    // drain: O += V(n_full-1)^T P(n_full-1)^T  (the last tile's V is in vb_prev after the final rotation)
    if ((n_full - 1) & 1) pv_tile<T>(sm.v[vb_prev], lane, hi, pB, oacc);
    else pv_tile<T>(sm.v[vb_prev], lane, hi, pA, oacc);
  }

  if (n_tiles > n_full) {  // ragged last tile: un-pipelined, keys past Nk masked, unconditional rescale
    f32x16 s[2];
    qk_tile<T>(sm.k[kb_cur], ln, hi, qf, s);  // after n_full rotations kb_cur is the buffer of tile n_full
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (n_full * KT + 32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3) >= p.Nk) s[kb][r] = -INFINITY;
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float use = fmaxf(run.mb, mx * sl);
    const float alpha = __builtin_amdgcn_exp2f(run.mb - use);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], sl, -use));
        s[kb][r] = e;
        psum += e;
      }
    run.l = __builtin_fmaf(run.l, alpha, psum);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      oacc[0][i] *= alpha;
      oacc[1][i] *= alpha;
    }
    typename T::v8 pf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      f32x8 pv;
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[st >> 1][8 * (st & 1) + j];
      pf[st] = T::pack(pv);
    }
    pv_tile<T>(sm.v[vb_cur], lane, hi, pf, oacc);
  }

  const float l_tot = run.l + __shfl_xor(run.l, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_row < p.Nq) {
    uint16_t* orow = og + (int64_t)q_row * p.o_sn;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x8 tmp;
#pragma unroll
        for (int e = 0; e < 4; ++e) tmp[e] = oacc[db][4 * g + e] * inv, tmp[4 + e] = 0.f;
        const Vec16 packed = __builtin_bit_cast(Vec16, T::pack(tmp));
        Vec8 out8 = {{packed.w[0], packed.w[1]}};
        *reinterpret_cast<Vec8*>(orow + 32 * db + 8 * g + 4 * hi) = out8;
      }
  }
}


}  // namespace

extern "C" int ed_x_flash_attention5_inregion(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H, int Nq, int Nk,
                                              int64_t q_sb, int64_t q_sn, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn,
                                              int64_t o_sb, int64_t o_sn, float scale, void* stream) {
  if (B == 0 || H == 0 || Nq == 0) return 0;
  if (Nk <= 0) return (int)hipErrorInvalidValue;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15u) || ((uintptr_t)out & 7u)) return (int)hipErrorInvalidValue;
  if ((q_sb | q_sn | k_sb | k_sn | v_sb | v_sn) % 8 || (o_sb | o_sn) % 4) return (int)hipErrorInvalidValue;
  if ((int64_t)(Nk + 2 * KT) * (k_sn > v_sn ? k_sn : v_sn) * 2 >= 0x7fffffffll) return (int)hipErrorInvalidValue;
  Params p;
  p.q = (const uint16_t*)q, p.k = (const uint16_t*)k, p.v = (const uint16_t*)v, p.o = (uint16_t*)out;
  p.Nq = Nq, p.Nk = Nk, p.H = H, p.BH = B * H, p.nqb = (Nq + QB - 1) / QB;
  p.q_sb = q_sb, p.q_sn = q_sn, p.k_sb = k_sb, p.k_sn = k_sn, p.v_sb = v_sb, p.v_sn = v_sn, p.o_sb = o_sb, p.o_sn = o_sn;
  p.scale_log2e = scale * 1.44269504088896340736f;
  const int64_t nb = (int64_t)p.BH * p.nqb;
  if (nb > 0x7fffffff) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == ED_BF16) k_flash_attn_pipe_ir<BF, true><<<dim3((unsigned)nb), dim3(256), 0, st>>>(p);
  else if (dtype == ED_F16) k_flash_attn_pipe_ir<HF, true><<<dim3((unsigned)nb), dim3(256), 0, st>>>(p);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}
