// attn16.hip -- EXPERIMENT for round 5 (not in libelastic_hip.so, nothing on the product path calls it).
//
// The product's pipelined flash attention (csrc/attention_kernels.hip: k_flash_attn_pipe, lazy maximum, included below for its
// helpers) rebuilt on v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16.  Why: profiles/r4_s20_bare_mfma_rate_by_shape_and_operands.jsonl
// -- a register-only loop of 32x32x16 MFMAs that cycles through 1, 2 or 4 accumulators runs at 0.68 of the 16x16x32 rate (1 636 vs
// 2 413 TFLOP/s on zeros, 1 250-1 350 vs 1 850-1 950 on random fp16 operands, two waves per SIMD); only with 8 accumulators in
// rotation does it reach the peak.  The product kernel has 2 (S) + 2 (O) accumulators per wave: its MFMA-only ablation (551 us =
// 1 558 TFLOP/s, profiles/r4_s6_attention_ablation.jsonl) IS that 0.68.  With 16x16x32 a wave's 32 query rows are two 16-column
// blocks, S^T and O^T are 8 accumulators of 4 registers each, every MFMA chain has distance 8, and the pipe runs at its full rate.
//
// Same design otherwise: S^T = K Q^T, O^T = V^T P^T (the lane is the query in both, P never leaves the registers), K one tile ahead
// of V, one softmax slice per MFMA gap, lazy maximum (numerators against the standing reference; a wave whose partial sums exceed
// 2^6 redoes the tile exactly), deferred O rescale.  Layouts (tools/emulate_flash_attention.py: run_block_p16, checked lane by lane
// with poisoned LDS before the first launch):
//   lane = (m = lane & 15, g = lane >> 4); queries of the lane: 32 wave + 16 qb + m (qb = 0, 1)
//   S accumulator s[kb][qb][r]  <-> key 16 kb + 4 g + r (kb = 0..3), query block qb
//   P operand p[st][qb] slot j  <-> key 16 (2 st + (j >> 2)) + 4 g + (j & 3): the registers of s[2 st][qb], s[2 st + 1][qb]
//   O accumulator o[db][qb][r]  <-> d = 16 db + 4 g + r
//   K tile in LDS: 8 subtiles [16 keys][32 d] of 1 KiB, csrc/gemm_kernels.hip's st_16x32 image (rows 8..15 with their 32-byte halves
//   swapped) -- the ds_read_b128 fragment read of the GEMM kernel; V tile row-major with 160-byte rows: the 32 8-byte addresses one
//   ds_read_b64_tr_b16 cycle serves (8 key rows x 4 chunks) fall on 64 distinct banks.
// Restrictions of the experiment: head_dim 64, Nk a multiple of 64 and >= 128 (the self-attention shapes), lazy variant only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include tools/attn16/attn16.hip -o tools/attn16/libattn16.so
#include "../../elasticdiffusion_official_amd/csrc/attention_kernels.hip"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BF16x {
  typedef bf16x8 v8;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ v8 pack(f32x8 p) { return __builtin_convertvector(p, v8); }
};
struct HF16x {
  typedef f16x8 v8;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ v8 pack(f32x8 p) { return __builtin_convertvector(p, v8); }
};

constexpr int V_LD_16 = 80;                 // V tile row pitch (elements): 160 B
constexpr int K_SUB = 512;                  // elements of one [16][32] K subtile (1 KiB)

struct Smem16 {
  uint16_t k[3][KT * D];
  uint16_t v[3][KT * V_LD_16];
};

__device__ __forceinline__ int swz_g(int p) { return p ^ (((p >> 9) & 1) << 5); }

struct Lane16 {
  int krd;    // byte offset of the lane's 16 bytes inside a K subtile: row m, d bytes 16 g (swizzled)
  int vrd;    // element offset of the lane's transpose-read address inside a V tile, key block 0, d block 0
};

__device__ __forceinline__ Vec16 k_frag16(const uint16_t* kt, const Lane16& L, int kb, int ks) {
  return *reinterpret_cast<const Vec16*>(reinterpret_cast<const uint8_t*>(kt) + (kb * 2 + ks) * (K_SUB * 2) + L.krd);
}
// V^T fragment of key step st (32 keys: key blocks 2 st, 2 st + 1), d block db (16 wide)
__device__ __forceinline__ Vec16 v_frag16(const uint16_t* vt, const Lane16& L, int st, int db) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const uint16_t* a = vt + L.vrd + (32 * st) * V_LD_16 + 16 * db;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 16 * V_LD_16));
  return __builtin_bit_cast(Vec16, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <typename T>
__device__ __forceinline__ void qk_tile16(const uint16_t* kt, const Lane16& L, const typename T::v8 (&qf)[2][2], f32x4 (&s)[4][2]) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) s[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const Vec16 kf = k_frag16(kt, L, kb, ks);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) s[kb][qb] = T::mfma(as_v8<typename T::v8>(kf), qf[qb][ks], s[kb][qb]);
    }
}

template <typename T>
__device__ __forceinline__ void pv_tile16(const uint16_t* vt, const Lane16& L, const typename T::v8 (&pf)[2][2], f32x4 (&o)[4][2]) {
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const Vec16 vf = v_frag16(vt, L, st, db);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) o[db][qb] = T::mfma(as_v8<typename T::v8>(vf), pf[st][qb], o[db][qb]);
    }
}

// online-softmax state of the lane's two query rows
struct Run16 {
  float mb[2];      // reference maximum in the exp2 domain
  float l[2];       // running sum of the numerators (this lane's quarter of the keys)
  float psum[2];    // this tile's partial sums
  float alpha[2];   // factor for everything accumulated before this tile
};

__device__ __forceinline__ float row_max4(float x) {   // maximum over the four lanes (g = 0..3) that share a query row
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  return fmaxf(x, __shfl_xor(x, 32, 64));
}

// slice i = 0..31 of a tile's softmax, issued behind MFMA i: one numerator; groups of 8 = one P operand (st, qb)
template <typename T>
__device__ __forceinline__ void softmax_slice16(int i, f32x4 (&s)[4][2], float sl, Run16& r, typename T::v8 (&pf)[2][2]) {
  if (i == 0) r.psum[0] = r.psum[1] = 0.f;
  const int G = i >> 3, e = i & 7, st = G >> 1, qb = G & 1, kb = 2 * st + (e >> 2), rr = e & 3;
  const float x = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][qb][rr], sl, -r.mb[qb]));
  s[kb][qb][rr] = x;
  r.psum[qb] += x;
  asm volatile("" : "+v"(r.psum[qb]));
  if (e == 7) {
    f32x8 pv;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = s[2 * st + (j >> 2)][qb][j & 3];
    pf[st][qb] = T::pack(pv);
    asm volatile("" : "+v"(pf[st][qb]));
  }
}

// exact softmax of the current tile (the lazy loop's slow path, and nothing else): S recomputed from the K tile still in LDS
template <typename T>
__device__ __forceinline__ void resoftmax_tile16(const uint16_t* kt, const Lane16& L, const typename T::v8 (&qf)[2][2], f32x4 (&s)[4][2],
                                                 float sl, Run16& r, typename T::v8 (&pf)[2][2]) {
  qk_tile16<T>(kt, L, qf, s);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = s[0][qb][0];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[kb][qb][j]);
    mx = row_max4(mx);
    const float use = fmaxf(r.mb[qb], mx * sl);
    r.alpha[qb] = __builtin_amdgcn_exp2f(r.mb[qb] - use);
    r.mb[qb] = use;
    float psum = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      f32x8 pv;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pv[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * st + (j >> 2)][qb][j & 3], sl, -use));
        psum += pv[j];
      }
      pf[st][qb] = T::pack(pv);
    }
    r.psum[qb] = psum;
  }
}

// per-segment cycle accounting (TIMING builds only): s_memtime stamps, wave-uniform, summed over the iterations of the wave
struct Tm {
  uint64_t last;
  uint32_t a[8];
};
template <bool TIMING>
__device__ __forceinline__ void stamp(Tm& tm, int k) {
  if (TIMING) {
    const uint64_t now = __builtin_amdgcn_s_memtime();
    tm.a[k] += (uint32_t)(now - tm.last);
    tm.last = now;
  }
}

// one iteration's compute: 32 MFMAs -- S_next = K(t+1) Q^T (16), then O += V(t-1)^T P(t-1)^T (16) -- each followed by one slice of
// the softmax of S_cur; a K / V fragment feeds two consecutive MFMAs (the two query blocks) and is fetched from LDS one fragment ahead
// `filler(i)` is called behind MFMA slot i (INREGION builds: the next tiles' global loads at slots 0..3, their LDS writes at 26..29 -- VMEM
// and DS instructions co-issue with the MFMA / VALU stream instead of standing alone between the region and the barrier)
template <typename T, bool HAS_PV, bool HAS_NEXT, bool TIMING, class Filler>
__device__ __forceinline__ void pipe_region16(Filler&& filler, const uint16_t* k_next, const uint16_t* v_prev, const Lane16& L,
                                              const typename T::v8 (&qf)[2][2], f32x4 (&s_cur)[4][2], f32x4 (&s_next)[4][2],
                                              const typename T::v8 (&p_prev)[2][2], typename T::v8 (&p_cur)[2][2], f32x4 (&o)[4][2],
                                              float sl, Run16& run, Tm& tm) {
  constexpr int FQK = HAS_NEXT ? 8 : 0, NF = FQK + (HAS_PV ? 8 : 0);    // fragments
  auto fetch = [&](int fi) -> Vec16 {
    if (fi < FQK) return k_frag16(k_next, L, fi & 3, fi >> 2);
    const int j = fi - FQK;
    return v_frag16(v_prev, L, j >> 2, j & 3);
  };
  if (HAS_NEXT) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) s_next[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  Vec16 ring[2] = {Vec16{{0u, 0u, 0u, 0u}}, Vec16{{0u, 0u, 0u, 0u}}};
  if (NF > 0) ring[0] = fetch(0);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int fi = i >> 1, qb = i & 1;
    if (fi < NF) {
      if (qb == 0 && fi + 1 < NF) ring[(fi + 1) & 1] = fetch(fi + 1);
      const Vec16 a_cur = ring[fi & 1];
      if (fi < FQK) {
        const int kb = fi & 3, ks = fi >> 2;
        s_next[kb][qb] = T::mfma(as_v8<typename T::v8>(a_cur), qf[qb][ks], s_next[kb][qb]);
        asm volatile("" : "+v"(s_next[kb][qb]));
      } else {
        const int j = fi - FQK, st = j >> 2, db = j & 3;
        o[db][qb] = T::mfma(as_v8<typename T::v8>(a_cur), p_prev[st][qb], o[db][qb]);
        asm volatile("" : "+v"(o[db][qb]));
      }
    }
    softmax_slice16<T>(i, s_cur, sl, run, p_cur);
    filler(i);
    __builtin_amdgcn_sched_barrier(0);
    if (i == 15) stamp<TIMING>(tm, 1);     // S_next half done
  }
  stamp<TIMING>(tm, 2);                     // P V half done
}

// NOCHECK (ablation only, wrong on data whose scores outgrow the first tile's maximum by 2^6): the lazy loop without its per-tile
// check -- what deferring the check into the next iteration's MFMA shadow could buy at most
template <typename T, bool TIMING = false, bool NOCHECK = false, bool INREGION = false>
__global__ void __launch_bounds__(256, 2)
k_flash_attn_p16(const Params p, uint32_t* __restrict__ tm_out) {
  __shared__ Smem16 sm;
  Tm tm;
  tm.last = TIMING ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) tm.a[i] = 0;
  const uint64_t t_begin = tm.last;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, g = lane >> 4;
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

  Lane16 L;
  L.krd = swz_g(m16 * 64 + g * 16);
  L.vrd = (4 * g + (m16 >> 2)) * V_LD_16 + 4 * (lane & 3);

  const int q_row0 = qblk * QB + wave * 32 + m16;       // query block 1: + 16
  typename T::v8 qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qb][ks] = as_v8<typename T::v8>(load_row16(qg, p.q_sn, q_row0 + 16 * qb, p.Nq, 32 * ks + 8 * g));

  const BufRsrc k_rs = make_rsrc(kg, ((int64_t)(p.Nk - 1) * p.k_sn + D) * 2);
  const BufRsrc v_rs = make_rsrc(vg, ((int64_t)(p.Nk - 1) * p.v_sn + D) * 2);
  const int st_row = tid >> 3, st_chunk = tid & 7;
  const uint32_t k_off = (uint32_t)(((int64_t)st_row * p.k_sn + 8 * st_chunk) * 2), k_half = (uint32_t)(32 * p.k_sn * 2);
  const uint32_t v_off = (uint32_t)(((int64_t)st_row * p.v_sn + 8 * st_chunk) * 2), v_half = (uint32_t)(32 * p.v_sn * 2);
  // LDS destinations of the thread's two rows (st_row, st_row + 32): K subtile (row >> 4, chunk >> 2), swizzled position; V row-major
  int k_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = st_row + 32 * i;
    k_dst[i] = ((row >> 4) * 2 + (st_chunk >> 2)) * (K_SUB * 2) + swz_g((row & 15) * 64 + (st_chunk & 3) * 16);   // bytes
  }
  Vec16 kreg[2], vreg[2];
  auto load_k = [&](int t) {
    const uint32_t base = k_off + (uint32_t)t * 2u * k_half;
#pragma unroll
    for (int i = 0; i < 2; ++i) kreg[i] = buf_load16(k_rs, base + i * k_half, 0);
  };
  auto load_v = [&](int t) {
    const uint32_t base = v_off + (uint32_t)t * 2u * v_half;
#pragma unroll
    for (int i = 0; i < 2; ++i) vreg[i] = buf_load16(v_rs, base + i * v_half, 0);
  };
  auto write_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<Vec16*>(reinterpret_cast<uint8_t*>(sm.k[buf]) + k_dst[i]) = kreg[i];
  };
  auto write_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<Vec16*>(&sm.v[buf][(st_row + 32 * i) * V_LD_16 + 8 * st_chunk]) = vreg[i];
  };

  f32x4 oacc[4][2];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) oacc[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  Run16 run;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) run.mb[qb] = -INFINITY, run.l[qb] = 0.f, run.psum[qb] = 0.f, run.alpha[qb] = 1.f;
  const float sl = p.scale_log2e;
  const int n_full = p.Nk / KT;        // (the launcher admits multiples of 64 only)

  load_k(0);
  load_v(0);
  write_k(0);
  write_v(0);
  load_k(1);
  write_k(1);
  __syncthreads();

  f32x4 sA[4][2], sB[4][2];
  typename T::v8 pA[2][2], pB[2][2];
  int vb_prev = 2, vb_cur = 0, vb_next = 1;
  int kb_cur = 0, kb_next = 1, kb_write = 2;

  auto iter = [&](auto has_pv, auto has_next, int t, f32x4 (&s_cur)[4][2], f32x4 (&s_next)[4][2], typename T::v8 (&p_prev)[2][2],
                  typename T::v8 (&p_cur)[2][2]) {
    stamp<TIMING>(tm, 7);                  // (whatever preceded the iteration: prologue / loop overhead)
    if (!INREGION) {
      load_k(t + 2);
      load_v(t + 1);
    }
    stamp<TIMING>(tm, 0);                  // global loads issued
    const uint32_t kbase = k_off + (uint32_t)(t + 2) * 2u * k_half, vbase = v_off + (uint32_t)(t + 1) * 2u * v_half;
    uint8_t* const kdst = reinterpret_cast<uint8_t*>(sm.k[kb_write]);
    uint16_t* const vdst = sm.v[vb_next];
    auto filler = [&](int i) {
      if (!INREGION) return;
      if (i == 0) kreg[0] = buf_load16(k_rs, kbase, 0);
      else if (i == 1) kreg[1] = buf_load16(k_rs, kbase + k_half, 0);
      else if (i == 2) vreg[0] = buf_load16(v_rs, vbase, 0);
      else if (i == 3) vreg[1] = buf_load16(v_rs, vbase + v_half, 0);
      else if (i == 26) *reinterpret_cast<Vec16*>(kdst + k_dst[0]) = kreg[0];
      else if (i == 27) *reinterpret_cast<Vec16*>(kdst + k_dst[1]) = kreg[1];
      else if (i == 28) *reinterpret_cast<Vec16*>(&vdst[st_row * V_LD_16 + 8 * st_chunk]) = vreg[0];
      else if (i == 29) *reinterpret_cast<Vec16*>(&vdst[(st_row + 32) * V_LD_16 + 8 * st_chunk]) = vreg[1];
    };
    pipe_region16<T, decltype(has_pv)::value, decltype(has_next)::value, TIMING>(filler, sm.k[kb_next], sm.v[vb_prev], L, qf, s_cur,
                                                                                s_next, p_prev, p_cur, oacc, sl, run, tm);
    // one vote per tile: the O rescale can only be needed after the exact redo (alpha is 1 otherwise), and that is a scalar flag
    run.alpha[0] = run.alpha[1] = 1.0f;
    bool redone = false;
    if (!NOCHECK && __any(!(run.psum[0] <= RESCALE_SUM_MAX && run.psum[1] <= RESCALE_SUM_MAX))) {
      resoftmax_tile16<T>(sm.k[kb_cur], L, qf, s_cur, sl, run, p_cur);
      redone = true;
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) run.l[qb] = __builtin_fmaf(run.l[qb], run.alpha[qb], run.psum[qb]);
    if (redone) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int j = 0; j < 4; ++j) oacc[db][qb][j] *= run.alpha[qb];
    }
    // (Measured: with two votes per tile -- this one and a second `__any(alpha != 1)` in front of the rescale, as the product kernel
    // has them -- the segment costs 180 cycles per tile; without any check the kernel is 5-11 % faster, profiles/r4_s21_*.  Taking the
    // vote before the LDS writes and the redo after them made hipcc give the redo path its own registers and put 33 v_mov on the FAST
    // path of the merge: slower.)
    stamp<TIMING>(tm, 3);                  // lazy check (+ slow path, rescale)
    if (!INREGION) {
      write_k(kb_write);
      write_v(vb_next);
    }
    stamp<TIMING>(tm, 4);                  // waited for the global loads, LDS writes issued
    const int tmp = vb_prev;
    vb_prev = vb_cur, vb_cur = vb_next, vb_next = tmp;
    const int ktmp = kb_cur;
    kb_cur = kb_next, kb_next = kb_write, kb_write = ktmp;
    __syncthreads();
    stamp<TIMING>(tm, 5);                  // barrier
  };

  // tile 0: S and its exact row maxima = the first reference; from there on every tile (tile 0 included) takes the lazy softmax
  qk_tile16<T>(sm.k[0], L, qf, sA);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = sA[0][qb][0];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(mx, sA[kb][qb][j]);
    run.mb[qb] = row_max4(mx) * sl;
  }
  __syncthreads();
  if (n_full == 1) {
    iter(False{}, False{}, 0, sA, sB, pB, pA);
  } else {
    iter(False{}, True{}, 0, sA, sB, pB, pA);
    int t = 1;
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
  if ((n_full - 1) & 1) pv_tile16<T>(sm.v[vb_prev], L, pB, oacc);
  else pv_tile16<T>(sm.v[vb_prev], L, pA, oacc);

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l_tot = run.l[qb] + __shfl_xor(run.l[qb], 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q_row = q_row0 + 16 * qb;
    if (q_row < p.Nq) {
      uint16_t* orow = og + (int64_t)q_row * p.o_sn;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        f32x8 tmp;
#pragma unroll
        for (int e = 0; e < 4; ++e) tmp[e] = oacc[db][qb][e] * inv, tmp[4 + e] = 0.f;
        const Vec16 packed = __builtin_bit_cast(Vec16, T::pack(tmp));
        Vec8 out8 = {{packed.w[0], packed.w[1]}};
        *reinterpret_cast<Vec8*>(orow + 16 * db + 4 * g) = out8;
      }
    }
  }
  if (TIMING) {
    stamp<TIMING>(tm, 6);                  // drain + epilogue
    if (lane == 0 && blockIdx.x < 64) {
      uint32_t* o = tm_out + (blockIdx.x * 4 + wave) * 10;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = tm.a[i];
      o[8] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_begin);
      o[9] = (uint32_t)n_full;
    }
  }
}

}  // namespace

extern "C" int ed_x_flash_attention16(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H, int Nq, int Nk,
                                      int64_t q_sb, int64_t q_sn, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn, int64_t o_sb,
                                      int64_t o_sn, float scale, void* timing_out, void* stream) {
  if (B == 0 || H == 0 || Nq == 0) return 0;
  if (Nk < 2 * KT || Nk % KT != 0) return (int)hipErrorInvalidValue;
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
  // timing_out (optional): 64 workgroups x 4 waves x 10 uint32 -- cycles per segment (see stamp<>), total, tiles
  if (timing_out == (void*)2) {    // variant: global loads and LDS writes inside the MFMA region
    if (dtype != ED_F16) return (int)hipErrorInvalidValue;
    k_flash_attn_p16<HF16x, false, false, true><<<dim3((unsigned)nb), dim3(256), 0, st>>>(p, nullptr);
  } else if (timing_out == (void*)3) {    // ... its s_memtime build (timing buffer = out + nothing: the caller passes it through q? no: see run.py)
    return (int)hipErrorInvalidValue;
  } else if (timing_out == (void*)1) {    // ablation: no per-tile check
    if (dtype != ED_F16) return (int)hipErrorInvalidValue;
    k_flash_attn_p16<HF16x, false, true><<<dim3((unsigned)nb), dim3(256), 0, st>>>(p, nullptr);
  } else if (timing_out) {
    if (dtype != ED_F16) return (int)hipErrorInvalidValue;
    k_flash_attn_p16<HF16x, true><<<dim3((unsigned)nb), dim3(256), 0, st>>>(p, (uint32_t*)timing_out);
  } else if (dtype == ED_BF16) k_flash_attn_p16<BF16x><<<dim3((unsigned)nb), dim3(256), 0, st>>>(p, nullptr);
  else if (dtype == ED_F16) k_flash_attn_p16<HF16x><<<dim3((unsigned)nb), dim3(256), 0, st>>>(p, nullptr);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}
