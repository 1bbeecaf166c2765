// attention_kernels.hip -- gfx950 (CDNA4) flash-attention forward for the UNet's transformer blocks (model side of the
// hot path's boundary, elastic_diffusion.py:422-426 `self.unet(...)`; diffusers' AttnProcessor2_0 calls
// F.scaled_dot_product_attention, which on ROCm dispatches to AOTriton's `attn_fwd` -- 15.7 % of GPU time in
// profiles/r1_bench_sdxl_1024x2048_kernel_stats_final.csv at ~450 TFLOP/s).  This is the one place on the path where
// the work is a dense contraction that no library call already covers well, so it is hand-written MFMA:
//
//   out[b, q, h, :] = softmax_k( scale * Q[b,q,h,:] . K[b,k,h,:] ) @ V[b,k,h,:]        head_dim D = 64, 16-bit I/O
//
// Design (64-wide wavefronts, v_mfma_f32_32x32x16_{bf16,f16}):
//   * workgroup = 4 waves = 128 query rows of one (batch, head); every wave owns 32 query rows for the whole kernel,
//     so the online-softmax state (running max m, running sum l) is ONE register per lane;
//   * both contractions are computed transposed -- S^T = K Q^T and O^T = V^T P^T -- so that the MFMA's N index (the
//     lane) is the query row in both: the S^T accumulator a lane holds (32 key values of its own query row per 64-key
//     tile) IS the B-operand register layout of the second MFMA after a 16-bit pack; no cross-lane shuffle, no LDS
//     round trip for P, and the per-row rescale of O is a per-lane multiply;
//   * the contraction index of an MFMA may be permuted freely as long as both operands use the same permutation:
//     step s of the P V product contracts over keys kappa(s,hi,j) = 16 s + 8 (j>>2) + 4 hi + (j&3), which is exactly
//     the order the S^T accumulator registers come in;
//   * K and V tiles (64 keys) are staged through LDS once per workgroup, double buffered, global loads for tile t+1
//     issued before the MFMAs of tile t and written to LDS after them (one barrier per tile); K rows are padded to
//     144 B so the 16-byte A-fragment reads of 32 consecutive keys spread over all banks;
//   * V needs a transpose (MFMA operands are contraction-index-contiguous per lane, V is head-dim-contiguous in
//     memory).  Two interchangeable paths (runtime flag, both tested): gfx950's LDS transpose read
//     `ds_read_b64_tr_b16` from a row-major V tile (192 B rows: conflict-free for the 4x16 blocks it gathers), or a
//     V^T tile written as packed key pairs (136 B rows) and read with plain 8-byte reads;
//   * exp2 with the softmax scale and log2(e) folded into one FMA per score; P is rounded to the I/O type before the
//     second MFMA (as every flash-attention implementation does), accumulation in fp32;
//   * blockIdx -> (batch*head, query block) is remapped so that the query blocks of one (batch, head) land on the
//     same XCD (workgroup b is placed on XCD b % 8): its K/V stay in that XCD's 4 MiB L2.
//
// Keys beyond Nk (cross-attention: 77 text tokens) are masked to -inf; query rows beyond Nq are computed on zeros and
// not stored.  Strides are per tensor (elements): batch and token strides are free, head stride = 64, unit d stride,
// so q/k/v may be column slices of one fused QKV projection output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elastic_hip.h"

namespace {

constexpr int D = 64;          // head dim
constexpr int QB = 128;        // query rows per workgroup (4 waves x 32)
constexpr int KT = 64;         // keys per tile
constexpr int K_LD = 72;       // K tile row pitch (elements): 144 B
constexpr int V_LD_TR = 96;    // row-major V tile pitch for ds_read_b64_tr_b16: 192 B
constexpr int V_LD_T = 68;     // V^T tile pitch (keys per d row): 136 B

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct alignas(16) Vec16 { uint32_t w[4]; };
struct alignas(8) Vec8 { uint32_t w[2]; };

struct BF {
  typedef bf16x8 v8;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v8 pack(f32x8 p) { return __builtin_convertvector(p, v8); }  // v_cvt_pk_bf16_f32, RNE
};
struct HF {
  typedef f16x8 v8;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v8 pack(f32x8 p) { return __builtin_convertvector(p, v8); }
};

template <typename V8>
__device__ __forceinline__ V8 as_v8(Vec16 x) {
  return __builtin_bit_cast(V8, x);
}

struct Params {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  int Nq, Nk, H, BH, nqb;
  int64_t q_sb, q_sn, k_sb, k_sn, v_sb, v_sn, o_sb, o_sn;  // element strides: batch, token (head stride 64, d stride 1)
  float scale_log2e;                                        // softmax scale * log2(e)
};

template <bool TR>
struct Smem {
  uint16_t k[2][KT * K_LD];
  uint16_t v[2][TR ? KT * V_LD_TR : D * V_LD_T];
};

// 16-byte global load of 8 consecutive head-dim elements of one token (zeros past the end of the sequence)
__device__ __forceinline__ Vec16 load_row16(const uint16_t* base, int64_t row_stride, int row, int n_rows, int col) {
  Vec16 z = {{0u, 0u, 0u, 0u}};
  if (row < n_rows) z = *reinterpret_cast<const Vec16*>(base + (int64_t)row * row_stride + col);
  return z;
}

// __launch_bounds__(256, 2): at least 2 waves per SIMD => a 256-register budget per lane, which also makes the compiler
// keep the MFMA accumulators in ordinary VGPRs (gfx950's unified file) instead of shuttling S and O through AGPRs with
// ~80 v_accvgpr moves per tile.
// QN = 32-row query blocks per wave (1: 128 query rows per workgroup, 3 waves/SIMD; 2: 256 rows per workgroup -- every K
// and V fragment read from LDS feeds two MFMAs, and the per-tile barrier, staging and LDS traffic are amortised over
// twice the MFMA work, at 2 waves/SIMD).
template <typename T, bool TR, int QN>
__global__ void __launch_bounds__(256, 2)
k_flash_attn_fwd(const Params p) {
  __shared__ Smem<TR> sm;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hi = lane >> 5;

  // ---- XCD-aware work mapping: 8 consecutive (batch, head) pairs are interleaved so each lands on one XCD ----------
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

  // ---- Q^T fragments (B operand of S^T = K Q^T): lane (q = ln, hi) holds d = 16 ks + 8 hi + [0,8) ------------------
  int q_row[QN];
  typename T::v8 qf[QN][4];
#pragma unroll
  for (int qn = 0; qn < QN; ++qn) {
    q_row[qn] = qblk * (QB * QN) + (wave * QN + qn) * 32 + ln;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[qn][ks] = as_v8<typename T::v8>(load_row16(qg, p.q_sn, q_row[qn], p.Nq, 16 * ks + 8 * hi));
  }

  // ---- staging registers for one K/V tile (2 x 16 B each per thread) ---------------------------------------------
  Vec16 kreg[2], vreg[2];
  auto issue_loads = [&](int t) {
    const int key0 = t * KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      kreg[i] = load_row16(kg, p.k_sn, key0 + (idx >> 3), p.Nk, (idx & 7) * 8);
    }
    if (TR) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        vreg[i] = load_row16(vg, p.v_sn, key0 + (idx >> 3), p.Nk, (idx & 7) * 8);
      }
    } else {  // key pair kp = tid >> 3 (keys 2kp, 2kp+1), d chunk c = tid & 7
      const int kp = tid >> 3, c = tid & 7;
      vreg[0] = load_row16(vg, p.v_sn, key0 + 2 * kp, p.Nk, c * 8);
      vreg[1] = load_row16(vg, p.v_sn, key0 + 2 * kp + 1, p.Nk, c * 8);
    }
  };
  auto write_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<Vec16*>(&sm.k[buf][(idx >> 3) * K_LD + (idx & 7) * 8]) = kreg[i];
    }
    if (TR) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        *reinterpret_cast<Vec16*>(&sm.v[buf][(idx >> 3) * V_LD_TR + (idx & 7) * 8]) = vreg[i];
      }
    } else {
      const int kp = tid >> 3, c = tid & 7;
      const uint16_t* a = reinterpret_cast<const uint16_t*>(&vreg[0]);
      const uint16_t* bb = reinterpret_cast<const uint16_t*>(&vreg[1]);
#pragma unroll
      for (int e = 0; e < 8; ++e)  // V^T[d = 8c+e][keys 2kp, 2kp+1] as one 32-bit word
        *reinterpret_cast<uint32_t*>(&sm.v[buf][(8 * c + e) * V_LD_T + 2 * kp]) = (uint32_t)a[e] | ((uint32_t)bb[e] << 16);
    }
  };

  f32x16 oacc[QN][2];
  float m_run[QN], l_run[QN];
#pragma unroll
  for (int qn = 0; qn < QN; ++qn) {
    m_run[qn] = -INFINITY, l_run[qn] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[qn][0][i] = oacc[qn][1][i] = 0.f;
  }
  const float sl = p.scale_log2e;
  const int n_tiles = (p.Nk + KT - 1) / KT;

  issue_loads(0);
  write_lds(0);
  __syncthreads();

  for (int t = 0; t < n_tiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < n_tiles) issue_loads(t + 1);

    // ---- S^T = K Q^T : two 32-key blocks x four 16-wide d steps (each K fragment feeds all QN query blocks) ---------
    f32x16 s[QN][2];
#pragma unroll
    for (int qn = 0; qn < QN; ++qn)
#pragma unroll
      for (int i = 0; i < 16; ++i) s[qn][0][i] = s[qn][1][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {  // the accumulator chains alternate: no back-to-back dependent MFMAs
        const Vec16 kf = *reinterpret_cast<const Vec16*>(&sm.k[buf][(32 * kb + ln) * K_LD + 16 * ks + 8 * hi]);
#pragma unroll
        for (int qn = 0; qn < QN; ++qn) s[qn][kb] = T::mfma(as_v8<typename T::v8>(kf), qf[qn][ks], s[qn][kb]);
      }
    // s[qn][kb][r] = score of key 32 kb + 8 (r>>2) + 4 hi + (r&3) against this lane's query row of block qn
    if ((t + 1) * KT > p.Nk) {  // ragged last tile: keys past Nk contribute nothing
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KT + 32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3) >= p.Nk) {
#pragma unroll
            for (int qn = 0; qn < QN; ++qn) s[qn][kb][r] = -INFINITY;
          }
    }

    // ---- online softmax (the two lanes of a query row, hi = 0/1, hold disjoint halves of its keys) ----------------
#pragma unroll
    for (int qn = 0; qn < QN; ++qn) {
      float mx = s[qn][0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qn][0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qn][1][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qn], mx);
      const float mb = m_new * sl;
      const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run[qn], sl, -mb));  // exp2(-inf) = 0 on the first tile
      m_run[qn] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qn][kb][r], sl, -mb));
          s[qn][kb][r] = e;
          psum += e;
        }
      l_run[qn] = __builtin_fmaf(l_run[qn], alpha, psum);  // per-lane partial; the two halves are added once, at the end
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        oacc[qn][0][i] *= alpha;
        oacc[qn][1][i] *= alpha;
      }
    }

    // ---- O^T += V^T P^T : four 16-key steps x two 32-wide d blocks (each V fragment feeds all QN query blocks) ------
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      typename T::v8 pf[QN];
#pragma unroll
      for (int qn = 0; qn < QN; ++qn) {
        f32x8 pv;
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = s[qn][st >> 1][8 * (st & 1) + j];
        pf[qn] = T::pack(pv);  // B operand: slot j <-> key 16 st + 8 (j>>2) + 4 hi + (j&3)
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        Vec16 vf;
        if (TR) {
          // 16-lane group g = (lane>>4) gathers the [4 keys][16 d] block: its lane i supplies row i>>2, columns
          // 4 (i&3)..+3, and receives column i (4 consecutive keys of d = 16 (g&1) + i)
          const int row = 16 * st + 4 * hi + ((lane & 15) >> 2);
          const int col = 32 * db + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
          typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&sm.v[buf][row * V_LD_TR + col]));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&sm.v[buf][(row + 8) * V_LD_TR + col]));
          const s16x8 both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          vf = __builtin_bit_cast(Vec16, both);
        } else {
          const uint16_t* vrow = &sm.v[buf][(32 * db + ln) * V_LD_T + 16 * st + 4 * hi];
          const Vec8 lo = *reinterpret_cast<const Vec8*>(vrow);
          const Vec8 hi8 = *reinterpret_cast<const Vec8*>(vrow + 8);
          vf.w[0] = lo.w[0], vf.w[1] = lo.w[1], vf.w[2] = hi8.w[0], vf.w[3] = hi8.w[1];
        }
#pragma unroll
        for (int qn = 0; qn < QN; ++qn) oacc[qn][db] = T::mfma(as_v8<typename T::v8>(vf), pf[qn], oacc[qn][db]);
      }
    }

    if (t + 1 < n_tiles) write_lds(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: O[q, d] = O^T[d, q] / l ; lane holds d = 32 db + 8 (r>>2) + 4 hi + (r&3) of its query row ----------
#pragma unroll
  for (int qn = 0; qn < QN; ++qn) {
    const float l_tot = l_run[qn] + __shfl_xor(l_run[qn], 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row[qn] < p.Nq) {
      uint16_t* orow = og + (int64_t)q_row[qn] * p.o_sn;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x8 tmp;
#pragma unroll
          for (int e = 0; e < 4; ++e) tmp[e] = oacc[qn][db][4 * g + e] * inv, tmp[4 + e] = 0.f;
          const Vec16 packed = __builtin_bit_cast(Vec16, T::pack(tmp));
          Vec8 out8 = {{packed.w[0], packed.w[1]}};
          *reinterpret_cast<Vec8*>(orow + 32 * db + 8 * g + 4 * hi) = out8;
        }
    }
  }
}


// =====================================================================================================================
// Round 3.  Two more kernels behind the same entry point (v_path bits 2 and 3).
//
// (1) k_flash_attn_pipe -- the self-attention kernel, software-pipelined inside the wave.  PMC on the round-2 kernel
//     (profiles/r2_unet_pmc.json) showed the MFMA pipe busy 0.34 of the time: every wave ran QK^T (8 MFMAs), THEN the
//     whole softmax (~145 VALU / transcendental issues per lane), THEN P V (8 MFMAs), and relied on its two SIMD
//     neighbours to fill the matrix pipe meanwhile.  Here one loop iteration holds, in ONE basic block,
//         MFMA :  S(t+1) = K(t+1) Q^T   and   O += V(t-1)^T P(t-1)^T          (16 MFMAs, no dependence on the VALU work)
//         VALU :  softmax of S(t) -> P(t)                                      (independent of both)
//     so the in-order wave can issue the softmax instructions into the 32-cycle shadow of each MFMA (up to ~5 issue
//     slots per MFMA; MI355X_MICROARCH.md "instructions hidden per MFMA gap").  K therefore runs one tile ahead of V:
//     K double-buffered, V triple-buffered in LDS (55 KiB per workgroup, 2 workgroups per CU at <= 256 VGPRs).
//     The O rescale of the online softmax is deferred (T13 in the CDNA guide): a row's reference maximum moves only
//     when the new tile's maximum exceeds it by more than 2^RESCALE_LOG2 (then, and on the first tile, everything at
//     the old reference -- O and l -- is scaled exactly once; P(t-1) has already been consumed by the MFMAs above).
//     P stays <= 2^RESCALE_LOG2 in between: harmless for a floating-point P (same relative rounding), fp32 sums.
//     A ragged last tile (Nk % 64 != 0) is handled after the loop by the un-pipelined masked code.
//
// (2) k_flash_attn_smallkv -- cross-attention against the 77 text tokens (Nk <= 96).  The generic kernel pads 77 keys
//     to two 64-key tiles (40 % of its MFMAs are masked work), runs the online-softmax machinery for a single tile and
//     re-stages K / V for every 128 query rows.  This kernel is a streaming kernel (the op is HBM-bound: 105 MB of q + out
//     per 8 GFLOP at batch 20): K and V of one (batch, head) -- three 32-key blocks, zero beyond Nk -- are staged ONCE
//     per workgroup, which then walks QBLOCKS x 128 query rows: per wave 12 + 12 MFMAs per 32 rows, plain softmax
//     (no running state, no rescale), keys >= Nk masked.
// =====================================================================================================================
constexpr float RESCALE_LOG2 = 6.0f;  // defer the O rescale until a row's maximum grows by more than 2^6 (exp2 domain)
constexpr float RESCALE_SUM_MAX = 64.0f;  // = 2^RESCALE_LOG2: the lazy variant's bound on a lane's sum of numerators

// NK = K tile buffers: 2 for the exact kernel; 3 for the LAZY one, whose slow path re-reads K(t) while faster waves of the
// workgroup already stage K(t+2) (with two buffers that write would land on K(t): a race, seen as wrong results on the GPU)
template <int NK>
struct SmemPipe {
  uint16_t k[NK][KT * K_LD];
  uint16_t v[3][KT * V_LD_TR];
};

template <typename T>
__device__ __forceinline__ void qk_tile(const uint16_t* kt, int ln, int hi, const typename T::v8 (&qf)[4], f32x16 (&s)[2]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) s[0][i] = s[1][i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const Vec16 kf = *reinterpret_cast<const Vec16*>(&kt[(32 * kb + ln) * K_LD + 16 * ks + 8 * hi]);
      s[kb] = T::mfma(as_v8<typename T::v8>(kf), qf[ks], s[kb]);
    }
}

// V^T fragment of key step st (16 keys), d block db (32 wide) from a row-major V tile via the LDS transpose read
__device__ __forceinline__ Vec16 v_frag_tr(const uint16_t* vt, int lane, int hi, int st, int db) {
  const int row = 16 * st + 4 * hi + ((lane & 15) >> 2);
  const int col = 32 * db + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&vt[row * V_LD_TR + col]));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&vt[(row + 8) * V_LD_TR + col]));
  return __builtin_bit_cast(Vec16, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <typename T>
__device__ __forceinline__ void pv_tile(const uint16_t* vt, int lane, int hi, const typename T::v8 (&pf)[4], f32x16 (&o)[2]) {
#pragma unroll
  for (int st = 0; st < 4; ++st)
#pragma unroll
    for (int db = 0; db < 2; ++db)
      o[db] = T::mfma(as_v8<typename T::v8>(v_frag_tr(vt, lane, hi, st, db)), pf[st], o[db]);
}

// raw buffer access (SRD in SGPRs): 32-bit offsets, out-of-range reads return zeros
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, int64_t bytes) {
  // wave-uniform by construction (blockIdx-derived); readfirstlane makes that provable, so no waterfall loops
  const uint64_t a = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const uint32_t n = __builtin_amdgcn_readfirstlane((int)(uint32_t)(bytes > 0xffffffffll ? 0xffffffffll : (bytes < 0 ? 0 : bytes)));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), /*stride*/ 0, (int)n, 0x00020000);
}
__device__ __forceinline__ Vec16 buf_load16(BufRsrc rs, uint32_t voff, uint32_t soff) {
  const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  Vec16 o = {{v[0], v[1], v[2], v[3]}};
  return o;
}


// Output row of one query (head dim 64) from the two lanes that hold it: lane (ln, hi) has dims 32 db + 8 g + 4 hi + e (e < 4) of the row,
// i.e. 8 bytes of every 16-byte group -- round 5 stored them as 8 `dwordx2` per lane.  The two lanes swap halves (lane ln keeps groups
// g = 0, 2 whole, lane ln + 32 groups g = 1, 3) and store 4 `dwordx4`: half the store instructions, each a full 16-byte vector
// (MI355X_MICROARCH.md: the attention store tail is store-ISSUE-bound; dwordx4 halves it).  Same values, same rounding: bit-identical.
template <typename T, int NDB = 2>
__device__ __forceinline__ void store_o_row(uint16_t* orow, const f32x16 (&o)[NDB], float inv, int hi, bool valid, int dh = 64) {
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    uint32_t part[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x8 tmp;
#pragma unroll
      for (int e = 0; e < 4; ++e) tmp[e] = o[db][4 * g + e] * inv, tmp[4 + e] = 0.f;
      const Vec16 packed = __builtin_bit_cast(Vec16, T::pack(tmp));
      part[g][0] = packed.w[0], part[g][1] = packed.w[1];
    }
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int ge = 2 * gp, go = 2 * gp + 1;
      // the partner (lane ^ 32, same query row) needs this lane's half of the group it stores: hi = 0 stores ge, hi = 1 stores go
      const uint32_t r0 = __shfl_xor(hi ? part[ge][0] : part[go][0], 32, 64);
      const uint32_t r1 = __shfl_xor(hi ? part[ge][1] : part[go][1], 32, 64);
      Vec16 out;
      if (hi) out = Vec16{{r0, r1, part[go][0], part[go][1]}};
      else out = Vec16{{part[ge][0], part[ge][1], r0, r1}};
      const int d0 = 32 * db + 8 * (hi ? go : ge);     // (head dims that are multiples of 8: a group of 8 is inside or outside as a whole)
      if (valid && d0 < dh) *reinterpret_cast<Vec16*>(orow + d0) = out;
    }
  }
}

struct True { static constexpr bool value = true; };
struct False { static constexpr bool value = false; };

// Online-softmax state of one 32-row query block in a wave (per lane: one query row, half of each tile's keys)
struct SoftmaxRun {
  float mb;     // reference maximum in the exp2 domain (max * scale * log2 e); deferred, see RESCALE_LOG2
  float l;      // running sum of the numerators (this lane's half of the keys)
  float mx;     // scratch: this tile's maximum
  float use;    // scratch: reference this tile's numerators are taken against
  float alpha;  // scratch: factor for everything accumulated before this tile (1 = unchanged)
  float psum;   // scratch
};

// The softmax of one 64-key tile cut into 16 slices of ~9 VALU / transcendental issues, one per MFMA gap of the
// pipelined loop (slice i is issued right behind MFMA i).  Flat value index f = 16 kb + r (kb: 32-key block, r: register).
//   0,1   running maximum of values 16 i .. 16 i + 15 (8 v_max3 each)
//   2     cross-half maximum, deferred-rescale decision, alpha
//   3-14  numerators: 3, 3, 2 values per slice (exp2(fma) + sum), and the 16-bit pack of each finished group of 8
//   15    l = l * alpha + sum
template <typename T>
__device__ __forceinline__ void softmax_slice(int i, f32x16 (&s)[2], float sl, SoftmaxRun& r, typename T::v8 (&pf)[4]) {
  if (i < 2) {
    float mx = (i == 0) ? -INFINITY : r.mx;
#pragma unroll
    for (int j = 0; j < 16; j += 2) mx = fmaxf(fmaxf(mx, s[i][j]), s[i][j + 1]);
    r.mx = mx;
    asm volatile("" : "+v"(r.mx));
  } else if (i == 2) {
    const float mx = fmaxf(r.mx, __shfl_xor(r.mx, 32, 64));  // the two lanes of a query row hold disjoint key halves
    const float mbn = mx * sl;                                 // sl > 0: scaling commutes with the maximum
    r.use = (mbn - r.mb > RESCALE_LOG2) ? mbn : r.mb;          // first tile: mb = -inf -> mbn
    r.alpha = __builtin_amdgcn_exp2f(r.mb - r.use);            // 1 when the reference stays; 0 on the first tile
    r.mb = r.use;
    r.psum = 0.f;
    asm volatile("" : "+v"(r.use), "+v"(r.alpha));
  } else if (i < 15) {
    const int g = (i - 3) / 3, w = (i - 3) % 3;        // 4 groups of 8 values: slices of 3, 3, 2
    const int f0 = 8 * g + 3 * w, n = (w == 2) ? 2 : 3;
#pragma unroll
    for (int j = 0; j < n; ++j) {
      const int f = f0 + j;
      const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[f >> 4][f & 15], sl, -r.use));
      s[f >> 4][f & 15] = e;
      r.psum += e;
    }
    asm volatile("" : "+v"(r.psum));  // the slice's arithmetic is issued HERE (pure ops would otherwise sink below the fences)
    if (w == 2) {  // group g complete: B operand of key step g (slot j <-> key 16 g + 8 (j>>2) + 4 hi + (j&3))
      f32x8 pv;
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[g >> 1][8 * (g & 1) + j];
      pf[g] = T::pack(pv);
      asm volatile("" : "+v"(pf[g]));
    }
  } else {
    r.l = __builtin_fmaf(r.l, r.alpha, r.psum);
  }
}

// LAZY variant (v_path 5): after the first tile the per-tile row maximum is not computed at all.  The numerators are taken
// against the standing reference mb; S is left intact and the only check is on the sum: every numerator is positive, so
// psum <= 2^RESCALE_LOG2 proves that none of them exceeded 2^RESCALE_LOG2.  When the check fails for any row of the wave
// (a row's scores grew by more than 2^RESCALE_LOG2 over the reference: rare) the tile's softmax is redone exactly
// (resoftmax_tile; S is recomputed from the K tile still in LDS) -- 16 v_max3 + the shuffle + the decision arithmetic per tile leave the loop (17 % of its VALU issues,
// and the max -> exp dependency chain at the head of every tile with them).  Slices: two numerators each, pack at 3, 7, 11, 15.
// EXP2 (v_path 6): the scores arrive as exponents already -- the caller folded scale * log2(e) into q and the reference -mb
// was the initial accumulator value of the S MFMA chain -- so a numerator is ONE v_exp_f32, no FMA.
template <typename T, bool EXP2>
__device__ __forceinline__ void softmax_slice_lazy(int i, f32x16 (&s)[2], float sl, SoftmaxRun& r, typename T::v8 (&pf)[4]) {
  if (i == 0) r.psum = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int f = 2 * i + j;
    const float e = EXP2 ? __builtin_amdgcn_exp2f(s[f >> 4][f & 15])
                         : __builtin_amdgcn_exp2f(__builtin_fmaf(s[f >> 4][f & 15], sl, -r.mb));
    s[f >> 4][f & 15] = e;
    r.psum += e;
  }
  asm volatile("" : "+v"(r.psum));
  if ((i & 3) == 3) {
    const int g = i >> 2;
    f32x8 pv;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = s[g >> 1][8 * (g & 1) + j];
    pf[g] = T::pack(pv);
    asm volatile("" : "+v"(pf[g]));
  }
}

// exact softmax of tile t (the LAZY slow path): S was consumed in place, so it is recomputed from the K tile that is still in
// LDS (it is overwritten only at the end of the iteration); the reference only ever rises
template <typename T>
__device__ __forceinline__ void resoftmax_tile(const uint16_t* kt, int ln, int hi, const typename T::v8 (&qf)[4],
                                               f32x16 (&s)[2], float sl, SoftmaxRun& r, typename T::v8 (&pf)[4]) {
  qk_tile<T>(kt, ln, hi, qf, s);
  float mx = s[0][0];
#pragma unroll
  for (int j = 1; j < 16; ++j) mx = fmaxf(mx, s[0][j]);
#pragma unroll
  for (int j = 0; j < 16; ++j) mx = fmaxf(mx, s[1][j]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float use = fmaxf(r.mb, mx * sl);
  r.alpha = __builtin_amdgcn_exp2f(r.mb - use);
  r.mb = use;
  float psum = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x8 pv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = 8 * g + j;
      pv[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[f >> 4][f & 15], sl, -use));
      psum += pv[j];
    }
    pf[g] = T::pack(pv);
  }
  r.psum = psum;
}

// One loop iteration's compute (see the header comment): 16 MFMAs -- S_next = K(t+1) Q^T, then O += V(t-1)^T P(t-1)^T --
// each followed by one slice of the softmax of S_cur; the A operand of MFMA i+1 is fetched from LDS before MFMA i is
// issued.  sched_barrier(0) pins that order (left to itself the scheduler clusters all MFMAs ahead of the softmax).
// EXP2: the two S chains start from `negm` (16 registers, every one = -mb of the lane's query row) instead of zeros.
// DEPTH = how many MFMAs ahead an A operand is fetched from LDS.  Round 3 used 1: the ISA shows `ds_read` -> a few VALU ->
// `s_waitcnt lgkmcnt` -> MFMA with one MFMA slot (~32-40 cycles) between a read and its use, against ~64-100+ cycles of LDS
// latency for a 16-byte read under load -- every MFMA waits for its operand, which is what holds SQ_VALU_MFMA_BUSY at 0.41
// whatever the VALU count (v_path 4 / 5 / 6 within 1 % of each other inside the UNet).  DEPTH 3 keeps three reads in flight
// (+8 VGPRs for the ring).
// `filler(i)` is issued behind MFMA slot i (the next tiles' global loads and LDS writes: VMEM / DS instructions co-issue with the MFMA and
// VALU stream here, while in front of and behind the region they stood alone -- tools/attn16: +0.4...4.2 %, bit-identical)
template <typename T, bool HAS_PV, bool HAS_NEXT, bool LAZY, bool EXP2, int DEPTH, class Filler>
__device__ __forceinline__ void pipe_region(Filler&& filler, const uint16_t* k_next, const uint16_t* v_prev, int lane, int ln, int hi,
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

// WAVES = waves per workgroup (4: 128 query rows, two workgroups per CU; 8: 256 query rows, one workgroup per CU -- the same
// two waves per SIMD, but every K / V tile is staged ONCE for twice the query rows: half the global loads, LDS writes and L2
// traffic per MFMA; the ablation of round 4 prices the staging at 17 % of the kernel's time, profiles/r4_s6_attention_ablation.jsonl)
template <typename T, bool LAZY, bool EXP2 = false, int DEPTH = 1, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, 2)
k_flash_attn_pipe(const Params p) {
  static_assert(LAZY || !EXP2, "the exponent-domain variant is built on the lazy-maximum loop");
  static_assert(WAVES == 4 || WAVES == 8, "4 or 8 waves per workgroup");
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

  // prologue: K(0), V(0) and K(1) in flight together (K(1) past the end reads as zeros into a buffer nobody reads)
  load_k(0);
  load_v(0);
  Vec16 k1reg[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) k1reg[i] = buf_load16(k_rs, k_off + 2u * k_half + i * k_half, 0);
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
    // the next tiles' global loads (K(t+2), V(t+1): unconditional, a tile past the end reads as zeros into a buffer nobody reads) go behind
    // MFMA slots 0 .. 2 NST - 1 of the region, their LDS writes behind slots 11 .. 11 + 2 NST - 1 (both target buffers are idle during the
    // iteration: K(t-1)'s and V(t-2)'s)
    const uint32_t kbase = k_off + (uint32_t)(t + 2) * 2u * k_half, vbase = v_off + (uint32_t)(t + 1) * 2u * v_half;
    uint16_t* const kdst = sm.k[kb_write];
    uint16_t* const vdst = sm.v[vb_next];
    auto filler = [&](int i) {
      if (i < NST) kreg[i] = buf_load16(k_rs, kbase + i * k_half, 0);
      else if (i < 2 * NST) vreg[i - NST] = buf_load16(v_rs, vbase + (i - NST) * v_half, 0);
      else if (i >= 11 && i < 11 + NST) *reinterpret_cast<Vec16*>(&kdst[(st_row + 32 * (i - 11)) * K_LD + st_col]) = kreg[i - 11];
      else if (i >= 11 + NST && i < 11 + 2 * NST)
        *reinterpret_cast<Vec16*>(&vdst[(st_row + 32 * (i - 11 - NST)) * V_LD_TR + st_col]) = vreg[i - 11 - NST];
    };
    // the first tile (no PV yet) takes the exact softmax; EXP2 found its maximum before the loop and is lazy throughout
    constexpr bool lazy = LAZY && (EXP2 || decltype(has_pv)::value);
    pipe_region<T, decltype(has_pv)::value, decltype(has_next)::value, lazy, EXP2, DEPTH>(filler, sm.k[kb_next], sm.v[vb_prev], lane, ln,
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
  store_o_row<T>(og + (int64_t)q_row * p.o_sn, oacc, inv, hi, q_row < p.Nq);
}

// ---------------------------------------------------------------------------------------------------------------------
// small-KV (cross-attention) kernel: Nk <= 32 * NKB keys, one pass, K / V staged once per workgroup
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SK_QBLOCKS = 4;  // 128-row query blocks one workgroup walks (K / V staged once for 512 query rows)

template <int NKB>
struct SmemSmall {
  uint16_t k[32 * NKB * K_LD];
  uint16_t v[32 * NKB * V_LD_TR];
};

template <typename T, int NKB>
__global__ void __launch_bounds__(256, 2)
k_flash_attn_smallkv(const Params p) {
  __shared__ SmemSmall<NKB> sm;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 31, hi = lane >> 5;
  const int nqg = p.nqb;  // query groups (of SK_QBLOCKS * 128 rows) per (batch, head)
  const int bh = blockIdx.x / nqg, qgrp = blockIdx.x - bh * nqg;
  const int b = bh / p.H, h = bh - b * p.H;
  const uint16_t* qg = p.q + b * p.q_sb + h * D;
  const uint16_t* kg = p.k + b * p.k_sb + h * D;
  const uint16_t* vg = p.v + b * p.v_sb + h * D;
  uint16_t* og = p.o + b * p.o_sb + h * D;

  // stage K and V (rows >= Nk are zeros: their scores are masked, their V rows multiply exact zeros)
  for (int idx = tid; idx < 32 * NKB * 8; idx += 256) {
    const int row = idx >> 3, col = (idx & 7) * 8;
    *reinterpret_cast<Vec16*>(&sm.k[row * K_LD + col]) = load_row16(kg, p.k_sn, row, p.Nk, col);
    *reinterpret_cast<Vec16*>(&sm.v[row * V_LD_TR + col]) = load_row16(vg, p.v_sn, row, p.Nk, col);
  }
  __syncthreads();
  const float sl = p.scale_log2e;

  // (Round 6: requesting the NEXT query block's rows before the current block's arithmetic -- a register prefetch of 16 VGPRs -- measured
  // 4-9 % SLOWER, profiles/r6_s14_ops_ab_xattn_prefetch.jsonl: the two workgroups of a CU already overlap each other's loads.)
  for (int qb = 0; qb < SK_QBLOCKS; ++qb) {
    const int q_row = (qgrp * SK_QBLOCKS + qb) * QB + wave * 32 + ln;
    if ((qgrp * SK_QBLOCKS + qb) * QB + wave * 32 >= p.Nq) break;  // wave-uniform: nothing left for this wave
    typename T::v8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = as_v8<typename T::v8>(load_row16(qg, p.q_sn, q_row, p.Nq, 16 * ks + 8 * hi));
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const Vec16 kf = *reinterpret_cast<const Vec16*>(&sm.k[(32 * kb + ln) * K_LD + 16 * ks + 8 * hi]);
        s[kb] = T::mfma(as_v8<typename T::v8>(kf), qf[ks], s[kb]);
      }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3) >= p.Nk) s[kb][r] = -INFINITY;
        mx = fmaxf(mx, s[kb][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mb = mx * sl;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], sl, -mb));
        s[kb][r] = e;
        psum += e;
      }
    psum += __shfl_xor(psum, 32, 64);
    const float inv = 1.0f / psum;
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[0][i] = o[1][i] = 0.f;
#pragma unroll
    for (int st = 0; st < 2 * NKB; ++st) {
      f32x8 pv;
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[st >> 1][8 * (st & 1) + j];
      const typename T::v8 pf = T::pack(pv);
#pragma unroll
      for (int db = 0; db < 2; ++db)
        o[db] = T::mfma(as_v8<typename T::v8>(v_frag_tr(sm.v, lane, hi, st, db)), pf, o[db]);
    }
    store_o_row<T>(og + (int64_t)q_row * p.o_sn, o, inv, hi, q_row < p.Nq);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: any head dimension that is a multiple of 8 up to 160 (SD 1.x: 8 heads of 40 / 80 / 160; SD 2.x: 64).  Round 3's
// kernels are head_dim 64 only, so the SD1.5 workload (BASELINE.json configs[1]) still went through SDPA -- AOTriton's
// `attn_fwd`, 25 % of its GPU time (profiles/r4_s5_bench_cfg2_sd15_512x1024_kernel_stats.csv).  Same design as
// k_flash_attn_fwd (transposed contractions, one query row per lane, P stays in registers, V through the LDS transpose
// read), with the head dimension padded to DP = a multiple of 32: q's pad columns are zeros in registers, K's pad columns
// are zeroed once in LDS (0 x garbage could be NaN), V's pad columns produce output rows that are never stored.
// ---------------------------------------------------------------------------------------------------------------------
template <int DH>
struct GenDims {
  static constexpr int DP = (DH + 31) / 32 * 32;   // padded head dim: 40 -> 64, 80 -> 96, 160 -> 160
  static constexpr int NKS = DP / 16;              // 16-wide contraction steps of S^T = K Q^T
  static constexpr int NDB = DP / 32;              // 32-wide d blocks of O^T = V^T P^T
  static constexpr int CH = DH / 8;                // 16-byte chunks per K / V row in memory
  static constexpr int KLD = DP + 8;               // K tile row pitch (elements): 144 / 208 / 336 B -- 16 rows hit 16 distinct bank quads
  static constexpr int VLD = DP == 64 ? 96 : DP + 16;  // V tile row pitch for ds_read_b64_tr_b16: 4 consecutive rows on distinct banks
  static constexpr int NLD = (KT * CH + 255) / 256;    // staging chunks per thread and tensor
};

template <int DH>
struct SmemGen {
  uint16_t k[2][KT * GenDims<DH>::KLD];
  uint16_t v[2][KT * GenDims<DH>::VLD];
};

// (head dims 40 / 80: <= 256 registers keep S and O in VGPRs, 2 waves per SIMD; 160: 80 accumulator registers for O alone)
template <typename T, int DH>
__global__ void __launch_bounds__(256, DH <= 80 ? 2 : 1)
k_flash_attn_gen(const Params p) {
  typedef GenDims<DH> G;
  __shared__ SmemGen<DH> sm;
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
  const uint16_t* qg = p.q + b * p.q_sb + h * DH;
  const uint16_t* kg = p.k + b * p.k_sb + h * DH;
  const uint16_t* vg = p.v + b * p.v_sb + h * DH;
  uint16_t* og = p.o + b * p.o_sb + h * DH;

  // Q^T fragments: lane (q = ln, hi) holds d = 16 ks + 8 hi + [0, 8); chunks at d >= DH are zeros
  const int q_row = qblk * QB + wave * 32 + ln;
  typename T::v8 qf[G::NKS];
#pragma unroll
  for (int ks = 0; ks < G::NKS; ++ks) {
    const int d0 = 16 * ks + 8 * hi;
    qf[ks] = as_v8<typename T::v8>(d0 < DH ? load_row16(qg, p.q_sn, q_row, p.Nq, d0) : Vec16{{0u, 0u, 0u, 0u}});
  }
  // K pad columns [DH, DP) of both buffers: zero once (the staging below never touches them)
  if (G::DP > DH) {
    constexpr int PADC = (G::DP - DH) / 8;
    for (int idx = tid; idx < 2 * KT * PADC; idx += 256) {
      const int buf = idx / (KT * PADC), rem = idx - buf * (KT * PADC), row = rem / PADC, c = rem - row * PADC;
      *reinterpret_cast<Vec16*>(&sm.k[buf][row * G::KLD + DH + 8 * c]) = Vec16{{0u, 0u, 0u, 0u}};
    }
  }

  Vec16 kreg[G::NLD], vreg[G::NLD];
  auto issue_loads = [&](int t) {
    const int key0 = t * KT;
#pragma unroll
    for (int i = 0; i < G::NLD; ++i) {
      const int idx = tid + 256 * i, row = idx / G::CH, c = idx - row * G::CH;
      if (idx < KT * G::CH) {
        kreg[i] = load_row16(kg, p.k_sn, key0 + row, p.Nk, c * 8);
        vreg[i] = load_row16(vg, p.v_sn, key0 + row, p.Nk, c * 8);
      }
    }
  };
  auto write_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < G::NLD; ++i) {
      const int idx = tid + 256 * i, row = idx / G::CH, c = idx - row * G::CH;
      if (idx < KT * G::CH) {
        *reinterpret_cast<Vec16*>(&sm.k[buf][row * G::KLD + c * 8]) = kreg[i];
        *reinterpret_cast<Vec16*>(&sm.v[buf][row * G::VLD + c * 8]) = vreg[i];
      }
    }
  };

  f32x16 oacc[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sl = p.scale_log2e;
  const int n_tiles = (p.Nk + KT - 1) / KT;

  issue_loads(0);
  write_lds(0);
  __syncthreads();

  for (int t = 0; t < n_tiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < n_tiles) issue_loads(t + 1);

    f32x16 s[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[0][i] = s[1][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const Vec16 kf = *reinterpret_cast<const Vec16*>(&sm.k[buf][(32 * kb + ln) * G::KLD + 16 * ks + 8 * hi]);
        s[kb] = T::mfma(as_v8<typename T::v8>(kf), qf[ks], s[kb]);
      }
    if ((t + 1) * KT > p.Nk) {  // ragged last tile
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KT + 32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3) >= p.Nk) s[kb][r] = -INFINITY;
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float mb = m_new * sl;
    const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, sl, -mb));
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], sl, -mb));
        s[kb][r] = e;
        psum += e;
      }
    l_run = __builtin_fmaf(l_run, alpha, psum);
#pragma unroll
    for (int db = 0; db < G::NDB; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;

#pragma unroll
    for (int st = 0; st < 4; ++st) {
      f32x8 pv;
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[st >> 1][8 * (st & 1) + j];
      const typename T::v8 pf = T::pack(pv);
#pragma unroll
      for (int db = 0; db < G::NDB; ++db) {
        const int row = 16 * st + 4 * hi + ((lane & 15) >> 2);
        const int col = 32 * db + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&sm.v[buf][row * G::VLD + col]));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&sm.v[buf][(row + 8) * G::VLD + col]));
        const Vec16 vf = __builtin_bit_cast(Vec16, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        oacc[db] = T::mfma(as_v8<typename T::v8>(vf), pf, oacc[db]);
      }
    }
    if (t + 1 < n_tiles) write_lds(buf ^ 1);
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  store_o_row<T, G::NDB>(og + (int64_t)q_row * p.o_sn, oacc, inv, hi, q_row < p.Nq, DH);
}

}  // namespace

extern "C" {

int ed_flash_attention(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H, int Nq, int Nk,
                       int head_dim, int64_t q_sb, int64_t q_sn, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn,
                       int64_t o_sb, int64_t o_sn, float scale, int v_path, void* stream) {
  if (B == 0 || H == 0 || Nq == 0) return 0;
  if (Nk <= 0) return (int)hipErrorInvalidValue;
  if (head_dim != D) {  // SD 1.x head dimensions: the generic kernel (v_path is ignored)
    if (head_dim != 40 && head_dim != 80 && head_dim != 160) return (int)hipErrorInvalidValue;
    if (dtype != ED_BF16 && dtype != ED_F16) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15u) || ((uintptr_t)out & 7u)) return (int)hipErrorInvalidValue;
    if ((q_sb | q_sn | k_sb | k_sn | v_sb | v_sn) % 8 || (o_sb | o_sn) % 4) return (int)hipErrorInvalidValue;
    Params p;
    p.q = (const uint16_t*)q, p.k = (const uint16_t*)k, p.v = (const uint16_t*)v, p.o = (uint16_t*)out;
    p.Nq = Nq, p.Nk = Nk, p.H = H, p.BH = B * H, p.nqb = (Nq + QB - 1) / QB;
    p.q_sb = q_sb, p.q_sn = q_sn, p.k_sb = k_sb, p.k_sn = k_sn, p.v_sb = v_sb, p.v_sn = v_sn, p.o_sb = o_sb, p.o_sn = o_sn;
    p.scale_log2e = scale * 1.44269504088896340736f;
    const int64_t nb = (int64_t)p.BH * p.nqb;
    if (nb > 0x7fffffff) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)nb), block(256);
    hipStream_t st = (hipStream_t)stream;
#define ED_FG(DHV)                                                                      \
  if (dtype == ED_BF16) k_flash_attn_gen<BF, DHV><<<grid, block, 0, st>>>(p);           \
  else k_flash_attn_gen<HF, DHV><<<grid, block, 0, st>>>(p);
    if (head_dim == 40) { ED_FG(40) } else if (head_dim == 80) { ED_FG(80) } else { ED_FG(160) }
#undef ED_FG
    return (int)hipGetLastError();
  }
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15u) || ((uintptr_t)out & 7u)) return (int)hipErrorInvalidValue;
  if ((q_sb | q_sn | k_sb | k_sn | v_sb | v_sn) % 8 || (o_sb | o_sn) % 4) return (int)hipErrorInvalidValue;
  Params p;
  p.q = (const uint16_t*)q, p.k = (const uint16_t*)k, p.v = (const uint16_t*)v, p.o = (uint16_t*)out;
  hipStream_t st = (hipStream_t)stream;
  if ((v_path >= 4 && v_path <= 10)) {  // 4 / 5 / 6 = software-pipelined (5: lazy maximum, 6: + exponent-domain q), 8 = small-KV
    if (v_path == 8 && Nk > 96) return (int)hipErrorInvalidValue;
    // the 8-wave variants' staging spans 64 key rows per pass, and 6 / 7 were only ever validated on full tiles: fewer than one
    // 64-key tile is rejected here, not only in the Python wrapper (which asks for >= 128)
    if ((v_path == 6 || v_path == 7 || v_path == 9 || v_path == 10) && Nk < 64) return (int)hipErrorInvalidValue;
    // the pipelined kernel addresses K / V with 32-bit byte offsets from the head's base pointer
    if (v_path != 8 && ((int64_t)(Nk + 2 * KT) * (k_sn > v_sn ? k_sn : v_sn) * 2 >= 0x7fffffffll)) return (int)hipErrorInvalidValue;
    const int rows_per_wg = v_path == 8 ? QB * SK_QBLOCKS : (v_path >= 9 ? 2 * QB : QB);
    p.Nq = Nq, p.Nk = Nk, p.H = H, p.BH = B * H, p.nqb = (Nq + rows_per_wg - 1) / rows_per_wg;
    p.q_sb = q_sb, p.q_sn = q_sn, p.k_sb = k_sb, p.k_sn = k_sn, p.v_sb = v_sb, p.v_sn = v_sn, p.o_sb = o_sb, p.o_sn = o_sn;
    p.scale_log2e = scale * 1.44269504088896340736f;
    const int64_t nb = (int64_t)p.BH * p.nqb;
    if (nb > 0x7fffffff) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)nb), block(256);
    if (dtype != ED_BF16 && dtype != ED_F16) return (int)hipErrorInvalidValue;
    if (v_path == 4) {
      if (dtype == ED_BF16) k_flash_attn_pipe<BF, false><<<grid, block, 0, st>>>(p);
      else k_flash_attn_pipe<HF, false><<<grid, block, 0, st>>>(p);
    } else if (v_path == 5) {
      if (dtype == ED_BF16) k_flash_attn_pipe<BF, true><<<grid, block, 0, st>>>(p);
      else k_flash_attn_pipe<HF, true><<<grid, block, 0, st>>>(p);
    } else if (v_path == 6) {
      if (dtype == ED_BF16) k_flash_attn_pipe<BF, true, true><<<grid, block, 0, st>>>(p);
      else k_flash_attn_pipe<HF, true, true><<<grid, block, 0, st>>>(p);
    } else if (v_path == 7) {  // 4 with the LDS operand reads three MFMAs ahead
      if (dtype == ED_BF16) k_flash_attn_pipe<BF, false, false, 3><<<grid, block, 0, st>>>(p);
      else k_flash_attn_pipe<HF, false, false, 3><<<grid, block, 0, st>>>(p);
    } else if (v_path == 9) {  // 4 with 8 waves (256 query rows) per workgroup
      if (dtype == ED_BF16) k_flash_attn_pipe<BF, false, false, 1, 8><<<grid, dim3(512), 0, st>>>(p);
      else k_flash_attn_pipe<HF, false, false, 1, 8><<<grid, dim3(512), 0, st>>>(p);
    } else if (v_path == 10) {  // 5 with 8 waves per workgroup
      if (dtype == ED_BF16) k_flash_attn_pipe<BF, true, false, 1, 8><<<grid, dim3(512), 0, st>>>(p);
      else k_flash_attn_pipe<HF, true, false, 1, 8><<<grid, dim3(512), 0, st>>>(p);
    } else if (Nk <= 64) {
      if (dtype == ED_BF16) k_flash_attn_smallkv<BF, 2><<<grid, block, 0, st>>>(p);
      else k_flash_attn_smallkv<HF, 2><<<grid, block, 0, st>>>(p);
    } else {
      if (dtype == ED_BF16) k_flash_attn_smallkv<BF, 3><<<grid, block, 0, st>>>(p);
      else k_flash_attn_smallkv<HF, 3><<<grid, block, 0, st>>>(p);
    }
    return (int)hipGetLastError();
  }
  // v_path: bit 0 = V staging path (0 transpose-read, 1 V^T tile); bit 1 = 64 query rows per wave (256 per workgroup)
  const int qn = (v_path & 2) ? 2 : 1;
  p.Nq = Nq, p.Nk = Nk, p.H = H, p.BH = B * H, p.nqb = (Nq + QB * qn - 1) / (QB * qn);
  p.q_sb = q_sb, p.q_sn = q_sn, p.k_sb = k_sb, p.k_sn = k_sn, p.v_sb = v_sb, p.v_sn = v_sn, p.o_sb = o_sb, p.o_sn = o_sn;
  p.scale_log2e = scale * 1.44269504088896340736f;
  const int64_t blocks = (int64_t)p.BH * p.nqb;
  if (blocks > 0x7fffffff) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(256);
#define ED_FA(T)                                                                 \
  if (v_path == 0) k_flash_attn_fwd<T, true, 1><<<grid, block, 0, st>>>(p);      \
  else if (v_path == 1) k_flash_attn_fwd<T, false, 1><<<grid, block, 0, st>>>(p); \
  else if (v_path == 2) k_flash_attn_fwd<T, true, 2><<<grid, block, 0, st>>>(p);  \
  else if (v_path == 3) k_flash_attn_fwd<T, false, 2><<<grid, block, 0, st>>>(p); \
  else return (int)hipErrorInvalidValue;
  if (dtype == ED_BF16) {
    ED_FA(BF)
  } else if (dtype == ED_F16) {
    ED_FA(HF)
  } else {
    return (int)hipErrorInvalidValue;
  }
#undef ED_FA
  return (int)hipGetLastError();
}

}  // extern "C"
