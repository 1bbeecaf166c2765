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

}  // namespace

extern "C" {

int ed_flash_attention(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H, int Nq, int Nk,
                       int head_dim, int64_t q_sb, int64_t q_sn, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn,
                       int64_t o_sb, int64_t o_sn, float scale, int v_path, void* stream) {
  if (B == 0 || H == 0 || Nq == 0) return 0;
  if (head_dim != D || Nk <= 0) return (int)hipErrorInvalidValue;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15u) || ((uintptr_t)out & 7u)) return (int)hipErrorInvalidValue;
  if ((q_sb | q_sn | k_sb | k_sn | v_sb | v_sn) % 8 || (o_sb | o_sn) % 4) return (int)hipErrorInvalidValue;
  Params p;
  p.q = (const uint16_t*)q, p.k = (const uint16_t*)k, p.v = (const uint16_t*)v, p.o = (uint16_t*)out;
  // v_path: bit 0 = V staging path (0 transpose-read, 1 V^T tile); bit 1 = 64 query rows per wave (256 per workgroup)
  const int qn = (v_path & 2) ? 2 : 1;
  p.Nq = Nq, p.Nk = Nk, p.H = H, p.BH = B * H, p.nqb = (Nq + QB * qn - 1) / (QB * qn);
  p.q_sb = q_sb, p.q_sn = q_sn, p.k_sb = k_sb, p.k_sn = k_sn, p.v_sb = v_sb, p.v_sn = v_sn, p.o_sb = o_sb, p.o_sn = o_sn;
  p.scale_log2e = scale * 1.44269504088896340736f;
  const int64_t blocks = (int64_t)p.BH * p.nqb;
  if (blocks > 0x7fffffff) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
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
