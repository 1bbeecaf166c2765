// unet_kernels.hip -- gfx950 fused memory-bound kernels INSIDE the UNet (model side of the hot path's boundary,
// elastic_diffusion.py:422-426 `self.unet(...)`).  rocprofv3 of the torch-ROCm SDXL UNet at batch 20
// (profiles/r1_unet_sdxl_b20_kernel_stats.csv) shows ~25 % of the forward in un-fused, partly strided elementwise
// kernels: GELU + MUL on chunk() views (GEGLU), GroupNorm = moments + affine + separate SiLU, NCHW->token copies.
// These kernels fuse them: one pass (GEGLU) / two passes (GroupNorm[+SiLU][+token layout]) with 16-byte vector
// access.  The GEMM / conv / attention contractions stay in hipBLASLt / MIOpen / SDPA (MFMA).
//
// Numerics follow the torch kernels they replace (fp32 math, bf16/f16 rounding at the same points), so swapping them
// in changes UNet outputs by at most a few low-order bits.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "elastic_hip.h"

namespace {

// ---- 16-bit element helpers ------------------------------------------------------------------------
struct BF16 {
  static __device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
  // gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN) -- one instruction instead of
  // the six-instruction integer sequence of rounds 1-2
  static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
};
struct F16 {
  static __device__ __forceinline__ float to_f32(uint16_t v) { return __half2float(__ushort_as_half(v)); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return __half_as_ushort(__float2half_rn(f)); }
};

struct alignas(16) U16x8 { uint16_t v[8]; };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
// x * rcp(1 + exp2(-x log2 e)): v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~20-instruction expf + IEEE division;
// the result is rounded to a 16-bit type right after, which absorbs the difference except at rounding ties
__device__ __forceinline__ float silu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}

// ---- GEGLU: out[m,i] = in[m,i] * gelu(in[m,I+i]) ---------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_geglu(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t M, int I) {
  const int vec_per_row = I / 8;
  const int64_t total = M * vec_per_row;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int64_t m = t / vec_per_row;
    int j = (int)(t - m * vec_per_row) * 8;
    const uint16_t* row = in + m * (2 * (int64_t)I);
    U16x8 h = *reinterpret_cast<const U16x8*>(row + j);
    U16x8 g = *reinterpret_cast<const U16x8*>(row + I + j);
    U16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float ge = T::to_f32(T::from_f32(gelu_erf(T::to_f32(g.v[e]))));  // torch rounds gelu(gate) to 16 bit first
      o.v[e] = T::from_f32(T::to_f32(h.v[e]) * ge);
    }
    *reinterpret_cast<U16x8*>(out + m * (int64_t)I + j) = o;
  }
}

// ---- GroupNorm (+SiLU) over NCHW, one workgroup per (sample, group) --------------------------------
struct Welford {
  float n, mean, m2;
};
__device__ __forceinline__ Welford merge(Welford a, Welford b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  float n = a.n + b.n;
  float d = b.mean - a.mean;
  Welford r;
  r.n = n;
  r.mean = a.mean + d * (b.n / n);
  r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / n);
  return r;
}

#define GN_THREADS 512

// ``cb`` (optional): per-channel bias of this (sample, group), cb[c - g*cpg]; the normalised tensor is
// round16(x + cb[c]) -- what torch's 16-bit `h + temb[:, :, None, None]` (ResnetBlock2D) hands to norm2.
// ``kb`` (optional): the producing convolution's bias, kb[c]: MIOpen leaves the bias of a convolution to a separate
// broadcast-add kernel, so the convolution is run bias-free and the add is folded in here, rounded to 16 bit like that
// kernel's output was, BEFORE the time-embedding add (which torch also rounds).
template <typename T>
__device__ __forceinline__ float biased(uint16_t x, float kb, bool has_kb, float cb, bool has_cb) {
  float f = T::to_f32(x);
  if (has_kb) f = T::to_f32(T::from_f32(f + kb));
  return has_cb ? T::to_f32(T::from_f32(f + cb)) : f;
}

template <typename T>
__device__ __forceinline__ void group_stats(const uint16_t* __restrict__ chunk, int64_t len, int HW,
                                            const uint16_t* __restrict__ kb, const uint16_t* __restrict__ cb, float eps,
                                            float& mean, float& rstd) {
  // pass 1: per-thread sum / sum of squares over 16-byte vectors (few hundred elements per thread), then Chan merge
  float s = 0.f, ss = 0.f, cnt = 0.f;
  const bool has_cb = cb != nullptr, has_kb = kb != nullptr;
  for (int64_t i = (int64_t)threadIdx.x * 8; i < len; i += GN_THREADS * 8) {
    U16x8 v = *reinterpret_cast<const U16x8*>(chunk + i);
    const float cbv = has_cb ? T::to_f32(cb[i / HW]) : 0.f;  // HW % 8 == 0: a vector never straddles channels
    const float kbv = has_kb ? T::to_f32(kb[i / HW]) : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = biased<T>(v.v[e], kbv, has_kb, cbv, has_cb);
      s += f;
      ss += f * f;
    }
    cnt += 8.f;
  }
  Welford w;
  w.n = cnt;
  w.mean = cnt > 0.f ? s / cnt : 0.f;
  w.m2 = cnt > 0.f ? fmaxf(ss - s * w.mean, 0.f) : 0.f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Welford o;
    o.n = __shfl_down(w.n, off, 64);
    o.mean = __shfl_down(w.mean, off, 64);
    o.m2 = __shfl_down(w.m2, off, 64);
    w = merge(w, o);
  }
  __shared__ Welford part[GN_THREADS / 64];
  __shared__ float sh_mean, sh_rstd;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    Welford t = part[0];
    for (int k = 1; k < GN_THREADS / 64; ++k) t = merge(t, part[k]);
    sh_mean = t.mean;
    sh_rstd = rsqrtf(t.m2 / t.n + eps);
  }
  __syncthreads();
  mean = sh_mean;
  rstd = sh_rstd;
}

// out layout NCHW (TOKENS == false) or [N, HW, C] (TOKENS == true, the transformer's token layout)
template <typename T, bool ACT, bool TOKENS>
__global__ void __launch_bounds__(GN_THREADS)
k_groupnorm(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
            const uint16_t* __restrict__ conv_bias, const uint16_t* __restrict__ chan_bias, uint16_t* __restrict__ out,
            int C, int HW, int G, float eps) {
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const int64_t len = (int64_t)cpg * HW;
  const uint16_t* chunk = x + ((int64_t)n * C + (int64_t)g * cpg) * HW;
  const uint16_t* cb = chan_bias ? chan_bias + (int64_t)n * C + (int64_t)g * cpg : nullptr;
  const uint16_t* kb = conv_bias ? conv_bias + (int64_t)g * cpg : nullptr;
  const bool has_cb = cb != nullptr, has_kb = kb != nullptr;
  float mean, rstd;
  group_stats<T>(chunk, len, HW, kb, cb, eps, mean, rstd);
  if (!TOKENS) {
    uint16_t* dst = out + ((int64_t)n * C + (int64_t)g * cpg) * HW;
    for (int64_t i = (int64_t)threadIdx.x * 8; i < len; i += GN_THREADS * 8) {
      int cl = (int)(i / HW);  // HW % 8 == 0: a vector never straddles channels
      int c = g * cpg + cl;
      float a = rstd * T::to_f32(gamma[c]);
      float b = fmaf(-a, mean, T::to_f32(beta[c]));
      const float cbv = has_cb ? T::to_f32(cb[cl]) : 0.f;
      const float kbv = has_kb ? T::to_f32(kb[cl]) : 0.f;
      U16x8 v = *reinterpret_cast<const U16x8*>(chunk + i), o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = T::to_f32(T::from_f32(fmaf(a, biased<T>(v.v[e], kbv, has_kb, cbv, has_cb), b)));  // torch rounds the norm output first
        o.v[e] = ACT ? T::from_f32(silu(y)) : T::from_f32(y);
      }
      *reinterpret_cast<U16x8*>(dst + i) = o;
    }
  } else {
    // one thread per token p: read its cpg channel values (coalesced across threads for each channel), write cpg
    // contiguous values of row p
    for (int p = threadIdx.x; p < HW; p += GN_THREADS) {
      uint16_t* row = out + ((int64_t)n * HW + p) * C + (int64_t)g * cpg;
      for (int cc = 0; cc < cpg; cc += 4) {  // cpg % 4 == 0 (checked on the host): 8-byte stores
        uint16_t o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int c = g * cpg + cc + e;
          float a = rstd * T::to_f32(gamma[c]);
          float b = fmaf(-a, mean, T::to_f32(beta[c]));
          const float cbv = has_cb ? T::to_f32(cb[cc + e]) : 0.f;
          const float kbv = has_kb ? T::to_f32(kb[cc + e]) : 0.f;
          float y = T::to_f32(T::from_f32(fmaf(a, biased<T>(chunk[(int64_t)(cc + e) * HW + p], kbv, has_kb, cbv, has_cb), b)));
          o4[e] = ACT ? T::from_f32(silu(y)) : T::from_f32(y);
        }
        uint2 pk;
        pk.x = (uint32_t)o4[0] | ((uint32_t)o4[1] << 16);
        pk.y = (uint32_t)o4[2] | ((uint32_t)o4[3] << 16);
        *reinterpret_cast<uint2*>(row + cc) = pk;
      }
    }
  }
}


// ---- GroupNorm for LARGE groups: statistics and apply as two fully parallel launches ---------------------------------
// One workgroup per (sample, group) leaves the chip under-filled when a group is hundreds of KB (SDXL level 1: 10-30
// channels x 128 x 128): N*G = 640 workgroups walk 0.3-1 MB each, twice (measured 2.4 TB/s).  Here every group is cut
// into `chunks` token ranges; k_gn_split_stats writes one Welford partial per (group, chunk), k_gn_split_apply merges the
// partials of its group (a few dozen floats) and normalises its own range.  Same traffic (2 reads + 1 write), all of it
// spread over thousands of workgroups.  Partials are merged in chunk order: deterministic.
#define GNS_THREADS 256

template <typename T>
__global__ void __launch_bounds__(GNS_THREADS)
k_gn_split_stats(const uint16_t* __restrict__ x, const uint16_t* __restrict__ conv_bias,
                 const uint16_t* __restrict__ chan_bias, float* __restrict__ partial, int C, int HW, int G, int chunks,
                 int span) {
  const int ng = blockIdx.y, chunk = blockIdx.x;
  const int n = ng / G, g = ng % G, cpg = C / G;
  const uint16_t* base = x + ((int64_t)n * C + (int64_t)g * cpg) * HW;
  const uint16_t* cb = chan_bias ? chan_bias + (int64_t)n * C + (int64_t)g * cpg : nullptr;
  const uint16_t* kb = conv_bias ? conv_bias + (int64_t)g * cpg : nullptr;
  const bool has_cb = cb != nullptr, has_kb = kb != nullptr;
  const int p0 = chunk * span, p1 = min(HW, p0 + span);
  const int vec_per_row = (p1 - p0) >> 3;  // span and HW are multiples of 8
  float s = 0.f, ss = 0.f, cnt = 0.f;
  for (int it = threadIdx.x; it < cpg * vec_per_row; it += GNS_THREADS) {
    const int c = it / vec_per_row, pv = it - c * vec_per_row;
    U16x8 v = *reinterpret_cast<const U16x8*>(base + (int64_t)c * HW + p0 + (pv << 3));
    const float cbv = has_cb ? T::to_f32(cb[c]) : 0.f, kbv = has_kb ? T::to_f32(kb[c]) : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = biased<T>(v.v[e], kbv, has_kb, cbv, has_cb);
      s += f;
      ss += f * f;
    }
    cnt += 8.f;
  }
  Welford w;
  w.n = cnt;
  w.mean = cnt > 0.f ? s / cnt : 0.f;
  w.m2 = cnt > 0.f ? fmaxf(ss - s * w.mean, 0.f) : 0.f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Welford o;
    o.n = __shfl_down(w.n, off, 64);
    o.mean = __shfl_down(w.mean, off, 64);
    o.m2 = __shfl_down(w.m2, off, 64);
    w = merge(w, o);
  }
  __shared__ Welford part[GNS_THREADS / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    Welford t = part[0];
    for (int k = 1; k < GNS_THREADS / 64; ++k) t = merge(t, part[k]);
    float* dst = partial + ((int64_t)ng * chunks + chunk) * 3;
    dst[0] = t.n, dst[1] = t.mean, dst[2] = t.m2;
  }
}

template <typename T, bool ACT, bool TOKENS>
__global__ void __launch_bounds__(GNS_THREADS)
k_gn_split_apply(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
                 const uint16_t* __restrict__ conv_bias, const uint16_t* __restrict__ chan_bias,
                 const float* __restrict__ partial, uint16_t* __restrict__ out, int C, int HW, int G, int chunks, int span,
                 float eps) {
  const int ng = blockIdx.y, chunk = blockIdx.x;
  const int n = ng / G, g = ng % G, cpg = C / G;
  __shared__ float sh_mean, sh_rstd;
  if (threadIdx.x == 0) {
    const float* src = partial + (int64_t)ng * chunks * 3;
    Welford t = {src[0], src[1], src[2]};
    for (int k = 1; k < chunks; ++k) t = merge(t, Welford{src[3 * k], src[3 * k + 1], src[3 * k + 2]});
    sh_mean = t.mean;
    sh_rstd = rsqrtf(t.m2 / t.n + eps);
  }
  __syncthreads();
  const float mean = sh_mean, rstd = sh_rstd;
  const uint16_t* base = x + ((int64_t)n * C + (int64_t)g * cpg) * HW;
  const uint16_t* cb = chan_bias ? chan_bias + (int64_t)n * C + (int64_t)g * cpg : nullptr;
  const uint16_t* kb = conv_bias ? conv_bias + (int64_t)g * cpg : nullptr;
  const bool has_cb = cb != nullptr, has_kb = kb != nullptr;
  const int p0 = chunk * span, p1 = min(HW, p0 + span);
  if (!TOKENS) {
    uint16_t* dst = out + ((int64_t)n * C + (int64_t)g * cpg) * HW;
    const int vec_per_row = (p1 - p0) >> 3;
    for (int it = threadIdx.x; it < cpg * vec_per_row; it += GNS_THREADS) {
      const int cl = it / vec_per_row, pv = it - cl * vec_per_row, c = g * cpg + cl;
      const int64_t off = (int64_t)cl * HW + p0 + (pv << 3);
      const float a = rstd * T::to_f32(gamma[c]);
      const float b = fmaf(-a, mean, T::to_f32(beta[c]));
      const float cbv = has_cb ? T::to_f32(cb[cl]) : 0.f, kbv = has_kb ? T::to_f32(kb[cl]) : 0.f;
      U16x8 v = *reinterpret_cast<const U16x8*>(base + off), o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = T::to_f32(T::from_f32(fmaf(a, biased<T>(v.v[e], kbv, has_kb, cbv, has_cb), b)));
        o.v[e] = ACT ? T::from_f32(silu(y)) : T::from_f32(y);
      }
      *reinterpret_cast<U16x8*>(dst + off) = o;
    }
  } else {
    for (int p = p0 + threadIdx.x; p < p1; p += GNS_THREADS) {
      uint16_t* row = out + ((int64_t)n * HW + p) * C + (int64_t)g * cpg;
      for (int cc = 0; cc < cpg; cc += 4) {
        uint16_t o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int c = g * cpg + cc + e;
          float a = rstd * T::to_f32(gamma[c]);
          float b = fmaf(-a, mean, T::to_f32(beta[c]));
          const float cbv = has_cb ? T::to_f32(cb[cc + e]) : 0.f, kbv = has_kb ? T::to_f32(kb[cc + e]) : 0.f;
          float y = T::to_f32(T::from_f32(fmaf(a, biased<T>(base[(int64_t)(cc + e) * HW + p], kbv, has_kb, cbv, has_cb), b)));
          o4[e] = ACT ? T::from_f32(silu(y)) : T::from_f32(y);
        }
        uint2 pk;
        pk.x = (uint32_t)o4[0] | ((uint32_t)o4[1] << 16);
        pk.y = (uint32_t)o4[2] | ((uint32_t)o4[3] << 16);
        *reinterpret_cast<uint2*>(row + cc) = pk;
      }
    }
  }
}

// ---- GroupNorm (+SiLU) over channels-last activations [N, HW, C] ------------------------------------
// Round 6: THREE launches again -- (1) per-block partial sums per group, (2) a finalize launch of N blocks that reduces a sample's
// partials to (mean, rstd) per group in double and fixed order, (3) apply, which reads its (at most two) groups' statistics with two
// 8-byte loads.  Round 3 had folded (2) into every apply block's prologue to save an eager-mode launch (17.5 us each in round 2);
// inside a hipGraph a dependent N-block kernel costs 2-3 us, and the folded prologue is what kept the chunks at 64 K elements (every
// apply block re-read ALL of its sample's partials, so there could not be many): [20, 1280, 32, 32] ran as 400 blocks of which 160 of
// 256 threads had a column -- 1.5 blocks per CU, 16 KiB of loads in flight per CU, 2.2 TB/s (profiles/r3_s3_probe_gn_bra.jsonl)
// against 4.4 TB/s for the large shapes.  Now the block size follows the column count (gn_plan: no thread without a column) and the
// chunks are sized for >= 8 blocks per CU.
// Reads 2x, writes 1x; every access is a full 16-byte channel vector, FOUR rows in flight per thread (the round-2 loops
// had one load outstanding per thread: 2.7-3.0 TB/s, latency-bound); the output *is* the transformer's token layout
// (no permute copy) and the layout MIOpen's CK convolutions consume.
#define GNL_THREADS 256

// Every thread owns a fixed 8-channel column (two when C > 2048) and walks rows, so its sums stay in registers; the
// reduction over row lanes and then over the columns of each group goes through LDS in a fixed order: no atomics,
// bit-reproducible statistics.
#define GNL_MAXCOL 2  // columns per thread: C <= 8 * 256 * GNL_MAXCOL = 4096
#define GNL_UNROLL 4  // rows in flight per thread

// Launch plan of the channels-last GroupNorm (round 6), shared by the launcher and the workspace query:
//   threads         VC = C / 8 column vectors; with VC <= 256 a block is R = 256 / VC row lanes of VC threads, rounded up to whole waves
//                   (C = 1280: 160 columns -> 192 threads instead of 256 with 96 idle); above, 256 threads own two columns each;
//   rows_per_block  sized for >= GNL_TARGET_BLOCKS blocks in the launch (8 per CU: ~100 KiB of loads in flight per CU), at least one
//                   unrolled sweep (GNL_UNROLL R rows) and at most the round-3 chunk (64 K elements).
#define GNL_TARGET_BLOCKS 2048
struct GnPlan {
  int threads, rows_per_block, nchunks;
};
static inline GnPlan gn_plan(int N, int C, int HW) {
  GnPlan pl;
  const int VC = C >> 3;
  int R = 1;
  if (VC <= GNL_THREADS) {
    R = GNL_THREADS / VC;
    pl.threads = ((VC * R + 63) / 64) * 64;
  } else {
    pl.threads = GNL_THREADS;
  }
  const int64_t want = ((int64_t)N * HW + GNL_TARGET_BLOCKS - 1) / GNL_TARGET_BLOCKS;   // rows per block for the target block count
  int rpb = (int)(want < 1 ? 1 : want);
  const int lo = GNL_UNROLL * R, hi = (65536 + C - 1) / C;
  if (rpb < lo) rpb = lo;
  if (rpb > hi) rpb = hi > lo ? hi : lo;
  rpb = ((rpb + R - 1) / R) * R;            // whole row-lane sweeps
  pl.rows_per_block = rpb;
  pl.nchunks = (HW + rpb - 1) / rpb;
  return pl;
}

template <typename T>
__device__ __forceinline__ void gn_accumulate(const U16x8& v, const U16x8& kbv, bool has_kb, const U16x8& cbv, bool has_cb,
                                              int split, float& s0, float& q0, float& s1, float& q1) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = biased<T>(v.v[e], has_kb ? T::to_f32(kbv.v[e]) : 0.f, has_kb, has_cb ? T::to_f32(cbv.v[e]) : 0.f, has_cb);
    if (e < split) {
      s0 += f;
      q0 += f * f;
    } else {
      s1 += f;
      q1 += f * f;
    }
  }
}

// X32 (round 6, the fp32-residual-stream mode of the UNet): x is an fp32 [N, HW, C] tensor -- 8 channels = two 16-byte loads -- and there
// are no folded biases; statistics, affine and output type are unchanged (the result is the 16-bit operand of the next GEMM).
template <bool X32>
struct Row8 {      // 8 consecutive channels of one pixel as loaded
  U16x8 h;
  float4 lo, hi;
  __device__ __forceinline__ void load(const void* base, int64_t elem) {
    if (X32) {
      const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
      lo = p[0], hi = p[1];
    } else {
      h = *reinterpret_cast<const U16x8*>(reinterpret_cast<const uint16_t*>(base) + elem);
    }
  }
  __device__ __forceinline__ float f32(int e) const {   // X32 only
    return e == 0 ? lo.x : e == 1 ? lo.y : e == 2 ? lo.z : e == 3 ? lo.w : e == 4 ? hi.x : e == 5 ? hi.y : e == 6 ? hi.z : hi.w;
  }
};

template <typename T, bool X32>
__device__ __forceinline__ void gn_accumulate8(const Row8<X32>& v, const U16x8& kbv, bool has_kb, const U16x8& cbv, bool has_cb,
                                               int split, float& s0, float& q0, float& s1, float& q1) {
  if (!X32) {
    gn_accumulate<T>(v.h, kbv, has_kb, cbv, has_cb, split, s0, q0, s1, q1);
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = v.f32(e);
    if (e < split) {
      s0 += f;
      q0 += f * f;
    } else {
      s1 += f;
      q1 += f * f;
    }
  }
}

template <typename T, bool X32 = false>
__global__ void __launch_bounds__(GNL_THREADS)
k_gn_nhwc_partial(const void* __restrict__ x, const void* __restrict__ x2, int C1, const uint16_t* __restrict__ conv_bias,
                  const uint16_t* __restrict__ chan_bias, float* __restrict__ partial, int C, int HW, int G,
                  int rows_per_block) {
  extern __shared__ float sh[];  // [lanes][VC][4]: (sum, sumsq) of the column's first / second group part
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int VC = C >> 3, cpg = C / G;
  const int NT = blockDim.x;                                  // gn_plan: a multiple of 64, <= GNL_THREADS
  const int ncol = (VC + NT - 1) / NT;                        // 1 or 2 columns per thread
  const int R = ncol == 1 ? NT / VC : 1;                      // row lanes
  const int my_r = ncol == 1 ? threadIdx.x / VC : 0;
  const int my_c = ncol == 1 ? threadIdx.x % VC : threadIdx.x;
  const bool active = ncol == 1 ? (int)threadIdx.x < R * VC : true;
  const int r0 = chunk * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  const uint16_t* cbn = chan_bias ? chan_bias + (int64_t)n * C : nullptr;
  const bool has_cb = cbn != nullptr, has_kb = conv_bias != nullptr;
  if (active) {
#pragma unroll
    for (int j = 0; j < GNL_MAXCOL; ++j) {
      const int vc = my_c + j * NT;
      if (j >= ncol || vc >= VC) break;
      const int c0 = vc << 3;
      const int split = (c0 / cpg + 1) * cpg - c0;  // channels [0, split) of the vector belong to its first group
      U16x8 kbv = {}, cbv = {};
      if (has_kb) kbv = *reinterpret_cast<const U16x8*>(conv_bias + c0);
      if (has_cb) cbv = *reinterpret_cast<const U16x8*>(cbn + c0);
      float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
      // two sources (ed_groupnorm_nhwc_cat: the normalised tensor is cat([x, x2], channels), which is never materialised): channels
      // [0, C1) are columns of x (C1 per pixel), [C1, C) of x2 (C - C1 per pixel); one source: C1 == C
      const bool second = c0 >= C1;
      const void* xs = second ? x2 : x;
      const int cs = second ? C - C1 : C1;
      const int64_t col = (int64_t)n * HW * cs + (second ? c0 - C1 : c0);
      const int64_t step = (int64_t)R * cs;
      int row = r0 + my_r;
      for (; row + (GNL_UNROLL - 1) * R < r1; row += GNL_UNROLL * R) {  // GNL_UNROLL independent (pairs of) 16-byte loads in flight
        const int64_t p0 = col + (int64_t)row * cs;
        Row8<X32> v[GNL_UNROLL];
#pragma unroll
        for (int u = 0; u < GNL_UNROLL; ++u) v[u].load(xs, p0 + u * step);
#pragma unroll
        for (int u = 0; u < GNL_UNROLL; ++u) gn_accumulate8<T, X32>(v[u], kbv, has_kb, cbv, has_cb, split, s0, q0, s1, q1);
      }
      for (; row < r1; row += R) {
        Row8<X32> v;
        v.load(xs, col + (int64_t)row * cs);
        gn_accumulate8<T, X32>(v, kbv, has_kb, cbv, has_cb, split, s0, q0, s1, q1);
      }
      float* slot = sh + ((int64_t)my_r * VC + vc) * 4;
      slot[0] = s0, slot[1] = q0, slot[2] = s1, slot[3] = q1;
    }
  }
  __syncthreads();
  // group g: its columns in ascending order, row lanes in ascending order inside each
  float* dst = partial + ((int64_t)n * gridDim.x + chunk) * 2 * G;
  for (int g = threadIdx.x; g < G; g += NT) {
    const int cfirst = (g * cpg) >> 3, clast = ((g + 1) * cpg - 1) >> 3;
    float s = 0.f, q = 0.f;
    for (int vc = cfirst; vc <= clast; ++vc) {
      const int part = ((vc << 3) / cpg == g) ? 0 : 1;  // this column's first group is g, or g is its second one
      for (int r = 0; r < R; ++r) {
        const float* slot = sh + ((int64_t)r * VC + vc) * 4 + 2 * part;
        s += slot[0];
        q += slot[1];
      }
    }
    dst[2 * g] = s;
    dst[2 * g + 1] = q;
  }
}

// Finalize: one block per sample reduces its partial sums [nchunks, G, 2] to (mean, rstd) per group -- thread (g, part) adds every
// parts-th chunk in double, then one thread per group adds the parts in ascending order (deterministic; the arithmetic of the round-3
// apply prologue, now run once per sample instead of once per apply block).
#define GNL_MAXG 256
__global__ void __launch_bounds__(GNL_THREADS)
k_gn_nhwc_finalize(const float* __restrict__ partial, float* __restrict__ stats_out, int G, int nchunks, double count, float eps) {
  __shared__ double red[2 * GNL_THREADS];
  const int n = blockIdx.x;
  const int parts = GNL_THREADS / G > 0 ? GNL_THREADS / G : 1;  // G <= 256
  const int g = threadIdx.x % G, part = threadIdx.x / G;
  double s = 0.0, q = 0.0;
  if (part < parts) {
    const float* p = partial + (int64_t)n * nchunks * 2 * G + 2 * g;
    int c = part;
    for (; c + 3 * parts < nchunks; c += 4 * parts) {          // four independent 8-byte loads in flight
      const float2 v0 = *reinterpret_cast<const float2*>(p + (int64_t)c * 2 * G);
      const float2 v1 = *reinterpret_cast<const float2*>(p + (int64_t)(c + parts) * 2 * G);
      const float2 v2 = *reinterpret_cast<const float2*>(p + (int64_t)(c + 2 * parts) * 2 * G);
      const float2 v3 = *reinterpret_cast<const float2*>(p + (int64_t)(c + 3 * parts) * 2 * G);
      s += (double)v0.x, q += (double)v0.y;
      s += (double)v1.x, q += (double)v1.y;
      s += (double)v2.x, q += (double)v2.y;
      s += (double)v3.x, q += (double)v3.y;
    }
    for (; c < nchunks; c += parts) {
      s += (double)p[(int64_t)c * 2 * G];
      q += (double)p[(int64_t)c * 2 * G + 1];
    }
  }
  red[2 * threadIdx.x] = s, red[2 * threadIdx.x + 1] = q;
  __syncthreads();
  if ((int)threadIdx.x < G) {
    double ss = 0.0, qq = 0.0;
    for (int k = 0; k < parts; ++k) {
      ss += red[2 * (k * G + threadIdx.x)];
      qq += red[2 * (k * G + threadIdx.x) + 1];
    }
    const double mean = ss / count;
    double var = qq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats_out[((int64_t)n * G + threadIdx.x) * 2] = (float)mean;
    stats_out[((int64_t)n * G + threadIdx.x) * 2 + 1] = rsqrtf((float)var + eps);
  }
}

// Same thread <-> column mapping as the statistics kernel: gamma / beta / folded biases / the (at most two) group statistics of a
// thread's 8 channels are loaded ONCE and it streams rows, GNL_UNROLL at a time.
template <typename T, bool ACT, bool X32 = false>
__global__ void __launch_bounds__(GNL_THREADS)
k_gn_nhwc_apply(const void* __restrict__ x, const void* __restrict__ x2, int C1, const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
                const uint16_t* __restrict__ conv_bias, const uint16_t* __restrict__ chan_bias,
                const float* __restrict__ stats_all, uint16_t* __restrict__ out, int C, int HW, int G, int rows_per_block) {
  // the statistics pass read the tensor front to back; this pass walks it BACK to front (blocks are dispatched in ascending order), so what
  // it reads first is what the last-level cache saw last -- a tensor larger than the cache is then half served from it instead of not at all
  const int n = gridDim.y - 1 - blockIdx.y, chunk = gridDim.x - 1 - blockIdx.x;
  const float* stats = stats_all + (int64_t)n * G * 2;
  const int VC = C >> 3, cpg = C / G;
  const int NT = blockDim.x;
  const int ncol = (VC + NT - 1) / NT;
  const int R = ncol == 1 ? NT / VC : 1;
  const int my_r = ncol == 1 ? threadIdx.x / VC : 0;
  const int my_c = ncol == 1 ? threadIdx.x % VC : threadIdx.x;
  if (ncol == 1 && (int)threadIdx.x >= R * VC) return;
  const int r0 = chunk * rows_per_block, r1 = min(HW, r0 + rows_per_block);
  const bool has_cb = chan_bias != nullptr, has_kb = conv_bias != nullptr;
  uint16_t* ob = out + (int64_t)n * HW * C;
#pragma unroll
  for (int j = 0; j < GNL_MAXCOL; ++j) {
    const int vc = my_c + j * NT;
    if (j >= ncol || vc >= VC) break;
    const int c0 = vc << 3;
    const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;  // C/G >= 8: a vector touches at most two groups
    const float mean0 = stats[2 * g0], rstd0 = stats[2 * g0 + 1];
    const float mean1 = split < 8 ? stats[2 * (g0 + 1)] : 0.f, rstd1 = split < 8 ? stats[2 * (g0 + 1) + 1] : 0.f;
    const U16x8 gm = *reinterpret_cast<const U16x8*>(gamma + c0), bt = *reinterpret_cast<const U16x8*>(beta + c0);
    U16x8 kbv = {}, cbv = {};
    if (has_kb) kbv = *reinterpret_cast<const U16x8*>(conv_bias + c0);
    if (has_cb) cbv = *reinterpret_cast<const U16x8*>(chan_bias + (int64_t)n * C + c0);
    float a[8], b[8], kb[8], cb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mean = e < split ? mean0 : mean1, rstd = e < split ? rstd0 : rstd1;
      a[e] = rstd * T::to_f32(gm.v[e]);
      b[e] = fmaf(-a[e], mean, T::to_f32(bt.v[e]));
      kb[e] = has_kb ? T::to_f32(kbv.v[e]) : 0.f;
      cb[e] = has_cb ? T::to_f32(cbv.v[e]) : 0.f;
    }
    auto norm = [&](const Row8<X32>& v) {
      U16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xin = X32 ? v.f32(e) : biased<T>(v.h.v[e], kb[e], has_kb, cb[e], has_cb);
        // 16-bit module: the normalised value is rounded before SiLU, as torch's two kernels round it; on the fp32 stream (the tolerance
        // mode) nothing asks for that intermediate rounding: one rounding, of the final result
        const float y = X32 ? fmaf(a[e], xin, b[e]) : T::to_f32(T::from_f32(fmaf(a[e], xin, b[e])));
        o.v[e] = ACT ? T::from_f32(silu_fast(y)) : T::from_f32(y);
      }
      return o;
    };
    const bool second = c0 >= C1;                               // (two sources: see k_gn_nhwc_partial)
    const void* xs = second ? x2 : x;
    const int cs = second ? C - C1 : C1;
    const int64_t xcol = (int64_t)n * HW * cs + (second ? c0 - C1 : c0);
    const int64_t step = (int64_t)R * C, xstep = (int64_t)R * cs;
    int row = r0 + my_r;
    for (; row + (GNL_UNROLL - 1) * R < r1; row += GNL_UNROLL * R) {
      const int64_t off = (int64_t)row * C + c0, xoff = xcol + (int64_t)row * cs;
      Row8<X32> v[GNL_UNROLL];
#pragma unroll
      for (int u = 0; u < GNL_UNROLL; ++u) v[u].load(xs, xoff + u * xstep);
#pragma unroll
      for (int u = 0; u < GNL_UNROLL; ++u) *reinterpret_cast<U16x8*>(ob + off + u * step) = norm(v[u]);
    }
    for (; row < r1; row += R) {
      const int64_t off = (int64_t)row * C + c0;
      Row8<X32> v;
      v.load(xs, xcol + (int64_t)row * cs);
      *reinterpret_cast<U16x8*>(ob + off) = norm(v);
    }
  }
}


// ---- LayerNorm over the last dimension: one wavefront per row, values kept in registers ------------------
// x / out [M, D] 16-bit, D % 8 == 0, D <= 64 lanes * 8 * LN_MAX_IT.  Two passes over registers (mean, then centred
// variance): no E[x^2]-mean^2 cancellation, one global read and one global write per element.
#define LN_MAX_IT 4
#define LN_ROWS_PER_BLOCK 4

template <typename T, bool X32 = false>     // X32: x is the fp32 residual stream (round 6); the result stays 16-bit
__global__ void __launch_bounds__(64 * LN_ROWS_PER_BLOCK)
k_layernorm(const void* __restrict__ x, const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ beta,
            uint16_t* __restrict__ out, int64_t M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const int VC = D >> 3;
  float v[LN_MAX_IT][8];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    int vc = lane + it * 64;
    if (vc < VC) {
      Row8<X32> u;
      u.load(x, row * (int64_t)D + (vc << 3));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[it][e] = X32 ? u.f32(e) : T::to_f32(u.h.v[e]);
        s += v[it][e];
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    int vc = lane + it * 64;
    if (vc < VC) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[it][e] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
  uint16_t* orow = out + row * (int64_t)D;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    int vc = lane + it * 64;
    if (vc < VC) {
      U16x8 g = *reinterpret_cast<const U16x8*>(gamma + (vc << 3));
      U16x8 b = *reinterpret_cast<const U16x8*>(beta + (vc << 3));
      U16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o.v[e] = T::from_f32(fmaf(T::to_f32(g.v[e]), rstd * (v[it][e] - mean), T::to_f32(b.v[e])));
      *reinterpret_cast<U16x8*>(orow + (vc << 3)) = o;
    }
  }
}

// ---- residual add + LayerNorm: sum = round16(a + b) (torch's 16-bit add), out = LayerNorm(sum) -------------------------
// BasicTransformerBlock: `x = attn(norm(x)) + x` followed by the next `norm(x)`: one read of each addend, one write of
// the new residual stream and one of its normalisation, instead of add (2r + 1w) + LayerNorm (1r + 1w).
// S32 (round 6): b and sum_out are the fp32 residual stream -- sum = a + b in fp32, stored unrounded; a and the normalised output stay 16-bit
template <typename T, bool S32 = false>
__global__ void __launch_bounds__(64 * LN_ROWS_PER_BLOCK)
k_add_layernorm(const uint16_t* __restrict__ a, const void* __restrict__ b, const uint16_t* __restrict__ gamma,
                const uint16_t* __restrict__ beta, void* __restrict__ sum_out, uint16_t* __restrict__ out, int64_t M,
                int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const int VC = D >> 3;
  const uint16_t* ar = a + row * (int64_t)D;
  float v[LN_MAX_IT][8];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    int vc = lane + it * 64;
    if (vc < VC) {
      U16x8 ua = *reinterpret_cast<const U16x8*>(ar + (vc << 3));
      Row8<S32> ub;
      ub.load(b, row * (int64_t)D + (vc << 3));
      if (S32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[it][e] = T::to_f32(ua.v[e]) + ub.f32(e);
          s += v[it][e];
        }
        float4* sp = reinterpret_cast<float4*>(reinterpret_cast<float*>(sum_out) + row * (int64_t)D + (vc << 3));
        sp[0] = float4{v[it][0], v[it][1], v[it][2], v[it][3]};
        sp[1] = float4{v[it][4], v[it][5], v[it][6], v[it][7]};
      } else {
        U16x8 us;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          us.v[e] = T::from_f32(T::to_f32(ua.v[e]) + T::to_f32(ub.h.v[e]));
          v[it][e] = T::to_f32(us.v[e]);
          s += v[it][e];
        }
        *reinterpret_cast<U16x8*>(reinterpret_cast<uint16_t*>(sum_out) + row * (int64_t)D + (vc << 3)) = us;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    int vc = lane + it * 64;
    if (vc < VC) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[it][e] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
  uint16_t* orow = out + row * (int64_t)D;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    int vc = lane + it * 64;
    if (vc < VC) {
      U16x8 g = *reinterpret_cast<const U16x8*>(gamma + (vc << 3));
      U16x8 bt = *reinterpret_cast<const U16x8*>(beta + (vc << 3));
      U16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o.v[e] = T::from_f32(fmaf(T::to_f32(g.v[e]), rstd * (v[it][e] - mean), T::to_f32(bt.v[e])));
      *reinterpret_cast<U16x8*>(orow + (vc << 3)) = o;
    }
  }
}

// ---- out[n,c,p] = x[n,c,p] + tok[n,p,c]: the residual add that closes a Transformer2DModel ----------------------------
// (diffusers: `hidden_states.reshape(B,H,W,C).permute(0,3,1,2) + residual`): a 64 x 64 (p, c) tile goes through LDS so
// that the token-major read and the channel-major read/write are all 16-byte coalesced.  torch ran this as a strided
// elementwise kernel (elementwise_kernel_manual_unroll, 3.2 % of GPU time in the round-1 profile).
#define TA_TILE 64
#define TA_LD (TA_TILE + 2)  // 132 B rows: the column reads of 8 consecutive p hit 8 different dwords mod 32
template <typename T>
__global__ void __launch_bounds__(256)
k_tokens_add_nchw(const uint16_t* __restrict__ x, const uint16_t* __restrict__ tok, uint16_t* __restrict__ out, int C,
                  int HW) {
  __shared__ uint16_t tile[TA_TILE * TA_LD];  // [p][c]
  const int n = blockIdx.z, c0 = blockIdx.y * TA_TILE, p0 = blockIdx.x * TA_TILE;
  const uint16_t* tb = tok + ((int64_t)n * HW + p0) * C + c0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int idx = threadIdx.x + 256 * i;  // 64 rows x 8 vectors
    int p = idx >> 3, cv = (idx & 7) * 8;
    U16x8 v = *reinterpret_cast<const U16x8*>(tb + (int64_t)p * C + cv);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&tile[p * TA_LD + cv]);  // rows are 4-byte aligned (132 B pitch)
    dst[0] = (uint32_t)v.v[0] | ((uint32_t)v.v[1] << 16);
    dst[1] = (uint32_t)v.v[2] | ((uint32_t)v.v[3] << 16);
    dst[2] = (uint32_t)v.v[4] | ((uint32_t)v.v[5] << 16);
    dst[3] = (uint32_t)v.v[6] | ((uint32_t)v.v[7] << 16);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int idx = threadIdx.x + 256 * i;  // 64 channels x 8 vectors of 8 tokens
    int c = idx >> 3, pv = (idx & 7) * 8;
    int64_t off = ((int64_t)n * C + c0 + c) * HW + p0 + pv;
    U16x8 xv = *reinterpret_cast<const U16x8*>(x + off), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = T::from_f32(T::to_f32(xv.v[e]) + T::to_f32(tile[(pv + e) * TA_LD + c]));
    *reinterpret_cast<U16x8*>(out + off) = o;
  }
}

// ---- out[n,c,p] = round16(res[n,c,p] (+ rb[c])) + round16(h[n,c,p] + hb[c]) : ResnetBlock2D's closing add ------------
// with the biases of conv2 (hb) and of the 1x1 shortcut convolution (rb) folded in.  torch / MIOpen ran this as one
// broadcast bias-add kernel per convolution (elementwise_kernel_manual_unroll, 1.8 % of GPU time) plus the residual add;
// every intermediate is rounded to 16 bit exactly where those kernels rounded.
template <typename T>
__global__ void __launch_bounds__(256)
k_bias_residual_add(const uint16_t* __restrict__ h, const uint16_t* __restrict__ hb, const uint16_t* __restrict__ res,
                    const uint16_t* __restrict__ rb, uint16_t* __restrict__ out, int C, int HW, int64_t total_vec,
                    int channels_last) {
  const uint32_t HW8 = (uint32_t)HW >> 3, VC = (uint32_t)C >> 3;  // total_vec < 2^31 (checked on the host): 32-bit math
  for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < (uint32_t)total_vec; t += gridDim.x * 256u) {
    // NCHW: the 8 elements share one channel (HW % 8 == 0); channels-last: they are 8 consecutive channels (C % 8 == 0)
    const int c = channels_last ? (int)((t % VC) << 3) : (int)((t / HW8) % (uint32_t)C);
    float b1[8], b2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ce = channels_last ? c + e : c;
      b1[e] = hb ? T::to_f32(hb[ce]) : 0.f;
      b2[e] = rb ? T::to_f32(rb[ce]) : 0.f;
    }
    const int64_t off = (int64_t)t * 8;
    U16x8 hv = *reinterpret_cast<const U16x8*>(h + off), rv = *reinterpret_cast<const U16x8*>(res + off), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = T::to_f32(hv.v[e]), r = T::to_f32(rv.v[e]);
      if (hb) a = T::to_f32(T::from_f32(a + b1[e]));
      if (rb) r = T::to_f32(T::from_f32(r + b2[e]));
      o.v[e] = T::from_f32(r + a);
    }
    *reinterpret_cast<U16x8*>(out + off) = o;
  }
}

// Channels-last form with the GroupNorm kernels' thread <-> column mapping: a thread owns one 8-channel column (its two
// bias vectors are loaded ONCE, as 16-byte vectors) and streams rows, GNL_UNROLL pairs of 16-byte loads in flight.  The
// flat kernel above pays a 32-bit modulo and sixteen 2-byte bias loads per vector with one load pair outstanding per
// thread: 2.7 TB/s on the 263 MB shapes (profiles/r2_final_bench_*).
template <typename T>
__global__ void __launch_bounds__(GNL_THREADS)
k_bias_residual_add_cl(const uint16_t* __restrict__ h, const uint16_t* __restrict__ hb, const uint16_t* __restrict__ res,
                       const uint16_t* __restrict__ rb, uint16_t* __restrict__ out, int C, int64_t rows, int rows_per_block) {
  const int VC = C >> 3;
  const int ncol = (VC + GNL_THREADS - 1) / GNL_THREADS;
  const int R = ncol == 1 ? GNL_THREADS / VC : 1;
  const int my_r = ncol == 1 ? threadIdx.x / VC : 0;
  const int my_c = ncol == 1 ? threadIdx.x % VC : threadIdx.x;
  if (ncol == 1 && (int)threadIdx.x >= R * VC) return;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int j = 0; j < ncol; ++j) {
    const int vc = my_c + j * GNL_THREADS;
    if (vc >= VC) break;
    const int c0 = vc << 3;
    float b1[8], b2[8];
    U16x8 hbv = {}, rbv = {};
    if (hb) hbv = *reinterpret_cast<const U16x8*>(hb + c0);
    if (rb) rbv = *reinterpret_cast<const U16x8*>(rb + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) b1[e] = T::to_f32(hbv.v[e]), b2[e] = T::to_f32(rbv.v[e]);
    auto add = [&](const U16x8& hv, const U16x8& rv) {
      U16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = T::to_f32(hv.v[e]), r = T::to_f32(rv.v[e]);
        if (hb) a = T::to_f32(T::from_f32(a + b1[e]));
        if (rb) r = T::to_f32(T::from_f32(r + b2[e]));
        o.v[e] = T::from_f32(r + a);
      }
      return o;
    };
    const int64_t step = (int64_t)R * C;
    int64_t row = r0 + my_r;
    for (; row + (GNL_UNROLL - 1) * R < r1; row += GNL_UNROLL * R) {
      const int64_t off = row * C + c0;
      U16x8 hv[GNL_UNROLL], rv[GNL_UNROLL];
#pragma unroll
      for (int u = 0; u < GNL_UNROLL; ++u) {
        hv[u] = *reinterpret_cast<const U16x8*>(h + off + u * step);
        rv[u] = *reinterpret_cast<const U16x8*>(res + off + u * step);
      }
#pragma unroll
      for (int u = 0; u < GNL_UNROLL; ++u) *reinterpret_cast<U16x8*>(out + off + u * step) = add(hv[u], rv[u]);
    }
    for (; row < r1; row += R) {
      const int64_t off = row * C + c0;
      *reinterpret_cast<U16x8*>(out + off) = add(*reinterpret_cast<const U16x8*>(h + off), *reinterpret_cast<const U16x8*>(res + off));
    }
  }
}

// ---- GroupNorm (+SiLU), fp32, NCHW: the VAE's normalisation layers ----------------------------------------------------
// torch's fp32 GroupNorm is one 501 us `RowwiseMomentsCUDAKernel` per call on the pad-strip encodes (ONE block per (sample,
// group): 160 blocks for a [5,128,256,1024] activation on a 256-CU chip, 1.3 TB/s) + an affine kernel + a separate SiLU kernel
// (profiles/r3_s3_bench_*: 610 ms per two images).  Here: (1) partial sums over 64 K-element chunks of every group -- all CUs
// busy, four 16-byte loads in flight per thread -- and (2) an apply pass whose blocks first combine their group's partials in
// double (fixed order: deterministic), then stream y = silu((x - mean) * rstd * gamma[c] + beta[c]).
#define GN32_THREADS 256
#define GN32_CHUNK 65536  // elements per block

__global__ void __launch_bounds__(GN32_THREADS)
k_gn32_partial(const float* __restrict__ x, float* __restrict__ partial, int64_t group_len, int nchunks) {
  __shared__ double red[2 * (GN32_THREADS / 64)];
  const int64_t ng = blockIdx.y;
  const int chunk = blockIdx.x;
  const float* base = x + ng * group_len;
  const int64_t e0 = (int64_t)chunk * GN32_CHUNK, e1 = e0 + GN32_CHUNK < group_len ? e0 + GN32_CHUNK : group_len;
  float s = 0.f, q = 0.f;
  int64_t i = e0 + 4 * (int64_t)threadIdx.x;
  const int64_t step = 4 * GN32_THREADS;
  for (; i + 3 * step + 3 < e1; i += 4 * step) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + i + u * step);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      q += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
    }
  }
  for (; i + 3 < e1; i += step) {
    const float4 v = *reinterpret_cast<const float4*>(base + i);
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  double ds = (double)s, dq = (double)q;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ds += __shfl_xor(ds, off, 64);
    dq += __shfl_xor(dq, off, 64);
  }
  if ((threadIdx.x & 63) == 0) red[2 * (threadIdx.x >> 6)] = ds, red[2 * (threadIdx.x >> 6) + 1] = dq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int w = 0; w < GN32_THREADS / 64; ++w) ts += red[2 * w], tq += red[2 * w + 1];
    float* dst = partial + (ng * nchunks + chunk) * 2;
    dst[0] = (float)ts, dst[1] = (float)tq;
  }
}

template <bool ACT>
__global__ void __launch_bounds__(GN32_THREADS)
k_gn32_apply(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
             const float* __restrict__ partial, float* __restrict__ out, int64_t group_len, int HW, int cpg, int G,
             int nchunks, float eps) {
  __shared__ float st[2];
  const int64_t ng = blockIdx.y;
  const int chunk = blockIdx.x;
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    const float* p = partial + ng * nchunks * 2;
    for (int c = 0; c < nchunks; ++c) ts += (double)p[2 * c], tq += (double)p[2 * c + 1];
    const double mean = ts / (double)group_len;
    double var = tq / (double)group_len - mean * mean;
    if (var < 0.0) var = 0.0;
    st[0] = (float)mean, st[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = st[0], rstd = st[1];
  const int g = (int)(ng % G);
  const float* base = x + ng * group_len;
  float* obase = out + ng * group_len;
  const int64_t e0 = (int64_t)chunk * GN32_CHUNK, e1 = e0 + GN32_CHUNK < group_len ? e0 + GN32_CHUNK : group_len;
  const int64_t step = 4 * GN32_THREADS;
  auto norm = [&](const float4& v, int64_t e) {  // HW % 4 == 0: the four elements share one channel
    const int c = g * cpg + (int)(e / HW);
    const float a = rstd * gamma[c], b = fmaf(-a, mean, beta[c]);
    float4 o;
    o.x = fmaf(a, v.x, b), o.y = fmaf(a, v.y, b), o.z = fmaf(a, v.z, b), o.w = fmaf(a, v.w, b);
    if (ACT) o.x = silu(o.x), o.y = silu(o.y), o.z = silu(o.z), o.w = silu(o.w);
    return o;
  };
  int64_t i = e0 + 4 * (int64_t)threadIdx.x;
  for (; i + 3 * step + 3 < e1; i += 4 * step) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + i + u * step);
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(obase + i + u * step) = norm(v[u], i + u * step);
  }
  for (; i + 3 < e1; i += step) *reinterpret_cast<float4*>(obase + i) = norm(*reinterpret_cast<const float4*>(base + i), i);
}

// ---- row softmax, fp32, in place: x[r, :] = softmax(scale * x[r, :]) ---------------------------------------------------
// The VAE mid-block attention (one head of dim 512 over up to 32768 tokens, fp32 like the rest of the VAE: ED:328 keeps
// the encoder out of autocast, the decoder runs after it, ED:1080-1121) as  S = Q K^T (fp32 GEMM) -> this kernel -> S V
// (fp32 GEMM), replacing F.scaled_dot_product_attention, whose ROCm backend is AOTriton's `attn_fwd` -- the last Triton
// kernel on the path (profiles/r2_final_bench_*: 0.8 % of GPU time).  One 256-thread block per row; the row (<= 128 KiB)
// is read twice -- online max / sum in one pass, then normalise -- and written once; the second read hits L2.
template <int NT>
__global__ void __launch_bounds__(NT)
k_softmax_rows_f32(float* __restrict__ x, int64_t cols, float scale_log2e) {
  __shared__ float red_m[NT / 64], red_s[NT / 64];
  float* row = x + (int64_t)blockIdx.x * cols;
  const int tid = threadIdx.x;
  float m = -INFINITY, sum = 0.f;
  const int64_t nv = cols >> 2;
  for (int64_t i = tid; i < nv; i += NT) {
    const float4 v = *reinterpret_cast<const float4*>(row + 4 * i);
    const float a = v.x * scale_log2e, b = v.y * scale_log2e, c = v.z * scale_log2e, d = v.w * scale_log2e;
    const float mx = fmaxf(fmaxf(a, b), fmaxf(c, d));
    if (mx > m) {
      sum *= __builtin_amdgcn_exp2f(m - mx);
      m = mx;
    }
    sum += __builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m) + __builtin_amdgcn_exp2f(d - m);
  }
  for (int64_t i = 4 * nv + tid; i < cols; i += NT) {
    const float a = row[i] * scale_log2e;
    if (a > m) {
      sum *= __builtin_amdgcn_exp2f(m - a);
      m = a;
    }
    sum += __builtin_amdgcn_exp2f(a - m);
  }
  // wave reduction of (m, sum) pairs, then across the block's waves
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(m, off, 64), os = __shfl_xor(sum, off, 64);
    const float nm = fmaxf(m, om);
    sum = (m == -INFINITY ? 0.f : sum * __builtin_amdgcn_exp2f(m - nm)) + (om == -INFINITY ? 0.f : os * __builtin_amdgcn_exp2f(om - nm));
    m = nm;
  }
  if ((tid & 63) == 0) red_m[tid >> 6] = m, red_s[tid >> 6] = sum;
  __syncthreads();
  float gm = red_m[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) gm = fmaxf(gm, red_m[w]);
  float gs = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) gs += red_m[w] == -INFINITY ? 0.f : red_s[w] * __builtin_amdgcn_exp2f(red_m[w] - gm);
  const float inv = 1.0f / gs;
  for (int64_t i = tid; i < nv; i += NT) {
    float4 v = *reinterpret_cast<const float4*>(row + 4 * i);
    v.x = __builtin_amdgcn_exp2f(v.x * scale_log2e - gm) * inv;
    v.y = __builtin_amdgcn_exp2f(v.y * scale_log2e - gm) * inv;
    v.z = __builtin_amdgcn_exp2f(v.z * scale_log2e - gm) * inv;
    v.w = __builtin_amdgcn_exp2f(v.w * scale_log2e - gm) * inv;
    *reinterpret_cast<float4*>(row + 4 * i) = v;
  }
  for (int64_t i = 4 * nv + tid; i < cols; i += NT) row[i] = __builtin_amdgcn_exp2f(row[i] * scale_log2e - gm) * inv;
}

inline int done() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int ed_geglu(const void* in, void* out, int dtype, int64_t M, int I, void* stream) {
  if (M == 0 || I == 0) return 0;
  if (I % 8 != 0 || (((uintptr_t)in | (uintptr_t)out) & 15u)) return (int)hipErrorInvalidValue;
  int64_t total = M * (I / 8);
  int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == ED_BF16)
    k_geglu<BF16><<<grid, 256, 0, (hipStream_t)stream>>>((const uint16_t*)in, (uint16_t*)out, M, I);
  else if (dtype == ED_F16)
    k_geglu<F16><<<grid, 256, 0, (hipStream_t)stream>>>((const uint16_t*)in, (uint16_t*)out, M, I);
  else
    return (int)hipErrorInvalidValue;
  return done();
}

// groups of more than GN_SPLIT_MIN elements are cut into chunks of ~GN_SPLIT_SPAN elements (see k_gn_split_*)
#define GN_SPLIT_MIN 65536
#define GN_SPLIT_SPAN 32768
static inline void gn_split_plan(int C, int HW, int G, int* chunks, int* span) {
  int64_t len = (int64_t)(C / G) * HW;
  int want = (int)((len + GN_SPLIT_SPAN - 1) / GN_SPLIT_SPAN);
  int sp = ((HW + want - 1) / want + 7) & ~7;  // tokens per chunk, multiple of 8
  *span = sp;
  *chunks = (HW + sp - 1) / sp;
}

int64_t ed_groupnorm_workspace(int N, int C, int HW, int G) {
  if (G <= 0 || C % G != 0 || (int64_t)(C / G) * HW <= GN_SPLIT_MIN) return 0;
  int chunks, span;
  gn_split_plan(C, HW, G, &chunks, &span);
  return (int64_t)N * G * chunks * 3 * (int64_t)sizeof(float);
}

int ed_groupnorm(const void* x, const void* gamma, const void* beta, const void* conv_bias, const void* chan_bias,
                 void* out, float* workspace, int dtype, int N, int C, int HW, int G, float eps, int act_silu,
                 int tokens_out, void* stream) {
  if (N == 0) return 0;
  if (C % G != 0 || HW % 8 != 0 || (((uintptr_t)x | (uintptr_t)out) & 15u)) return (int)hipErrorInvalidValue;
  if (tokens_out && (C / G) % 4 != 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  if (workspace && (int64_t)(C / G) * HW > GN_SPLIT_MIN) {
    int chunks, span;
    gn_split_plan(C, HW, G, &chunks, &span);
    if ((int64_t)N * G > 65535) return (int)hipErrorInvalidValue;
    dim3 sgrid(chunks, N * G), sblock(GNS_THREADS);
#define GNS_APPLY(T, A, K)                                                                                              \
  k_gn_split_apply<T, A, K><<<sgrid, sblock, 0, s>>>((const uint16_t*)x, (const uint16_t*)gamma, (const uint16_t*)beta, \
                                                     (const uint16_t*)conv_bias, (const uint16_t*)chan_bias, workspace,  \
                                                     (uint16_t*)out, C, HW, G, chunks, span, eps)
#define GNS_RUN(T)                                                                                                      \
  k_gn_split_stats<T><<<sgrid, sblock, 0, s>>>((const uint16_t*)x, (const uint16_t*)conv_bias,                          \
                                               (const uint16_t*)chan_bias, workspace, C, HW, G, chunks, span);           \
  if (act_silu && !tokens_out) GNS_APPLY(T, true, false);                                                               \
  else if (act_silu) GNS_APPLY(T, true, true);                                                                          \
  else if (!tokens_out) GNS_APPLY(T, false, false);                                                                     \
  else GNS_APPLY(T, false, true);
    if (dtype == ED_BF16) {
      GNS_RUN(BF16)
    } else if (dtype == ED_F16) {
      GNS_RUN(F16)
    } else {
      return (int)hipErrorInvalidValue;
    }
#undef GNS_RUN
#undef GNS_APPLY
    return done();
  }
  dim3 grid(N * G), block(GN_THREADS);
#define GN_LAUNCH(T, A, K) \
  k_groupnorm<T, A, K><<<grid, block, 0, s>>>((const uint16_t*)x, (const uint16_t*)gamma, (const uint16_t*)beta, \
                                              (const uint16_t*)conv_bias, (const uint16_t*)chan_bias, (uint16_t*)out, C, \
                                              HW, G, eps)
#define GN_DISPATCH(T)                         \
  if (act_silu && !tokens_out) GN_LAUNCH(T, true, false);  \
  else if (act_silu) GN_LAUNCH(T, true, true);  \
  else if (!tokens_out) GN_LAUNCH(T, false, false); \
  else GN_LAUNCH(T, false, true);
  if (dtype == ED_BF16) {
    GN_DISPATCH(BF16)
  } else if (dtype == ED_F16) {
    GN_DISPATCH(F16)
  } else {
    return (int)hipErrorInvalidValue;
  }
#undef GN_DISPATCH
#undef GN_LAUNCH
  return done();
}

static int gn_nhwc_launch(const void* x, const void* x2, int C1, const void* gamma, const void* beta, const void* conv_bias, const void* chan_bias,
                          void* out, float* workspace, int dtype, int N, int C, int HW, int G, float eps, int act_silu, bool x32,
                          void* stream) {
  if (N == 0) return 0;
  // C/G >= 8: an 8-channel vector then touches at most two groups (what the partial-sum kernel bins into)
  if (C % G != 0 || C % 8 != 0 || C / G < 8 || G > 256 || C > 8 * GNL_THREADS * GNL_MAXCOL || (x32 && (conv_bias || chan_bias)) ||
      (((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15u))
    return (int)hipErrorInvalidValue;
  if (x2 ? (C1 <= 0 || C1 >= C || C1 % 8 != 0 || ((uintptr_t)x2 & 15u) || conv_bias || chan_bias) : C1 != C) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const GnPlan pl = gn_plan(N, C, HW);
  float* partial = workspace;                                // [N, nchunks, G, 2]
  float* stats = workspace + (int64_t)N * pl.nchunks * G * 2;  // [N, G, 2]: (mean, rstd), written by the finalize launch
  dim3 grid1(pl.nchunks, N);
  const int VC_ = C / 8;
  size_t lds = sizeof(float) * 4 * (size_t)VC_ * (VC_ <= pl.threads ? pl.threads / VC_ : 1);
  const int rows_per_block = pl.rows_per_block;
#define GNL_RUN2(T, X32)                                                                                               \
  k_gn_nhwc_partial<T, X32><<<grid1, pl.threads, lds, s>>>(x, x2, C1, (const uint16_t*)conv_bias,                              \
                                                           (const uint16_t*)chan_bias, partial, C, HW, G, rows_per_block); \
  k_gn_nhwc_finalize<<<N, GNL_THREADS, 0, s>>>(partial, stats, G, pl.nchunks, (double)HW * (C / G), eps);             \
  if (act_silu)                                                                                                        \
    k_gn_nhwc_apply<T, true, X32><<<grid1, pl.threads, 0, s>>>(x, x2, C1, (const uint16_t*)gamma,                              \
                                                         (const uint16_t*)beta, (const uint16_t*)conv_bias,            \
                                                         (const uint16_t*)chan_bias, stats, (uint16_t*)out, C, HW,     \
                                                         G, rows_per_block);                                           \
  else                                                                                                                 \
    k_gn_nhwc_apply<T, false, X32><<<grid1, pl.threads, 0, s>>>(x, x2, C1, (const uint16_t*)gamma,                             \
                                                          (const uint16_t*)beta, (const uint16_t*)conv_bias,           \
                                                          (const uint16_t*)chan_bias, stats, (uint16_t*)out, C, HW,    \
                                                          G, rows_per_block);
#define GNL_RUN(T)        \
  if (x32) {              \
    GNL_RUN2(T, true)     \
  } else {                \
    GNL_RUN2(T, false)    \
  }
  if (dtype == ED_BF16) {
    GNL_RUN(BF16)
  } else if (dtype == ED_F16) {
    GNL_RUN(F16)
  } else {
    return (int)hipErrorInvalidValue;
  }
#undef GNL_RUN
#undef GNL_RUN2
  return done();
}

int ed_groupnorm_nhwc(const void* x, const void* gamma, const void* beta, const void* conv_bias, const void* chan_bias,
                      void* out, float* workspace, int dtype, int N, int C, int HW, int G, float eps, int act_silu,
                      void* stream) {
  return gn_nhwc_launch(x, nullptr, C, gamma, beta, conv_bias, chan_bias, out, workspace, dtype, N, C, HW, G, eps, act_silu, false, stream);
}

int ed_groupnorm_nhwc_cat(const void* x1, const void* x2, const void* gamma, const void* beta, void* out, float* workspace, int dtype,
                          int N, int C1, int C2, int HW, int G, float eps, int act_silu, void* stream) {
  if (!x1 || !x2 || C1 <= 0 || C2 <= 0) return (int)hipErrorInvalidValue;
  return gn_nhwc_launch(x1, x2, C1, gamma, beta, nullptr, nullptr, out, workspace, dtype, N, C1 + C2, HW, G, eps, act_silu, false, stream);
}

int ed_groupnorm_nhwc_s32(const void* x, const void* gamma, const void* beta, void* out, float* workspace, int dtype, int N, int C,
                          int HW, int G, float eps, int act_silu, void* stream) {
  return gn_nhwc_launch(x, nullptr, C, gamma, beta, nullptr, nullptr, out, workspace, dtype, N, C, HW, G, eps, act_silu, true, stream);
}

static int layernorm_launch(const void* x, const void* gamma, const void* beta, void* out, int dtype, int64_t M, int D, float eps,
                            bool x32, void* stream) {
  if (M == 0) return 0;
  if (D % 8 != 0 || D > 64 * 8 * LN_MAX_IT ||
      (((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15u))
    return (int)hipErrorInvalidValue;
  int64_t blocks = (M + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
  if (blocks > 0x7fffffff) return (int)hipErrorInvalidValue;
  dim3 grid((unsigned)blocks), block(64 * LN_ROWS_PER_BLOCK);
  hipStream_t s = (hipStream_t)stream;
#define LN_RUN(T, X32) k_layernorm<T, X32><<<grid, block, 0, s>>>(x, (const uint16_t*)gamma, (const uint16_t*)beta, (uint16_t*)out, M, D, eps)
  if (dtype == ED_BF16) {
    if (x32) LN_RUN(BF16, true);
    else LN_RUN(BF16, false);
  } else if (dtype == ED_F16) {
    if (x32) LN_RUN(F16, true);
    else LN_RUN(F16, false);
  } else {
    return (int)hipErrorInvalidValue;
  }
#undef LN_RUN
  return done();
}

int ed_layernorm(const void* x, const void* gamma, const void* beta, void* out, int dtype, int64_t M, int D, float eps,
                 void* stream) {
  return layernorm_launch(x, gamma, beta, out, dtype, M, D, eps, false, stream);
}

int ed_layernorm_s32(const void* x, const void* gamma, const void* beta, void* out, int dtype, int64_t M, int D, float eps,
                     void* stream) {
  return layernorm_launch(x, gamma, beta, out, dtype, M, D, eps, true, stream);
}

static int add_layernorm_launch(const void* a, const void* b, const void* gamma, const void* beta, void* sum_out, void* out,
                                int dtype, int64_t M, int D, float eps, bool s32, void* stream) {
  if (M == 0) return 0;
  if (D % 8 != 0 || D > 64 * 8 * LN_MAX_IT ||
      (((uintptr_t)a | (uintptr_t)b | (uintptr_t)sum_out | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15u))
    return (int)hipErrorInvalidValue;
  int64_t blocks = (M + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
  if (blocks > 0x7fffffff) return (int)hipErrorInvalidValue;
  dim3 grid((unsigned)blocks), block(64 * LN_ROWS_PER_BLOCK);
  hipStream_t s = (hipStream_t)stream;
#define ALN_RUN(T, S32)                                                                                                        \
  k_add_layernorm<T, S32><<<grid, block, 0, s>>>((const uint16_t*)a, b, (const uint16_t*)gamma, (const uint16_t*)beta, sum_out, \
                                                 (uint16_t*)out, M, D, eps)
  if (dtype == ED_BF16) {
    if (s32) ALN_RUN(BF16, true);
    else ALN_RUN(BF16, false);
  } else if (dtype == ED_F16) {
    if (s32) ALN_RUN(F16, true);
    else ALN_RUN(F16, false);
  } else {
    return (int)hipErrorInvalidValue;
  }
#undef ALN_RUN
  return done();
}

int ed_add_layernorm(const void* a, const void* b, const void* gamma, const void* beta, void* sum_out, void* out,
                     int dtype, int64_t M, int D, float eps, void* stream) {
  return add_layernorm_launch(a, b, gamma, beta, sum_out, out, dtype, M, D, eps, false, stream);
}

int ed_add_layernorm_s32(const void* a, const void* b, const void* gamma, const void* beta, void* sum_out, void* out,
                         int dtype, int64_t M, int D, float eps, void* stream) {
  return add_layernorm_launch(a, b, gamma, beta, sum_out, out, dtype, M, D, eps, true, stream);
}

int ed_bias_residual_add(const void* h, const void* h_bias, const void* res, const void* res_bias, void* out, int dtype,
                         int N, int C, int HW, int channels_last, void* stream) {
  if (N == 0) return 0;
  if ((channels_last ? C % 8 : HW % 8) != 0 || (((uintptr_t)h | (uintptr_t)res | (uintptr_t)out) & 15u))
    return (int)hipErrorInvalidValue;
  int64_t total_vec = (int64_t)N * C * HW / 8;
  if (total_vec >= 0x7fffffff) return (int)hipErrorInvalidValue;
  int grid = (int)((total_vec + 255) / 256 < 16384 ? (total_vec + 255) / 256 : 16384);
  hipStream_t s = (hipStream_t)stream;
  if (channels_last && C <= 8 * GNL_THREADS * GNL_MAXCOL && (dtype == ED_BF16 || dtype == ED_F16) &&
      !(((uintptr_t)h_bias | (uintptr_t)res_bias) & 15u)) {
    const int64_t rows = (int64_t)N * HW;
    int rows_per_block = (65536 + C - 1) / C;  // ~64 K elements per block, as the GroupNorm kernels
    const int64_t nblk = (rows + rows_per_block - 1) / rows_per_block;
    if (nblk <= 0x7fffffff) {
      if (dtype == ED_BF16)
        k_bias_residual_add_cl<BF16><<<(unsigned)nblk, GNL_THREADS, 0, s>>>((const uint16_t*)h, (const uint16_t*)h_bias, (const uint16_t*)res,
                                                                         (const uint16_t*)res_bias, (uint16_t*)out, C, rows, rows_per_block);
      else
        k_bias_residual_add_cl<F16><<<(unsigned)nblk, GNL_THREADS, 0, s>>>((const uint16_t*)h, (const uint16_t*)h_bias, (const uint16_t*)res,
                                                                        (const uint16_t*)res_bias, (uint16_t*)out, C, rows, rows_per_block);
      return done();
    }
  }
  if (dtype == ED_BF16)
    k_bias_residual_add<BF16><<<grid, 256, 0, s>>>((const uint16_t*)h, (const uint16_t*)h_bias, (const uint16_t*)res,
                                                   (const uint16_t*)res_bias, (uint16_t*)out, C, HW, total_vec,
                                                   channels_last);
  else if (dtype == ED_F16)
    k_bias_residual_add<F16><<<grid, 256, 0, s>>>((const uint16_t*)h, (const uint16_t*)h_bias, (const uint16_t*)res,
                                                  (const uint16_t*)res_bias, (uint16_t*)out, C, HW, total_vec,
                                                  channels_last);
  else
    return (int)hipErrorInvalidValue;
  return done();
}

int ed_tokens_add_nchw(const void* x, const void* tokens, void* out, int dtype, int N, int C, int HW, void* stream) {
  if (N == 0) return 0;
  if (C % TA_TILE != 0 || HW % TA_TILE != 0 || N > 65535 || C / TA_TILE > 65535 ||
      (((uintptr_t)x | (uintptr_t)tokens | (uintptr_t)out) & 15u))
    return (int)hipErrorInvalidValue;
  dim3 grid(HW / TA_TILE, C / TA_TILE, N), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ED_BF16)
    k_tokens_add_nchw<BF16><<<grid, block, 0, s>>>((const uint16_t*)x, (const uint16_t*)tokens, (uint16_t*)out, C, HW);
  else if (dtype == ED_F16)
    k_tokens_add_nchw<F16><<<grid, block, 0, s>>>((const uint16_t*)x, (const uint16_t*)tokens, (uint16_t*)out, C, HW);
  else
    return (int)hipErrorInvalidValue;
  return done();
}

int64_t ed_groupnorm_f32_workspace(int N, int C, int HW, int G) {
  if (G <= 0 || C % G) return 0;
  const int64_t group_len = (int64_t)(C / G) * HW;
  const int64_t nchunks = (group_len + GN32_CHUNK - 1) / GN32_CHUNK;
  return (int64_t)N * G * nchunks * 2 * (int64_t)sizeof(float);
}

int ed_groupnorm_f32(const void* x, const void* gamma, const void* beta, void* out, float* workspace, int N, int C, int HW,
                     int G, float eps, int act_silu, void* stream) {
  if (N == 0) return 0;
  if (G <= 0 || C % G != 0 || HW % 4 != 0 || (((uintptr_t)x | (uintptr_t)out) & 15u) || (int64_t)N * G > 65535)
    return (int)hipErrorInvalidValue;  // (sample, group) is the grid's y dimension
  const int cpg = C / G;
  const int64_t group_len = (int64_t)cpg * HW;
  const int64_t nchunks = (group_len + GN32_CHUNK - 1) / GN32_CHUNK;
  if (nchunks > 0x7fffffffll) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)nchunks, (unsigned)((int64_t)N * G));
  k_gn32_partial<<<grid, GN32_THREADS, 0, s>>>((const float*)x, workspace, group_len, (int)nchunks);
  if (act_silu)
    k_gn32_apply<true><<<grid, GN32_THREADS, 0, s>>>((const float*)x, (const float*)gamma, (const float*)beta, workspace,
                                                    (float*)out, group_len, HW, cpg, G, (int)nchunks, eps);
  else
    k_gn32_apply<false><<<grid, GN32_THREADS, 0, s>>>((const float*)x, (const float*)gamma, (const float*)beta, workspace,
                                                     (float*)out, group_len, HW, cpg, G, (int)nchunks, eps);
  return done();
}

int ed_softmax_rows(void* x, int64_t rows, int64_t cols, float scale, void* stream) {
  if (rows == 0 || cols == 0) return 0;
  if (rows < 0 || cols < 0 || rows > 0x7fffffffll || ((uintptr_t)x & 15u) || (cols % 4 != 0 && rows > 1))
    return (int)hipErrorInvalidValue;  // rows start 16-byte aligned when cols % 4 == 0
  k_softmax_rows_f32<256><<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>((float*)x, cols, scale * 1.44269504088896340736f);
  return done();
}

int64_t ed_groupnorm_nhwc_workspace(int N, int C, int HW, int G) {
  if (N <= 0 || C <= 0 || HW <= 0 || G <= 0) return 0;
  const GnPlan pl = gn_plan(N, C, HW);
  return ((int64_t)N * pl.nchunks * G * 2 + (int64_t)N * G * 2) * (int64_t)sizeof(float);
}

}  // extern "C"
