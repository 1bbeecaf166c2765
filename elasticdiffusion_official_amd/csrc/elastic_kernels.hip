// elastic_kernels.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for the ElasticDiffusion hot path.
//
// Every kernel here is HBM/latency bound latent-space glue (gather / scatter / select / a few fp32 flops per
// element); tensors are 64 KiB - 12 MiB, so the design rules are: one launch per logical phase, x-fastest
// coalesced addressing (64 lanes x 4 B or 16 B contiguous), 16-byte vector access on the pure streaming kernels,
// no atomics, no read-modify-write, no host sync.  MFMA is deliberately not used: nothing here is a contraction.
//
// fp32 arithmetic is written with explicit __fmul_rn/__fadd_rn/__fsub_rn/__fdiv_rn and the file is compiled with
// -ffp-contract=off so no FMA is formed: results are bit-identical to the reference's torch-CPU op sequence.
//
// Interface documentation (and the reference file:line each entry replaces) lives in include/elastic_hip.h.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "elastic_hip.h"

#define ED_ABI_VERSION 9
#define ED_BLOCK 256

namespace {

// ---- element type adapters (model-boundary tensors may be f32 / f16 / bf16) -----------------------
struct F32 { using type = float; };
struct F16 { using type = __half; };
struct BF16 { using type = uint16_t; };

template <typename Tag> __device__ __forceinline__ float ld(const void* p, int64_t i);
template <> __device__ __forceinline__ float ld<F32>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float ld<F16>(const void* p, int64_t i) { return __half2float(((const __half*)p)[i]); }
template <> __device__ __forceinline__ float ld<BF16>(const void* p, int64_t i) {
  return __uint_as_float(((uint32_t)((const uint16_t*)p)[i]) << 16);
}

template <typename Tag> __device__ __forceinline__ void st(void* p, int64_t i, float v);
template <> __device__ __forceinline__ void st<F32>(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void st<F16>(void* p, int64_t i, float v) { ((__half*)p)[i] = __float2half_rn(v); }
template <> __device__ __forceinline__ void st<BF16>(void* p, int64_t i, float v) {
  uint32_t u = __float_as_uint(v);
  uint16_t r;
  if ((u & 0x7fffffffu) > 0x7f800000u) {
    r = 0x7fc0;  // NaN
  } else {
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even, like torch
    r = (uint16_t)(u >> 16);
  }
  ((uint16_t*)p)[i] = r;
}

// 4 consecutive elements: 16-byte (f32) or 8-byte (16-bit) vector access; callers guarantee alignment
template <typename Tag> __device__ __forceinline__ void st4(void* p, int64_t i, float a, float b, float c, float d);
template <> __device__ __forceinline__ void st4<F32>(void* p, int64_t i, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>((float*)p + i) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void st4<F16>(void* p, int64_t i, float a, float b, float c, float d) {
  uint2 pk;
  pk.x = (uint32_t)__half_as_ushort(__float2half_rn(a)) | ((uint32_t)__half_as_ushort(__float2half_rn(b)) << 16);
  pk.y = (uint32_t)__half_as_ushort(__float2half_rn(c)) | ((uint32_t)__half_as_ushort(__float2half_rn(d)) << 16);
  *reinterpret_cast<uint2*>((uint16_t*)p + i) = pk;
}
__device__ __forceinline__ uint32_t bf16_bits(float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
template <> __device__ __forceinline__ void st4<BF16>(void* p, int64_t i, float a, float b, float c, float d) {
  uint2 pk;
  pk.x = bf16_bits(a) | (bf16_bits(b) << 16);
  pk.y = bf16_bits(c) | (bf16_bits(d) << 16);
  *reinterpret_cast<uint2*>((uint16_t*)p + i) = pk;
}
template <typename Tag> __device__ __forceinline__ float4 ld4(const void* p, int64_t i);
template <> __device__ __forceinline__ float4 ld4<F32>(const void* p, int64_t i) {
  return *reinterpret_cast<const float4*>((const float*)p + i);
}
template <> __device__ __forceinline__ float4 ld4<F16>(const void* p, int64_t i) {
  uint2 pk = *reinterpret_cast<const uint2*>((const uint16_t*)p + i);
  return make_float4(__half2float(__ushort_as_half((uint16_t)(pk.x & 0xffffu))), __half2float(__ushort_as_half((uint16_t)(pk.x >> 16))),
                     __half2float(__ushort_as_half((uint16_t)(pk.y & 0xffffu))), __half2float(__ushort_as_half((uint16_t)(pk.y >> 16))));
}
template <> __device__ __forceinline__ float4 ld4<BF16>(const void* p, int64_t i) {
  uint2 pk = *reinterpret_cast<const uint2*>((const uint16_t*)p + i);
  return make_float4(__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                     __uint_as_float(pk.y & 0xffff0000u));
}

inline int grid_for(int64_t n, int per_thread = 1) {
  int64_t t = (n + per_thread - 1) / per_thread;
  int64_t g = (t + ED_BLOCK - 1) / ED_BLOCK;
  return (int)(g < 1 ? 1 : g);
}

inline int done() { return (int)hipGetLastError(); }

// ---- ed_gather_views / ed_tile_gather_pad ----------------------------------------------------------
template <typename Tag>
__device__ __forceinline__ void gather_windows_body(int64_t t, const float* __restrict__ latent, void* __restrict__ out, int B, int C, int H, int W,
                 const int32_t* __restrict__ wy0, const int32_t* __restrict__ wx0, int V, int Sh, int Sw,
                 int PH, int PW, int off_y, int off_x, const float* __restrict__ frame, float divisor, int use_div) {
  int64_t n = (int64_t)V * B * C * PH * PW;
  if (t >= n) return;
  int x = (int)(t % PW);
  int64_t r = t / PW;
  int y = (int)(r % PH);
  r /= PH;
  int c = (int)(r % C);
  r /= C;  // row = v*B + b
  int b = (int)(r % B);
  int v = (int)(r / B);
  int yy = y - off_y, xx = x - off_x;
  float val;
  if (yy >= 0 && yy < Sh && xx >= 0 && xx < Sw) {
    int sy = wy0[v] + yy, sx = wx0[v] + xx;
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
      val = latent[(((int64_t)b * C + c) * H + sy) * W + sx];
      if (use_div) val = __fdiv_rn(val, divisor);
    } else {
      val = 0.0f;
    }
  } else {
    val = frame ? frame[((int64_t)c * PH + y) * PW + x] : 0.0f;
  }
  st<Tag>(out, t, val);
}

template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_gather_windows(const float* __restrict__ latent, void* __restrict__ out, int B, int C, int H, int W,
                 const int32_t* __restrict__ wy0, const int32_t* __restrict__ wx0, int V, int Sh, int Sw,
                 int PH, int PW, int off_y, int off_x, const float* __restrict__ frame, float divisor, int use_div) {
  gather_windows_body<Tag>((int64_t)blockIdx.x * ED_BLOCK + threadIdx.x, latent, out, B, C, H, W, wy0, wx0, V, Sh, Sw, PH, PW, off_y, off_x, frame, divisor, use_div);
}

// 4 consecutive x per thread; requires PW, Sw, off_x multiples of 4 (a group is entirely inside or outside the window)
template <typename Tag>
__device__ __forceinline__ void gather_windows_x4_body(int64_t t, const float* __restrict__ latent, void* __restrict__ out, int B, int C, int H, int W,
                    const int32_t* __restrict__ wy0, const int32_t* __restrict__ wx0, int V, int Sh, int Sw,
                    int PH, int PW, int off_y, int off_x, const float* __restrict__ frame, float divisor, int use_div) {
  int PW4 = PW >> 2;
  int64_t n = (int64_t)V * B * C * PH * PW4;
  if (t >= n) return;
  int x = (int)(t % PW4) << 2;
  int64_t r = t / PW4;
  int y = (int)(r % PH);
  r /= PH;
  int c = (int)(r % C);
  r /= C;
  int b = (int)(r % B);
  int v = (int)(r / B);
  int yy = y - off_y, xx = x - off_x;
  float val[4];
  if (yy >= 0 && yy < Sh && xx >= 0 && xx < Sw) {
    int sy = wy0[v] + yy, sx0 = wx0[v] + xx;
    const float* src = latent + (((int64_t)b * C + c) * H + sy) * W;
    bool row_ok = sy >= 0 && sy < H;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int sx = sx0 + e;
      float f = (row_ok && sx >= 0 && sx < W) ? src[sx] : 0.0f;
      val[e] = (use_div && row_ok && sx >= 0 && sx < W) ? __fdiv_rn(f, divisor) : f;
    }
  } else if (frame) {
    float4 f = *reinterpret_cast<const float4*>(frame + ((int64_t)c * PH + y) * PW + x);
    val[0] = f.x; val[1] = f.y; val[2] = f.z; val[3] = f.w;
  } else {
    val[0] = val[1] = val[2] = val[3] = 0.0f;
  }
  st4<Tag>(out, t << 2, val[0], val[1], val[2], val[3]);
}

template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_gather_windows_x4(const float* __restrict__ latent, void* __restrict__ out, int B, int C, int H, int W,
                    const int32_t* __restrict__ wy0, const int32_t* __restrict__ wx0, int V, int Sh, int Sw,
                    int PH, int PW, int off_y, int off_x, const float* __restrict__ frame, float divisor, int use_div) {
  gather_windows_x4_body<Tag>((int64_t)blockIdx.x * ED_BLOCK + threadIdx.x, latent, out, B, C, H, W, wy0, wx0, V, Sh, Sw, PH, PW, off_y, off_x, frame, divisor, use_div);
}

// ---- ed_scatter_centres ----------------------------------------------------------------------------
template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_scatter_centres(const void* __restrict__ pred, float* __restrict__ local, int B, int C, int H, int W,
                  int PH, int PW, int ncb, const int32_t* __restrict__ row_blk, const int32_t* __restrict__ row_src,
                  const int32_t* __restrict__ col_blk, const int32_t* __restrict__ col_src) {
  int64_t n = (int64_t)B * C * H * W;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  int X = (int)(t % W);
  int64_t r = t / W;
  int Y = (int)(r % H);
  r /= H;
  int c = (int)(r % C);
  int b = (int)(r / C);
  float cur = 0.0f;
  bool settled = false;
#pragma unroll
  for (int kr = 0; kr < 2; ++kr) {
    int rb = row_blk[Y * 2 + kr];
    if (rb < 0 || settled) continue;
    int sy = row_src[Y * 2 + kr];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      int cb = col_blk[X * 2 + kc];
      if (cb < 0 || settled) continue;
      int sx = col_src[X * 2 + kc];
      int64_t row = (int64_t)(rb * ncb + cb) * B + b;
      cur = ld<Tag>(pred, ((row * C + c) * PH + sy) * PW + sx);
      if (cur != 0.0f) settled = true;  // NaN != 0 is true, as in torch
    }
  }
  local[t] = cur;
}

// ---- ed_pick_assemble ------------------------------------------------------------------------------
template <typename Tag>
__device__ __forceinline__ void pick_assemble_body(int64_t t, const float* __restrict__ latent, const uint8_t* __restrict__ idx,
                const int32_t* __restrict__ src_row, const int32_t* __restrict__ src_col,
                const float* __restrict__ frame, void* __restrict__ out, float* __restrict__ low,
                int K, int B, int C, int H, int W, int h, int w, int PH, int PW, int off_y, int off_x) {
  int64_t n = (int64_t)K * B * C * PH * PW;
  if (t >= n) return;
  int x = (int)(t % PW);
  int64_t r = t / PW;
  int y = (int)(r % PH);
  r /= PH;
  int c = (int)(r % C);
  r /= C;
  int b = (int)(r % B);
  int k = (int)(r / B);
  int i = y - off_y, j = x - off_x;
  float val;
  if (i >= 0 && i < h && j >= 0 && j < w) {
    int q = idx[(int64_t)k * h * w + (int64_t)i * w + j];
    int sy = src_row[2 * i + (q >> 1)];
    int sx = src_col[2 * j + (q & 1)];
    val = latent[(((int64_t)b * C + c) * H + sy) * W + sx];
    if (low) low[((((int64_t)k * B + b) * C + c) * h + i) * w + j] = val;
  } else {
    val = frame ? frame[((int64_t)c * PH + y) * PW + x] : 0.0f;
  }
  int64_t plane = (int64_t)PH * PW;
  int64_t e = ((int64_t)c * PH + y) * PW + x;
  int64_t row_u = ((int64_t)k * 2 + 0) * B + b;
  int64_t row_c = ((int64_t)k * 2 + 1) * B + b;
  st<Tag>(out, row_u * C * plane + e, val);
  st<Tag>(out, row_c * C * plane + e, val);
}

template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_pick_assemble(const float* __restrict__ latent, const uint8_t* __restrict__ idx,
                const int32_t* __restrict__ src_row, const int32_t* __restrict__ src_col,
                const float* __restrict__ frame, void* __restrict__ out, float* __restrict__ low,
                int K, int B, int C, int H, int W, int h, int w, int PH, int PW, int off_y, int off_x) {
  pick_assemble_body<Tag>((int64_t)blockIdx.x * ED_BLOCK + threadIdx.x, latent, idx, src_row, src_col, frame, out, low, K, B, C, H, W, h, w, PH, PW, off_y, off_x);
}

// 4 consecutive x per thread; requires PW, w, off_x multiples of 4
template <typename Tag>
__device__ __forceinline__ void pick_assemble_x4_body(int64_t t, const float* __restrict__ latent, const uint8_t* __restrict__ idx,
                   const int32_t* __restrict__ src_row, const int32_t* __restrict__ src_col,
                   const float* __restrict__ frame, void* __restrict__ out, float* __restrict__ low,
                   int K, int B, int C, int H, int W, int h, int w, int PH, int PW, int off_y, int off_x) {
  int PW4 = PW >> 2;
  int64_t n = (int64_t)K * B * C * PH * PW4;
  if (t >= n) return;
  int x = (int)(t % PW4) << 2;
  int64_t r = t / PW4;
  int y = (int)(r % PH);
  r /= PH;
  int c = (int)(r % C);
  r /= C;
  int b = (int)(r % B);
  int k = (int)(r / B);
  int i = y - off_y, j = x - off_x;
  float val[4];
  if (i >= 0 && i < h && j >= 0 && j < w) {
    uint32_t q4 = *reinterpret_cast<const uint32_t*>(idx + (int64_t)k * h * w + (int64_t)i * w + j);
    const float* plane = latent + ((int64_t)b * C + c) * H * W;
    int r0 = src_row[2 * i], r1 = src_row[2 * i + 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int q = (q4 >> (8 * e)) & 0xff;
      int sy = (q >> 1) ? r1 : r0;
      int sx = src_col[2 * (j + e) + (q & 1)];
      val[e] = plane[(int64_t)sy * W + sx];
    }
    if (low)
      *reinterpret_cast<float4*>(low + ((((int64_t)k * B + b) * C + c) * h + i) * w + j) =
          make_float4(val[0], val[1], val[2], val[3]);
  } else if (frame) {
    float4 f = *reinterpret_cast<const float4*>(frame + ((int64_t)c * PH + y) * PW + x);
    val[0] = f.x; val[1] = f.y; val[2] = f.z; val[3] = f.w;
  } else {
    val[0] = val[1] = val[2] = val[3] = 0.0f;
  }
  int64_t plane_sz = (int64_t)PH * PW;
  int64_t e0 = ((int64_t)c * PH + y) * PW + x;
  int64_t row_u = ((int64_t)k * 2 + 0) * B + b;
  int64_t row_c = ((int64_t)k * 2 + 1) * B + b;
  st4<Tag>(out, row_u * C * plane_sz + e0, val[0], val[1], val[2], val[3]);
  st4<Tag>(out, row_c * C * plane_sz + e0, val[0], val[1], val[2], val[3]);
}

template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_pick_assemble_x4(const float* __restrict__ latent, const uint8_t* __restrict__ idx,
                   const int32_t* __restrict__ src_row, const int32_t* __restrict__ src_col,
                   const float* __restrict__ frame, void* __restrict__ out, float* __restrict__ low,
                   int K, int B, int C, int H, int W, int h, int w, int PH, int PW, int off_y, int off_x) {
  pick_assemble_x4_body<Tag>((int64_t)blockIdx.x * ED_BLOCK + threadIdx.x, latent, idx, src_row, src_col, frame, out, low, K, B, C, H, W, h, w, PH, PW, off_y, off_x);
}

// ---- ed_unpad_direction ----------------------------------------------------------------------------
template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_unpad_direction(const void* __restrict__ uo, float* __restrict__ dirs, float* __restrict__ uncond_last,
                  int K, int B, int C, int h, int w, int PH, int PW, int off_y, int off_x) {
  int64_t n = (int64_t)K * B * C * h * w;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  int j = (int)(t % w);
  int64_t r = t / w;
  int i = (int)(r % h);
  r /= h;
  int c = (int)(r % C);
  r /= C;
  int b = (int)(r % B);
  int k = (int)(r / B);
  int64_t plane = (int64_t)PH * PW;
  int64_t e = ((int64_t)c * PH + (i + off_y)) * PW + (j + off_x);
  float u = ld<Tag>(uo, (((int64_t)k * 2 + 0) * B + b) * C * plane + e);
  float cd = ld<Tag>(uo, (((int64_t)k * 2 + 1) * B + b) * C * plane + e);
  dirs[t] = __fsub_rn(cd, u);
  if (uncond_last && k == K - 1) uncond_last[(((int64_t)b * C + c) * h + i) * w + j] = u;
}

// 4 consecutive j per thread; requires w, PW, off_x multiples of 4
template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_unpad_direction_x4(const void* __restrict__ uo, float* __restrict__ dirs, float* __restrict__ uncond_last,
                     int K, int B, int C, int h, int w, int PH, int PW, int off_y, int off_x) {
  int w4 = w >> 2;
  int64_t n = (int64_t)K * B * C * h * w4;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  int j = (int)(t % w4) << 2;
  int64_t r = t / w4;
  int i = (int)(r % h);
  r /= h;
  int c = (int)(r % C);
  r /= C;
  int b = (int)(r % B);
  int k = (int)(r / B);
  int64_t plane = (int64_t)PH * PW;
  int64_t e = ((int64_t)c * PH + (i + off_y)) * PW + (j + off_x);
  float4 u = ld4<Tag>(uo, (((int64_t)k * 2 + 0) * B + b) * C * plane + e);
  float4 cd = ld4<Tag>(uo, (((int64_t)k * 2 + 1) * B + b) * C * plane + e);
  float4 d = make_float4(__fsub_rn(cd.x, u.x), __fsub_rn(cd.y, u.y), __fsub_rn(cd.z, u.z), __fsub_rn(cd.w, u.w));
  *reinterpret_cast<float4*>(dirs + (t << 2)) = d;
  if (uncond_last && k == K - 1)
    *reinterpret_cast<float4*>(uncond_last + (((int64_t)b * C + c) * h + i) * w + j) = u;
}

// ---- ed_fill_directions ----------------------------------------------------------------------------
// stamp[n*4 + q] = last resampling step whose pick at reduced pixel n was q (int8, -1 = never); built on the host next
// to the draws.  A full-res pixel is covered by step k iff one of its (<= 2 x 2) pick-grid cells (rr,cc) was picked.
__device__ __forceinline__ int last_covering_step(const int8_t* __restrict__ stamp, const int32_t* __restrict__ inv_row,
                                                  const int32_t* __restrict__ inv_col, int K, int w, int Y, int X) {
  int r0 = inv_row[Y * 2], r1 = inv_row[Y * 2 + 1];
  int c0 = inv_col[X * 2], c1 = inv_col[X * 2 + 1];
  int best = -1;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    int rr = a ? r1 : r0;
    if (rr < 0) continue;
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      int cc = bq ? c1 : c0;
      if (cc < 0) continue;
      int k = stamp[((int64_t)(rr >> 1) * w + (cc >> 1)) * 4 + ((rr & 1) * 2 + (cc & 1))];
      best = k > best ? k : best;
    }
  }
  return best < 0 ? K - 1 : best;  // fill_all: what no pick reached takes the last step's upsample (ED:643-644)
}

__global__ void __launch_bounds__(ED_BLOCK)
k_fill_directions(const float* __restrict__ dirs, const int8_t* __restrict__ stamp,
                  const int32_t* __restrict__ inv_row, const int32_t* __restrict__ inv_col,
                  const int32_t* __restrict__ up_row, const int32_t* __restrict__ up_col,
                  const int32_t* __restrict__ down_row, const int32_t* __restrict__ down_col,
                  float* __restrict__ target, float* __restrict__ low_dir,
                  int K, int B, int C, int H, int W, int h, int w) {
  int64_t nfull = (int64_t)H * W;
  int64_t nlow = low_dir ? (int64_t)h * w : 0;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= nfull + nlow) return;
  int Y, X;
  float* dst;
  int64_t dplane, doff;
  if (t < nfull) {
    Y = (int)(t / W);
    X = (int)(t % W);
    dst = target;
    dplane = nfull;
    doff = t;
  } else {
    int64_t u = t - nfull;
    int i = (int)(u / w), j = (int)(u % w);
    Y = down_row[i];
    X = down_col[j];
    dst = low_dir;
    dplane = (int64_t)h * w;
    doff = u;
  }
  int k = last_covering_step(stamp, inv_row, inv_col, K, w, Y, X);
  int64_t lplane = (int64_t)h * w;
  int64_t src = (int64_t)up_row[Y] * w + up_col[X];
  int BC = B * C;
  for (int bc = 0; bc < BC; ++bc)
    dst[(int64_t)bc * dplane + doff] = dirs[((int64_t)k * BC + bc) * lplane + src];
}

// ---- ed_cfg_ddim_step ------------------------------------------------------------------------------
__device__ __forceinline__ void ddim_one(float l, float d, float xv, float g, float sb, float sa, float sp, float sd,
                                         float& prev, float& x0) {
  float eps = __fadd_rn(l, __fmul_rn(g, d));
  x0 = __fdiv_rn(__fsub_rn(xv, __fmul_rn(sb, eps)), sa);
  prev = __fadd_rn(__fmul_rn(sp, x0), __fmul_rn(sd, eps));
}

__global__ void __launch_bounds__(ED_BLOCK)
k_cfg_ddim_v4(const float4* __restrict__ local, const float4* __restrict__ dir, const float4* __restrict__ x,
              float4* __restrict__ prev, float4* __restrict__ x0, float g, float sb, float sa, float sp, float sd,
              int64_t n4) {
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n4) return;
  float4 l = local[t], d = dir[t], xv = x[t], p, o;
  ddim_one(l.x, d.x, xv.x, g, sb, sa, sp, sd, p.x, o.x);
  ddim_one(l.y, d.y, xv.y, g, sb, sa, sp, sd, p.y, o.y);
  ddim_one(l.z, d.z, xv.z, g, sb, sa, sp, sd, p.z, o.z);
  ddim_one(l.w, d.w, xv.w, g, sb, sa, sp, sd, p.w, o.w);
  prev[t] = p;
  x0[t] = o;
}

__global__ void __launch_bounds__(ED_BLOCK)
k_cfg_ddim_s(const float* __restrict__ local, const float* __restrict__ dir, const float* __restrict__ x,
             float* __restrict__ prev, float* __restrict__ x0, float g, float sb, float sa, float sp, float sd,
             int64_t n) {
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  float p, o;
  ddim_one(local[t], dir[t], x[t], g, sb, sa, sp, sd, p, o);
  prev[t] = p;
  x0[t] = o;
}

// ---- ed_assemble_rows: ed_pick_assemble + ed_gather_views in ONE launch ----------------------------
// The first `pick_blocks` workgroups assemble the K CFG pairs, the rest gather the V context crops; both write rows of
// the same fused model batch.  PX4 / GX4: the 4-wide variants (host checks the alignment conditions per part).
struct AssembleArgs {
  const float* latent;
  int B, C, H, W;
  // global (pick) part
  const uint8_t* idx;
  const int32_t *src_row, *src_col;
  const float* gframe;
  void* g_out;
  float* low;
  int K, h, w, gPH, gPW, g_off_y, g_off_x;
  // view part
  const int32_t *win_y0, *win_x0;
  const float* vframe;
  void* v_out;
  int V, Sh, Sw, vPH, vPW, v_off_y, v_off_x;
  int pick_blocks;
};

template <typename Tag, bool PX4, bool GX4>
__global__ void __launch_bounds__(ED_BLOCK)
k_assemble_rows(const AssembleArgs a) {
  if ((int)blockIdx.x < a.pick_blocks) {
    int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (PX4)
      pick_assemble_x4_body<Tag>(t, a.latent, a.idx, a.src_row, a.src_col, a.gframe, a.g_out, a.low, a.K, a.B, a.C, a.H, a.W,
                                 a.h, a.w, a.gPH, a.gPW, a.g_off_y, a.g_off_x);
    else
      pick_assemble_body<Tag>(t, a.latent, a.idx, a.src_row, a.src_col, a.gframe, a.g_out, a.low, a.K, a.B, a.C, a.H, a.W,
                              a.h, a.w, a.gPH, a.gPW, a.g_off_y, a.g_off_x);
  } else {
    int64_t t = (int64_t)((int)blockIdx.x - a.pick_blocks) * ED_BLOCK + threadIdx.x;
    if (GX4)
      gather_windows_x4_body<Tag>(t, a.latent, a.v_out, a.B, a.C, a.H, a.W, a.win_y0, a.win_x0, a.V, a.Sh, a.Sw, a.vPH, a.vPW,
                                  a.v_off_y, a.v_off_x, a.vframe, 1.0f, 0);
    else
      gather_windows_body<Tag>(t, a.latent, a.v_out, a.B, a.C, a.H, a.W, a.win_y0, a.win_x0, a.V, a.Sh, a.Sw, a.vPH, a.vPW,
                               a.v_off_y, a.v_off_x, a.vframe, 1.0f, 0);
  }
}

// ---- ed_phase_epilogue: unpad + fill + scatter + CFG/DDIM (+ RRG) in ONE launch ---------------------
// Everything downstream of the model call is a per-output-pixel gather: which resampling step covers the pixel (stamp
// table) -> cond - uncond of that step's rows; the first non-zero covering view centre; the DDIM update; optionally the
// reduced-resolution-guidance term of ED:886-940 / ED:1078.  Same fp32 operation order as the separate kernels
// (__f*_rn, no contraction): results are bit-identical to ed_unpad_direction -> ed_fill_directions ->
// ed_scatter_centres -> ed_cfg_ddim_step [-> ed_rrg_update].
struct EpilogueArgs {
  const void *g_out, *v_out;
  const float* x;
  const int8_t* stamp;
  const int32_t *inv_row, *inv_col, *up_row, *up_col, *down_row, *down_col;
  const int32_t *row_blk, *row_src, *col_blk, *col_src;
  const float* low_latent;                       // [B,C,h,w] last picked reduced latent (RRG) or NULL
  float *prev, *x0, *x_next, *low_dir, *uncond_last, *direction, *local;  // x_next/direction/local optional
  int K, B, C, H, W, h, w, gPH, gPW, g_off_y, g_off_x, vPH, vPW, ncb;
  float g, sb, sa, sp, sd, rrg_norm, rrg_weight;
};

template <typename Tag>
__device__ __forceinline__ float direction_at(const EpilogueArgs& a, int b, int c, int Y, int X) {
  int k = last_covering_step(a.stamp, a.inv_row, a.inv_col, a.K, a.w, Y, X);
  int64_t plane = (int64_t)a.gPH * a.gPW;
  int64_t e = ((int64_t)c * a.gPH + (a.up_row[Y] + a.g_off_y)) * a.gPW + (a.up_col[X] + a.g_off_x);
  float u = ld<Tag>(a.g_out, (((int64_t)k * 2 + 0) * a.B + b) * a.C * plane + e);
  float cd = ld<Tag>(a.g_out, (((int64_t)k * 2 + 1) * a.B + b) * a.C * plane + e);
  return __fsub_rn(cd, u);
}

template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_phase_epilogue(const EpilogueArgs a) {
  const int64_t nfull = (int64_t)a.B * a.C * a.H * a.W;
  const int64_t nlow = (int64_t)a.B * a.C * a.h * a.w;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= nfull + nlow) return;
  if (t >= nfull) {  // reduced-resolution by-products for RRG / the caller: direction sampled at the nearest-downsample points
    int64_t u = t - nfull;
    int j = (int)(u % a.w);
    int64_t r = u / a.w;
    int i = (int)(r % a.h);
    r /= a.h;
    int c = (int)(r % a.C);
    int b = (int)(r / a.C);
    if (a.low_dir) a.low_dir[u] = direction_at<Tag>(a, b, c, a.down_row[i], a.down_col[j]);
    if (a.uncond_last) {
      int64_t plane = (int64_t)a.gPH * a.gPW;
      int64_t e = ((int64_t)c * a.gPH + (i + a.g_off_y)) * a.gPW + (j + a.g_off_x);
      a.uncond_last[u] = ld<Tag>(a.g_out, (((int64_t)(a.K - 1) * 2 + 0) * a.B + b) * a.C * plane + e);
    }
    return;
  }
  int X = (int)(t % a.W);
  int64_t r = t / a.W;
  int Y = (int)(r % a.H);
  r /= a.H;
  int c = (int)(r % a.C);
  int b = (int)(r / a.C);
  // local unconditional score: first covering view whose centre value is non-zero (ED:852-861)
  float loc = 0.0f;
  bool settled = false;
#pragma unroll
  for (int kr = 0; kr < 2; ++kr) {
    int rb = a.row_blk[Y * 2 + kr];
    if (rb < 0 || settled) continue;
    int sy = a.row_src[Y * 2 + kr];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      int cb = a.col_blk[X * 2 + kc];
      if (cb < 0 || settled) continue;
      int sx = a.col_src[X * 2 + kc];
      int64_t row = (int64_t)(rb * a.ncb + cb) * a.B + b;
      loc = ld<Tag>(a.v_out, ((row * a.C + c) * a.vPH + sy) * a.vPW + sx);
      if (loc != 0.0f) settled = true;
    }
  }
  float d = direction_at<Tag>(a, b, c, Y, X);
  float pv, z0;
  ddim_one(loc, d, a.x[t], a.g, a.sb, a.sa, a.sp, a.sd, pv, z0);
  a.prev[t] = pv;
  a.x0[t] = z0;
  if (a.direction) a.direction[t] = d;
  if (a.local) a.local[t] = loc;
  if (a.x_next) {  // ED:886-940 closed form + ED:1078, as k_rrg_update
    int i = a.up_row[Y], j = a.up_col[X];
    int64_t plane = (int64_t)a.gPH * a.gPW;
    int64_t e = ((int64_t)c * a.gPH + (i + a.g_off_y)) * a.gPW + (j + a.g_off_x);
    float lu = ld<Tag>(a.g_out, (((int64_t)(a.K - 1) * 2 + 0) * a.B + b) * a.C * plane + e);
    float ldir = direction_at<Tag>(a, b, c, a.down_row[i], a.down_col[j]);
    float eps = __fadd_rn(lu, __fmul_rn(a.g, ldir));
    float up = __fdiv_rn(__fsub_rn(a.low_latent[(((int64_t)b * a.C + c) * a.h + i) * a.w + j], __fmul_rn(a.sb, eps)), a.sa);
    float grad = __fmul_rn(__fmul_rn(a.rrg_norm, __fsub_rn(z0, up)), a.rrg_weight);
    a.x_next[t] = __fadd_rn(pv, -grad);
  }
}

// ---- ed_undo_step ----------------------------------------------------------------------------------
// n_sub independent 16-byte loads per lane are issued before the dependent chain -> deep memory-level parallelism.
__global__ void __launch_bounds__(ED_BLOCK)
k_undo_v4(const float4* __restrict__ x_in, const float4* __restrict__ noise, const float2* __restrict__ coef,
          float4* __restrict__ x_out, int n_sub, int64_t n4) {
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n4) return;
  float4 xv = x_in[t];
  int k = 0;
  for (; k + 4 <= n_sub; k += 4) {
    float4 z0 = noise[(int64_t)(k + 0) * n4 + t];
    float4 z1 = noise[(int64_t)(k + 1) * n4 + t];
    float4 z2 = noise[(int64_t)(k + 2) * n4 + t];
    float4 z3 = noise[(int64_t)(k + 3) * n4 + t];
    float2 c0 = coef[k], c1 = coef[k + 1], c2 = coef[k + 2], c3 = coef[k + 3];
#define ED_UNDO(NZ, CF)                                             \
  xv.x = __fadd_rn(__fmul_rn(CF.x, xv.x), __fmul_rn(CF.y, NZ.x)); \
  xv.y = __fadd_rn(__fmul_rn(CF.x, xv.y), __fmul_rn(CF.y, NZ.y)); \
  xv.z = __fadd_rn(__fmul_rn(CF.x, xv.z), __fmul_rn(CF.y, NZ.z)); \
  xv.w = __fadd_rn(__fmul_rn(CF.x, xv.w), __fmul_rn(CF.y, NZ.w));
    ED_UNDO(z0, c0) ED_UNDO(z1, c1) ED_UNDO(z2, c2) ED_UNDO(z3, c3)
  }
  for (; k < n_sub; ++k) {
    float4 z = noise[(int64_t)k * n4 + t];
    float2 c = coef[k];
    ED_UNDO(z, c)
  }
#undef ED_UNDO
  x_out[t] = xv;
}

__global__ void __launch_bounds__(ED_BLOCK)
k_undo_s(const float* __restrict__ x_in, const float* __restrict__ noise, const float2* __restrict__ coef,
         float* __restrict__ x_out, int n_sub, int64_t n) {
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  float xv = x_in[t];
  for (int k = 0; k < n_sub; ++k) {
    float2 c = coef[k];
    xv = __fadd_rn(__fmul_rn(c.x, xv), __fmul_rn(c.y, noise[(int64_t)k * n + t]));
  }
  x_out[t] = xv;
}

// ---- ed_rrg_update ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(ED_BLOCK)
k_rrg_update(const float* __restrict__ prev, const float* __restrict__ x0, const float* __restrict__ low_latent,
             const float* __restrict__ low_uncond, const float* __restrict__ low_dir,
             const int32_t* __restrict__ up_row, const int32_t* __restrict__ up_col, float* __restrict__ out,
             float g, float sb, float sa, float norm, float weight, int B, int C, int H, int W, int h, int w) {
  int64_t n = (int64_t)B * C * H * W;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  int X = (int)(t % W);
  int64_t r = t / W;
  int Y = (int)(r % H);
  int64_t bc = r / H;
  int64_t s = (bc * h + up_row[Y]) * w + up_col[X];
  float eps = __fadd_rn(low_uncond[s], __fmul_rn(g, low_dir[s]));
  float up = __fdiv_rn(__fsub_rn(low_latent[s], __fmul_rn(sb, eps)), sa);
  float grad = __fmul_rn(__fmul_rn(norm, __fsub_rn(x0[t], up)), weight);  // d/dx0 of weight*mse(up, x0)
  out[t] = __fadd_rn(prev[t], -grad);
}

__global__ void __launch_bounds__(ED_BLOCK)
k_rrg_update_x4(const float* __restrict__ prev, const float* __restrict__ x0, const float* __restrict__ low_latent,
                const float* __restrict__ low_uncond, const float* __restrict__ low_dir,
                const int32_t* __restrict__ up_row, const int32_t* __restrict__ up_col, float* __restrict__ out,
                float g, float sb, float sa, float norm, float weight, int B, int C, int H, int W, int h, int w) {
  int W4 = W >> 2;
  int64_t n = (int64_t)B * C * H * W4;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  int X = (int)(t % W4) << 2;
  int64_t r = t / W4;
  int Y = (int)(r % H);
  int64_t bc = r / H;
  int64_t base = (bc * h + up_row[Y]) * w;
  float4 p = *reinterpret_cast<const float4*>(prev + (t << 2));
  float4 z = *reinterpret_cast<const float4*>(x0 + (t << 2));
  float pv[4] = {p.x, p.y, p.z, p.w}, zv[4] = {z.x, z.y, z.z, z.w}, o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int64_t s = base + up_col[X + e];
    float eps = __fadd_rn(low_uncond[s], __fmul_rn(g, low_dir[s]));
    float up = __fdiv_rn(__fsub_rn(low_latent[s], __fmul_rn(sb, eps)), sa);
    float grad = __fmul_rn(__fmul_rn(norm, __fsub_rn(zv[e], up)), weight);
    o[e] = __fadd_rn(pv[e], -grad);
  }
  *reinterpret_cast<float4*>(out + (t << 2)) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- ed_gather2d -----------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void __launch_bounds__(ED_BLOCK)
k_gather2d(const void* __restrict__ in, void* __restrict__ out, int C, int H, int W,
           const int32_t* __restrict__ src_n, const int32_t* __restrict__ rows, const int32_t* __restrict__ cols,
           int N, int oh, int ow) {
  int64_t n = (int64_t)N * C * oh * ow;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  int j = (int)(t % ow);
  int64_t r = t / ow;
  int i = (int)(r % oh);
  r /= oh;
  int c = (int)(r % C);
  int m = (int)(r / C);
  int sy = rows[(int64_t)m * oh + i], sx = cols[(int64_t)m * ow + j];
  float v = 0.0f;
  if (sy >= 0 && sx >= 0) v = ld<TI>(in, (((int64_t)src_n[m] * C + c) * H + sy) * W + sx);
  st<TO>(out, t, v);
}

// ---- ed_tile_accumulate_normalise ------------------------------------------------------------------
template <typename Tag>
__global__ void __launch_bounds__(ED_BLOCK)
k_tile_accumulate(const void* __restrict__ dec, float* __restrict__ image, int B, int Cimg, int HP, int WP, int TP,
                  int nct, const int32_t* __restrict__ row_tile, const int32_t* __restrict__ row_src,
                  const int32_t* __restrict__ col_tile, const int32_t* __restrict__ col_src) {
  int64_t n = (int64_t)B * Cimg * HP * WP;
  int64_t t = (int64_t)blockIdx.x * ED_BLOCK + threadIdx.x;
  if (t >= n) return;
  int X = (int)(t % WP);
  int64_t r = t / WP;
  int Y = (int)(r % HP);
  r /= HP;
  int c = (int)(r % Cimg);
  int b = (int)(r / Cimg);
  float sum = 0.0f, cnt = 0.0f;
  for (int kr = 0; kr < ED_TILE_MAXC; ++kr) {
    int rb = row_tile[Y * ED_TILE_MAXC + kr];
    if (rb < 0) break;
    int sy = row_src[Y * ED_TILE_MAXC + kr];
    for (int kc = 0; kc < ED_TILE_MAXC; ++kc) {
      int cb = col_tile[X * ED_TILE_MAXC + kc];
      if (cb < 0) break;
      int sx = col_src[X * ED_TILE_MAXC + kc];
      int64_t row = (int64_t)(rb * nct + cb) * B + b;
      float v = ld<Tag>(dec, ((row * Cimg + c) * TP + sy) * TP + sx);
      v = __fadd_rn(__fdiv_rn(v, 2.0f), 0.5f);
      v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);  // clamp(0,1); NaN stays NaN like torch.clamp
      sum = __fadd_rn(sum, v);
      cnt = __fadd_rn(cnt, 1.0f);
    }
  }
  image[t] = __fdiv_rn(sum, cnt);
}

}  // namespace

// ====================================================================================================
// C ABI
// ====================================================================================================
#define ED_LAUNCH_T(dtype, KERNEL, n, ...)                                                       \
  switch (dtype) {                                                                               \
    case ED_F32: KERNEL<F32><<<grid_for(n), ED_BLOCK, 0, (hipStream_t)stream>>>(__VA_ARGS__); break;   \
    case ED_F16: KERNEL<F16><<<grid_for(n), ED_BLOCK, 0, (hipStream_t)stream>>>(__VA_ARGS__); break;   \
    case ED_BF16: KERNEL<BF16><<<grid_for(n), ED_BLOCK, 0, (hipStream_t)stream>>>(__VA_ARGS__); break; \
    default: return (int)hipErrorInvalidValue;                                                   \
  }
#define ED_LAUNCH(KERNEL, n, ...) KERNEL<<<grid_for(n), ED_BLOCK, 0, (hipStream_t)stream>>>(__VA_ARGS__)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

extern "C" {

int ed_version(void) { return ED_ABI_VERSION; }

const char* ed_error_string(int err) { return hipGetErrorString((hipError_t)err); }

int ed_gather_views(const float* latent, void* out, int dtype, int B, int C, int H, int W, const int32_t* win_y0,
                    const int32_t* win_x0, int V, int Sh, int Sw, int PH, int PW, int off_y, int off_x,
                    const float* frame, float divisor, void* stream) {
  int64_t n = (int64_t)V * B * C * PH * PW;
  if (n == 0) return 0;
  int use_div = divisor != 1.0f;
  if ((PW & 3) == 0 && (Sw & 3) == 0 && (off_x & 3) == 0 && aligned16(out) && (!frame || aligned16(frame))) {
    n >>= 2;
    ED_LAUNCH_T(dtype, k_gather_windows_x4, n, latent, out, B, C, H, W, win_y0, win_x0, V, Sh, Sw, PH, PW, off_y, off_x,
                frame, divisor, use_div);
    return done();
  }
  ED_LAUNCH_T(dtype, k_gather_windows, n, latent, out, B, C, H, W, win_y0, win_x0, V, Sh, Sw, PH, PW, off_y, off_x,
                                         frame, divisor, use_div);
  return done();
}

int ed_tile_gather_pad(const float* latent, void* tiles, int dtype, int B, int C, int H, int W, const int32_t* tile_y0,
                       const int32_t* tile_x0, int T_, int Ts, float scaling_factor, void* stream) {
  return ed_gather_views(latent, tiles, dtype, B, C, H, W, tile_y0, tile_x0, T_, Ts, Ts, Ts, Ts, 0, 0, nullptr,
                         scaling_factor, stream);
}

int ed_scatter_centres(const void* pred, int dtype, float* local, int B, int C, int H, int W, int PH, int PW,
                       int n_col_blocks, const int32_t* row_blk, const int32_t* row_src, const int32_t* col_blk,
                       const int32_t* col_src, void* stream) {
  int64_t n = (int64_t)B * C * H * W;
  if (n == 0) return 0;
  ED_LAUNCH_T(dtype, k_scatter_centres, n, pred, local, B, C, H, W, PH, PW, n_col_blocks, row_blk, row_src, col_blk,
                                         col_src);
  return done();
}

int ed_pick_assemble(const float* latent, const uint8_t* idx, const int32_t* src_row, const int32_t* src_col,
                     const float* frame, void* out, int dtype, float* low, int K, int B, int C, int H, int W, int h,
                     int w, int PH, int PW, int off_y, int off_x, void* stream) {
  int64_t n = (int64_t)K * B * C * PH * PW;
  if (n == 0) return 0;
  if ((PW & 3) == 0 && (w & 3) == 0 && (off_x & 3) == 0 && aligned16(out) && aligned16(idx) &&
      (!frame || aligned16(frame)) && (!low || aligned16(low))) {
    n >>= 2;
    ED_LAUNCH_T(dtype, k_pick_assemble_x4, n, latent, idx, src_row, src_col, frame, out, low, K, B, C, H, W, h, w, PH,
                PW, off_y, off_x);
    return done();
  }
  ED_LAUNCH_T(dtype, k_pick_assemble, n, latent, idx, src_row, src_col, frame, out, low, K, B, C, H, W, h, w, PH, PW,
                                         off_y, off_x);
  return done();
}

int ed_assemble_rows(const float* latent, int B, int C, int H, int W, const uint8_t* idx, const int32_t* src_row,
                     const int32_t* src_col, const float* gframe, void* g_rows, float* low, int K, int h, int w, int gPH,
                     int gPW, int g_off_y, int g_off_x, const int32_t* win_y0, const int32_t* win_x0,
                     const float* vframe, void* v_rows, int V, int Sh, int Sw, int vPH, int vPW, int v_off_y, int v_off_x,
                     int dtype, void* stream) {
  int64_t n_g = (int64_t)K * B * C * gPH * gPW, n_v = (int64_t)V * B * C * vPH * vPW;
  if (n_g + n_v == 0) return 0;
  bool px4 = (gPW & 3) == 0 && (w & 3) == 0 && (g_off_x & 3) == 0 && aligned16(g_rows) && aligned16(idx) &&
             (!gframe || aligned16(gframe)) && (!low || aligned16(low));
  bool gx4 = (vPW & 3) == 0 && (Sw & 3) == 0 && (v_off_x & 3) == 0 && aligned16(v_rows) && (!vframe || aligned16(vframe));
  AssembleArgs a;
  a.latent = latent, a.B = B, a.C = C, a.H = H, a.W = W;
  a.idx = idx, a.src_row = src_row, a.src_col = src_col, a.gframe = gframe, a.g_out = g_rows, a.low = low;
  a.K = K, a.h = h, a.w = w, a.gPH = gPH, a.gPW = gPW, a.g_off_y = g_off_y, a.g_off_x = g_off_x;
  a.win_y0 = win_y0, a.win_x0 = win_x0, a.vframe = vframe, a.v_out = v_rows;
  a.V = V, a.Sh = Sh, a.Sw = Sw, a.vPH = vPH, a.vPW = vPW, a.v_off_y = v_off_y, a.v_off_x = v_off_x;
  a.pick_blocks = n_g ? grid_for(px4 ? n_g >> 2 : n_g) : 0;
  int view_blocks = n_v ? grid_for(gx4 ? n_v >> 2 : n_v) : 0;
  dim3 grid(a.pick_blocks + view_blocks), block(ED_BLOCK);
  hipStream_t st_ = (hipStream_t)stream;
#define ED_ASM(T)                                                              \
  if (px4 && gx4) k_assemble_rows<T, true, true><<<grid, block, 0, st_>>>(a);  \
  else if (px4) k_assemble_rows<T, true, false><<<grid, block, 0, st_>>>(a);   \
  else if (gx4) k_assemble_rows<T, false, true><<<grid, block, 0, st_>>>(a);   \
  else k_assemble_rows<T, false, false><<<grid, block, 0, st_>>>(a);
  switch (dtype) {
    case ED_F32: ED_ASM(F32) break;
    case ED_F16: ED_ASM(F16) break;
    case ED_BF16: ED_ASM(BF16) break;
    default: return (int)hipErrorInvalidValue;
  }
#undef ED_ASM
  return done();
}

int ed_phase_epilogue(const void* g_out, const void* v_out, int dtype, const float* x, const int8_t* stamp,
                      const int32_t* inv_row, const int32_t* inv_col, const int32_t* up_row, const int32_t* up_col,
                      const int32_t* down_row, const int32_t* down_col, const int32_t* row_blk, const int32_t* row_src,
                      const int32_t* col_blk, const int32_t* col_src, const float* low_latent, float* prev, float* x0,
                      float* x_next, float* low_dir, float* uncond_last, float* direction, float* local, int K, int B,
                      int C, int H, int W, int h, int w, int gPH, int gPW, int g_off_y, int g_off_x, int vPH, int vPW,
                      int n_col_blocks, float g, float sqrt_beta_t, float sqrt_alpha_t, float sqrt_alpha_prev,
                      float sqrt_1m_alpha_prev, float rrg_norm, float rrg_weight, void* stream) {
  int64_t n = (int64_t)B * C * H * W + (int64_t)B * C * h * w;
  if ((int64_t)B * C * H * W == 0) return 0;
  if (x_next && !low_latent) return (int)hipErrorInvalidValue;
  EpilogueArgs a;
  a.g_out = g_out, a.v_out = v_out, a.x = x, a.stamp = stamp;
  a.inv_row = inv_row, a.inv_col = inv_col, a.up_row = up_row, a.up_col = up_col, a.down_row = down_row, a.down_col = down_col;
  a.row_blk = row_blk, a.row_src = row_src, a.col_blk = col_blk, a.col_src = col_src;
  a.low_latent = low_latent, a.prev = prev, a.x0 = x0, a.x_next = x_next, a.low_dir = low_dir, a.uncond_last = uncond_last;
  a.direction = direction, a.local = local;
  a.K = K, a.B = B, a.C = C, a.H = H, a.W = W, a.h = h, a.w = w, a.gPH = gPH, a.gPW = gPW, a.g_off_y = g_off_y;
  a.g_off_x = g_off_x, a.vPH = vPH, a.vPW = vPW, a.ncb = n_col_blocks;
  a.g = g, a.sb = sqrt_beta_t, a.sa = sqrt_alpha_t, a.sp = sqrt_alpha_prev, a.sd = sqrt_1m_alpha_prev;
  a.rrg_norm = rrg_norm, a.rrg_weight = rrg_weight;
  ED_LAUNCH_T(dtype, k_phase_epilogue, n, a);
  return done();
}

int ed_unpad_direction(const void* unet_out, int dtype, float* dirs, float* uncond_last, int K, int B, int C, int h,
                       int w, int PH, int PW, int off_y, int off_x, void* stream) {
  int64_t n = (int64_t)K * B * C * h * w;
  if (n == 0) return 0;
  if ((PW & 3) == 0 && (w & 3) == 0 && (off_x & 3) == 0 && aligned16(unet_out) && aligned16(dirs) &&
      (!uncond_last || aligned16(uncond_last))) {
    n >>= 2;
    ED_LAUNCH_T(dtype, k_unpad_direction_x4, n, unet_out, dirs, uncond_last, K, B, C, h, w, PH, PW, off_y, off_x);
    return done();
  }
  ED_LAUNCH_T(dtype, k_unpad_direction, n, unet_out, dirs, uncond_last, K, B, C, h, w, PH, PW, off_y, off_x);
  return done();
}

int ed_fill_directions(const float* dirs, const int8_t* stamp, const int32_t* inv_row, const int32_t* inv_col,
                       const int32_t* up_row, const int32_t* up_col, const int32_t* down_row, const int32_t* down_col,
                       float* target, float* low_dir, int K, int B, int C, int H, int W, int h, int w, void* stream) {
  int64_t n = (int64_t)H * W + (low_dir ? (int64_t)h * w : 0);
  if (n == 0 || K <= 0) return K <= 0 ? (int)hipErrorInvalidValue : 0;
  ED_LAUNCH(k_fill_directions, n, dirs, stamp, inv_row,
                     inv_col, up_row, up_col, down_row, down_col, target, low_dir, K, B, C, H, W, h, w);
  return done();
}

int ed_cfg_ddim_step(const float* local, const float* direction, const float* x, float* prev, float* x0, float g,
                     float sqrt_beta_t, float sqrt_alpha_t, float sqrt_alpha_prev, float sqrt_one_minus_alpha_prev,
                     int64_t n, void* stream) {
  if (n == 0) return 0;
  if ((n & 3) == 0 && aligned16(local) && aligned16(direction) && aligned16(x) && aligned16(prev) && aligned16(x0)) {
    ED_LAUNCH(k_cfg_ddim_v4, n / 4, (const float4*)local, (const float4*)direction, (const float4*)x, (float4*)prev, (float4*)x0, g,
                       sqrt_beta_t, sqrt_alpha_t, sqrt_alpha_prev, sqrt_one_minus_alpha_prev, n / 4);
  } else {
    ED_LAUNCH(k_cfg_ddim_s, n, local, direction, x,
                       prev, x0, g, sqrt_beta_t, sqrt_alpha_t, sqrt_alpha_prev, sqrt_one_minus_alpha_prev, n);
  }
  return done();
}

int ed_undo_step(const float* x_in, const float* noise, const float* coef, float* x_out, int n_sub, int64_t n,
                 void* stream) {
  if (n == 0) return 0;
  if ((n & 3) == 0 && aligned16(x_in) && aligned16(noise) && aligned16(x_out)) {
    ED_LAUNCH(k_undo_v4, n / 4, (const float4*)x_in,
                       (const float4*)noise, (const float2*)coef, (float4*)x_out, n_sub, n / 4);
  } else {
    ED_LAUNCH(k_undo_s, n, x_in, noise,
                       (const float2*)coef, x_out, n_sub, n);
  }
  return done();
}

int ed_rrg_update(const float* prev, const float* x0, const float* low_latent, const float* low_uncond,
                  const float* low_dir, const int32_t* up_row, const int32_t* up_col, float* out, float g,
                  float sqrt_beta_t, float sqrt_alpha_t, float norm, float weight, int B, int C, int H, int W, int h,
                  int w, void* stream) {
  int64_t n = (int64_t)B * C * H * W;
  if (n == 0) return 0;
  if ((W & 3) == 0 && aligned16(prev) && aligned16(x0) && aligned16(out)) {
    n >>= 2;
    ED_LAUNCH(k_rrg_update_x4, n, prev, x0, low_latent, low_uncond, low_dir, up_row, up_col, out, g, sqrt_beta_t,
              sqrt_alpha_t, norm, weight, B, C, H, W, h, w);
    return done();
  }
  ED_LAUNCH(k_rrg_update, n, prev, x0, low_latent,
                     low_uncond, low_dir, up_row, up_col, out, g, sqrt_beta_t, sqrt_alpha_t, norm, weight, B, C, H, W,
                     h, w);
  return done();
}

int ed_gather2d(const void* in, int in_dtype, void* out, int out_dtype, int C, int H, int W, const int32_t* src_n,
                const int32_t* rows, const int32_t* cols, int N, int oh, int ow, void* stream) {
  int64_t n = (int64_t)N * C * oh * ow;
  if (n == 0) return 0;
#define ED_G2D(TI, TO) \
  k_gather2d<TI, TO><<<grid_for(n), ED_BLOCK, 0, (hipStream_t)stream>>>(in, out, C, H, W, src_n, rows, cols, N, oh, ow)
  int key = in_dtype * 3 + out_dtype;
  switch (key) {
    case 0: ED_G2D(F32, F32); break;
    case 1: ED_G2D(F32, F16); break;
    case 2: ED_G2D(F32, BF16); break;
    case 3: ED_G2D(F16, F32); break;
    case 4: ED_G2D(F16, F16); break;
    case 5: ED_G2D(F16, BF16); break;
    case 6: ED_G2D(BF16, F32); break;
    case 7: ED_G2D(BF16, F16); break;
    case 8: ED_G2D(BF16, BF16); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef ED_G2D
  return done();
}

int ed_tile_accumulate_normalise(const void* decoded, int dtype, float* image, int B, int Cimg, int HP, int WP, int TP,
                                 int n_col_tiles, const int32_t* row_tile, const int32_t* row_src,
                                 const int32_t* col_tile, const int32_t* col_src, void* stream) {
  int64_t n = (int64_t)B * Cimg * HP * WP;
  if (n == 0) return 0;
  ED_LAUNCH_T(dtype, k_tile_accumulate, n, decoded, image, B, Cimg, HP, WP, TP, n_col_tiles, row_tile, row_src, col_tile,
                                         col_src);
  return done();
}

}  // extern "C"
