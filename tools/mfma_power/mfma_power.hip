// mfma_power.hip -- MEASUREMENT ONLY (not in libelastic_hip.so).  A register-only MFMA loop: what does the MI355X sustain on 16-bit MFMAs with
// nothing else going on, as a function of the MFMA shape (16x16x32 vs 32x32x16), the type (f16 / bf16) and the operands' bit activity
// (random vs zeros)?  2 waves per SIMD (512-thread workgroups, one per CU and per launch slot), 64 accumulator registers per lane,
// 16 (8) MFMAs per iteration = 131072 MACs per wave and iteration for either shape, cycling through `nacc` accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/mfma_power/mfma_power.hip -o tools/mfma_power/libmfma_power.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool BF, int NACC>
__global__ void __launch_bounds__(512, 2) k_mfma_loop(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 512 + threadIdx.x;
  u32x4 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = in[(size_t)t * 8 + i];
    rb[i] = in[(size_t)t * 8 + 4 + i];
  }
  float sum = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (BF)
          acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[j & 3]), __builtin_bit_cast(bf16x8, rb[j >> 2]), acc[j % NACC], 0, 0, 0);
        else
          acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ra[j & 3]), __builtin_bit_cast(f16x8, rb[j >> 2]), acc[j % NACC], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NACC; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  } else {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (BF)
          acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[j & 3]), __builtin_bit_cast(bf16x8, rb[(j >> 1) & 3]), acc[j % NACC], 0, 0, 0);
        else
          acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[j & 3]), __builtin_bit_cast(f16x8, rb[(j >> 1) & 3]), acc[j % NACC], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[j][e];
  }
  out[t] = sum;
}

// nacc = accumulators cycled through (= distance, in MFMAs, between two MFMAs on the same accumulator): 16x16x32: 2 4 8 16; 32x32x16: 1 2 4 8
extern "C" int ed_mfma_loop(int shape, int bf, int nacc, const void* in, void* out, int blocks, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define GO(SH, B, NA) k_mfma_loop<SH, B, NA><<<blocks, 512, 0, s>>>((const u32x4*)in, (float*)out, iters)
#define PICK(SH, NA)                  \
  if (shape == SH && nacc == NA) {    \
    if (bf) GO(SH, true, NA);         \
    else GO(SH, false, NA);           \
    return (int)hipGetLastError();    \
  }
  PICK(16, 2) PICK(16, 4) PICK(16, 8) PICK(16, 16) PICK(32, 1) PICK(32, 2) PICK(32, 4) PICK(32, 8)
#undef PICK
#undef GO
  return (int)hipErrorInvalidValue;
}

// ---- VALU issue rates next to it: what does one softmax numerator cost?  8 independent chains per lane, OP per chain element:
//   0  v_fma_f32         1  v_exp_f32         2  v_fma_f32 + v_exp_f32 + v_add_f32 (one lazy-softmax numerator)
//   3  as 2, and one 16x16x32 MFMA per 2 numerators (4 per iteration)
//   4  as 2, and one 32x32x16 MFMA per 4 numerators (2 per iteration: the same MACs as 3), 8 accumulators in rotation
//   5  4 MFMAs 16x16x32 alone          6  2 MFMAs 32x32x16 alone (8 accumulators in rotation)
template <int OP>
__global__ void __launch_bounds__(512, 2) k_valu_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 512 + threadIdx.x;
  float x[8], acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = in[(size_t)t * 8 + i];
  const float a = in[0] * 1e-9f + 0.999f, b = in[1] * 1e-9f - 0.01f;
  f32x4 macc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) macc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f16x8 ma = __builtin_bit_cast(f16x8, u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u});
  f32x16 bacc[8];
  if (OP == 4 || OP == 6) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) bacc[j][e] = 0.f;
  }
  for (int it4 = 0; it4 < iters; it4 += 4) {
#pragma unroll
   for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 5) {
        if (i & 1) macc[i >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ma, ma, macc[i >> 1], 0, 0, 0);
      } else if (OP == 6) {
        if ((i & 3) == 3) bacc[2 * u + (i >> 2)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ma, ma, bacc[2 * u + (i >> 2)], 0, 0, 0);
      } else if (OP == 0) {
        x[i] = __builtin_fmaf(x[i], a, b);
      } else if (OP == 1) {
        x[i] = __builtin_amdgcn_exp2f(x[i]) - 1.0f;     // (keeps the chain bounded: one extra v_add per exp, reported as such)
      } else {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], a, b));
        acc += e;
        x[i] = e - 1.0f;
        if (OP == 3 && (i & 1)) macc[i >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ma, ma, macc[i >> 1], 0, 0, 0);
        if (OP == 4 && (i & 3) == 3) bacc[2 * u + (i >> 2)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ma, ma, bacc[2 * u + (i >> 2)], 0, 0, 0);
      }
    }
   }
  }
  if (OP == 4 || OP == 6) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += bacc[j][0];
  }
  float s = acc;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) s += macc[j][0];
  out[t] = s;
}

extern "C" int ed_valu_loop(int op, const void* in, void* out, int blocks, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (op == 0) k_valu_loop<0><<<blocks, 512, 0, s>>>((const float*)in, (float*)out, iters);
  else if (op == 1) k_valu_loop<1><<<blocks, 512, 0, s>>>((const float*)in, (float*)out, iters);
  else if (op == 2) k_valu_loop<2><<<blocks, 512, 0, s>>>((const float*)in, (float*)out, iters);
  else if (op == 3) k_valu_loop<3><<<blocks, 512, 0, s>>>((const float*)in, (float*)out, iters);
  else if (op == 4) k_valu_loop<4><<<blocks, 512, 0, s>>>((const float*)in, (float*)out, iters);
  else if (op == 5) k_valu_loop<5><<<blocks, 512, 0, s>>>((const float*)in, (float*)out, iters);
  else if (op == 6) k_valu_loop<6><<<blocks, 512, 0, s>>>((const float*)in, (float*)out, iters);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}
