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
