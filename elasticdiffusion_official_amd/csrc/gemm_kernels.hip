// gemm_kernels.hip -- the UNet's dense contractions where a hand-written gfx950 kernel beats the library call it replaces
// (model side of the hot path's boundary, elastic_diffusion.py:422-426 `self.unet(...)`; diffusers' GEGLU / Linear /
// Conv2d modules):
//
//   ed_geglu_gemm    out[m, n] = (x[m,:] . W[n,:] + b[n]) * gelu(x[m,:] . W[I+n,:] + b[I+n])     x [M,K], W [2I,K], out [M,I]
//                    the transformer feed-forward's first projection with the GEGLU product in the epilogue: removes
//                    `ed_geglu` (reads 2I and writes I right after the GEMM wrote 2I) -- 1.15..1.32 x the hipBLASLt GEMM +
//                    ed_geglu pair on the SDXL shapes (profiles/r4_s1_gemm_first_run_fp16.json)
//   ed_linear        out = x W^T + b (+ residual)                                                   W [N,K] (torch Linear)
//                    the same main loop as a plain projection: 1.1..1.7 x hipBLASLt on the K = 640 projections
//   ed_conv3x3_nhwc  3x3 / stride 1 / pad 1 convolution of an NHWC image as an implicit GEMM (K = 9 Cin: tap offset per K
//                    tile, out-of-image taps read as zeros through the buffer range check), + bias + per-sample channel
//                    bias (the time-embedding add) + residual in the epilogue: 1.2..1.6 x MIOpen's CK kernels on the UNet's
//                    ResnetBlock convolutions
//
// History: written at the end of round 3 with no GPU at hand -- every address and the schedule's LDS hazard intervals were
// replayed lane by lane on the CPU (tools/emulate_gemm_kernel.py, kept as a test) -- and correct on its first hardware run
// in round 4 (bit-identical over 20 launches on every shape; tools/probe_gemm.py is that harness).
//
// Structure: the 256 x 256 x 64, 8-wave, 8-phase schedule of the CDNA4 guide (cdna_hip_programming.md section 5, "The
// 256^2 8-phase template"), with the operands arranged for this epilogue:
//   * a workgroup owns 256 rows of x and 128 VALUE columns + the matching 128 GATE columns of W: a 256 x 256 MFMA tile
//     whose result is a 256 x 128 tile of `out` (plain projection / convolution: two 128-column halves, 256 output columns);
//   * wave (wr, wc), wr = wave >> 2, wc = wave & 3: rows [128 wr, +128), value columns [32 wc, +32) and the same gate
//     columns: 8 x (2 + 2) accumulators of 16 x 16 (128 registers);
//   * the MFMA is issued as D = W_frag . X_frag^T (A operand = W rows, B operand = x rows), so a lane's 4 accumulator
//     registers are 4 CONSECUTIVE output columns of ONE row; W rows are staged in the order
//     n(i, f) = 8 (i >> 2) + (i & 3) + 4 f   (i = MFMA row index, f = fragment 0 / 1), which makes the two fragments of a
//     lane 8 consecutive columns: one 16-byte store per (lane, 16-row block);
//   * LDS image = 1-KiB subtiles of 16 rows x 32 k (64-byte rows), rows 8..15 with the two 32-byte halves swapped
//     (st_16x32 swizzle: conflict-free ds_read_b128 for the 16x16x32 operand layout); a subtile is exactly one
//     wave-wide LDS-DMA (64 lanes x 16 B), the swizzle is applied to the per-lane SOURCE address and to the read address;
//   * K tile t lives in buffer t & 1 (2 x 64 KiB: x 32 KiB + W 32 KiB); a tile is consumed in 4 phases of 16 MFMAs
//     (m half 0 x value, m half 0 x gate, m half 1 x gate, m half 1 x value); one half tile (16 KiB, 2 LDS-DMAs per wave)
//     is staged per phase, 4..7 phases ahead of its first read; `vmcnt(6)` once per tile, never 0 in the steady state;
//   * the two wave rows run half a phase apart (the second one passes one extra barrier first): while one issues its
//     LDS reads and DMAs the other owns the matrix pipe (the two share every SIMD).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "elastic_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct BF {
  typedef bf16x8 v8;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
  // two RNE conversions in one instruction (v_cvt_pk_bf16_f32): lo | hi << 16
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));
  }
};
struct HF {
  typedef f16x8 v8;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {      // v_cvt_pk_f16_f32 (RNE)
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, h2));
  }
};

constexpr int BM = 256;            // x rows per workgroup
constexpr int BN = 128;            // value columns per workgroup (+ the same number of gate columns)
constexpr int BK = 64;             // k per tile
constexpr int SUB = 1024;          // bytes of one 16-row x 32-k subtile
constexpr int W_REGION = 32768;    // byte offset of the W subtiles inside a buffer (x: 16 row groups x 2 k halves first)
constexpr int BUF = 65536;         // bytes per K-tile buffer

// ---- index algebra (mirrored one to one by emulate_geglu_gemm.py) ---------------------------------------------------
// position p (bytes) inside a subtile <-> element (row, k byte): rows 8..15 have their 32-byte halves swapped
__device__ __forceinline__ int swz(int p) { return p ^ (((p >> 9) & 1) << 5); }
// subtile byte offsets inside a buffer
__device__ __forceinline__ constexpr int x_sub(int rg, int kh) { return (rg * 2 + kh) * SUB; }
__device__ __forceinline__ constexpr int w_sub(int rg, int kh) { return W_REGION + (rg * 2 + kh) * SUB; }

// gelu(x) = x Phi(x) with erfc from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 on erf): q = erfc(|x| / sqrt 2)
__device__ __forceinline__ float gelu_as(float x) {
  const float z = __builtin_fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float q = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float h = 0.5f * x * q;           // x Phi(x) for x < 0
  return x > 0.f ? x - h : h;             // x (1 - q/2) for x > 0
}

// The GEGLU epilogue's arithmetic, v * gelu(g), in as few VALU issues as the formula allows (round 6).  The epilogue is pure VALU time
// during which the matrix pipe idles (PMC: MFMA busy 0.55 on GEGLU against 0.91 for the same main loop on a long-K convolution), so it
// is paid per issue slot -- and on gfx950 a packed fp32 instruction costs two slots (SIMD-32: v_fma_f32 2 cycles, v_pk_fma_f32 4;
// MI355X_MICROARCH.md), so what counts is the number of scalar operations per element: 18 here against 23.4 in the round-5 epilogue
// (gelu_as + the select: 17.4 issues per element of which 6 packed).  Same values as gelu_as bit for bit: the polynomial's coefficients
// carry the 0.5 of x Phi(x) (an exact power-of-two scaling of every Horner step),  x > 0 ? x - h : h  with h = (0.5 x) q  becomes
// max(x, 0) - |x| (0.5 q)  (sign-symmetric products; |.| is a source modifier), and nothing is re-associated.
// |a| * b with the absolute value as a source modifier (hipcc's SLP vectoriser otherwise packs the multiplies and pays a v_and per element
// for the |.| a packed instruction cannot express)
__device__ __forceinline__ float abs_mul(float a, float b) {
  float r;
  asm("v_mul_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max(a, 0) in ONE instruction (the builtin adds a canonicalising v_max a, a in front: a is a sum here, already canonical)
__device__ __forceinline__ float relu1(float a) {
  float r;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ float geglu_one(float v, float g) {
  const float z = abs_mul(g, 0.70710678118654752f);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = __builtin_fmaf(p, t, 0.5f * 1.421413741f);
  p = __builtin_fmaf(p, t, 0.5f * -0.284496736f);
  p = __builtin_fmaf(p, t, 0.5f * 0.254829592f);
  const float hq = (p * t) * __builtin_amdgcn_exp2f((-1.4426950408889634f * z) * z);     // 0.5 erfc(|g| / sqrt 2)
  return v * (relu1(g) - abs_mul(g, hq));
}

#define ED_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define ED_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define ED_BARRIER()                          \
  do {                                        \
    asm volatile("s_barrier" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

struct Ctx {
  __amdgpu_buffer_rsrc_t xr, wr_;   // buffer descriptors: rows beyond M (or 2I) read as zeros
  // per-lane byte offsets of the lane's 16 bytes of its wave's subtile (k tile 0, k half 0): [m half] / [value, gate]; the K tile
  // and k half are added per DMA (rows beyond the tensor are beyond the buffer's range and read as zeros)
  int x_voff[2], w_voff[2];
  int px_mask[2];                   // CONV: bit t set = tap t of the lane's pixel (m half 0 / 1) lies inside the image
  int img_w, cin2, cpt;             // CONV: image width (pixels), bytes per pixel (2 Cin), K tiles per tap (Cin / 64)
  int wave;                         // wave id (wave-uniform)
  int xrd, wrd;                     // per-lane LDS byte addresses of fragment (row group 0 of this wave, k half 0), buffer 0
};

// position of a K tile (wave-uniform).  GEMM: k = 64 tile.  CONV (3x3, stride 1, pad 1, NHWC): K = 9 taps x Cin, tile -> (tap, 64-channel
// block ct); the A operand row of output pixel m at tap (dy, dx) is the Cin vector of pixel m + dy W + dx, or zeros outside the image.
// `tile` is the position in the WALK, `w` the K-tile index into W's rows (= tap cpt + ct; = tile for a GEMM).
// Round 6: the convolution walks CHANNEL-BLOCK-major -- all 9 taps of channel block 0, then of block 1, ... -- instead of tap-major.
// The 9 taps of one channel block read the same 128-byte segments of a tile's (256 + halo) pixels, shifted by a pixel or an image row:
// ~64 KB per tile that stays in L2 across the 9 K tiles and is never needed again, where the tap-major walk came back to every pixel's
// whole Cin vector (328 KB per tile at 128 x 128 x 320, 10 MB for the 32 tiles an XCD runs against 4 MiB of L2) nine times: PMC
// FETCH_SIZE 6.8 x the algorithmic bytes (profiles/r5_unet_pmc.json).  Same products, another summation order.
constexpr int CONV_UP2X = 2;   // template value CONV = 2: the 3x3 convolution of the 2x nearest-upsampled input (ed_conv3x3_nhwc_up2x)
constexpr int CONV_S2 = 3;     // CONV = 3: stride 2, padding 1 (ed_conv3x3_nhwc_s2): output pixel (y, x) at tap (dy, dx) reads input pixel
                               // (2 y + dy, 2 x + dx) -- the lane's base offset is its input pixel, the tap offsets stay wave-uniform
constexpr int CONV_S2P0 = 4;   // CONV = 4: stride 2 after F.pad(x, (0, 1, 0, 1)) (the VAE encoder's Downsample2D, ed_conv3x3_nhwc_f32out_s2): tap
                               // (ty, tx) in 0..2 reads input (2 y + ty, 2 x + tx), zeros past the bottom / right edge: the base is pixel (2 y + 1, 2 x + 1)
struct KPos {
  int tile, tap, ct, w;
};
template <int CONV>
__device__ __forceinline__ KPos k_next(KPos p, int cpt) {
  ++p.tile;
  if (CONV) {
    ++p.tap;
    p.w += cpt;
    if (p.tap == 9) {
      p.tap = 0;
      ++p.ct;
      p.w = p.ct;
    }
  } else {
    p.w = p.tile;
  }
  return p;
}

// one half tile = 16 subtiles: wave w fills row group `rg` (both k halves) -- 2 LDS-DMAs of 1 KiB
template <int BUFI, int CONV>
__device__ __forceinline__ void stage_x(uint8_t* lds, const Ctx& c, KPos p, int h) {
  const int rg = (c.wave & 3) + 8 * (c.wave >> 2) + 4 * h;   // rows read in phase 1 (h = 0) / phase 3 (h = 1) of either wave row
  uint8_t* dst = lds + BUFI * BUF + x_sub(0, 0) + rg * (2 * SUB);
  // LDS-DMA with everything in the VGPR offset: immediate and scalar offsets stay 0 (the immediate also moves the LDS address;
  // composable_kernel's ck_tile forces the scalar offset to 0 for these loads on gfx950 -- amd_async_buffer_load)
  if (CONV) {
    const int dy = p.tap / 3 - 1, dx = p.tap - 3 * (p.tap / 3) - 1;              // scalar
    int delta;
    if (CONV == CONV_UP2X) {
      // the A operand is the nearest-neighbour 2x upsampling of x, never written: output pixel (y, x) at tap (dy, dx) reads source pixel
      // ((y + dy) >> 1, (x + dx) >> 1) = the lane's own source pixel (x_voff) moved by ((y & 1) + dy) >> 1 rows and ((x & 1) + dx) >> 1
      // pixels of the SOURCE image (c.img_w = its width): per lane, from the two parity bits kept beside the tap mask
      const int yp = (c.px_mask[h] >> 16) & 1, xp = (c.px_mask[h] >> 17) & 1;
      delta = (((yp + dy) >> 1) * c.img_w + ((xp + dx) >> 1)) * c.cin2 + p.ct * (BK * 2);
    } else {
      delta = (dy * c.img_w + dx) * c.cin2 + p.ct * (BK * 2);                     // scalar, may be negative
    }
    const int vo = ((c.px_mask[h] >> p.tap) & 1) ? c.x_voff[h] + delta : (int)0x80000000;   // outside the image: out of range = zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)(dst + SUB), 16, vo + 64, 0, 0, 0);
  } else {
    const int vo = c.x_voff[h] + p.tile * (BK * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)(dst + SUB), 16, vo + 64, 0, 0, 0);
  }
}
template <int BUFI>
__device__ __forceinline__ void stage_w(uint8_t* lds, const Ctx& c, int widx, int g) {    // widx = KPos::w of the K tile
  const int rg = 8 * g + c.wave;                              // value row groups 0..7, gate row groups 8..15
  const int vo = c.w_voff[g] + widx * (BK * 2);
  uint8_t* dst = lds + BUFI * BUF + w_sub(0, 0) + rg * (2 * SUB);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.wr_, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.wr_, (lds_ptr_t)(dst + SUB), 16, vo + 64, 0, 0, 0);
}

template <class T>
struct Frags {
  typename T::v8 x[4][2];    // 4 row blocks of the current m half x 2 k halves   (MFMA B operand)
  typename T::v8 wv[2][2];   // value fragments f = 0, 1 x 2 k halves              (MFMA A operand)
  typename T::v8 wg[2][2];   // gate fragments
};

template <class T, int BUFI>
__device__ __forceinline__ void read_x(const uint8_t* lds, const Ctx& c, Frags<T>& f, int mh) {
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
      f.x[mf][kh] = *reinterpret_cast<const typename T::v8*>(lds + c.xrd + BUFI * BUF + x_sub(mh * 4 + mf, kh));
}
template <class T, int BUFI, int G>
__device__ __forceinline__ void read_w(const uint8_t* lds, const Ctx& c, Frags<T>& f) {
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      typename T::v8 v = *reinterpret_cast<const typename T::v8*>(lds + c.wrd + BUFI * BUF + w_sub(8 * G + nf, kh) - W_REGION);
      if (G == 0) f.wv[nf][kh] = v; else f.wg[nf][kh] = v;
    }
}

// 16 MFMAs: m half `MH` x (value | gate) x both k halves
template <class T, int MH, int G>
__device__ __forceinline__ void mma16(f32x4 (&acc)[8][4], const Frags<T>& f) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int kh = 0; kh < 2; ++kh)
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
        acc[MH * 4 + mf][G * 2 + nf] = T::mfma(G == 0 ? f.wv[nf][kh] : f.wg[nf][kh], f.x[mf][kh], acc[MH * 4 + mf][G * 2 + nf]);
  __builtin_amdgcn_s_setprio(0);
}

// the four phases of K tile `tile` (buffer BUFI).  s1: tile + 1 exists, s2: tile + 2 exists (wave-uniform); p1 / p2 their positions.
// FIRST ("early start", round 4): K tile 0 of a workgroup whose prologue waited only for what phase 1 reads -- the W value rows
// and x m-half 0, 4 of its 14 LDS-DMAs per wave -- so the MFMAs start after 32 KiB instead of 64 KiB have landed (a tile's fixed
// cost is ~7 us against 1.5 us per K tile: 19 % of a K = 1280 tile, 33 % of a K = 640 one).  The gate rows (read in phase 2) and x
// m-half 1 (phase 3) are retired by a `vmcnt(10)` in front of BOTH barriers of phases 1 and 2: the two wave rows run one barrier
// apart, so the barrier that publishes a wave's DMAs to the row that reads first is that row's second barrier of the phase and
// the other row's first one (16 / 18 DMAs issued by then: 10 outstanding = the first 6 / 8 landed).  Replayed under both
// adversarial timings in tools/emulate_gemm_kernel.py (`--break early` weakens the count and is caught).
template <class T, int BUFI, int CONV, bool FIRST = false>
__device__ __forceinline__ void tile_phases(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], int tile, bool s1,
                                            bool s2, KPos p1, KPos p2) {
  // ---- phase 1: m half 0 x value.  12 fragment reads: the 4 W reads first, so that lgkmcnt(8) retires them before the
  // barrier and the value half of this buffer may be re-staged one phase from now.
  read_w<T, BUFI, 0>(lds, c, f);
  __builtin_amdgcn_sched_barrier(0);
  read_x<T, BUFI>(lds, c, f, 0);
  if (s1) stage_x<BUFI ^ 1, CONV>(lds, c, p1, 1);       // x m-half 1 of tile + 1: its buffer's copy was last read 2 phases ago
  ED_WAIT_LGKM(8);
  if (FIRST) ED_WAIT_VM(10);
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16<T, 0, 0>(acc, f);
  if (FIRST) ED_WAIT_VM(10);
  ED_BARRIER();
  // ---- phase 2: m half 0 x gate
  read_w<T, BUFI, 1>(lds, c, f);
  if (s2) stage_w<BUFI>(lds, c, p2.w, 0);           // value rows of tile + 2 (read in phase 1, retired before its barrier)
  if (FIRST) ED_WAIT_VM(10);
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16<T, 0, 1>(acc, f);
  if (FIRST) ED_WAIT_VM(10);
  ED_BARRIER();
  // ---- phase 3: m half 1 x gate
  read_x<T, BUFI>(lds, c, f, 1);
  if (s2) stage_x<BUFI, CONV>(lds, c, p2, 0);           // x m-half 0 of tile + 2 (read in phase 1)
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16<T, 1, 1>(acc, f);
  ED_BARRIER();
  // ---- phase 4: m half 1 x value (fragments already in registers)
  if (s2) {
    stage_w<BUFI>(lds, c, p2.w, 1);                 // gate rows of tile + 2 (read in phase 2)
    ED_WAIT_VM(6);                                      // all of tile + 1 has landed; 3 half tiles of tile + 2 stay in flight
  } else {
    ED_WAIT_VM(0);                                      // last two tiles: nothing newer to leave in flight
  }
  ED_BARRIER();
  mma16<T, 1, 0>(acc, f);
  ED_BARRIER();
}

// LONG K (round 5; the convolutions: K = 9 Cin >= 2880): the same K tile in 4 barrier intervals of 32 MFMAs instead of 8 of 16 -- half the
// barriers and pipe drains per K tile.  Measured on the MI355X against the 8-phase loop, bit-identical (profiles/r5_s2_long_k_patch_0006_*):
// convolutions +4 ... 12 % (32 x 32 1280 -> 1280: 1205 -> 1354 TFLOP/s), plain projections K = 1280 / 2560 -0.5 ... -2.7 %, K = 640 -1 ... -3.5 %
// (profiles/r4_s15_*) -- hence a separate instantiation (TWO) that the launcher picks for CONVOLUTIONS of at least TWO_MIN_TILES K tiles
// only.  No early start (the prologue waits for all of K tile 0).
//   R1  W value + gate rows and x m-half 0 (16 fragment reads); x m-half 1 of tile + 1 -> the other buffer
//   M1  32 MFMAs (m half 0 x value, gate)
//   R2  x m-half 1 (8 reads); W value rows, x m-half 0, W gate rows of tile + 2 -> this buffer (all last read in R1, by either wave row one
//       interval ago); vmcnt(6): all of tile + 1 has landed
//   M2  32 MFMAs (m half 1 x gate, value)
// A wave retires its fragment reads before its barrier.  Replay: tools/emulate_gemm_kernel.py --sched two_read (incl. the convolution).
constexpr int TWO_MIN_TILES = 20;
template <class T, int BUFI, int CONV>
__device__ __forceinline__ void tile_phases_two(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], int tile, bool s1, bool s2,
                                                KPos p1, KPos p2) {
  read_w<T, BUFI, 0>(lds, c, f);
  read_x<T, BUFI>(lds, c, f, 0);
  read_w<T, BUFI, 1>(lds, c, f);
  if (s1) stage_x<BUFI ^ 1, CONV>(lds, c, p1, 1);
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 0, 0>(acc, f);
  mma16<T, 0, 1>(acc, f);
  ED_BARRIER();
  read_x<T, BUFI>(lds, c, f, 1);
  if (s2) {
    stage_w<BUFI>(lds, c, p2.w, 0);
    stage_x<BUFI, CONV>(lds, c, p2, 0);
    stage_w<BUFI>(lds, c, p2.w, 1);
    ED_WAIT_VM(6);
  } else {
    ED_WAIT_VM(0);
  }
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 1, 1>(acc, f);
  mma16<T, 1, 0>(acc, f);
  ED_BARRIER();
}

// HALF tile (round 5): a column tile of the plain projection / convolution whose second 128-column half lies entirely beyond the output
// width (N mod 256 in (0, 128]: the last tile of N = 320, 640, 1920) runs the first half only -- 256 x 128 outputs: no DMAs and no
// fragment reads for the second half's W rows, 32 instead of 64 MFMAs per K tile and wave (those 32 multiplied zero-filled rows and
// their results were never stored: 25 % of the N = 320 convolutions' MFMAs, 17 % at N = 640).  Four barrier intervals per K tile:
//   R1  W rows + x m-half 0 (12 fragment reads); x m-half 1 of tile + 1 -> the other buffer (its copy was last read in the previous R2)
//   M1  16 MFMAs (m half 0)
//   R2  x m-half 1 (8 reads); W rows and x m-half 0 of tile + 2 -> this buffer (both last read in R1, by either wave row one interval ago);
//       vmcnt(4): all of tile + 1 has landed, those two half tiles stay in flight
//   M2  16 MFMAs (m half 1, the W fragments of R1)
// A wave retires its fragment reads before its barrier (the other wave row is busy with 16 MFMAs meanwhile).  Replayed under both
// adversarial timings in tools/emulate_gemm_kernel.py --half (--break half_raw weakens the count and is caught).
template <class T, int BUFI, int CONV>
__device__ __forceinline__ void tile_phases_half(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], int tile, bool s1, bool s2,
                                                 KPos p1, KPos p2) {
  read_w<T, BUFI, 0>(lds, c, f);
  read_x<T, BUFI>(lds, c, f, 0);
  if (s1) stage_x<BUFI ^ 1, CONV>(lds, c, p1, 1);
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 0, 0>(acc, f);
  ED_BARRIER();
  read_x<T, BUFI>(lds, c, f, 1);
  if (s2) {
    stage_w<BUFI>(lds, c, p2.w, 0);
    stage_x<BUFI, CONV>(lds, c, p2, 0);
    ED_WAIT_VM(4);
  } else {
    ED_WAIT_VM(0);
  }
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 1, 0>(acc, f);
  ED_BARRIER();
}

// ROWS (round 6): a 128-row tile -- each wave row runs its m-half 0 only (64 rows), both column halves -- for grids that leave the chip
// under-filled with 256-row tiles: the batch-6 forward's 32 x 32 convolutions are 120 tiles on 256 CUs (240 half-height ones fill 94 % of one
// round at 0.72 of a tile's time: 1.35-1.45 x, profiles/r6_s4_gemm_rows_mode.jsonl), a 3-row per-rank forward of the multi-GPU layout 60.  Four barrier intervals per K tile, the structure of
// tile_phases_half with the roles of "x m-half 1" and "W gate rows" exchanged (so the LDS hazard analysis is that one's; replay:
// tools/emulate_gemm_kernel.py --rows):
//   R1  W value rows + x m-half 0 (12 fragment reads); W GATE rows of tile + 1 -> the other buffer (its copy was last read in the previous R2)
//   M1  16 MFMAs (m half 0 x value)
//   R2  W gate rows (4 reads); W value rows and x m-half 0 of tile + 2 -> this buffer (both last read in R1, by either wave row one interval
//       ago); vmcnt(4): all of tile + 1 has landed, those two half tiles stay in flight
//   M2  16 MFMAs (m half 0 x gate)
template <class T, int BUFI, int CONV>
__device__ __forceinline__ void tile_phases_rows(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], bool s1, bool s2, KPos p1,
                                                 KPos p2) {
  read_w<T, BUFI, 0>(lds, c, f);
  read_x<T, BUFI>(lds, c, f, 0);
  if (s1) stage_w<BUFI ^ 1>(lds, c, p1.w, 1);
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 0, 0>(acc, f);
  ED_BARRIER();
  read_w<T, BUFI, 1>(lds, c, f);
  if (s2) {
    stage_w<BUFI>(lds, c, p2.w, 0);
    stage_x<BUFI, CONV>(lds, c, p2, 0);
    ED_WAIT_VM(4);
  } else {
    ED_WAIT_VM(0);
  }
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 0, 1>(acc, f);
  ED_BARRIER();
}

// EPI 0: GEGLU -- W is [2 I, K], the two 128-row halves of the tile are value rows n0.. and gate rows I + n0.., out is [M, I]
// EPI 1: plain projection + bias -- W is [I, K] (I = output columns), the halves are rows n0.. and n0 + 128.., out is [M, I];
//        the same main loop, kept so that the schedule can be timed against hipBLASLt on every projection of the block
// CONV (with EPI 1): x is an NHWC image [B, img_h, img_w, Cin], W is [I, 3, 3, Cin] (a torch Conv2d weight in channels_last memory
//        format), K = 9 Cin, M = B img_h img_w, out is NHWC [M, I]: 3x3, stride 1, zero padding 1 as an implicit GEMM -- only the
//        addresses of the A operand differ (stage_x)
// Plain-projection epilogue extras (EPI 1; each optional): row_bias [M / rows_per_sample, I] is added to every row of its sample
// (ResnetBlock2D's time-embedding add), residual [M, I] element-wise (the block's closing residual / a transformer's skip):
//     out = round16(acc + bias[n] + row_bias[m / rows_per_sample, n] + residual[m, n])        one rounding, fp32 sums
// ADD (EPI 1): the launch has at least one epilogue addend (row_bias / residual).  Without the flag the epilogue converted and added two
// pairs of zero vectors per store even when both pointers were null: 1.7-3.7 % of a plain projection (profiles/r4_s15_*, r4_s17_*).
// OUT32 (EPI 1, round 5: the fp32 VAE's convolutions on split 16-bit operands, vae_kernels.hip): `bias`, `residual` and `out` point to
// fp32 data ([I], [M, I], [M, I]); out = out_scale * acc + bias + residual with NO rounding to 16 bits -- the accumulators already are
// the fp32 result; out_scale (a power of two) undoes the pre-scaling of the split weights.  row_bias is not used.
// TWO: the long-K loop (tile_phases_two) instead of the 8-phase one -- a separate instantiation (both loops in one kernel made hipcc spill).
// ROWS: 128-row tiles (tile_phases_rows) -- a separate instantiation the launcher picks for under-filled grids (launch: rows_mode_pays).
template <class T, int EPI, int CONV, bool ADD = true, bool OUT32 = false, bool TWO = false, bool ROWS = false>
__global__ void __launch_bounds__(512, 2)
k_gemm_8phase(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
              const uint16_t* __restrict__ row_bias, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int M,
              int K, int I, int n_blocks_n, int n_blocks, int img_h, int img_w, int rows_per_sample, float out_scale,
              const float* __restrict__ act_absmax) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * BUF];

  // workgroup -> (row block, column block).  Id b runs on XCD b % 8: give every XCD a contiguous run of tile ids, and walk
  // the ids in groups of 8 row blocks (rows fastest inside a group), so the 32 workgroups an XCD runs at a time are 8 row
  // blocks x 4 column blocks: 12 operand slabs for 32 tiles in that XCD's L2 instead of 22..34 (bijective for any grid)
  const int bid = blockIdx.x;
  const int q = n_blocks >> 3, r = n_blocks & 7, xcd = bid & 7;
  const int tid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int n_blocks_m = n_blocks / n_blocks_n;
  const int per_group = 8 * n_blocks_n, grp = tid / per_group, first = grp * 8;
  const int rows_here = n_blocks_m - first < 8 ? n_blocks_m - first : 8;
  constexpr int TBM = ROWS ? BM / 2 : BM;            // rows per tile
  constexpr int WROWS = TBM / 2;                     // rows per wave row
  constexpr int NMB = WROWS / 16;                    // 16-row blocks per wave row
  const int m0 = (first + (tid % per_group) % rows_here) * TBM;
  const int n0 = ((tid % per_group) / rows_here) * (EPI == 0 ? BN : 2 * BN);
  const int gap = EPI == 0 ? I : BN;                 // W rows (= output columns for EPI 1) from the first half to the second

  const int lane = threadIdx.x & 63;
  Ctx c;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wrow = c.wave >> 2, wcol = c.wave & 3;

  // a lane's 16 bytes of a subtile: position 16 lane, element (row, k byte) after the swizzle
  const int ps = swz(16 * lane), srow = ps >> 6, skb = ps & 63;
  const int row_bytes = K * 2;                                   // a W row; for a GEMM also an x row
  const int x_row_bytes = CONV ? row_bytes / 9 : row_bytes;      // CONV: one pixel's Cin values
  c.xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((int64_t)(CONV == CONV_UP2X ? M / 4 : CONV >= CONV_S2 ? 4 * (int64_t)M : M) * x_row_bytes),
                                           0x00020000);
  c.wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((int64_t)(EPI == 0 ? 2 : 1) * I * row_bytes), 0x00020000);
  const int xrow0 = m0 + ((c.wave & 3) + (ROWS ? 4 : 8) * (c.wave >> 2)) * 16 + srow;   // the lane's row of m half 0; m half 1 is 64 rows on (ROWS: none)
  c.x_voff[0] = xrow0 * x_row_bytes + skb;
  c.x_voff[1] = c.x_voff[0] + 64 * x_row_bytes;
  c.img_w = CONV == CONV_UP2X ? img_w / 2 : CONV >= CONV_S2 ? 2 * img_w : img_w;    // width of the SOURCE image (stage_x's tap offsets)
  c.cin2 = x_row_bytes;
  c.cpt = CONV ? K / (9 * BK) : 1;
  c.px_mask[0] = c.px_mask[1] = 0;
  if (CONV) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = xrow0 + 64 * h;
      if (m < M) {
        const int rem = m % (img_h * img_w), py = rem / img_w, px = rem - py * img_w;
        int mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (CONV == CONV_S2) {       // input pixel (2 py + dy, 2 px + dx) of the 2 img_h x 2 img_w input: only -1 can fall outside
            if (2 * py + t / 3 - 1 >= 0 && 2 * px + t % 3 - 1 >= 0) mask |= 1 << t;
          } else if (CONV == CONV_S2P0) {   // input pixel (2 py + ty, 2 px + tx), ty / tx = 0..2: only the row / column past the edge falls outside
            if (2 * py + t / 3 < 2 * img_h && 2 * px + t % 3 < 2 * img_w) mask |= 1 << t;
          } else {
            const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
            if (yy >= 0 && yy < img_h && xx >= 0 && xx < img_w) mask |= 1 << t;
          }
        }
        c.px_mask[h] = mask;
        if (CONV == CONV_UP2X) {       // the lane's source pixel and the parities of its output pixel
          const int hs = img_h / 2, ws = img_w / 2;
          c.px_mask[h] = mask | ((py & 1) << 16) | ((px & 1) << 17);
          c.x_voff[h] = ((m / (img_h * img_w)) * (hs * ws) + (py >> 1) * ws + (px >> 1)) * x_row_bytes + skb;
        }
        if (CONV >= CONV_S2) {         // the lane's base input pixel: (2 py, 2 px), or (2 py + 1, 2 px + 1) for the pad-(0, 1, 0, 1) variant
          const int o1 = CONV == CONV_S2P0 ? 1 : 0;
          c.x_voff[h] = (((m / (img_h * img_w)) * (2 * img_h) + 2 * py + o1) * (2 * img_w) + 2 * px + o1) * x_row_bytes + skb;
        }
      }
    }
  }
  // LDS row (row group w, row i) of the value / gate half holds W row n0 + 32 (w >> 1) + 8 (i >> 2) + (i & 3) + 4 (w & 1)
  c.w_voff[0] = (n0 + 32 * (c.wave >> 1) + 8 * (srow >> 2) + (srow & 3) + 4 * (c.wave & 1)) * row_bytes + skb;
  c.w_voff[1] = c.w_voff[0] + gap * row_bytes;
  // fragment read: row lane & 15, k bytes 16 (lane >> 4), swizzled; this wave's first row group
  const int rd = swz((lane & 15) * 64 + (lane >> 4) * 16);
  c.xrd = rd + wrow * 8 * (2 * SUB);
  c.wrd = rd + W_REGION + wcol * 2 * (2 * SUB);

  // bias: lane holds columns ncol + 4 f + j (f = fragment, j = accumulator register): 8 consecutive values, one 16-byte load each
  const int ncol = n0 + 32 * wcol + 8 * (lane >> 4);
  // The two loads are issued BEFORE the first LDS-DMA and their values are first touched in the epilogue: they are the oldest
  // entries of the VM queue, so every counted vmcnt of the schedule retires them on the way (in-order return), and the wave does
  // not spend an L2 / HBM round trip waiting for 32 bytes before it may start staging (round 4; until then the values were
  // converted -- i.e. waited for -- right here).
  u32x4 bias_v = {0, 0, 0, 0}, bias_g = {0, 0, 0, 0};
  if (bias && !OUT32) {
    if (EPI == 0 || ncol < I) bias_v = *reinterpret_cast<const u32x4*>(bias + ncol);
    if (EPI == 0 || ncol + gap < I) bias_g = *reinterpret_cast<const u32x4*>(bias + gap + ncol);
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  Frags<T> f;

  const int nt = K / BK;
  // prologue: all of tile 0, then the three half tiles of tile 1 the loop does not stage itself
  const KPos p0 = {0, 0, 0, 0};
  KPos pa = k_next<CONV>(p0, c.cpt);      // position of tile t + 1
  KPos pb = k_next<CONV>(pa, c.cpt);      // position of tile t + 2
  const bool half = !ROWS && EPI == 1 && n0 + BN >= I;     // wave-uniform: nothing of this tile's second half is inside the output (tile_phases_half)
  if (ROWS) {
    stage_w<0>(lds, c, 0, 0);
    stage_x<0, CONV>(lds, c, p0, 0);
    stage_w<0>(lds, c, 0, 1);
    if (nt > 1) {
      stage_w<1>(lds, c, pa.w, 0);
      stage_x<1, CONV>(lds, c, pa, 0);
      ED_WAIT_VM(4);              // all of tile 0 (6 DMAs); tile 1's two half tiles stay in flight
    } else {
      ED_WAIT_VM(0);
    }
    ED_BARRIER();
    if (wrow == 1) ED_BARRIER();
    int tr = 0;
    for (; tr + 1 < nt; tr += 2) {
      tile_phases_rows<T, 0, CONV>(lds, c, f, acc, true, tr + 2 < nt, pa, pb);
      pa = pb;
      pb = k_next<CONV>(pb, c.cpt);
      tile_phases_rows<T, 1, CONV>(lds, c, f, acc, tr + 2 < nt, tr + 3 < nt, pa, pb);
      pa = pb;
      pb = k_next<CONV>(pb, c.cpt);
    }
    if (tr < nt) tile_phases_rows<T, 0, CONV>(lds, c, f, acc, false, false, pa, pb);
    if (wrow == 0) ED_BARRIER();
  } else if (half) {
    stage_w<0>(lds, c, 0, 0);
    stage_x<0, CONV>(lds, c, p0, 0);
    stage_x<0, CONV>(lds, c, p0, 1);
    if (nt > 1) {
      stage_w<1>(lds, c, pa.w, 0);
      stage_x<1, CONV>(lds, c, pa, 0);
      ED_WAIT_VM(4);              // all of tile 0 (6 DMAs); tile 1's two half tiles stay in flight
    } else {
      ED_WAIT_VM(0);
    }
    ED_BARRIER();
    if (wrow == 1) ED_BARRIER();
    int th = 0;
    for (; th + 1 < nt; th += 2) {
      tile_phases_half<T, 0, CONV>(lds, c, f, acc, th, true, th + 2 < nt, pa, pb);
      pa = pb;
      pb = k_next<CONV>(pb, c.cpt);
      tile_phases_half<T, 1, CONV>(lds, c, f, acc, th + 1, th + 2 < nt, th + 3 < nt, pa, pb);
      pa = pb;
      pb = k_next<CONV>(pb, c.cpt);
    }
    if (th < nt) tile_phases_half<T, 0, CONV>(lds, c, f, acc, th, false, false, pa, pb);
    if (wrow == 0) ED_BARRIER();
  } else {
  stage_w<0>(lds, c, 0, 0);
  stage_x<0, CONV>(lds, c, p0, 0);
  stage_w<0>(lds, c, 0, 1);
  stage_x<0, CONV>(lds, c, p0, 1);
  constexpr bool two = TWO;               // long K: 4 barrier intervals per K tile (tile_phases_two)
  const bool early = nt >= 3 && !two;     // (every real shape: K >= 320)
  if (nt > 1) {
    stage_w<1>(lds, c, pa.w, 0);
    stage_x<1, CONV>(lds, c, pa, 0);
    stage_w<1>(lds, c, pa.w, 1);
    if (early) ED_WAIT_VM(10);    // only what phase 1 of tile 0 reads (tile_phases<.., FIRST>)
    else ED_WAIT_VM(6);
  } else {
    ED_WAIT_VM(0);
  }
  ED_BARRIER();
  if (wrow == 1) ED_BARRIER();    // second wave row runs half a phase behind

  int t = 0;
  if (two) {
    for (; t + 1 < nt; t += 2) {
      tile_phases_two<T, 0, CONV>(lds, c, f, acc, t, true, t + 2 < nt, pa, pb);
      pa = pb;
      pb = k_next<CONV>(pb, c.cpt);
      tile_phases_two<T, 1, CONV>(lds, c, f, acc, t + 1, t + 2 < nt, t + 3 < nt, pa, pb);
      pa = pb;
      pb = k_next<CONV>(pb, c.cpt);
    }
    if (t < nt) tile_phases_two<T, 0, CONV>(lds, c, f, acc, t, false, false, pa, pb);
    t = nt;
  }
  if (early) {                    // the first pair of K tiles, tile 0 in its early-start form (s1 = s2 = true: nt >= 3)
    tile_phases<T, 0, CONV, true>(lds, c, f, acc, 0, true, true, pa, pb);
    pa = pb;
    pb = k_next<CONV>(pb, c.cpt);
    tile_phases<T, 1, CONV>(lds, c, f, acc, 1, true, 3 < nt, pa, pb);
    pa = pb;
    pb = k_next<CONV>(pb, c.cpt);
    t = 2;
  }
  for (; t + 1 < nt; t += 2) {
    tile_phases<T, 0, CONV>(lds, c, f, acc, t, true, t + 2 < nt, pa, pb);
    pa = pb;
    pb = k_next<CONV>(pb, c.cpt);
    tile_phases<T, 1, CONV>(lds, c, f, acc, t + 1, t + 2 < nt, t + 3 < nt, pa, pb);
    pa = pb;
    pb = k_next<CONV>(pb, c.cpt);
  }
  if (t < nt) tile_phases<T, 0, CONV>(lds, c, f, acc, t, false, false, pa, pb);
  if (wrow == 0) ED_BARRIER();    // pair the extra barrier of the second wave row
  }   // (!half)

  if (OUT32) {
    // fp32 epilogue: a lane's 8 consecutive columns of a row are two 16-byte fp32 vectors (fragment f = columns ncol + 4 f ..)
    const float* biasf = reinterpret_cast<const float*>(bias);
    const float* resf = reinterpret_cast<const float*>(residual);
    float* outf = reinterpret_cast<float*>(out);
    if (act_absmax) {     // the raw-stream split scaled the activation by 2^-e (vae_kernels.hip, split_exponent): undo it, exactly
      const int E = (int)((__builtin_bit_cast(uint32_t, *act_absmax) >> 23) & 0xffu) - 127;
      const int e = E > 14 ? E - 14 : 0;
      out_scale *= __builtin_bit_cast(float, (uint32_t)(127 + e) << 23);
    }
    const bool col_v = ncol < I, col_g = ncol + gap < I;
    f32x4 b32[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
    if (biasf) {
#pragma unroll
      for (int fr = 0; fr < 2; ++fr) {
        if (col_v) b32[0][fr] = *reinterpret_cast<const f32x4*>(biasf + ncol + 4 * fr);
        if (col_g) b32[1][fr] = *reinterpret_cast<const f32x4*>(biasf + ncol + gap + 4 * fr);
      }
    }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const int m = m0 + WROWS * wrow + 16 * mb + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {      // value half / second ("gate") half of the 256 columns
        if (!(hf == 0 ? col_v : col_g)) continue;
        const int64_t o = (int64_t)m * I + ncol + hf * gap;
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
          f32x4 v = acc[mb][2 * hf + fr] * out_scale + b32[hf][fr];
          if (ADD && resf) v += *reinterpret_cast<const f32x4*>(resf + o + 4 * fr);
          *reinterpret_cast<f32x4*>(outf + o + 4 * fr) = v;
        }
      }
    }
    return;
  }
  // epilogue: one 16-byte store per (lane, 16-row block): 8 consecutive columns of one row
  float bv[2][4], bg[2][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bv[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_v[e >> 1] >> (16 * (e & 1))));
    bg[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_g[e >> 1] >> (16 * (e & 1))));
  }
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) {
    const int m = m0 + WROWS * wrow + 16 * mb + (lane & 15);
    if (EPI == 0) {
      uint32_t pk[4];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const float o0 = geglu_one(acc[mb][nf][2 * jj] + bv[nf][2 * jj], acc[mb][2 + nf][2 * jj] + bg[nf][2 * jj]);
          const float o1 = geglu_one(acc[mb][nf][2 * jj + 1] + bv[nf][2 * jj + 1], acc[mb][2 + nf][2 * jj + 1] + bg[nf][2 * jj + 1]);
          pk[nf * 2 + jj] = T::pack2(o0, o1);
        }
      if (m < M) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol) = u32x4{pk[0], pk[1], pk[2], pk[3]};
    } else {
      const bool ok_v = m < M && ncol < I, ok_g = m < M && ncol + gap < I;
      // the (optional) addends of this row's two 8-column groups: 16-byte loads, zeros when absent / out of range
      u32x4 av[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, ag[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
      if (ADD && row_bias) {
        const int64_t rb = (int64_t)(m / rows_per_sample) * I;
        if (ok_v) av[0] = *reinterpret_cast<const u32x4*>(row_bias + rb + ncol);
        if (ok_g) ag[0] = *reinterpret_cast<const u32x4*>(row_bias + rb + ncol + gap);
      }
      if (ADD && residual) {
        if (ok_v) av[1] = *reinterpret_cast<const u32x4*>(residual + (int64_t)m * I + ncol);
        if (ok_g) ag[1] = *reinterpret_cast<const u32x4*>(residual + (int64_t)m * I + ncol + gap);
      }
      uint32_t pv[4], pg[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {   // column e of the group = fragment e >> 2, accumulator register e & 3
        float v0 = acc[mb][e >> 2][e & 3] + bv[e >> 2][e & 3], v1 = acc[mb][e >> 2][(e & 3) + 1] + bv[e >> 2][(e & 3) + 1];
        float g0 = acc[mb][2 + (e >> 2)][e & 3] + bg[e >> 2][e & 3], g1 = acc[mb][2 + (e >> 2)][(e & 3) + 1] + bg[e >> 2][(e & 3) + 1];
        if (ADD) {
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            v0 += T::to_f32((uint16_t)av[a][e >> 1]), v1 += T::to_f32((uint16_t)(av[a][e >> 1] >> 16));
            g0 += T::to_f32((uint16_t)ag[a][e >> 1]), g1 += T::to_f32((uint16_t)(ag[a][e >> 1] >> 16));
          }
        }
        pv[e >> 1] = (uint32_t)T::from_f32(v0) | ((uint32_t)T::from_f32(v1) << 16);
        pg[e >> 1] = (uint32_t)T::from_f32(g0) | ((uint32_t)T::from_f32(g1) << 16);
      }
      if (ok_v) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol) = u32x4{pv[0], pv[1], pv[2], pv[3]};
      if (ok_g) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol + gap) = u32x4{pg[0], pg[1], pg[2], pg[3]};
    }
  }
}

// ---- persistent GEGLU (round 5) -----------------------------------------------------------------------------------------------------
// The same main loop as k_gemm_8phase<T, 0, false>, one workgroup per CU walking the tile ids b, b + G, b + 2G, ... (G a multiple of 8:
// a workgroup stays on its XCD and an XCD's workgroups walk its 8 x 4 tile groups together).  The NEXT tile's bias loads and 14
// prologue LDS-DMAs are issued right after the barrier pair that ends the current tile, BEFORE its epilogue -- both LDS buffers are
// free there (every fragment read precedes the MFMAs that precede the barrier) and the GELU epilogue (~2.5 us) touches registers and
// global memory only, so the next fill runs under it.  Same arithmetic in the same order: bit-identical to the one-tile-per-workgroup
// kernel (round 4, tools/gemm_persist: +2.5 ... 5.6 % on the UNet's GEGLU shapes).  LDS hazards of the overlap replayed in
// tools/emulate_gemm_kernel.py --persist.
struct TilePos {
  int m0, n0;
};
__device__ __forceinline__ TilePos geglu_tile_of(int bid, int n_blocks, int n_blocks_n) {
  const int q = n_blocks >> 3, r = n_blocks & 7, xcd = bid & 7;
  const int tid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int n_blocks_m = n_blocks / n_blocks_n;
  const int per_group = 8 * n_blocks_n, grp = tid / per_group, first = grp * 8;
  const int rows_here = n_blocks_m - first < 8 ? n_blocks_m - first : 8;
  return TilePos{(first + (tid % per_group) % rows_here) * BM, ((tid % per_group) / rows_here) * BN};
}

template <class T>
__global__ void __launch_bounds__(512, 2)
k_geglu_persist(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
                uint16_t* __restrict__ out, int M, int K, int I, int n_blocks_n, int n_blocks) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * BUF];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wrow = wave >> 2, wcol = wave & 3;
  const int ps = swz(16 * lane), srow = ps >> 6, skb = ps & 63;
  const int row_bytes = K * 2;
  const int rd = swz((lane & 15) * 64 + (lane >> 4) * 16);
  const int nt = K / BK;
  const bool early = nt >= 3;

  auto setup = [&](TilePos tp) {
    Ctx c;
    c.wave = wave;
    c.xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((int64_t)M * row_bytes), 0x00020000);
    c.wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((int64_t)2 * I * row_bytes), 0x00020000);
    c.x_voff[0] = (tp.m0 + ((wave & 3) + 8 * (wave >> 2)) * 16 + srow) * row_bytes + skb;
    c.x_voff[1] = c.x_voff[0] + 64 * row_bytes;
    c.img_w = 0, c.cin2 = row_bytes, c.cpt = 1;
    c.px_mask[0] = c.px_mask[1] = 0;
    c.w_voff[0] = (tp.n0 + 32 * (wave >> 1) + 8 * (srow >> 2) + (srow & 3) + 4 * (wave & 1)) * row_bytes + skb;
    c.w_voff[1] = c.w_voff[0] + I * row_bytes;
    c.xrd = rd + wrow * 8 * (2 * SUB);
    c.wrd = rd + W_REGION + wcol * 2 * (2 * SUB);
    return c;
  };
  auto load_bias = [&](int ncol, u32x4& bv_raw, u32x4& bg_raw) {
    bv_raw = u32x4{0, 0, 0, 0}, bg_raw = u32x4{0, 0, 0, 0};
    if (bias) {
      bv_raw = *reinterpret_cast<const u32x4*>(bias + ncol);
      bg_raw = *reinterpret_cast<const u32x4*>(bias + I + ncol);
    }
  };
  auto issue_prologue = [&](const Ctx& c) {   // all of K tile 0, then the three half tiles of K tile 1 the loop does not stage itself
    const KPos p0 = {0, 0, 0, 0}, p1 = {1, 0, 0, 1};
    stage_w<0>(lds, c, 0, 0);
    stage_x<0, false>(lds, c, p0, 0);
    stage_w<0>(lds, c, 0, 1);
    stage_x<0, false>(lds, c, p0, 1);
    if (nt > 1) {
      stage_w<1>(lds, c, 1, 0);
      stage_x<1, false>(lds, c, p1, 0);
      stage_w<1>(lds, c, 1, 1);
    }
  };

  int bid = blockIdx.x;
  TilePos tp = geglu_tile_of(bid, n_blocks, n_blocks_n);
  Ctx c = setup(tp);
  u32x4 bias_v, bias_g;
  load_bias(tp.n0 + 32 * wcol + 8 * (lane >> 4), bias_v, bias_g);
  issue_prologue(c);

  for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frags<T> f;
    KPos pa = {1, 0, 0, 1}, pb = {2, 0, 0, 2};
    if (nt > 1) {
      if (early) ED_WAIT_VM(10);
      else ED_WAIT_VM(6);
    } else {
      ED_WAIT_VM(0);
    }
    ED_BARRIER();
    if (wrow == 1) ED_BARRIER();

    int t = 0;
    if (early) {
      tile_phases<T, 0, false, true>(lds, c, f, acc, 0, true, true, pa, pb);
      pa = pb, pb = KPos{pb.tile + 1, 0, 0, pb.tile + 1};
      tile_phases<T, 1, false>(lds, c, f, acc, 1, true, 3 < nt, pa, pb);
      pa = pb, pb = KPos{pb.tile + 1, 0, 0, pb.tile + 1};
      t = 2;
    }
    for (; t + 1 < nt; t += 2) {
      tile_phases<T, 0, false>(lds, c, f, acc, t, true, t + 2 < nt, pa, pb);
      pa = pb, pb = KPos{pb.tile + 1, 0, 0, pb.tile + 1};
      tile_phases<T, 1, false>(lds, c, f, acc, t + 1, t + 2 < nt, t + 3 < nt, pa, pb);
      pa = pb, pb = KPos{pb.tile + 1, 0, 0, pb.tile + 1};
    }
    if (t < nt) tile_phases<T, 0, false>(lds, c, f, acc, t, false, false, pa, pb);
    if (wrow == 0) ED_BARRIER();

    // the VM queue is empty here (the last K tiles end with vmcnt(0)): converting the bias costs no wait
    const int m0 = tp.m0, ncol = tp.n0 + 32 * wcol + 8 * (lane >> 4);
    float bv[2][4], bg[2][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bv[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_v[e >> 1] >> (16 * (e & 1))));
      bg[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_g[e >> 1] >> (16 * (e & 1))));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bv[e >> 2][e & 3]), "+v"(bg[e >> 2][e & 3]));

    const int next = bid + (int)gridDim.x;
    const bool has_next = next < n_blocks;     // wave-uniform
    Ctx c2 = c;
    TilePos tp2 = tp;
    if (has_next) {     // next tile: addresses, its bias loads (the oldest entries of the queue), then its prologue DMAs
      tp2 = geglu_tile_of(next, n_blocks, n_blocks_n);
      c2 = setup(tp2);
      load_bias(tp2.n0 + 32 * wcol + 8 * (lane >> 4), bias_v, bias_g);
      __builtin_amdgcn_sched_barrier(0);     // the two loads stay AHEAD of the DMAs (left alone, hipcc sinks them below the epilogue)
      issue_prologue(c2);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const int m = m0 + 128 * wrow + 16 * mb + (lane & 15);
      uint32_t pk[4];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const float o0 = geglu_one(acc[mb][nf][2 * jj] + bv[nf][2 * jj], acc[mb][2 + nf][2 * jj] + bg[nf][2 * jj]);
          const float o1 = geglu_one(acc[mb][nf][2 * jj + 1] + bv[nf][2 * jj + 1], acc[mb][2 + nf][2 * jj + 1] + bg[nf][2 * jj + 1]);
          pk[nf * 2 + jj] = T::pack2(o0, o1);
        }
      if (m < M) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol) = u32x4{pk[0], pk[1], pk[2], pk[3]};
    }
    if (!has_next) break;
    bid = next, tp = tp2, c = c2;
  }
}

}  // namespace

// CUs of the current device rounded down to a multiple of 8 (one 128-KiB-LDS workgroup per CU; cached per device)
static int gemm_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) return 0;
    cus[dev] = n / 8 * 8;
  }
  return cus[dev];
}
constexpr double ROWS_TILE_COST = 0.72;   // a round of 128-row tiles relative to a round of 256-row ones, measured: 0.71-0.72
                                          // (profiles/r6_s4_gemm_rows_mode.jsonl: 120 tiles 201.9 -> 144.8 us; 288 tiles 188.8 (2 rounds) vs 200.6 (3))

// C ABI (include/elastic_hip.h).  Returns 0, a hipError_t, or hipErrorInvalidValue for a shape the kernel does not take.
template <int EPI, int CONV, bool OUT32 = false>
static int launch(const void* x, const void* w, const void* bias, const void* row_bias, const void* residual, void* out, int dtype,
                  int64_t M, int K, int I, int img_h, int img_w, int rows_per_sample, void* stream, float out_scale = 1.0f,
                  const float* act_absmax = nullptr) {
  if (M == 0) return 0;
  const int bad = (int)hipErrorInvalidValue;
  if (M < 0 || K % BK != 0 || K < BK || I <= 0 || (EPI == 0 ? I % BN != 0 : I % 8 != 0)) return bad;
  if (CONV && (K % (9 * BK) != 0 || img_h <= 0 || img_w <= 0 || M % ((int64_t)img_h * img_w) != 0)) return bad;
  if (CONV == CONV_UP2X && (img_h % 2 != 0 || img_w % 2 != 0 || row_bias || residual)) return bad;   // (H, W = the OUTPUT size; bias only)
  if (CONV >= CONV_S2 && (row_bias || residual || 4 * M * (int64_t)(K / 9) * 2 >= 0x7ffffff0ll)) return bad;   // (H, W = the OUTPUT size; the input is 2H x 2W)
  if (row_bias && (rows_per_sample <= 0 || M % rows_per_sample != 0)) return bad;
  if (M * (int64_t)(CONV ? K / 9 : K) * 2 >= 0x7ffffff0ll || (int64_t)(EPI == 0 ? 2 : 1) * I * K * 2 >= 0x7ffffff0ll) return bad;   // 32-bit buffer offsets
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)row_bias | (uintptr_t)residual) & 15u)) return bad;
  const int nbn = EPI == 0 ? I / BN : (I + 2 * BN - 1) / (2 * BN);
  const int64_t nb256 = ((M + BM - 1) / BM) * nbn, nb128 = ((M + BM / 2 - 1) / (BM / 2)) * nbn;
  if (nb128 >= (1ll << 31) || M >= (1ll << 31)) return bad;
  hipStream_t s = (hipStream_t)stream;
  const bool add = EPI == 1 && (row_bias || residual);
  const bool two = CONV && K / BK >= TWO_MIN_TILES;     // the long-K loop: convolutions only (measured negative on the plain projections)
  // 128-row tiles (ROWS) where they finish sooner: one workgroup per CU, so a grid runs in rounds of `cus` tiles; a half-height tile costs
  // ROWS_TILE_COST of a full one (W staged and read for half the rows).  120 full tiles (the batch-6 forward's 32 x 32 convolutions):
  // 1 round vs 0.72; 288: 2 rounds vs 3 x 0.72 -- stays (measured 0.94 x); 400: 2 vs 4 x 0.72 -- stays.  ED_GEMM_ROWS=0 / 1 forces it off / on (measurement only).
  bool rows = false;
  if (!OUT32 && EPI == 1 && CONV < CONV_UP2X) {
    const int cus = gemm_cus();
    const double r256 = (double)((nb256 + cus - 1) / cus), r128 = (double)((nb128 + cus - 1) / cus) * ROWS_TILE_COST;
    rows = cus > 0 && r128 < r256 - 1e-9;
    if (const char* e = getenv("ED_GEMM_ROWS")) rows = e[0] == '1' ? true : (e[0] == '0' ? false : rows);
  }
  const int64_t nb = rows ? nb128 : nb256;
#define ED_LAUNCH1(TT, ADD_, TWO_, ROWS_)                                                                                            \
  k_gemm_8phase<TT, EPI, CONV, ADD_, OUT32, TWO_, ROWS_><<<(int)nb, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, \
                                                             (const uint16_t*)row_bias, (const uint16_t*)residual, (uint16_t*)out,  \
                                                             (int)M, K, I, nbn, (int)nb, img_h, img_w, rows_per_sample > 0 ? rows_per_sample : 1, out_scale, \
                                                             act_absmax)
#define ED_LAUNCH(TT, ADD_)                                   \
  do {                                                        \
    if constexpr (OUT32 || EPI == 0) {                        \
      if constexpr (CONV) {                                   \
        if (two) ED_LAUNCH1(TT, ADD_, true, false);           \
        else ED_LAUNCH1(TT, ADD_, false, false);              \
      } else {                                                \
        ED_LAUNCH1(TT, ADD_, false, false);                   \
      }                                                       \
    } else if (CONV < CONV_UP2X && rows) {                    \
      if constexpr (CONV < CONV_UP2X) ED_LAUNCH1(TT, ADD_, false, true);   \
    } else if constexpr (CONV) {                              \
      if (two) ED_LAUNCH1(TT, ADD_, true, false);             \
      else ED_LAUNCH1(TT, ADD_, false, false);                \
    } else {                                                  \
      ED_LAUNCH1(TT, ADD_, false, false);                     \
    }                                                         \
  } while (0)
  if constexpr (OUT32) {       // split-fp16 operands only (bf16's 8 significand bits would need three terms per operand)
    if (dtype != ED_F16 || row_bias) return bad;
    if constexpr (CONV >= CONV_UP2X) {
      ED_LAUNCH(HF, false);      // (bias only: no ADD instantiation)
    } else {
      if (add) ED_LAUNCH(HF, true);
      else ED_LAUNCH(HF, false);
    }
  } else {
    if constexpr (CONV >= CONV_UP2X) {   // the up / down-samplers' convolutions: bias only -- no ADD instantiation
      if (dtype == ED_BF16) ED_LAUNCH(BF, false);
      else if (dtype == ED_F16) ED_LAUNCH(HF, false);
      else return bad;
    } else if (dtype == ED_BF16) {
      if (EPI == 0 || add) ED_LAUNCH(BF, true);
      else ED_LAUNCH(BF, EPI == 0);      // (false for the plain projection / convolution; no second GEGLU instantiation)
    } else if (dtype == ED_F16) {
      if (EPI == 0 || add) ED_LAUNCH(HF, true);
      else ED_LAUNCH(HF, EPI == 0);
    } else {
      return bad;
    }
  }
#undef ED_LAUNCH
#undef ED_LAUNCH1
  return (int)hipGetLastError();
}

extern "C" {

int ed_geglu_gemm(const void* x, const void* w, const void* bias, void* out, int dtype, int64_t M, int K, int I, void* stream) {
  if (M == 0) return 0;
  const int bad = (int)hipErrorInvalidValue;
  if (M < 0 || K % BK != 0 || K < BK || I <= 0 || I % BN != 0) return bad;
  if (M * (int64_t)K * 2 >= 0x7ffffff0ll || (int64_t)2 * I * K * 2 >= 0x7ffffff0ll) return bad;   // 32-bit buffer offsets
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias) & 15u)) return bad;
  const int nbn = I / BN;
  const int64_t nb = ((M + BM - 1) / BM) * nbn;
  if (nb >= (1ll << 31) || M >= (1ll << 31)) return bad;
  // one workgroup per CU (128 KiB of LDS each); a multiple of 8 keeps a workgroup on its XCD for all of its tiles
  const int ncu = gemm_cus();
  if (ncu <= 0) return (int)hipErrorInvalidDevice;
  const int grid = nb < ncu ? (int)nb : ncu;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ED_BF16)
    k_geglu_persist<BF><<<grid, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)out, (int)M, K, I, nbn, (int)nb);
  else if (dtype == ED_F16)
    k_geglu_persist<HF><<<grid, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)out, (int)M, K, I, nbn, (int)nb);
  else
    return bad;
  return (int)hipGetLastError();
}

int ed_linear(const void* x, const void* w, const void* bias, const void* residual, void* out, int dtype, int64_t M, int K, int N,
              void* stream) {
  return launch<1, 0>(x, w, bias, nullptr, residual, out, dtype, M, K, N, 0, 0, 0, stream);
}

int ed_conv3x3_nhwc(const void* x, const void* w, const void* bias, const void* sample_bias, const void* residual, void* out,
                    int dtype, int B, int H, int W, int Cin, int N, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0) return (int)hipErrorInvalidValue;
  return launch<1, 1>(x, w, bias, sample_bias, residual, out, dtype, (int64_t)B * H * W, 9 * Cin, N, H, W, H * W, stream);
}

int ed_conv3x3_nhwc_up2x(const void* x, const void* w, const void* bias, void* out, int dtype, int B, int H, int W, int Cin, int N,
                         void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0) return (int)hipErrorInvalidValue;
  if ((int64_t)B * (H / 2) * (W / 2) * Cin * 2 >= 0x7ffffff0ll) return (int)hipErrorInvalidValue;
  return launch<1, CONV_UP2X>(x, w, bias, nullptr, nullptr, out, dtype, (int64_t)B * H * W, 9 * Cin, N, H, W, H * W, stream);
}

int ed_conv3x3_nhwc_s2(const void* x, const void* w, const void* bias, void* out, int dtype, int B, int H, int W, int Cin, int N,
                       void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0) return (int)hipErrorInvalidValue;
  return launch<1, CONV_S2>(x, w, bias, nullptr, nullptr, out, dtype, (int64_t)B * H * W, 9 * Cin, N, H, W, H * W, stream);
}

int ed_conv3x3_nhwc_f32out(const void* x, const void* w, const float* bias, const float* residual, float* out, int dtype, int B, int H, int W,
                           int Cin, int N, float out_scale, const float* act_absmax, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || ((uintptr_t)act_absmax & 3u)) return (int)hipErrorInvalidValue;
  return launch<1, 1, true>(x, w, bias, nullptr, residual, out, dtype, (int64_t)B * H * W, 9 * Cin, N, H, W, H * W, stream, out_scale,
                               act_absmax);
}

int ed_conv3x3_nhwc_f32out_s2(const void* x, const void* w, const float* bias, float* out, int dtype, int B, int H, int W, int Cin, int N,
                              float out_scale, const float* act_absmax, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || ((uintptr_t)act_absmax & 3u)) return (int)hipErrorInvalidValue;
  return launch<1, CONV_S2P0, true>(x, w, bias, nullptr, nullptr, out, dtype, (int64_t)B * H * W, 9 * Cin, N, H, W, H * W, stream, out_scale,
                                    act_absmax);
}

}  // extern "C"
