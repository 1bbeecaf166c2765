// gemm_sched.hip -- EXPERIMENT for round 5 (not in libelastic_hip.so, nothing on the product path calls it).
//
// Where inside a K tile should the 8 LDS-DMA pieces of a wave be issued, and how many barriers does a K tile need?  The product's
// main loop (csrc/gemm_kernels.hip, included below for its helpers) issues one half tile (2 pieces) in the READ half of each of
// its 4 phases, next to that phase's ds_reads; MI355X_MICROARCH.md prices a piece at 100-185 cycles there, ~60 among bare MFMAs
// and 25-60 in a later gap.  This file rebuilds the same loop (plain GEMM / GEGLU, no convolution) from a schedule descriptor:
//
//   pieces, in issue order:  A0 A1 = x m-half 1 of tile + 1 (other buffer),  B0 B1 = W value rows of tile + 2 (own buffer),
//                            C0 C1 = x m-half 0 of tile + 2,                  D0 D1 = W gate rows of tile + 2
//   slot[i] = barrier interval of the K tile the piece is issued in: 0 R1, 1 M1, 2 R2, 3 M2, 4 R3, 5 M3, 6 R4, 7 M4
//             (R = the read half of a phase, pieces go behind its ds_reads; M = the MFMA half)
//   pos[i]  = M slots: the piece goes in front of MFMA number pos (0..15), 16 = behind the last one
//
// Schedule 8 (EPI 1 only): the product order plus HALF tiles -- a column tile whose second 128-column half lies beyond N runs the value half
// only (tile_phases_half; replay: emulate_gemm_kernel.py --half, --break half_raw is caught).
// Convolutions (ed_s_conv3x3_nhwc: the product kernel's implicit-GEMM addresses through the same pieces) run the control and the 4-interval loop.
// Same arithmetic in the same order for every schedule: results must be bit-identical to the product kernel's.  The counted waits
// follow from the descriptor (in-order return): the wait at the end of R4 leaves the B / C / D pieces issued by then in flight
// (A must be issued by then); the early-start waits of K tile 0 in front of both barriers of phases 1 / 2 leave everything but the
// first 6 / 8 prologue pieces in flight.  WAR / RAW hazards of a descriptor are replayed in tools/emulate_gemm_kernel.py
// (`--sched NAME`) before it is run.  A second loop (`TWO`) merges phases 1+2 and 3+4: 4 barrier intervals of 32 MFMAs per K tile.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -I include tools/gemm_sched/gemm_sched.hip -o tools/gemm_sched/libgemm_sched.so
#include <type_traits>

#include "../../elasticdiffusion_official_amd/csrc/gemm_kernels.hip"

namespace {

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- schedule descriptors ---------------------------------------------------------------------------------------------------
struct NoConv {
  static constexpr bool conv = false;
};
template <class S0>
struct Conv : S0 {   // the same schedule for the 3x3 convolution's A operand addresses
  static constexpr bool conv = true;
};
struct S_product : NoConv {   // the product's placement, through this file's code path (control)
  static constexpr int slot[8] = {0, 0, 2, 2, 4, 4, 6, 6};
  static constexpr int pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
struct S_late : NoConv {      // A, B, C behind the MFMAs of their phase (D stays in R4: it must precede the tile's wait)
  static constexpr int slot[8] = {1, 1, 3, 3, 5, 5, 6, 6};
  static constexpr int pos[8] = {16, 16, 16, 16, 16, 16, 0, 0};
};
struct S_mid : NoConv {       // A, B, C inside the MFMA burst of their phase (in front of MFMAs 4 and 12)
  static constexpr int slot[8] = {1, 1, 3, 3, 5, 5, 6, 6};
  static constexpr int pos[8] = {4, 12, 4, 12, 4, 12, 0, 0};
};
struct S_spread : NoConv {    // one piece per barrier interval: R1 M1 R2 M2 R3 M3 R4 R4
  static constexpr int slot[8] = {0, 1, 2, 3, 4, 5, 6, 6};
  static constexpr int pos[8] = {0, 8, 0, 8, 0, 8, 0, 0};
};
struct S_mfma_all : NoConv {  // every piece in an MFMA half (M1 M1 M2 M2 M3 M3, D in M4 = waited for one tile later)
  static constexpr int slot[8] = {1, 1, 3, 3, 5, 5, 7, 7};
  static constexpr int pos[8] = {4, 12, 4, 12, 4, 12, 4, 12};
};
struct S_r4 : NoConv {        // phase 4 reads nothing: give it two half tiles (C, D), phases 2 / 3 one piece less each
  static constexpr int slot[8] = {0, 0, 2, 4, 6, 6, 6, 6};
  static constexpr int pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

template <class S>
constexpr int pieces_upto(int last_slot, int first_id = 0) {   // pieces first_id..7 issued in slots <= last_slot
  int n = 0;
  for (int i = first_id; i < 8; ++i) n += S::slot[i] <= last_slot;
  return n;
}

// piece i of K tile `tile` in buffer BUFI
// what a piece needs to know about its K tile (wave-uniform): tile index, whether tiles + 1 / + 2 exist, their K positions
struct TA {
  int tile;
  bool s1, s2;
  KPos p1, p2;
};

// x address of m half h at K position p: GEMM = the K tile's columns; CONV = the tap's pixel (or the out-of-range sentinel) and channel block
template <bool CONV>
__device__ __forceinline__ int x_vo(const Ctx& c, KPos p, int h) {
  if (CONV) {
    const int dy = p.tap / 3 - 1, dx = p.tap - 3 * (p.tap / 3) - 1;
    const int delta = (dy * c.img_w + dx) * c.cin2 + p.ct * (BK * 2);
    return ((c.px_mask[h] >> p.tap) & 1) ? c.x_voff[h] + delta : (int)0x80000000;
  }
  return c.x_voff[h] + p.tile * (BK * 2);
}

template <int BUFI, int I, bool CONV>
__device__ __forceinline__ void piece(uint8_t* lds, const Ctx& c, TA a) {
  constexpr int op = I >> 1, k = I & 1;
  const int tile = a.tile;
  const bool s1 = a.s1, s2 = a.s2;
  const KPos p1 = a.p1, p2 = a.p2;
  if (op == 0) {
    if (s1) {
      const int rg = (c.wave & 3) + 8 * (c.wave >> 2) + 4;
      uint8_t* dst = lds + (BUFI ^ 1) * BUF + x_sub(0, 0) + rg * (2 * SUB) + k * SUB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)dst, 16, x_vo<CONV>(c, p1, 1) + 64 * k, 0, 0, 0);
    }
  } else if (op == 2) {
    if (s2) {
      const int rg = (c.wave & 3) + 8 * (c.wave >> 2);
      uint8_t* dst = lds + BUFI * BUF + x_sub(0, 0) + rg * (2 * SUB) + k * SUB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)dst, 16, x_vo<CONV>(c, p2, 0) + 64 * k, 0, 0, 0);
    }
  } else {
    if (s2) {
      constexpr int g = op == 1 ? 0 : 1;
      const int rg = 8 * g + c.wave;
      uint8_t* dst = lds + BUFI * BUF + w_sub(0, 0) + rg * (2 * SUB) + k * SUB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(c.wr_, (lds_ptr_t)dst, 16, c.w_voff[g] + (tile + 2) * (BK * 2) + 64 * k, 0, 0, 0);
    }
  }
}

template <class S, int BUFI, int SLOT, int POS, int I = 0>
__device__ __forceinline__ void issue_at(uint8_t* lds, const Ctx& c, TA a) {
  if constexpr (I < 8) {
    if constexpr (S::slot[I] == SLOT && ((SLOT & 1) == 0 || S::pos[I] == POS)) piece<BUFI, I, S::conv>(lds, c, a);
    issue_at<S, BUFI, SLOT, POS, I + 1>(lds, c, a);
  }
}

// 16 MFMAs (the product's order) with the schedule's pieces of slot SLOT between them
template <class T, class S, int BUFI, int SLOT, int MH, int G, int N = 0>
__device__ __forceinline__ void mma16_s(uint8_t* lds, const Ctx& c, f32x4 (&acc)[8][4], const Frags<T>& f, TA a) {
  if constexpr (N == 0) __builtin_amdgcn_s_setprio(1);
  issue_at<S, BUFI, SLOT, N>(lds, c, a);
  if constexpr (N < 16) {
    constexpr int kh = N >> 3, mf = (N >> 1) & 3, nf = N & 1;
    acc[MH * 4 + mf][G * 2 + nf] = T::mfma(G == 0 ? f.wv[nf][kh] : f.wg[nf][kh], f.x[mf][kh], acc[MH * 4 + mf][G * 2 + nf]);
    mma16_s<T, S, BUFI, SLOT, MH, G, N + 1>(lds, c, acc, f, a);
  } else {
    __builtin_amdgcn_s_setprio(0);
  }
}

template <class T, int BUFI, class S, bool FIRST>
__device__ __forceinline__ void tile_phases_s(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], TA a) {
  static_assert(S::slot[0] <= 6 && S::slot[1] <= 6, "the A pieces must precede the tile's wait");
  // early start: 14 prologue pieces + this tile's so far; the first 6 (8) must have landed at the barriers of phase 1 (2)
  constexpr int E1a = 14 + pieces_upto<S>(0) - 6, E1b = 14 + pieces_upto<S>(1) - 6;
  constexpr int E2a = 14 + pieces_upto<S>(2) - 8, E2b = 14 + pieces_upto<S>(3) - 8;
  read_w<T, BUFI, 0>(lds, c, f);
  __builtin_amdgcn_sched_barrier(0);
  read_x<T, BUFI>(lds, c, f, 0);
  issue_at<S, BUFI, 0, 0>(lds, c, a);
  ED_WAIT_LGKM(8);
  if (FIRST) wait_vm<E1a>();
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16_s<T, S, BUFI, 1, 0, 0>(lds, c, acc, f, a);
  if (FIRST) wait_vm<E1b>();
  ED_BARRIER();
  read_w<T, BUFI, 1>(lds, c, f);
  issue_at<S, BUFI, 2, 0>(lds, c, a);
  if (FIRST) wait_vm<E2a>();
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16_s<T, S, BUFI, 3, 0, 1>(lds, c, acc, f, a);
  if (FIRST) wait_vm<E2b>();
  ED_BARRIER();
  read_x<T, BUFI>(lds, c, f, 1);
  issue_at<S, BUFI, 4, 0>(lds, c, a);
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16_s<T, S, BUFI, 5, 1, 1>(lds, c, acc, f, a);
  ED_BARRIER();
  issue_at<S, BUFI, 6, 0>(lds, c, a);
  if (a.s2) wait_vm<pieces_upto<S>(6, 2)>();      // all of tile + 1 has landed; the B / C / D pieces issued so far stay in flight
  else wait_vm<0>();
  ED_BARRIER();
  mma16_s<T, S, BUFI, 7, 1, 0>(lds, c, acc, f, a);
  ED_BARRIER();
}

// ---- TWO: 4 barrier intervals per K tile.  R1 = W value + gate + x m-half 0 (16 reads) and A;  M1 = 32 MFMAs;  R2 = x m-half 1
// (8 reads), B C D and the tile's wait;  M2 = 32 MFMAs.  The reading row has a whole 32-MFMA interval of the other row to spend, so it
// retires its fragment reads BEFORE its barrier (no read is in flight across a barrier: the regions read in R1 may be re-staged by
// either row from the next interval on).
struct S2_read : NoConv {     // A in R1; B C D in R2
  static constexpr int slot[8] = {0, 0, 2, 2, 2, 2, 2, 2};
  static constexpr int pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
struct S2_mfma : NoConv {     // A in R1; B C D inside M2 (waited for one tile later: the wait at the end of R2 leaves nothing in flight)
  static constexpr int slot[8] = {0, 0, 3, 3, 3, 3, 3, 3};
  static constexpr int pos[8] = {0, 0, 4, 8, 12, 16, 20, 24};
};

template <class S, int BUFI, int SLOT, int POS, int I = 0>
__device__ __forceinline__ void issue2_at(uint8_t* lds, const Ctx& c, TA a) {
  if constexpr (I < 8) {
    if constexpr (S::slot[I] == SLOT && ((SLOT & 1) == 0 || S::pos[I] == POS)) piece<BUFI, I, S::conv>(lds, c, a);
    issue2_at<S, BUFI, SLOT, POS, I + 1>(lds, c, a);
  }
}
// 32 MFMAs: m half MH x (value, gate | gate, value): G0 first
template <class T, class S, int BUFI, int SLOT, int MH, int G0, int N = 0>
__device__ __forceinline__ void mma32_s(uint8_t* lds, const Ctx& c, f32x4 (&acc)[8][4], const Frags<T>& f, TA a) {
  if constexpr (N == 0) __builtin_amdgcn_s_setprio(1);
  issue2_at<S, BUFI, SLOT, N>(lds, c, a);
  if constexpr (N < 32) {
    constexpr int G = (N >> 4) ^ G0, kh = (N >> 3) & 1, mf = (N >> 1) & 3, nf = N & 1;
    acc[MH * 4 + mf][G * 2 + nf] = T::mfma(G == 0 ? f.wv[nf][kh] : f.wg[nf][kh], f.x[mf][kh], acc[MH * 4 + mf][G * 2 + nf]);
    mma32_s<T, S, BUFI, SLOT, MH, G0, N + 1>(lds, c, acc, f, a);
  } else {
    __builtin_amdgcn_s_setprio(0);
  }
}
template <class T, int BUFI, class S>
__device__ __forceinline__ void tile_phases_2(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], TA a) {
  static_assert(S::slot[0] <= 2 && S::slot[1] <= 2, "the A pieces must precede the tile's wait");
  read_w<T, BUFI, 0>(lds, c, f);
  read_x<T, BUFI>(lds, c, f, 0);
  read_w<T, BUFI, 1>(lds, c, f);
  issue2_at<S, BUFI, 0, 0>(lds, c, a);
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma32_s<T, S, BUFI, 1, 0, 0>(lds, c, acc, f, a);    // the MFMA order of phases 1, 2
  ED_BARRIER();
  read_x<T, BUFI>(lds, c, f, 1);
  issue2_at<S, BUFI, 2, 0>(lds, c, a);
  if (a.s2) wait_vm<pieces_upto<S>(2, 2)>();
  else wait_vm<0>();
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma32_s<T, S, BUFI, 3, 1, 1>(lds, c, acc, f, a);    // ... of phases 3, 4 (gate first)
  ED_BARRIER();
}

// ---- HALF: a column tile whose second 128-column half lies entirely beyond N (N mod 256 in (0, 128]: the last tile of N = 320, 640, 1920)
// runs the value half only -- 256 x 128 outputs: no gate-row DMAs, no gate fragment reads, 32 instead of 64 MFMAs per K tile and wave.
// 4 barrier intervals per K tile: R1 = W value + x m-half 0 (12 reads) and x m-half 1 of tile + 1; M1 = 16 MFMAs; R2 = x m-half 1 (8 reads),
// W value rows and x m-half 0 of tile + 2, the tile's wait (those 4 pieces stay in flight); M2 = 16 MFMAs with the W fragments of R1.
template <class T, int BUFI, bool CONV>
__device__ __forceinline__ void tile_phases_half(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], TA a) {
  read_w<T, BUFI, 0>(lds, c, f);
  read_x<T, BUFI>(lds, c, f, 0);
  piece<BUFI, 0, CONV>(lds, c, a);
  piece<BUFI, 1, CONV>(lds, c, a);
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 0, 0>(acc, f);
  ED_BARRIER();
  read_x<T, BUFI>(lds, c, f, 1);
  piece<BUFI, 2, CONV>(lds, c, a);
  piece<BUFI, 3, CONV>(lds, c, a);
  piece<BUFI, 4, CONV>(lds, c, a);
  piece<BUFI, 5, CONV>(lds, c, a);
  if (a.s2) wait_vm<4>();
  else wait_vm<0>();
  ED_WAIT_LGKM(0);
  ED_BARRIER();
  mma16<T, 1, 0>(acc, f);
  ED_BARRIER();
}

template <class T, int EPI, class S, bool TWO, bool HALF = false>
__global__ void __launch_bounds__(512, 2)
k_gemm_sched(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
             uint16_t* __restrict__ out, int M, int K, int I, int n_blocks_n, int n_blocks, int img_h, int img_w) {
  constexpr bool CONV = S::conv;      // 3x3 convolution of an NHWC image as an implicit GEMM (EPI 1), as in the product kernel
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * BUF];
  const int bid = blockIdx.x;
  const int q = n_blocks >> 3, r = n_blocks & 7, xcd = bid & 7;
  const int tid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int n_blocks_m = n_blocks / n_blocks_n;
  const int per_group = 8 * n_blocks_n, grp = tid / per_group, first = grp * 8;
  const int rows_here = n_blocks_m - first < 8 ? n_blocks_m - first : 8;
  const int m0 = (first + (tid % per_group) % rows_here) * BM;
  const int n0 = ((tid % per_group) / rows_here) * (EPI == 0 ? BN : 2 * BN);
  const int gap = EPI == 0 ? I : BN;
  const int lane = threadIdx.x & 63;
  Ctx c;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wrow = c.wave >> 2, wcol = c.wave & 3;
  const int ps = swz(16 * lane), srow = ps >> 6, skb = ps & 63;
  const int row_bytes = K * 2;
  const int x_row_bytes = CONV ? row_bytes / 9 : row_bytes;
  c.xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((int64_t)M * x_row_bytes), 0x00020000);
  c.wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((int64_t)(EPI == 0 ? 2 : 1) * I * row_bytes), 0x00020000);
  const int xrow0 = m0 + ((c.wave & 3) + 8 * (c.wave >> 2)) * 16 + srow;
  c.x_voff[0] = xrow0 * x_row_bytes + skb;
  c.x_voff[1] = c.x_voff[0] + 64 * x_row_bytes;
  c.img_w = img_w, c.cin2 = x_row_bytes, c.cpt = CONV ? K / (9 * BK) : 1;
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
          const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
          if (yy >= 0 && yy < img_h && xx >= 0 && xx < img_w) mask |= 1 << t;
        }
        c.px_mask[h] = mask;
      }
    }
  }
  c.w_voff[0] = (n0 + 32 * (c.wave >> 1) + 8 * (srow >> 2) + (srow & 3) + 4 * (c.wave & 1)) * row_bytes + skb;
  c.w_voff[1] = c.w_voff[0] + gap * row_bytes;
  const int rd = swz((lane & 15) * 64 + (lane >> 4) * 16);
  c.xrd = rd + wrow * 8 * (2 * SUB);
  c.wrd = rd + W_REGION + wcol * 2 * (2 * SUB);
  const int ncol = n0 + 32 * wcol + 8 * (lane >> 4);
  u32x4 bias_v = {0, 0, 0, 0}, bias_g = {0, 0, 0, 0};
  if (bias) {
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
  const KPos p0 = {0, 0, 0};
  KPos pa = k_next<CONV>(p0, c.cpt), pb = k_next<CONV>(pa, c.cpt);     // positions of tiles t + 1, t + 2
  const bool half = HALF && EPI == 1 && n0 + BN >= I;                  // wave-uniform: the gate half of this tile is beyond N
  if (half) {
    stage_w<0>(lds, c, 0, 0);
    stage_x<0, CONV>(lds, c, p0, 0);
    stage_x<0, CONV>(lds, c, p0, 1);
    if (nt > 1) {
      stage_w<1>(lds, c, 1, 0);
      stage_x<1, CONV>(lds, c, pa, 0);
      ED_WAIT_VM(4);
    } else {
      ED_WAIT_VM(0);
    }
    ED_BARRIER();
    if (wrow == 1) ED_BARRIER();
    int th = 0;
    for (; th + 1 < nt; th += 2) {
      tile_phases_half<T, 0, CONV>(lds, c, f, acc, TA{th, true, th + 2 < nt, pa, pb});
      pa = pb, pb = k_next<CONV>(pb, c.cpt);
      tile_phases_half<T, 1, CONV>(lds, c, f, acc, TA{th + 1, th + 2 < nt, th + 3 < nt, pa, pb});
      pa = pb, pb = k_next<CONV>(pb, c.cpt);
    }
    if (th < nt) tile_phases_half<T, 0, CONV>(lds, c, f, acc, TA{th, false, false, pa, pb});
    if (wrow == 0) ED_BARRIER();
  } else {
  stage_w<0>(lds, c, 0, 0);
  stage_x<0, CONV>(lds, c, p0, 0);
  stage_w<0>(lds, c, 0, 1);
  stage_x<0, CONV>(lds, c, p0, 1);
  const bool early = !TWO && nt >= 3;
  if (nt > 1) {
    stage_w<1>(lds, c, 1, 0);
    stage_x<1, CONV>(lds, c, pa, 0);
    stage_w<1>(lds, c, 1, 1);
    if (early) ED_WAIT_VM(10);
    else ED_WAIT_VM(6);
  } else {
    ED_WAIT_VM(0);
  }
  ED_BARRIER();
  if (wrow == 1) ED_BARRIER();

  int t = 0;
  auto step = [&]() { pa = pb, pb = k_next<CONV>(pb, c.cpt); };
  if (TWO) {
    for (; t + 1 < nt; t += 2) {
      tile_phases_2<T, 0, S>(lds, c, f, acc, TA{t, true, t + 2 < nt, pa, pb});
      step();
      tile_phases_2<T, 1, S>(lds, c, f, acc, TA{t + 1, t + 2 < nt, t + 3 < nt, pa, pb});
      step();
    }
    if (t < nt) tile_phases_2<T, 0, S>(lds, c, f, acc, TA{t, false, false, pa, pb});
  } else {
    if (early) {
      tile_phases_s<T, 0, S, true>(lds, c, f, acc, TA{0, true, true, pa, pb});
      step();
      tile_phases_s<T, 1, S, false>(lds, c, f, acc, TA{1, true, 3 < nt, pa, pb});
      step();
      t = 2;
    }
    for (; t + 1 < nt; t += 2) {
      tile_phases_s<T, 0, S, false>(lds, c, f, acc, TA{t, true, t + 2 < nt, pa, pb});
      step();
      tile_phases_s<T, 1, S, false>(lds, c, f, acc, TA{t + 1, t + 2 < nt, t + 3 < nt, pa, pb});
      step();
    }
    if (t < nt) tile_phases_s<T, 0, S, false>(lds, c, f, acc, TA{t, false, false, pa, pb});
  }
  if (wrow == 0) ED_BARRIER();
  }   // (!half)
  if (S::slot[7] == 7 || (TWO && S::slot[7] == 3)) wait_vm<0>();   // (nothing is in flight here: s2 was false for the last two tiles)

  float bv[2][4], bg[2][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bv[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_v[e >> 1] >> (16 * (e & 1))));
    bg[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_g[e >> 1] >> (16 * (e & 1))));
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const int m = m0 + 128 * wrow + 16 * mb + (lane & 15);
    if (EPI == 0) {
      uint32_t pk[4];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float o0 = (acc[mb][nf][2 * jj] + bv[nf][2 * jj]) * gelu_as(acc[mb][2 + nf][2 * jj] + bg[nf][2 * jj]);
          float o1 = (acc[mb][nf][2 * jj + 1] + bv[nf][2 * jj + 1]) * gelu_as(acc[mb][2 + nf][2 * jj + 1] + bg[nf][2 * jj + 1]);
          pk[nf * 2 + jj] = (uint32_t)T::from_f32(o0) | ((uint32_t)T::from_f32(o1) << 16);
        }
      if (m < M) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol) = u32x4{pk[0], pk[1], pk[2], pk[3]};
    } else {
      const bool ok_v = m < M && ncol < I, ok_g = m < M && ncol + gap < I;
      uint32_t pv[4], pg[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        float v0 = acc[mb][e >> 2][e & 3] + bv[e >> 2][e & 3], v1 = acc[mb][e >> 2][(e & 3) + 1] + bv[e >> 2][(e & 3) + 1];
        float g0 = acc[mb][2 + (e >> 2)][e & 3] + bg[e >> 2][e & 3], g1 = acc[mb][2 + (e >> 2)][(e & 3) + 1] + bg[e >> 2][(e & 3) + 1];
        pv[e >> 1] = (uint32_t)T::from_f32(v0) | ((uint32_t)T::from_f32(v1) << 16);
        pg[e >> 1] = (uint32_t)T::from_f32(g0) | ((uint32_t)T::from_f32(g1) << 16);
      }
      if (ok_v) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol) = u32x4{pv[0], pv[1], pv[2], pv[3]};
      if (ok_g) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol + gap) = u32x4{pg[0], pg[1], pg[2], pg[3]};
    }
  }
}

template <int EPI, bool CONV = false>
int launch_sched(int sched, const void* x, const void* w, const void* bias, void* out, int dtype, int64_t M, int K, int I, void* stream,
                 int img_h = 0, int img_w = 0) {
  if (M == 0) return 0;
  const int bad = (int)hipErrorInvalidValue;
  if (dtype != ED_F16 || M < 0 || K % BK != 0 || K < BK || I <= 0 || (EPI == 0 ? I % BN != 0 : I % 8 != 0)) return bad;
  if (CONV && (K % (9 * BK) != 0 || img_h <= 0 || img_w <= 0 || M % ((int64_t)img_h * img_w) != 0)) return bad;
  if (M * (int64_t)(CONV ? K / 9 : K) * 2 >= 0x7ffffff0ll || (int64_t)(EPI == 0 ? 2 : 1) * I * K * 2 >= 0x7ffffff0ll) return bad;
  const int nbn = EPI == 0 ? I / BN : (I + 2 * BN - 1) / (2 * BN);
  const int64_t nb = ((M + BM - 1) / BM) * nbn;
  if (nb >= (1ll << 31)) return bad;
  hipStream_t s = (hipStream_t)stream;
#define ED_GO(SS, TWO_)                                                                                                            \
  k_gemm_sched<HF, EPI, SS, TWO_><<<(int)nb, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)out, \
                                                        (int)M, K, I, nbn, (int)nb, img_h, img_w)
#define ED_GO_HALF(SS)                                                                                                             \
  k_gemm_sched<HF, EPI, SS, false, true><<<(int)nb, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, (uint16_t*)out, \
                                                              (int)M, K, I, nbn, (int)nb, img_h, img_w)
  if (CONV) {      // the convolution runs the control (product order), the 4-interval loop and the control with half tiles
    if (sched == 0) ED_GO(Conv<S_product>, false);
    else if (sched == 6) ED_GO(Conv<S2_read>, true);
    else if (sched == 8) ED_GO_HALF(Conv<S_product>);
    else return bad;
    return (int)hipGetLastError();
  }
  if (sched == 8) {   // product order; a last column tile with nothing in its second half runs the value half only
    if (EPI != 1) return bad;
    ED_GO_HALF(S_product);
    return (int)hipGetLastError();
  }
  switch (sched) {
    case 0: ED_GO(S_product, false); break;
    case 1: ED_GO(S_late, false); break;
    case 2: ED_GO(S_mid, false); break;
    case 3: ED_GO(S_spread, false); break;
    case 4: ED_GO(S_mfma_all, false); break;
    case 5: ED_GO(S_r4, false); break;
    case 6: ED_GO(S2_read, true); break;
    case 7: ED_GO(S2_mfma, true); break;
    default: return bad;
  }
#undef ED_GO
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {
int ed_s_geglu_gemm(int sched, const void* x, const void* w, const void* bias, void* out, int dtype, int64_t M, int K, int I, void* stream) {
  return launch_sched<0>(sched, x, w, bias, out, dtype, M, K, I, stream);
}
int ed_s_linear(int sched, const void* x, const void* w, const void* bias, void* out, int dtype, int64_t M, int K, int N, void* stream) {
  return launch_sched<1>(sched, x, w, bias, out, dtype, M, K, N, stream);
}
// 3x3 convolution, NHWC, stride 1, zero padding 1, bias only: sched 0 (the product's order) or 6 (4 barrier intervals per K tile)
int ed_s_conv3x3_nhwc(int sched, const void* x, const void* w, const void* bias, void* out, int dtype, int B, int H, int W, int Cin, int N, void* stream) {
  return launch_sched<1, true>(sched, x, w, bias, out, dtype, (int64_t)B * H * W, 9 * Cin, N, stream, H, W);
}
int ed_s_count(void) { return 8; }   // (schedule 8 = half tiles: EPI 1 only, not part of the sweep)
}
