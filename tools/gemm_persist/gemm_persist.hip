// gemm_persist.hip -- EXPERIMENT for round 5 (not in libelastic_hip.so, nothing on the product path calls it).
//
// The product's 8-phase GEMM main loop (csrc/gemm_kernels.hip, included below for its helpers) as a PERSISTENT kernel: one
// workgroup per CU walks the tile ids b, b + G, b + 2G, ... and issues the NEXT tile's 14 prologue LDS-DMAs right after the last
// barrier of the current tile, i.e. BEFORE the current tile's epilogue (GELU, packing, stores) -- both LDS buffers are free at
// that point, the epilogue touches registers and global memory only.  Why: a tile's fixed cost (prologue fill + epilogue +
// workgroup turnaround) is ~7 us against ~1.5 us per K tile: 19 % of a K = 1280 tile, 33 % of a K = 640 one.
//
//   * tile id -> (row block, column block) exactly as in the product kernel; G is a multiple of 8, so a workgroup stays on its
//     XCD and an XCD's 32 workgroups walk its 8 x 4 tile groups together;
//   * bias: the current tile's values are converted (= waited for) right after the last barrier, when the VM queue is empty;
//     the next tile's two loads are issued before its prologue DMAs (oldest in the queue, retired by the counted waits);
//   * tiles whose epilogue reads global memory (per-sample bias / residual of ed_conv3x3_nhwc, ed_linear's residual) do NOT
//     overlap: an ordinary load next to 14 DMAs in flight makes hipcc wait vmcnt(0) before its first use;
//   * everything inside a tile -- phases, counted waits, early start, the half-phase stagger of the two wave rows -- is the product's.
// LDS hazard of the overlap: after the barrier pair that ends a tile every wave has retired all its fragment reads (they precede the
// MFMAs that precede the barrier), so the next prologue may overwrite both buffers; replayed in tools/emulate_gemm_kernel.py
// (`--persist`).
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -I include tools/gemm_persist/gemm_persist.hip -o tools/gemm_persist/libgemm_persist.so
#include "../../elasticdiffusion_official_amd/csrc/gemm_kernels.hip"

namespace {

struct TilePos {
  int m0, n0;
};

template <int EPI>
__device__ __forceinline__ TilePos tile_of(int bid, int n_blocks, int n_blocks_n) {
  const int q = n_blocks >> 3, r = n_blocks & 7, xcd = bid & 7;
  const int tid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int n_blocks_m = n_blocks / n_blocks_n;
  const int per_group = 8 * n_blocks_n, grp = tid / per_group, first = grp * 8;
  const int rows_here = n_blocks_m - first < 8 ? n_blocks_m - first : 8;
  return TilePos{(first + (tid % per_group) % rows_here) * BM, ((tid % per_group) / rows_here) * (EPI == 0 ? BN : 2 * BN)};
}

template <class T, int EPI, bool CONV>
__global__ void __launch_bounds__(512, 2)
k_gemm_persist(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
               const uint16_t* __restrict__ row_bias, const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int M,
               int K, int I, int n_blocks_n, int n_blocks, int img_h, int img_w, int rows_per_sample) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * BUF];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wrow = wave >> 2, wcol = wave & 3;
  const int gap = EPI == 0 ? I : BN;
  const int ps = swz(16 * lane), srow = ps >> 6, skb = ps & 63;
  const int row_bytes = K * 2;
  const int x_row_bytes = CONV ? row_bytes / 9 : row_bytes;
  const int rd = swz((lane & 15) * 64 + (lane >> 4) * 16);
  const int nt = K / BK;
  const bool early = nt >= 3;
  const bool overlap = !(row_bias || residual);     // see the header: epilogue loads would drain the next tile's DMAs

  auto setup = [&](TilePos tp) {
    Ctx c;
    c.wave = wave;
    c.xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((int64_t)M * x_row_bytes), 0x00020000);
    c.wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((int64_t)(EPI == 0 ? 2 : 1) * I * row_bytes), 0x00020000);
    const int xrow0 = tp.m0 + ((wave & 3) + 8 * (wave >> 2)) * 16 + srow;
    c.x_voff[0] = xrow0 * x_row_bytes + skb;
    c.x_voff[1] = c.x_voff[0] + 64 * x_row_bytes;
    c.img_w = img_w;
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
            const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
            if (yy >= 0 && yy < img_h && xx >= 0 && xx < img_w) mask |= 1 << t;
          }
          c.px_mask[h] = mask;
        }
      }
    }
    c.w_voff[0] = (tp.n0 + 32 * (wave >> 1) + 8 * (srow >> 2) + (srow & 3) + 4 * (wave & 1)) * row_bytes + skb;
    c.w_voff[1] = c.w_voff[0] + gap * row_bytes;
    c.xrd = rd + wrow * 8 * (2 * SUB);
    c.wrd = rd + W_REGION + wcol * 2 * (2 * SUB);
    return c;
  };
  auto load_bias = [&](int ncol, u32x4& bv_raw, u32x4& bg_raw) {
    bv_raw = u32x4{0, 0, 0, 0}, bg_raw = u32x4{0, 0, 0, 0};
    if (bias) {
      if (EPI == 0 || ncol < I) bv_raw = *reinterpret_cast<const u32x4*>(bias + ncol);
      if (EPI == 0 || ncol + gap < I) bg_raw = *reinterpret_cast<const u32x4*>(bias + gap + ncol);
    }
  };
  auto issue_prologue = [&](const Ctx& c) {   // all of K tile 0, then the three half tiles of K tile 1 the loop does not stage itself
    const KPos p0 = {0, 0, 0};
    const KPos pa = k_next<CONV>(p0, c.cpt);
    stage_w<0>(lds, c, 0, 0);
    stage_x<0, CONV>(lds, c, p0, 0);
    stage_w<0>(lds, c, 0, 1);
    stage_x<0, CONV>(lds, c, p0, 1);
    if (nt > 1) {
      stage_w<1>(lds, c, 1, 0);
      stage_x<1, CONV>(lds, c, pa, 0);
      stage_w<1>(lds, c, 1, 1);
    }
  };

  int bid = blockIdx.x;
  TilePos tp = tile_of<EPI>(bid, n_blocks, n_blocks_n);
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
    const KPos p0 = {0, 0, 0};
    KPos pa = k_next<CONV>(p0, c.cpt);
    KPos pb = k_next<CONV>(pa, c.cpt);
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
    if (has_next && overlap) {                // next tile: addresses, its bias loads (oldest in the queue), then its prologue DMAs
      tp2 = tile_of<EPI>(next, n_blocks, n_blocks_n);
      c2 = setup(tp2);
      load_bias(tp2.n0 + 32 * wcol + 8 * (lane >> 4), bias_v, bias_g);
      __builtin_amdgcn_sched_barrier(0);     // the two loads stay AHEAD of the DMAs (left alone, hipcc sinks them below the epilogue)
      issue_prologue(c2);
      __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue of the current tile (registers and global stores only when `overlap`)
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
        u32x4 av[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, ag[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
        if (row_bias) {
          const int64_t rb = (int64_t)(m / rows_per_sample) * I;
          if (ok_v) av[0] = *reinterpret_cast<const u32x4*>(row_bias + rb + ncol);
          if (ok_g) ag[0] = *reinterpret_cast<const u32x4*>(row_bias + rb + ncol + gap);
        }
        if (residual) {
          if (ok_v) av[1] = *reinterpret_cast<const u32x4*>(residual + (int64_t)m * I + ncol);
          if (ok_g) ag[1] = *reinterpret_cast<const u32x4*>(residual + (int64_t)m * I + ncol + gap);
        }
        uint32_t pv[4], pg[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float v0 = acc[mb][e >> 2][e & 3] + bv[e >> 2][e & 3], v1 = acc[mb][e >> 2][(e & 3) + 1] + bv[e >> 2][(e & 3) + 1];
          float g0 = acc[mb][2 + (e >> 2)][e & 3] + bg[e >> 2][e & 3], g1 = acc[mb][2 + (e >> 2)][(e & 3) + 1] + bg[e >> 2][(e & 3) + 1];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            v0 += T::to_f32((uint16_t)av[a][e >> 1]), v1 += T::to_f32((uint16_t)(av[a][e >> 1] >> 16));
            g0 += T::to_f32((uint16_t)ag[a][e >> 1]), g1 += T::to_f32((uint16_t)(ag[a][e >> 1] >> 16));
          }
          pv[e >> 1] = (uint32_t)T::from_f32(v0) | ((uint32_t)T::from_f32(v1) << 16);
          pg[e >> 1] = (uint32_t)T::from_f32(g0) | ((uint32_t)T::from_f32(g1) << 16);
        }
        if (ok_v) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol) = u32x4{pv[0], pv[1], pv[2], pv[3]};
        if (ok_g) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol + gap) = u32x4{pg[0], pg[1], pg[2], pg[3]};
      }
    }
    if (!has_next) break;
    if (!overlap) {                            // the same steps after the epilogue: nothing of the next tile was in flight
      tp2 = tile_of<EPI>(next, n_blocks, n_blocks_n);
      c2 = setup(tp2);
      load_bias(tp2.n0 + 32 * wcol + 8 * (lane >> 4), bias_v, bias_g);
      issue_prologue(c2);
    }
    bid = next, tp = tp2, c = c2;
  }
}

// ---- v2: the NEXT tile's K tile 0 is staged during the LAST K tile of the current tile ------------------------------------------
// For an even number of K tiles the last one lives in buffer 1 and buffer 0 (K tile nt - 2) is dead from its end on: the next
// output tile's K tile 0 goes there, one half tile per phase (W value rows, x m-half 0, W gate rows, x m-half 1 -- each region 5+
// phases after its last read by either wave row), so that by the time the epilogue is done the fill has long landed; K tile 1's three
// half tiles follow after the tile's last barrier pair as in v1.  Plain GEMM only (a convolution's next tile needs new in-image
// masks: it keeps v1's order); the next tile's per-lane offsets are the current ones plus wave-uniform deltas.
template <int BUFI>
__device__ __forceinline__ void stage_x_d(uint8_t* lds, const Ctx& c, int dx, int h) {
  const int rg = (c.wave & 3) + 8 * (c.wave >> 2) + 4 * h;
  uint8_t* dst = lds + BUFI * BUF + x_sub(0, 0) + rg * (2 * SUB);
  const int vo = c.x_voff[h] + dx;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.xr, (lds_ptr_t)(dst + SUB), 16, vo + 64, 0, 0, 0);
}
template <int BUFI>
__device__ __forceinline__ void stage_w_d(uint8_t* lds, const Ctx& c, int dw, int tile, int g) {
  const int rg = 8 * g + c.wave;
  const int vo = c.w_voff[g] + dw + tile * (BK * 2);
  uint8_t* dst = lds + BUFI * BUF + w_sub(0, 0) + rg * (2 * SUB);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.wr_, (lds_ptr_t)dst, 16, vo, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.wr_, (lds_ptr_t)(dst + SUB), 16, vo + 64, 0, 0, 0);
}

// K tile `tile` in buffer 1 (the second of a pair), as the product's tile_phases<T, 1, false>, plus: `pfn` (wave-uniform) = this is the
// LAST K tile of the output tile and the workgroup has a next one -- nothing of this tile is left to stage (s1 = s2 = false), so each
// phase stages one half of the NEXT output tile's K tile 0 into buffer 0 instead, and phase 4 does not wait (the queue was emptied by
// K tile nt - 2's phase 4: everything in flight belongs to the next tile).  One body for both cases keeps the loop's shape -- separate
// inlined copies of the phases for the prefetching pair made hipcc spill 181 registers.
template <class T>
__device__ __forceinline__ void tile_phases_b1(uint8_t* lds, const Ctx& c, Frags<T>& f, f32x4 (&acc)[8][4], int tile, bool s1, bool s2,
                                               KPos p1, KPos p2, bool pfn, int dx, int dw) {
  read_w<T, 1, 0>(lds, c, f);
  __builtin_amdgcn_sched_barrier(0);
  read_x<T, 1>(lds, c, f, 0);
  if (pfn) stage_w_d<0>(lds, c, dw, 0, 0);
  else if (s1) stage_x<0, false>(lds, c, p1, 1);
  ED_WAIT_LGKM(8);
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16<T, 0, 0>(acc, f);
  ED_BARRIER();
  read_w<T, 1, 1>(lds, c, f);
  if (pfn) stage_x_d<0>(lds, c, dx, 0);
  else if (s2) stage_w<1>(lds, c, tile + 2, 0);
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16<T, 0, 1>(acc, f);
  ED_BARRIER();
  read_x<T, 1>(lds, c, f, 1);
  if (pfn) stage_w_d<0>(lds, c, dw, 0, 1);
  else if (s2) stage_x<1, false>(lds, c, p2, 0);
  ED_BARRIER();
  ED_WAIT_LGKM(0);
  __builtin_amdgcn_sched_barrier(0);
  mma16<T, 1, 1>(acc, f);
  ED_BARRIER();
  if (pfn) {
    stage_x_d<0>(lds, c, dx, 1);
  } else if (s2) {
    stage_w<1>(lds, c, tile + 2, 1);
    ED_WAIT_VM(6);
  } else {
    ED_WAIT_VM(0);
  }
  ED_BARRIER();
  mma16<T, 1, 0>(acc, f);
  ED_BARRIER();
}

template <class T, int EPI>
__global__ void __launch_bounds__(512, 2)
k_gemm_persist2(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
                const uint16_t* __restrict__ residual, uint16_t* __restrict__ out, int M, int K, int I, int n_blocks_n, int n_blocks) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * BUF];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wrow = wave >> 2, wcol = wave & 3;
  const int gap = EPI == 0 ? I : BN;
  const int ps = swz(16 * lane), srow = ps >> 6, skb = ps & 63;
  const int row_bytes = K * 2;
  const int rd = swz((lane & 15) * 64 + (lane >> 4) * 16);
  const int nt = K / BK;
  const bool early = nt >= 3;
  const bool prefetch = nt >= 6 && (nt & 1) == 0;    // the last K tile in buffer 1, buffer 0 free for the next tile
  const bool overlap = !residual;

  int bid = blockIdx.x;
  TilePos tp = tile_of<EPI>(bid, n_blocks, n_blocks_n);
  Ctx c;
  c.wave = wave;
  c.xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((int64_t)M * row_bytes), 0x00020000);
  c.wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((int64_t)(EPI == 0 ? 2 : 1) * I * row_bytes), 0x00020000);
  c.x_voff[0] = (tp.m0 + ((wave & 3) + 8 * (wave >> 2)) * 16 + srow) * row_bytes + skb;
  c.x_voff[1] = c.x_voff[0] + 64 * row_bytes;
  c.img_w = 0, c.cin2 = row_bytes, c.cpt = 1;
  c.px_mask[0] = c.px_mask[1] = 0;
  c.w_voff[0] = (tp.n0 + 32 * (wave >> 1) + 8 * (srow >> 2) + (srow & 3) + 4 * (wave & 1)) * row_bytes + skb;
  c.w_voff[1] = c.w_voff[0] + gap * row_bytes;
  c.xrd = rd + wrow * 8 * (2 * SUB);
  c.wrd = rd + W_REGION + wcol * 2 * (2 * SUB);
  auto load_bias = [&](int ncol, u32x4& bv_raw, u32x4& bg_raw) {
    bv_raw = u32x4{0, 0, 0, 0}, bg_raw = u32x4{0, 0, 0, 0};
    if (bias) {
      if (EPI == 0 || ncol < I) bv_raw = *reinterpret_cast<const u32x4*>(bias + ncol);
      if (EPI == 0 || ncol + gap < I) bg_raw = *reinterpret_cast<const u32x4*>(bias + gap + ncol);
    }
  };
  u32x4 bias_v, bias_g;
  load_bias(tp.n0 + 32 * wcol + 8 * (lane >> 4), bias_v, bias_g);
  const KPos p0 = {0, 0, 0}, p1 = {1, 0, 0};
  stage_w<0>(lds, c, 0, 0);
  stage_x<0, false>(lds, c, p0, 0);
  stage_w<0>(lds, c, 0, 1);
  stage_x<0, false>(lds, c, p0, 1);
  if (nt > 1) {
    stage_w<1>(lds, c, 1, 0);
    stage_x<1, false>(lds, c, p1, 0);
    stage_w<1>(lds, c, 1, 1);
  }

  for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frags<T> f;
    KPos pa = p1, pb = {2, 0, 0};
    const int next = bid + (int)gridDim.x;
    const bool has_next = next < n_blocks;
    TilePos tp2 = tp;
    int dx = 0, dw = 0;
    if (has_next) {
      tp2 = tile_of<EPI>(next, n_blocks, n_blocks_n);
      dx = (tp2.m0 - tp.m0) * row_bytes, dw = (tp2.n0 - tp.n0) * row_bytes;      // wave-uniform byte deltas of the next tile
    }
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
      pa = pb, pb = KPos{pb.tile + 1, 0, 0};
      tile_phases<T, 1, false>(lds, c, f, acc, 1, true, 3 < nt, pa, pb);
      pa = pb, pb = KPos{pb.tile + 1, 0, 0};
      t = 2;
    }
    const bool pf = prefetch && has_next;          // wave-uniform: the last K tile (nt even: the second of the last pair) prefetches
    for (; t + 1 < nt; t += 2) {
      tile_phases<T, 0, false>(lds, c, f, acc, t, true, t + 2 < nt, pa, pb);
      pa = pb, pb = KPos{pb.tile + 1, 0, 0};
      tile_phases_b1<T>(lds, c, f, acc, t + 1, t + 2 < nt, t + 3 < nt, pa, pb, pf && t + 2 == nt, dx, dw);
      pa = pb, pb = KPos{pb.tile + 1, 0, 0};
    }
    if (t < nt) tile_phases<T, 0, false>(lds, c, f, acc, t, false, false, pa, pb);
    if (wrow == 0) ED_BARRIER();

    const int m0 = tp.m0, ncol = tp.n0 + 32 * wcol + 8 * (lane >> 4);
    float bv[2][4], bg[2][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bv[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_v[e >> 1] >> (16 * (e & 1))));
      bg[e >> 2][e & 3] = T::to_f32((uint16_t)(bias_g[e >> 1] >> (16 * (e & 1))));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bv[e >> 2][e & 3]), "+v"(bg[e >> 2][e & 3]));

    auto rest_of_next_prologue = [&]() {       // c already points at the next tile
      load_bias(tp2.n0 + 32 * wcol + 8 * (lane >> 4), bias_v, bias_g);
      __builtin_amdgcn_sched_barrier(0);
      if (!(prefetch)) {
        stage_w<0>(lds, c, 0, 0);
        stage_x<0, false>(lds, c, p0, 0);
        stage_w<0>(lds, c, 0, 1);
        stage_x<0, false>(lds, c, p0, 1);
      }
      if (nt > 1) {
        stage_w<1>(lds, c, 1, 0);
        stage_x<1, false>(lds, c, p1, 0);
        stage_w<1>(lds, c, 1, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (has_next) {
      c.x_voff[0] += dx, c.x_voff[1] += dx, c.w_voff[0] += dw, c.w_voff[1] += dw;
      if (overlap) rest_of_next_prologue();
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
        u32x4 av = {0, 0, 0, 0}, ag = {0, 0, 0, 0};
        if (residual) {
          if (ok_v) av = *reinterpret_cast<const u32x4*>(residual + (int64_t)m * I + ncol);
          if (ok_g) ag = *reinterpret_cast<const u32x4*>(residual + (int64_t)m * I + ncol + gap);
        }
        uint32_t pv[4], pg[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float v0 = acc[mb][e >> 2][e & 3] + bv[e >> 2][e & 3], v1 = acc[mb][e >> 2][(e & 3) + 1] + bv[e >> 2][(e & 3) + 1];
          float g0 = acc[mb][2 + (e >> 2)][e & 3] + bg[e >> 2][e & 3], g1 = acc[mb][2 + (e >> 2)][(e & 3) + 1] + bg[e >> 2][(e & 3) + 1];
          v0 += T::to_f32((uint16_t)av[e >> 1]), v1 += T::to_f32((uint16_t)(av[e >> 1] >> 16));
          g0 += T::to_f32((uint16_t)ag[e >> 1]), g1 += T::to_f32((uint16_t)(ag[e >> 1] >> 16));
          pv[e >> 1] = (uint32_t)T::from_f32(v0) | ((uint32_t)T::from_f32(v1) << 16);
          pg[e >> 1] = (uint32_t)T::from_f32(g0) | ((uint32_t)T::from_f32(g1) << 16);
        }
        if (ok_v) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol) = u32x4{pv[0], pv[1], pv[2], pv[3]};
        if (ok_g) *reinterpret_cast<u32x4*>(out + (int64_t)m * I + ncol + gap) = u32x4{pg[0], pg[1], pg[2], pg[3]};
      }
    }
    if (!has_next) break;
    if (!overlap) rest_of_next_prologue();
    bid = next, tp = tp2;
  }
}

template <int EPI, bool CONV>
static int launch_persist(const void* x, const void* w, const void* bias, const void* row_bias, const void* residual, void* out,
                          int dtype, int64_t M, int K, int I, int img_h, int img_w, int rows_per_sample, int grid_cap, void* stream) {
  if (M == 0) return 0;
  const int bad = (int)hipErrorInvalidValue;
  if (M < 0 || K % BK != 0 || K < BK || I <= 0 || (EPI == 0 ? I % BN != 0 : I % 8 != 0)) return bad;
  if (CONV && (K % (9 * BK) != 0 || img_h <= 0 || img_w <= 0 || M % ((int64_t)img_h * img_w) != 0)) return bad;
  if (M * (int64_t)(CONV ? K / 9 : K) * 2 >= 0x7ffffff0ll || (int64_t)(EPI == 0 ? 2 : 1) * I * K * 2 >= 0x7ffffff0ll) return bad;
  const int nbn = EPI == 0 ? I / BN : (I + 2 * BN - 1) / (2 * BN);
  const int64_t nb = ((M + BM - 1) / BM) * nbn;
  if (nb >= (1ll << 31) || grid_cap <= 0 || grid_cap % 8 != 0) return bad;
  const int grid = nb < grid_cap ? (int)nb : grid_cap;     // one workgroup per CU (grid_cap = 256 on the MI355X); a multiple of 8
  hipStream_t s = (hipStream_t)stream;
#define ED_LAUNCH_P(TT)                                                                                                        \
  k_gemm_persist<TT, EPI, CONV><<<grid, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias,            \
                                                     (const uint16_t*)row_bias, (const uint16_t*)residual, (uint16_t*)out,    \
                                                     (int)M, K, I, nbn, (int)nb, img_h, img_w, rows_per_sample > 0 ? rows_per_sample : 1)
  if (dtype == ED_BF16) {
    ED_LAUNCH_P(BF);
  } else if (dtype == ED_F16) {
    ED_LAUNCH_P(HF);
  } else {
    return bad;
  }
#undef ED_LAUNCH_P
  return (int)hipGetLastError();
}

}  // namespace

template <int EPI>
static int launch_persist2(const void* x, const void* w, const void* bias, const void* residual, void* out, int dtype, int64_t M, int K,
                           int I, int grid_cap, void* stream) {
  if (M == 0) return 0;
  const int bad = (int)hipErrorInvalidValue;
  if (M < 0 || K % BK != 0 || K < BK || I <= 0 || (EPI == 0 ? I % BN != 0 : I % 8 != 0)) return bad;
  if (M * (int64_t)K * 2 >= 0x7ffffff0ll || (int64_t)(EPI == 0 ? 2 : 1) * I * K * 2 >= 0x7ffffff0ll) return bad;
  const int nbn = EPI == 0 ? I / BN : (I + 2 * BN - 1) / (2 * BN);
  const int64_t nb = ((M + BM - 1) / BM) * nbn;
  if (nb >= (1ll << 31) || grid_cap <= 0 || grid_cap % 8 != 0) return bad;
  const int grid = nb < grid_cap ? (int)nb : grid_cap;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ED_BF16)
    k_gemm_persist2<BF, EPI><<<grid, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, (const uint16_t*)residual,
                                                   (uint16_t*)out, (int)M, K, I, nbn, (int)nb);
  else if (dtype == ED_F16)
    k_gemm_persist2<HF, EPI><<<grid, 512, 0, s>>>((const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias, (const uint16_t*)residual,
                                                   (uint16_t*)out, (int)M, K, I, nbn, (int)nb);
  else
    return bad;
  return (int)hipGetLastError();
}

extern "C" {
int ed_p2_geglu_gemm(const void* x, const void* w, const void* bias, void* out, int dtype, int64_t M, int K, int I, int grid_cap, void* stream) {
  return launch_persist2<0>(x, w, bias, nullptr, out, dtype, M, K, I, grid_cap, stream);
}
int ed_p2_linear(const void* x, const void* w, const void* bias, const void* residual, void* out, int dtype, int64_t M, int K, int N,
                 int grid_cap, void* stream) {
  return launch_persist2<1>(x, w, bias, residual, out, dtype, M, K, N, grid_cap, stream);
}
int ed_p_geglu_gemm(const void* x, const void* w, const void* bias, void* out, int dtype, int64_t M, int K, int I, int grid_cap, void* stream) {
  return launch_persist<0, false>(x, w, bias, nullptr, nullptr, out, dtype, M, K, I, 0, 0, 0, grid_cap, stream);
}
int ed_p_linear(const void* x, const void* w, const void* bias, const void* residual, void* out, int dtype, int64_t M, int K, int N,
                int grid_cap, void* stream) {
  return launch_persist<1, false>(x, w, bias, nullptr, residual, out, dtype, M, K, N, 0, 0, 0, grid_cap, stream);
}
int ed_p_conv3x3_nhwc(const void* x, const void* w, const void* bias, const void* sample_bias, const void* residual, void* out, int dtype,
                      int B, int H, int W, int Cin, int N, int grid_cap, void* stream) {
  return launch_persist<1, true>(x, w, bias, sample_bias, residual, out, dtype, (int64_t)B * H * W, 9 * Cin, N, H, W, H * W, grid_cap, stream);
}
}
