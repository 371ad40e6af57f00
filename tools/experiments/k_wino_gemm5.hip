// agz_wino5.hip -- the exact-f32 Winograd tower layer with TWO workgroups per compute unit (round 3).
//
// k_wino_gemm4 (agz_wino.hip) holds a 64 tile x 64 cout x 25 plane tile in the 400 accumulator registers of one wave per
// SIMD.  Its K loop runs within 10 % of the MFMA floor, but every non-MFMA phase of a workgroup -- the inverse transform,
// the copy of y, the next layer's input transform: 0.40-0.45 ms of a 2.35 ms layer -- is executed by that lone wave at
// VALU / LDS *latency* with the matrix pipe idle: nothing else is resident on the CU (VERDICT r2 weak #6, 19 % of the
// launch).  A wave with 400 accumulators cannot share its SIMD, so the tile is halved instead:
//
//   workgroup  = 64 tile rows x 32 couts x 25 planes, four waves, each 32 rows x 16 couts as 2 x 1 tiles of
//                v_mfma_f32_16x16x4_f32 (4 accumulator registers per tile and plane: 200 per wave, <= 256 in all)
//   CU         = two such workgroups (2 x 80 KB of LDS, 2 waves per SIMD), independent of each other: while one is in
//                its epilogue the other's K loop has the matrix pipe to itself, and while both are in their K loops
//                they alternate on it.  The f32 16x16x4 form issues every 32 cycles with a 40-cycle dependent latency;
//                a wave's 50 MFMAs per stage are all independent.
//
// Arithmetic: a 16x16x4 MFMA is a k-ordered fmaf chain over the 4 channels of a stage (k = channel; k_wino_gemm4's two
// 32x32x2 MFMAs sum them in the order 0, 2, 1, 3: same products, last-bit differences in the sums); the epilogue uses
// the same formulas.  V (activations) keeps the stage-image layout of agz_wino.hip -- k_wino_in and both kernels' fused
// transforms are interchangeable producers; U (weights) has its own layout (wino5_pack_weights).
//
// A-operand rows: a V row is 16 bytes (4 channels), so 16 CONSECUTIVE rows read with one ds_read_b32 per lane
// (lanes 0-31 = 16 rows x channels 0, 1) would hit every bank twice.  The image swaps the two channel pairs of a row
// when bit 4 of the row is set (wino_v_off), so an MFMA tile takes 8 rows with bit 4 clear and the 8 rows 16 further
// on: tile t of a wave = rows {8 t + i, 8 t + 16 + i : i < 8} of its 32 -- channels 0, 1 of the first eight sit in
// dwords 0, 1 of their banks' window, those of the second eight in dwords 2, 3: 32 lanes, 32 banks.
//
// LDS: a stage = 4 input channels x (64 tile rows + 32 couts) x 25 planes = 25 KB of V + 13 KB of U, double-buffered
// (76 KB) + the 2.3 KB point table: 80,128 B per workgroup.  Double, not triple buffering: the second workgroup of the
// CU is what covers a DMA wait, and 2 x 38 KB in flight per CU is what round 2's kernel had (2 x 52 KB).
#include "agz_wino_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace agz {

constexpr int G5C = 32;                          // couts per workgroup
constexpr int G5_A = WXI * WT * 4;               // floats of V in an LDS stage: 25 planes x 64 rows x 4 channels (6400)
constexpr int G5_BPIECES = 13;                   // U of a stage: 25 planes x 32 couts x 4 k = 12.5 KB -> 13 pieces of 1 KB
constexpr int G5_B = G5_BPIECES * 256;           // 3328 floats
constexpr int G5_STAGE = G5_A + G5_B;            // 9728 floats = 38,912 B
constexpr int G5_NPIECE = WXI + G5_BPIECES;      // 38 LDS-DMA instructions per stage
constexpr int G5_IMG = WT * 9 * G5C;             // epilogue tile image: 18,432 floats = 73,728 B
static_assert(G5_IMG <= 2 * G5_STAGE, "the tile image lives in the two stage buffers");

// row (of the wave's 32) behind A-operand lane index / C-D row index i of tile t: see "A-operand rows" above
__device__ __forceinline__ int g5_row(int t, int i) { return 8 * t + (i & 7) + 16 * (i >> 3); }

// epilogue tile image img[X][32 couts], X = output k * 64 + tile row; the 16-byte unit of channel group g (4 couts)
// sits at position g ^ ((X >> 1) & 7) of the 128-byte row: rows are 128 B, so two consecutive X share a 256-byte bank
// window and eight consecutive pairs rotate through the eight unit positions -- a wave whose lanes are consecutive X
// reads one channel group with a conflict-free ds_read_b128, and the accumulator layout (lane = cout, 4 rows apart)
// writes with 2-way conflicts, which a ds_write_b32 hides.
__device__ __forceinline__ int g5_img_off(int X, int c) { return X * G5C + 4 * ((c >> 2) ^ ((X >> 1) & 7)) + (c & 3); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef AGZ_TIMING_EXPERIMENTS
__device__ int g5_census[2];       // resident workgroups now / at most
#endif

// MODE bit 0: write y (affine, residual, ReLU applied); bit 1: emit the next layer's V stage images.
// NS: stages of the K loop (input channels / 4): 64, or 8 for the stem
// X: timing experiments, instantiated only under -DAGZ_TIMING_EXPERIMENTS (results are WRONG for X != 0):
//   1 = K loop only; 4 = no DMA after the prologue; 5 = no MFMA; 6 = no LDS operand reads; 7 = no stage barrier / DMA wait
template <int MODE, int NS, int X = 0>
__global__ __launch_bounds__(256, 2) void k_wino_gemm5(
    const float* __restrict__ vimg, const float* __restrict__ uimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ vnext, const int* __restrict__ d_count, int N, int T, int relu) {
  __shared__ __attribute__((aligned(16))) float lds[2 * G5_STAGE];
  __shared__ int ptab[WT * 9];     // element offset of output point X in y / res, or -1 (off the board / dead row)
  const int P = N * N, TT = T * T;
  const int RPB = wino_rows_per_block(T);
  const long Mt = (long)(*d_count) * TT;
  // workgroup -> (tile block, cout block): block b runs on XCD b % 8.  An XCD works on four of the eight cout blocks
  // (half of U, 3.3 MB, stays in its 4 MB L2) and on every fourth tile block; the four cout blocks of a tile block
  // are consecutive on one XCD and share its V slab in that L2.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jb = bid >> 3;
  const int cb = 4 * (xcd & 1) + (jb & 3);
  const int tb = (xcd >> 1) + 4 * (jb >> 2);
  if ((long)tb * RPB >= Mt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l15 = lane & 15, kq = lane >> 4;
#ifdef AGZ_TIMING_EXPERIMENTS
  if (tid == 0) atomicMax(&g5_census[1], atomicAdd(&g5_census[0], 1) + 1);
  struct Leave { int tid; __device__ ~Leave() { if (tid == 0) atomicSub(&g5_census[0], 1); } } leave{tid};
  {   // stagger experiment: relu bits 8.. = number of ~3.7 us sleeps, bits 4..7 = which workgroups sleep
    const int loops = relu >> 8, how = (relu >> 4) & 15;
    const bool me = how == 1 ? (bid >= 256 && bid < 512) : how == 2 ? ((bid >> 3) & 1) && bid < 512 : how == 3 ? ((bid >> 8) & 1) : false;
    if (me)
      for (int i = 0; i < loops; ++i) __builtin_amdgcn_s_sleep(127);
    relu &= 1;
  }
#endif
  const float* asrc = vimg + (long)tb * NS * A_STAGE;
  const float* bsrc = uimg + (long)cb * NS * G5_B;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0];

  // piece j of a stage: j < 25 plane j of V, else piece j - 25 of U; wave w moves pieces w, w + 4, ... (10, 10, 9, 9)
  auto dma = [&](int st, int buf, int i) {
    const int j = wave + 4 * i;
    if (j >= G5_NPIECE) return;
    if (X == 8 || X == 9 || X == 81) st &= 1;                 // every stage from the same two (L2-resident) images
    const float* g = j < WXI ? (X == 8 || X == 9 || X == 81 ? vimg : asrc) + (long)st * A_STAGE + j * 256
                             : (X == 9 ? uimg : bsrc) + (long)st * G5_B + (j - WXI) * 256;
    const unsigned dst = j < WXI ? (unsigned)(buf * G5_STAGE + j * 256) : (unsigned)(buf * G5_STAGE + G5_A + (j - WXI) * 256);
    glds16s(g, (unsigned)lane * 16u, lds0 + dst * 4u);
  };
  constexpr int NDMA = (G5_NPIECE + 3) / 4;      // 10 issue slots per wave and stage
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma(0, 0, i);

  for (int idx = tid; idx < WT * 9; idx += 256) {     // (published by the barrier in front of the first operand reads)
    const int row = idx & (WT - 1), k = idx >> 6;       // X = k * 64 + row
    const long tile = (long)tb * RPB + row;
    int off = -1;
    if (row < RPB && tile < Mt) {
      const int b = (int)(tile / TT), t = (int)(tile % TT);
      const int pi = 3 * (t / T) + k / 3, pj = 3 * (t % T) + k % 3;
      if (pi < N && pj < N) off = (b * P + pi + N * pj) * kC + cb * G5C;      // < 2^31: 8192 x 361 x 256 = 7.6e8
    }
    ptab[idx] = off;
  }

  f32x4 acc[WXI][2];
#pragma unroll
  for (int i = 0; i < WXI; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // operand addresses (floats, within a stage): lane (l15, kq) supplies A[row g5_row(t, l15)][channel kq]; the two
  // channel pairs of a V row are swapped when bit 4 of the row is set (wino_v_off), i.e. for l15 >= 8
  const int apos = kq ^ (2 * (l15 >> 3));
  const int aoff0 = (wm * 32 + g5_row(0, l15)) * 4 + apos;
  const int aoff1 = (wm * 32 + g5_row(1, l15)) * 4 + apos;
  const int boff = G5_A + wn * 64 + kq * 16 + l15;           // U of a plane: [cout half][k 4][cout 16]: 32 lanes, 32 banks
  constexpr int LA = 3, RING = 4;                             // operands are read LA planes ahead of their MFMAs
  float ra0[RING], ra1[RING], rb[RING];
  auto load = [&](const float* L, int xi, int slot) {
    if (X == 6) { ra0[slot] = (float)lane; ra1[slot] = 1.f; rb[slot] = (float)xi; return; }
    ra0[slot] = L[aoff0 + xi * 256];
    ra1[slot] = L[aoff1 + xi * 256];
    rb[slot] = L[boff + xi * 128];
  };

  int buf = 0;
#pragma unroll 1
  for (int st = 0; st < NS; ++st) {
    // stage st has landed (this wave's pieces: vmcnt(0); everybody's: the barrier), and every wave has finished
    // reading stage st - 1, whose buffer the DMA of stage st + 1 now overwrites
    if (X != 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const float* L = lds + buf * G5_STAGE;
    const bool more = st + 1 < NS;
#pragma unroll
    for (int k = 0; k < LA; ++k) load(L, k, k % RING);
#pragma unroll
    for (int k = 0; k < WXI; ++k) {
      if (k + LA < WXI) load(L, k + LA, (k + LA) % RING);
      if (X == 21 || X == 22) {                     // experiment: all pieces in the first five planes
        if (k < NDMA / 2) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) { dma(st + 1, buf ^ 1, 2 * k); dma(st + 1, buf ^ 1, 2 * k + 1); }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (k < NDMA) {
        __builtin_amdgcn_sched_barrier(0);
        if (more && X != 4) dma(st + 1, buf ^ 1, k);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (X == 5) {
        asm volatile("" ::"v"(ra0[k % RING]), "v"(ra1[k % RING]), "v"(rb[k % RING]));
        continue;
      }
      acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra0[k % RING], rb[k % RING], acc[k][0], 0, 0, 0);
      acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[k % RING], rb[k % RING], acc[k][1], 0, 0, 0);
    }
    buf ^= 1;
  }
  if (X == 1 || X == 81 || X == 21) {
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < WXI; ++k) keep += acc[k][0][0] + acc[k][1][3];
    if (keep == 123.456f) y[0] = keep;
    return;
  }

  // ---- epilogue.  The stage buffers become the tile image img (layout: g5_img_off).  Phases:
  //   0   residual -> img by LDS-DMA (72 KB in 72 instructions), in flight during the register work of phase 1
  //   1   inverse transform A^T M A + BatchNorm affine in registers, then img = ReLU(img (the residual) + value)
  //   1b  img -> y: 128-byte runs per output point, 16 B per lane; only where y is wanted (MODE & 1)
  //   2   next layer's input transform V = B^T d B from img -> HBM stage images   (MODE & 2)
  __syncthreads();
  float* img = lds;
  if (res) {
    // instruction i fills points 8 i .. 8 i + 7: lane = (point, unit u) fetches channel group u ^ ((X >> 1) & 7)
    for (int i = wave; i < WT * 9 / 8; i += 4) {
      const int Xp = 8 * i + (lane >> 3), u = lane & 7;
      const int off = ptab[Xp];
      const float* g = res + (off >= 0 ? off + 4 * (u ^ ((Xp >> 1) & 7)) : 0);      // dead points: any valid address
      glds16(g, lds0 + (unsigned)(i * 256) * 4u);
    }
  }
  {
    // phase 1.  C/D map of the 16x16 MFMA: col (cout) = lane & 15, row index i = 4 (lane >> 4) + e -> row g5_row(t, i)
    const int col = wn * 16 + l15;
    const float sc = scale[cb * G5C + col], sh = shift[cb * G5C + col];
    const bool relu_now = relu != 0;
    f32x4 o[2][9];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 tmp[3][5];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const f32x4 m0 = acc[0 * 5 + j][t], m1 = acc[1 * 5 + j][t], m2 = acc[2 * 5 + j][t], m3 = acc[3 * 5 + j][t], m4 = acc[4 * 5 + j][t];
        tmp[0][j] = ((m0 + m1) + m2) + m3;
        tmp[1][j] = (m1 - m2) + 2.f * m3;
        tmp[2][j] = ((m1 + m2) + 4.f * m3) + m4;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        o[t][i * 3 + 0] = ((tmp[i][0] + tmp[i][1]) + tmp[i][2]) + tmp[i][3];
        o[t][i * 3 + 1] = (tmp[i][1] - tmp[i][2]) + 2.f * tmp[i][3];
        o[t][i * 3 + 2] = ((tmp[i][1] + tmp[i][2]) + 4.f * tmp[i][3]) + tmp[i][4];
      }
    }
    if (res) {                                    // the residual tile has landed, for every wave
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    auto rows = [&](auto with_res) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float* p0[4];
        float rr[4][9];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = wm * 32 + g5_row(t, 4 * kq + e);
          p0[e] = img + g5_img_off(row, col);          // output k at p0 + k * 64 * 32 (64 k is a multiple of 16: same swizzle)
#pragma unroll
          for (int k = 0; k < 9; ++k) rr[e][k] = decltype(with_res)::value ? p0[e][k * (WT * G5C)] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            float v = o[t][k][e] * sc + sh + rr[e][k];
            if (relu_now) v = fmaxf(v, 0.f);
            p0[e][k * (WT * G5C)] = v;
          }
      }
    };
    if (res) rows(std::true_type{});
    else rows(std::false_type{});
  }
  __syncthreads();
  if (MODE & 1) {
    // phase 1b: element = (point X, 16-byte unit); eight consecutive lanes cover the 128 contiguous bytes of one point.
    // element i of thread tid: X = (tid >> 3) + 32 i, unit position tid & 7 -> channel group (tid ^ (tid >> 4)) & 7 for
    // every i ((X >> 1) & 7 = (tid >> 4) & 7): one LDS address and one channel offset per thread
    constexpr int PER = WT * 9 * (G5C / 4) / 256;     // 18 per thread
    const int cg4 = 4 * ((tid ^ (tid >> 4)) & 7);
    const f32x4* ip0 = reinterpret_cast<const f32x4*>(img) + tid;
    const int* pt0 = ptab + (tid >> 3);
    int offs[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) offs[i] = pt0[32 * i];
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += 9) {
      f32x4 v[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) v[j] = ip0[256 * (i0 + j)];
#pragma unroll
      for (int j = 0; j < 9; ++j)
        if (offs[i0 + j] >= 0) *reinterpret_cast<f32x4*>(y + offs[i0 + j] + cg4) = v[j];
    }
  }
  if (!(MODE & 2)) return;

  // ---- phase 2: the next layer's input transform for this workgroup's 32 channels (= stages 8 cb .. 8 cb + 7 of the
  // next layer's K loop).  Task = (tile row, stage): lane = row, so that the 64 lanes of a wave fill 64 consecutive
  // 16-byte rows of a stage image plane (1 KB per store instruction); wave w takes stages w and w + 4.
  {
    const int row = lane;
    const long tile = (long)tb * RPB + row;
    const bool live = row < RPB && tile < Mt;
    const int t = live ? (int)(tile % TT) : 0, lb = row / TT;
    const int ti = t / T, tj = t % T;
    int pbase[25], pxm[25];            // float offset of point X's row in img (or -1), and its swizzle (X >> 1) & 7
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
      for (int v = 0; v < 5; ++v) {
        const int pi = 3 * ti - 1 + u, pj = 3 * tj - 1 + v;
        const bool ok = live && pi >= 0 && pi < N && pj >= 0 && pj < N;
        // the point lives in tile (pi / 3, pj / 3) of the same board, output k = (pi % 3) * 3 + pj % 3
        const int Xq = ok ? ((pi % 3) * 3 + pj % 3) * WT + lb * TT + (pi / 3) * T + pj / 3 : -1;
        pbase[u * 5 + v] = ok ? Xq * G5C : -1;
        pxm[u * 5 + v] = (Xq >> 1) & 7;
      }
    const bool swap = (row >> 4) & 1;
#pragma unroll 1
    for (int sl = wave; sl < G5C / WK; sl += 4) {
      f32x4 d[25];
#pragma unroll
      for (int q = 0; q < 25; ++q) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        d[q] = pbase[q] >= 0 ? *reinterpret_cast<const f32x4*>(img + pbase[q] + 4 * (sl ^ pxm[q])) : z;
      }
      float* g = vnext + ((long)tb * WNS + (cb * (G5C / WK) + sl)) * A_STAGE + row * 4;
      // B^T d B on channel PAIRS, shared subexpressions: the arithmetic of k_wino_gemm4's phase 2, operation for operation
      auto bt5p = [](f32x2 x0, f32x2 x1, f32x2 x2, f32x2 x3, f32x2 x4, f32x2* r) {
        r[3] = x3 - x1;
        r[0] = 2.f * (x0 - x2) + r[3];
        r[4] = (x4 - x2) - 2.f * r[3];
        r[1] = 2.f * x1 + (x2 - x3);
        r[2] = (3.f * x2 - x3) - 2.f * x1;
      };
      f32x2 vv[25][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x2 tx[25];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
          f32x2 r[5], c[5];
#pragma unroll
          for (int u = 0; u < 5; ++u) c[u] = (f32x2){d[u * 5 + v][2 * h], d[u * 5 + v][2 * h + 1]};
          bt5p(c[0], c[1], c[2], c[3], c[4], r);
#pragma unroll
          for (int i = 0; i < 5; ++i) tx[i * 5 + v] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          f32x2 r[5];
          bt5p(tx[i * 5 + 0], tx[i * 5 + 1], tx[i * 5 + 2], tx[i * 5 + 3], tx[i * 5 + 4], r);
#pragma unroll
          for (int j = 0; j < 5; ++j) vv[i * 5 + j][h] = r[j];
        }
      }
#pragma unroll
      for (int xi = 0; xi < 26; ++xi) {
        const f32x2 z2 = {0.f, 0.f};
        const f32x2 p0 = xi < 25 ? vv[xi < 25 ? xi : 0][0] : z2, p1 = xi < 25 ? vv[xi < 25 ? xi : 0][1] : z2;
        const f32x2 lo = swap ? p1 : p0, hi2 = swap ? p0 : p1;
        const f32x4 v4 = {lo[0], lo[1], hi2[0], hi2[1]};
        __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(g + xi * 256));
      }
    }
  }
}

// ------------------------------------------------------------------ host side

// Flux [kw,kh,cin,cout] column-major -> U5[cout block 8][stage ns][plane 25 (+1 pad)][cout half 2][k 4][cout 16],
// U_xi = G k G^T in float64 (the arithmetic of wino_pack_weights); k = channel of the stage.
void wino5_pack_weights(const ConvHost& c, float* out, int ns) {
  static const double G[5][3] = {{0.5, 0.0, 0.0}, {0.5, 0.5, 0.5}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                                 {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
  const int cin = c.cin, cout = c.cout;
  std::memset(out, 0, sizeof(float) * wino5_weight_floats(ns));
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci) {
      double k[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) k[a][b] = c.w[(2 - a) + 3 * ((2 - b) + 3 * (ci + (size_t)cin * o))];
      const int cb = o / G5C, ol = o % G5C, st = ci / WK, cl = ci % WK;
      for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
          double u = 0.0;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) u += G[i][a] * k[a][b] * G[j][b];
          const int xi = i * 5 + j;
          out[((size_t)cb * ns + st) * G5_B + (size_t)xi * (G5C * 4) + (ol >> 4) * 64 + cl * 16 + (ol & 15)] = (float)u;
        }
    }
}

size_t wino5_weight_floats(int ns) { return (size_t)(kC / G5C) * ns * G5_B; }

void launch_wino_gemm5(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                       float* y, float* vnext, const int* d_count, int bcap, int N, int relu, hipStream_t s, int ns) {
  const int T = (N + 2) / 3;
  const long rpb = wino_rows_per_block(T);
  const int blocks = (int)(((long)bcap * T * T + rpb - 1) / rpb);
  const int per_xcd = 4 * ((blocks + 3) / 4);   // see the placement comment in k_wino_gemm5
  const dim3 grid(8 * per_xcd), block(256);
  if (ns == kWinoStemStages) {                   // the stem: its output is always wanted in HBM (block 0's residual)
    constexpr int S = kWinoStemStages;
    if (vnext) hipLaunchKernelGGL((k_wino_gemm5<3, S>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
    else hipLaunchKernelGGL((k_wino_gemm5<1, S>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
    return;
  }
#ifdef AGZ_TIMING_EXPERIMENTS
  static const int xp = getenv("AGZ_WINO_X") ? atoi(getenv("AGZ_WINO_X")) : 0;
  static const int stg = getenv("AGZ_WINO_STAGGER") ? atoi(getenv("AGZ_WINO_STAGGER")) : 0;     // loops * 256 + how * 16
  relu |= stg;
  static bool once = false;
  if (!once) {
    once = true;
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wino_gemm5<3, WNS, 0>, 256, 0);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)k_wino_gemm5<3, WNS, 0>);
    int zero[2] = {0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g5_census), zero, sizeof(zero));
    hipLaunchKernelGGL((k_wino_gemm5<3, WNS, 1>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(zero, HIP_SYMBOL(g5_census), sizeof(zero));
    fprintf(stderr, "[gemm5] census: resident now %d, max %d\n", zero[0], zero[1]);
    fprintf(stderr, "[gemm5] occupancy API: %d blocks/CU; regs %d, static LDS %zu, grid %d\n", nb, fa.numRegs, fa.sharedSizeBytes, (int)grid.x);
  }
  if (xp && y && vnext && res) {
    auto kern = xp == 1 ? k_wino_gemm5<3, WNS, 1> : xp == 4 ? k_wino_gemm5<3, WNS, 4> : xp == 5 ? k_wino_gemm5<3, WNS, 5>
              : xp == 6 ? k_wino_gemm5<3, WNS, 6> : xp == 7 ? k_wino_gemm5<3, WNS, 7> : xp == 8 ? k_wino_gemm5<3, WNS, 8> : xp == 9 ? k_wino_gemm5<3, WNS, 9> : xp == 81 ? k_wino_gemm5<3, WNS, 81> : xp == 21 ? k_wino_gemm5<3, WNS, 21> : xp == 22 ? k_wino_gemm5<3, WNS, 22> : xp == 11 ? k_wino_gemm5<1, WNS, 0>
              : xp == 12 ? k_wino_gemm5<2, WNS, 0> : k_wino_gemm5<3, WNS, 0>;
    hipLaunchKernelGGL(kern, grid, block, 0, s, vimg, uimg, scale, shift, xp == 12 ? nullptr : res, y, vnext, d_count, N, T, relu);
    return;
  }
#endif
  if (y && vnext)
    hipLaunchKernelGGL((k_wino_gemm5<3, WNS>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
  else if (vnext)
    hipLaunchKernelGGL((k_wino_gemm5<2, WNS>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
  else
    hipLaunchKernelGGL((k_wino_gemm5<1, WNS>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
}

}  // namespace agz
