// agz_wino4.hip -- the 3x3 256->256 tower convolution as Winograd F(4x4, 3x3) on the f32 MFMA, for boards of 13x13 and
// larger (BASELINE configs[3]: 19x19, tower 20).
//
// Why a second Winograd kernel: on the exact-f32 matrix pipe the layer is bound by MFMA work (and by the power that work
// costs: DESIGN.md 4f), so the lever is fewer multiplies.  F(3x3,3x3) (agz_wino.hip) tiles a 19x19 board as 7x7 tiles of
// 3 over a 21-wide cover: 49 x 25 = 1225 multiplies per (cin, cout) and board, 3.39 per output point.  F(4x4,3x3) tiles
// it as 5x5 tiles of 4 over a 20-wide cover: 25 x 36 = 900, 2.49 per point -- 27 % less MFMA work, 25 % less V traffic --
// and f32 has four orders of magnitude of head-room on this network (|d pi| ~ 1e-8 against the 1e-4 bar).
//
//   Y = A^T [ (G k G^T) .* (B^T d B) ] A          interpolation points {0, 1, -1, 2, -2, inf}
//   B^T = [ 4  0 -5  0  1  0 ]   A^T = [ 1  1  1  1  1  0 ]   G = [  1/4    0     0   ]
//         [ 0 -4 -4  1  1  0 ]         [ 0  1 -1  2 -2  0 ]       [ -1/6  -1/6  -1/6  ]
//         [ 0  4 -4 -1  1  0 ]         [ 0  1  1  4  4  0 ]       [ -1/6   1/6  -1/6  ]
//         [ 0 -2 -1  2  1  0 ]         [ 0  1 -1  8 -8  1 ]       [ 1/24  1/12   1/6  ]
//         [ 0  2 -1 -2  1  0 ]                                     [ 1/24 -1/12   1/6  ]
//         [ 0  4  0 -5  0  1 ]                                     [  0     0     1   ]
//   (B^T is integral, so the input transform is exact up to its own sums; G is applied on the host in float64.)
//
// The problem this kernel is built around: a workgroup tile of 64 tiles x 64 couts (the one that gives 16 flop per
// LDS-DMA byte, agz_wino.hip) needs 36 planes x 16 accumulator registers = 576 per lane with one wave per SIMD -- more
// than the 512 a lane owns.  So the 36 planes are multiplied in four PASSES over the input channels, and each pass's
// planes are folded into the inverse transform as soon as its K loop ends.  With M[i][j] the plane of transform row i
// and column j,  Y = A^T M A = sum_i A^T[:, i] (x) (A^T M[i][:]) :
//     pass A: rows 1, 2  (12 planes)   t_i = A^T M[i][:]  (four 16-register tuples per row);  S = t1 + t2, D = t1 - t2
//     pass B: rows 3, 4  (12 planes)   s = t3 + t4, d = t3 - t4;  Y0 = S + s, Y1 = D + 2d, Y2 = S + 4s, Y3 = D + 8d
//     pass C: row 0      ( 6 planes)   Y0 += t0
//     pass D: row 5      ( 6 planes)   Y3 += t5
// Live registers: 192 accumulators (pass A), 192 + 128 (B), 96 + 256 (C, D): the last two passes, where the 16 output
// tuples are complete but for one term, run with the fewest accumulators.  Every byte and every MFMA of the one-pass form
// is kept: a pass moves only its own planes of V and U.
//
// Stage = 24 UNITS; a unit = one plane x 4 input channels = 64 rows x 16 B of V and of U (1 KB each), two MFMAs per wave.
// Passes A, B: 12 planes x 2 channel groups per stage (32 stages each); passes C, D: 6 planes x 4 groups (16 stages each):
// 96 stages of 48 KB, triple-buffered in LDS and filled by LDS-DMA exactly like agz_wino.hip's, 12 pieces per wave and stage.
// The stage sequence is one flat list in HBM ([pass][stage][unit]), so the DMA stream runs across pass boundaries.
//
// Accumulators are TRANSPOSED (D = U^T-rows x V-rows: lane = tile row, register = cout): a lane holds, per output point,
// four consecutive couts in a register quad, so the epilogue's image pass moves 16-byte units, and the 64 couts split
// into two halves BY REGISTER INDEX (all four waves work on either half).  That matters because the tile image of 64
// tiles x 16 outputs x 64 couts (256 KB) does not fit the LDS: the epilogue runs twice, on a 128 KB image of 32 couts.
//
// Epilogue per half: residual half-tile -> image by LDS-DMA; BatchNorm affine in registers; image = ReLU(image + value);
// image -> y; the NEXT layer's input transform V = B^T d B (6x6 patches from the image, lane = tile row, 1 KB contiguous
// stores) for every tile whose patch lies in this tile block (whole-board blocks for N = 13..16: 16 tiles per board, 4
// boards per block; dense blocks above, where k_wino4_in<FIXUP> does the block ends -- same split as agz_wino.hip).
#include "agz_nn.h"
#include "agz_glds.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace agz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4T = 64;                       // tile rows per workgroup
constexpr int W4C = 64;                       // couts per workgroup
constexpr int W4UNITS = 24;                   // units per stage
constexpr int W4UNIT = W4T * 4;               // floats of a unit image (64 rows x 4 channels): 1 KB
constexpr int W4HALF = W4UNITS * W4UNIT;      // one operand's part of a stage: 6144 floats = 24 KB
constexpr int W4STAGE = 2 * W4HALF;           // 48 KB
constexpr int W4NST = kWino4Stages;           // 96 stages: 32 (rows 1,2) + 32 (rows 3,4) + 16 (row 0) + 16 (row 5)
constexpr int W4BLOCK = W4NST * W4HALF;       // floats of V per tile block / of U per cout block: 589,824 (2.25 MB)
constexpr int W4CP = 32;                      // couts per epilogue half
constexpr int W4IMG = 16 * W4T * W4CP;        // floats of the half image: 128 KB
static_assert(W4NST == 96 && kC == 256, "pass structure below assumes 64 channel groups");

// where plane (i, j) of 4-channel group c (0..63) lives: stage of the flat list, unit within the stage
__host__ __device__ __forceinline__ void w4_slot(int i, int j, int c, int& stage, int& unit) {
  if (i == 1 || i == 2) { stage = c >> 1; unit = (c & 1) * 12 + (i - 1) * 6 + j; }
  else if (i == 3 || i == 4) { stage = 32 + (c >> 1); unit = (c & 1) * 12 + (i - 3) * 6 + j; }
  else if (i == 0) { stage = 64 + (c >> 2); unit = (c & 3) * 6 + j; }
  else { stage = 80 + (c >> 2); unit = (c & 3) * 6 + j; }
}
// a unit image row (either operand) is the unit's 4 channels as two pairs, pair h at slot (h + (row >> 4)) & 1: rows r and
// r + 16 start on the same bank and take different slots, so the 32-row ds_read_b64 of the K loop is conflict-free, and a
// producer whose lane is the row writes whole 16-byte rows (agz_wino.hip's V layout; here for U as well)
__host__ __device__ __forceinline__ int w4_off(int row, int h) { return row * 4 + 2 * ((h + (row >> 4)) & 1); }
// U image row r of a cout block -> cout within the block.  The transposed accumulators give lane (l31, hi) of wave wn the
// couts i = (e & 3) + 8 (e >> 2) + 4 hi, e = register; registers 0..7 (i < 16) are epilogue half 0, 8..15 half 1.  Mapping
// i -> (i >> 4) * 32 + wn * 16 + (i & 15) makes each half 32 CONSECUTIVE couts (128-byte runs of y, 8 channel groups of
// the next layer's K loop).
__host__ __device__ __forceinline__ int w4_cout_of_urow(int r) { return (((r & 31) >> 4) << 5) + ((r >> 5) << 4) + (r & 15); }

// tile geometry (shared with agz_wino.hip's rules, for T = ceil(N / 4)): whole boards per 64-row block when they pack
// with <= 10 % waste (T*T = 16: N = 13..16), else dense packing
__host__ __device__ inline int w4_rows_per_block(int T) {
  const int tt = T * T;
  const int whole = (W4T / tt) * tt;
  return (tt <= W4T && whole * 10 >= W4T * 9) ? whole : W4T;
}
__host__ __device__ inline bool w4_whole_boards(int T) { return w4_rows_per_block(T) % (T * T) == 0 && T * T <= W4T; }
// dense blocks: is the 6x6 input patch of tile (ti, tj) in row `row` made of tiles of the same block?  (its neighbours
// (ti + di, tj + dj) are rows row + di T + dj)
__host__ __device__ __forceinline__ bool w4_tile_fused(int T, int row, int ti, int tj) {
  const int lo = (ti > 0 ? T : 0) + (tj > 0 ? 1 : 0), hi = (ti < T - 1 ? T : 0) + (tj < T - 1 ? 1 : 0);
  return row - lo >= 0 && row + hi < W4T;
}

// ------------------------------------------------------------------ B^T x, six values
// ONE arithmetic for both producers of V (k_wino4_in: scalars; the GEMM epilogue: channel pairs), every multiply-add
// an explicit fma: with dense tile blocks a tile is transformed by one or the other depending on its batch row, and a
// network output must not depend on the batch row (tree parity rests on it).
template <typename V>
__device__ __forceinline__ V w4_fma(float c, V a, V b) {
  if constexpr (std::is_same<V, float>::value) return __builtin_fmaf(c, a, b);
  else return __builtin_elementwise_fma((V){c, c}, a, b);
}
template <typename V>
__device__ __forceinline__ void bt6(V x0, V x1, V x2, V x3, V x4, V x5, V* r) {
  const V a = w4_fma<V>(-4.f, x2, x4), b = w4_fma<V>(-4.f, x1, x3);
  const V c = x4 - x2, d = x3 - x1;
  r[1] = a + b;
  r[2] = a - b;
  r[3] = w4_fma<V>(2.f, d, c);
  r[4] = w4_fma<V>(-2.f, d, c);
  r[0] = w4_fma<V>(4.f, x0, w4_fma<V>(-5.f, x2, x4));
  r[5] = w4_fma<V>(4.f, x1, w4_fma<V>(-5.f, x3, x5));
}

// ------------------------------------------------------------------ input transform (first layer; block ends)
// x[M][256] -> V stage images.  grid = 2 x tile blocks (32-tile halves); 256 threads = 32 tiles x 8 lanes (a lane = one
// channel pair of one of 4 channel groups): the eight lanes of a tile read one 64-byte run of every patch point.  A pass
// covers 16 channels; the 36 transformed planes go to an LDS copy of this half-block's part of their unit images and leave
// for HBM as whole 512-byte runs, 16 B per lane.  FIXUP (dense tile blocks only): only the tiles the previous layer's GEMM
// epilogue could not emit (!w4_tile_fused) and the rows past the batch (zeros) are transformed and stored.
template <bool FIXUP>
__global__ __launch_bounds__(256) void k_wino4_in(const float* __restrict__ x, float* __restrict__ vimg,
                                                  const int* __restrict__ d_count, int N, int T) {
  constexpr int TPB = 32, CH = TPB * 4;          // floats per chunk: 32 rows of one unit
  constexpr int IMG = 36 * CH + 8;               // stride between the four groups' copies (+8: bank skew of the 8-byte writes)
  __shared__ __attribute__((aligned(16))) float img[4 * IMG];
  const int P = N * N, TT = T * T;
  const int RPB = w4_rows_per_block(T);
  const long Mt = (long)(*d_count) * TT;
  const int tb = blockIdx.x >> 1, part = blockIdx.x & 1;
  if ((long)tb * RPB + part * TPB >= Mt) return;
  const int hs = threadIdx.x & 7, h = hs & 1, sl = hs >> 1;
  const int tl = threadIdx.x >> 3;
  const int row = part * TPB + tl;
  const long tile = (long)tb * RPB + row;
  bool live = row < RPB && tile < Mt;
  const int b = live ? (int)(tile / TT) : 0, t = live ? (int)(tile % TT) : 0;
  const int ti = t / T, tj = t % T;
  if (FIXUP && live && w4_tile_fused(T, row, ti, tj)) live = false;       // in place already
  int off[36];
#pragma unroll
  for (int u = 0; u < 6; ++u)
#pragma unroll
    for (int v = 0; v < 6; ++v) {
      const int pi = 4 * ti - 1 + u, pj = 4 * tj - 1 + v;
      const bool ok = live && pi >= 0 && pi < N && pj >= 0 && pj < N;
      off[u * 6 + v] = ok ? (b * P + pi + N * pj) * kC : -1;               // < 2^31 (checked by the launcher)
    }
  float* mine = img + sl * IMG + tl * 4 + 2 * ((h + (row >> 4)) & 1);
  float* gdst = vimg + (long)tb * W4BLOCK + part * CH;
  const int cq = threadIdx.x >> 5, cl = threadIdx.x & 31;                 // copy-out: 8 chunks per round, 32 lanes each
  bool copy_row = true;
  if (FIXUP) {
    const int crow = part * TPB + cl;
    const long ctile = (long)tb * RPB + crow;
    const int ct = (int)(ctile % TT);
    copy_row = !(crow < RPB && ctile < Mt && w4_tile_fused(T, crow, ct / T, ct % T));
  }
  for (int pass = 0; pass < kC / 16; ++pass) {
    const int ch = pass * 16 + sl * 4 + 2 * h;
    f32x2 d[36];
#pragma unroll
    for (int q = 0; q < 36; ++q)
      d[q] = off[q] >= 0 ? *reinterpret_cast<const f32x2*>(x + off[q] + ch) : (f32x2){0.f, 0.f};
    f32x2 tx[36];
#pragma unroll
    for (int v = 0; v < 6; ++v) {
      f32x2 r[6];
      bt6<f32x2>(d[0 * 6 + v], d[1 * 6 + v], d[2 * 6 + v], d[3 * 6 + v], d[4 * 6 + v], d[5 * 6 + v], r);
#pragma unroll
      for (int i = 0; i < 6; ++i) tx[i * 6 + v] = r[i];
    }
    if (pass) __syncthreads();                     // the previous pass has left the LDS image
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      f32x2 r[6];
      bt6<f32x2>(tx[i * 6 + 0], tx[i * 6 + 1], tx[i * 6 + 2], tx[i * 6 + 3], tx[i * 6 + 4], tx[i * 6 + 5], r);
#pragma unroll
      for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2*>(mine + (i * 6 + j) * CH) = r[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 18; ++r) {
      const int c = r * 8 + cq, s4 = c / 36, pl = c - s4 * 36;            // chunk c = (group s4 of this pass, plane pl)
      const f32x4 v = *reinterpret_cast<const f32x4*>(img + s4 * IMG + pl * CH + cl * 4);
      int stage, unit;
      w4_slot(pl / 6, pl % 6, pass * 4 + s4, stage, unit);
      f32x4* gp = reinterpret_cast<f32x4*>(gdst + (long)stage * W4HALF + unit * W4UNIT + cl * 4);
      if (FIXUP && !copy_row) continue;
      __builtin_nontemporal_store(v, gp);
    }
  }
}

// ------------------------------------------------------------------ GEMM in four passes + output transform + next input transform

// A^T m for one transform row: six planes -> four tuples
__device__ __forceinline__ void w4_fold_row(const f32x16& m0, const f32x16& m1, const f32x16& m2, const f32x16& m3,
                                            const f32x16& m4, const f32x16& m5, f32x16* t) {
  const f32x16 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  t[0] = (m0 + s12) + s34;
  t[1] = d12 + 2.f * d34;
  t[2] = s12 + 4.f * s34;
  t[3] = (d12 + 8.f * d34) + m5;
}

// MODE bit 0: write y (affine, residual, ReLU applied); bit 1: emit the next layer's V stage images.
// X (timing experiments, -DAGZ_TIMING_EXPERIMENTS only; results WRONG): 1 = K loops and folds only; 2 = no phase 2;
// 4 = no DMA after the prologue; 5 = no MFMA
template <int MODE, int X>
__global__ __launch_bounds__(256, 1) void k_wino4_gemm(
    const float* __restrict__ vimg, const float* __restrict__ uimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ vnext, const int* __restrict__ d_count, int N, int T, int relu) {
  __shared__ __attribute__((aligned(256))) float lds[3 * W4STAGE];      // 144 KB: stage buffers, then the half image
  __shared__ int ptab[W4T * 16];            // element offset of output point X = k * 64 + row in y / res, or -1
  __shared__ __attribute__((aligned(256))) float zeros[64];      // what phase 2 reads for a patch point off the board
  const int P = N * N, TT = T * T;
  const int RPB = w4_rows_per_block(T);
  const long Mt = (long)(*d_count) * TT;
  // workgroup -> (tile block, cout block): the four cout blocks of a tile block are four consecutive workgroups of one
  // XCD (block b runs on XCD b % 8), so its V slab comes out of HBM once (agz_wino.hip's placement)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jb = bid >> 3;
  const int cb = jb & 3;
  const int tb = xcd + 8 * (jb >> 2);
  if ((long)tb * RPB >= Mt) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const float* asrc = vimg + (long)tb * W4BLOCK;
  const float* bsrc = uimg + (long)cb * W4BLOCK;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0];

  // a stage is 48 pieces of 1 KB: 0..23 V units, 24..47 U units; wave w moves pieces 12 w .. 12 w + 11 (waves 0, 1: V;
  // 2, 3: U), wave-uniform base in SGPRs + a 32-bit lane offset
  auto dma = [&](int st, int buf, int j) {
    const int c = 12 * wave + j;
    const float* g = wave < 2 ? asrc + (long)st * W4HALF + c * W4UNIT : bsrc + (long)st * W4HALF + (c - 24) * W4UNIT;
    glds16s(g, (unsigned)lane * 16u, lds0 + (unsigned)(buf * W4STAGE + c * W4UNIT) * 4u);
  };
#pragma unroll
  for (int j = 0; j < 12; ++j) dma(0, 0, j);

  if (tid < 64) zeros[tid] = 0.f;
  for (int idx = tid; idx < W4T * 16; idx += 256) {     // (published by the barrier in front of the first operand reads)
    const int row = idx & (W4T - 1), k = idx >> 6;
    const long tile = (long)tb * RPB + row;
    int off = -1;
    if (row < RPB && tile < Mt) {
      const unsigned tile32 = (unsigned)tile, b = tile32 / (unsigned)TT, t = tile32 - b * (unsigned)TT;
      const unsigned ti = t / (unsigned)T;
      const int pi = (int)(4 * ti) + (k >> 2), pj = (int)(4 * (t - ti * T)) + (k & 3);
      if (pi < N && pj < N) off = ((int)b * P + pi + N * pj) * kC + cb * W4C;
    }
    ptab[idx] = off;
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) dma(1, 1, j);

  f32x16 acc[12];
  const int arow = wm * 32 + l31, brow = wn * 32 + l31;
  const int aoff = w4_off(arow, hi), boff = W4HALF + w4_off(brow, hi);
  constexpr int LA = 5, RING = LA + 1;       // 24 % RING == 0: ring slots are compile-time within a stage
  float2 ra[RING], rb[RING];
  auto load = [&](const float* L, int u, float2& a, float2& b) {
    a = *reinterpret_cast<const float2*>(L + aoff + u * W4UNIT);
    b = *reinterpret_cast<const float2*>(L + boff + u * W4UNIT);
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // transposed: srcA = U (its rows become D's rows = registers: couts), srcB = V (D's columns = lanes: tile rows)
  auto mma0 = [&](int p, const float2& a, const float2& b) {
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, zero16, 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[p], 0, 0, 0);
  };
  auto mma = [&](int p, const float2& a, const float2& b) {
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[p], 0, 0, 0);
    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[p], 0, 0, 0);
  };

  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // stage 0 has landed (stage 1's 12 pieces may be in flight)
  __syncthreads();
#pragma unroll
  for (int u = 0; u < LA; ++u) load(lds, u, ra[u], rb[u]);

  // One stage of a pass with PL planes (unit u = group * PL + plane).  MORE: stage st + 2 exists and is fetched during
  // this stage; NEXT: stage st + 1 exists; FIRST: the pass's first stage (its planes' first MFMAs take C = 0).
  int buf = 0;
  auto stage = [&](int st, auto pl_c, auto more_c, auto next_c, auto first_c) {
    constexpr int PL = decltype(pl_c)::value;
    constexpr bool more = decltype(more_c)::value, next = decltype(next_c)::value, first = decltype(first_c)::value;
    const int nbuf = buf == 2 ? 0 : buf + 1;
    const int dbuf = buf == 0 ? 2 : buf - 1;
    const float* L = lds + buf * W4STAGE;
    const float* Ln = lds + nbuf * W4STAGE;
#pragma unroll
    for (int u = 0; u < W4UNITS; ++u) {
      const int t = u + LA;
      if (t == W4UNITS && next) {
        // everything this wave owes to stage st + 1 has landed (its 12 pieces of stage st + 2, all issued by now, may be
        // in flight); hipcc adds lgkmcnt(0) in front of the barrier: all reads of stage st are back
        if (more && X != 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (t < W4UNITS) load(L, t, ra[t % RING], rb[t % RING]);
      else if (next) load(Ln, t - W4UNITS, ra[t % RING], rb[t % RING]);
      constexpr int D0 = 6;
      if (more && u >= D0 && u < D0 + 12) {
        __builtin_amdgcn_sched_barrier(0);
        if (X != 4) dma(st + 2, dbuf, u - D0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (X != 5) {
        if (first && u < PL) mma0(u % PL, ra[u % RING], rb[u % RING]);
        else mma(u % PL, ra[u % RING], rb[u % RING]);
      }
    }
    buf = nbuf;
  };
  using c12 = std::integral_constant<int, 12>;
  using c6 = std::integral_constant<int, 6>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  if (X == 5) {
#pragma unroll
    for (int p = 0; p < 12; ++p) acc[p] = zero16;
  }

  f32x16 Y[16];      // Y[4 i' + j'] = output point (i', j') of the tile; complete after pass D
  // ---- pass A: transform rows 1, 2
  stage(0, c12{}, T_{}, T_{}, T_{});
  for (int st = 1; st < 32; ++st) stage(st, c12{}, T_{}, T_{}, F_{});
  {
    f32x16 t1[4], t2[4];
    w4_fold_row(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], t1);
    w4_fold_row(acc[6], acc[7], acc[8], acc[9], acc[10], acc[11], t2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Y[0 + j] = t1[j] + t2[j];      // S_j (goes into Y0 and Y2)
      Y[4 + j] = t1[j] - t2[j];      // D_j (goes into Y1 and Y3)
      asm volatile("" : "+v"(Y[0 + j]));      // pin the fold here: hipcc otherwise sinks it towards the epilogue and
      asm volatile("" : "+v"(Y[4 + j]));      // keeps every pass's accumulators alive in scratch
    }
  }
  // ---- pass B: rows 3, 4
  stage(32, c12{}, T_{}, T_{}, T_{});
  for (int st = 33; st < 64; ++st) stage(st, c12{}, T_{}, T_{}, F_{});
  {
    f32x16 t3[4], t4[4];
    w4_fold_row(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], t3);
    w4_fold_row(acc[6], acc[7], acc[8], acc[9], acc[10], acc[11], t4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x16 s = t3[j] + t4[j], d = t3[j] - t4[j];
      const f32x16 S = Y[0 + j], D = Y[4 + j];
      Y[0 + j] = S + s;
      Y[4 + j] = D + 2.f * d;
      Y[8 + j] = S + 4.f * s;
      Y[12 + j] = D + 8.f * d;
      asm volatile("" : "+v"(Y[0 + j]));
      asm volatile("" : "+v"(Y[4 + j]));
      asm volatile("" : "+v"(Y[8 + j]));
      asm volatile("" : "+v"(Y[12 + j]));
    }
  }
  // ---- pass C: row 0
  stage(64, c6{}, T_{}, T_{}, T_{});
  for (int st = 65; st < 80; ++st) stage(st, c6{}, T_{}, T_{}, F_{});
  {
    f32x16 t0[4];
    w4_fold_row(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], t0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Y[0 + j] += t0[j];
      asm volatile("" : "+v"(Y[0 + j]));
    }
  }
  // ---- pass D: row 5
  stage(80, c6{}, T_{}, T_{}, T_{});
  for (int st = 81; st < W4NST - 2; ++st) stage(st, c6{}, T_{}, T_{}, F_{});
  stage(W4NST - 2, c6{}, F_{}, T_{}, F_{});
  stage(W4NST - 1, c6{}, F_{}, F_{}, F_{});
  {
    f32x16 t5[4];
    w4_fold_row(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], t5);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Y[12 + j] += t5[j];
      asm volatile("" : "+v"(Y[12 + j]));
    }
  }
  if (X == 1) {
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) keep += Y[k][0] + Y[k][15];
    if (keep == 123.456f) y[0] = keep;
    return;
  }

  // ---- epilogue, once per half of the 64 couts (registers 8 hh .. 8 hh + 7 of every tuple = couts 32 hh .. 32 hh + 31).
  // Half image img[X][8 units of 16 B], X = k * 64 + tile row; unit u of row X sits at slot u ^ ((X >> 1) & 7): the 16
  // lanes of a ds_read_b128 group (consecutive rows) then cover 64 banks once, and an LDS-DMA instruction fills eight
  // rows (1 KB), each lane choosing the global 16 B that belong in its slot.
  float* img = lds;
  
  const int trow = wm * 32 + l31;                       // this lane's tile row
  float sc[16], sh[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = (e & 3) + 8 * (e >> 2) + 4 * hi;
    const int co = cb * W4C + (i >> 4) * 32 + wn * 16 + (i & 15);
    sc[e] = scale[co];
    sh[e] = shift[co];
  }
  const float relu_lo = relu ? 0.f : -3.0e38f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                       // every wave has left the K loop: the stage buffers are dead

  auto half = [&](auto hh_c) {
    constexpr int hh = decltype(hh_c)::value;
    if (res) {
      // instruction n of wave w fills image rows 8 (w + 4 n) .. + 7: lane = (row X, slot s) fetches unit s ^ ((X >> 1) & 7)
#pragma unroll 4
      for (int n = 0; n < 32; ++n) {
        const int i = wave + 4 * n;
        const int Xp = 8 * i + (lane >> 3), u = (lane & 7) ^ ((Xp >> 1) & 7);
        const int off = ptab[Xp];
        const unsigned boff = off >= 0 ? 4u * (unsigned)(off + hh * W4CP + 4 * u) : 0u;
        glds16s(res, boff, lds0 + (unsigned)(i * 256) * 4u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the residual half-tile has landed, for every wave
      __syncthreads();
    }
    // img = ReLU(img (the residual) + scale * value + shift): the lane's two register quads of this half for each of the
    // 16 output points; unit wn * 4 + 2 qd + hi of row X = k * 64 + trow
    {
      const unsigned rowb = lds0 + 4u * (unsigned)(trow * W4CP);
      const int swz = (trow >> 1) & 7;                   // (X >> 1) & 7 = (trow >> 1) & 7: k * 64 does not reach bits 1..3
      const unsigned a0 = rowb + 16u * (unsigned)((wn * 4 + hi) ^ swz), a1 = rowb + 16u * (unsigned)((wn * 4 + 2 + hi) ^ swz);
#pragma unroll
      for (int k0 = 0; k0 < 16; k0 += 4) {
        f32x4 rr[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const unsigned kb = 4u * (unsigned)((k0 + kk) * W4T * W4CP);
          rr[kk][0] = res ? *(const __attribute__((address_space(3))) f32x4*)(size_t)(a0 + kb) : (f32x4){0.f, 0.f, 0.f, 0.f};
          rr[kk][1] = res ? *(const __attribute__((address_space(3))) f32x4*)(size_t)(a1 + kb) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const unsigned kb = 4u * (unsigned)((k0 + kk) * W4T * W4CP);
#pragma unroll
          for (int qd = 0; qd < 2; ++qd) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              constexpr int e0 = 8 * hh;
              const int e = e0 + 4 * qd + c;
              v[c] = fmaxf(__builtin_fmaf(Y[k0 + kk][e], sc[e], sh[e]) + rr[kk][qd][c], relu_lo);
            }
            *(__attribute__((address_space(3))) f32x4*)(size_t)((qd ? a1 : a0) + kb) = v;
          }
        }
      }
    }
    __syncthreads();

    if (MODE & 1) {
      // image -> y: 8 consecutive lanes cover the 128 contiguous bytes of one output point's half; thread tid handles
      // points X = (tid >> 3) + 32 i, always slot tid & 7 = unit (tid & 7) ^ ((tid >> 4) & 7) (32 i does not reach bits 1..3)
      const int cg4 = hh * W4CP + 4 * ((tid & 7) ^ ((tid >> 4) & 7));
      const f32x4* ip0 = reinterpret_cast<const f32x4*>(img) + tid;
      const int* pt0 = ptab + (tid >> 3);
#pragma unroll
      for (int i0 = 0; i0 < 32; i0 += 8) {
        f32x4 v[8];
        int offs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = ip0[256 * (i0 + j)];
          offs[j] = pt0[32 * (i0 + j)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (offs[j] >= 0) *reinterpret_cast<f32x4*>(y + offs[j] + cg4) = v[j];
      }
    }

    if ((MODE & 2) && X != 2) {
      // ---- the next layer's input transform for this half's 32 channels = 8 channel groups (groups cb * 16 + hh * 8 + g
      // of the next layer's 64).  Task = (tile row, group): lane = row, wave w takes groups w and w + 4.
      const int row = lane;
      const long tile = (long)tb * RPB + row;
      const bool live = row < RPB && tile < Mt;
      const int t = live ? (int)(tile % TT) : 0;
      const int ti = t / T, tj = t % T;
      const bool emit = w4_whole_boards(T) || (live && w4_tile_fused(T, row, ti, tj));
      // Patch point (u, v) of tile (ti, tj) is board point (4 ti - 1 + u, 4 tj - 1 + v): output (ku, kv) of the tile du
      // tile rows / dv tiles further on, (du, ku) = (-1, 3), (0, 0..3), (1, 0) for u = 0..5.  A point off the board (or a
      // lane that emits nothing) reads a block of zeros instead of being masked out.  adr[q]: LDS byte address of the
      // point's unit for group `wave` (+ the lane's pair order: rows with bit 4 set store pair 1 first, so they read it
      // first); group wave + 4 is the same address with bit 6 flipped (slot ^ 4) -- also inside the 256-byte zeros block.
      unsigned adr[36];
      const unsigned zadr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&zeros[0];
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v = 0; v < 6; ++v) {
          const int du = u == 0 ? -1 : (u == 5 ? 1 : 0), ku = u == 0 ? 3 : (u == 5 ? 0 : u - 1);
          const int dv = v == 0 ? -1 : (v == 5 ? 1 : 0), kv = v == 0 ? 3 : (v == 5 ? 0 : v - 1);
          const int pi = 4 * ti - 1 + u, pj = 4 * tj - 1 + v;
          const bool ok = live && emit && pi >= 0 && pi < N && pj >= 0 && pj < N;
          const int Xq = (ku * 4 + kv) * W4T + row + du * T + dv;
          adr[u * 6 + v] = ok ? lds0 + 4u * (unsigned)(Xq * W4CP + 4 * (wave ^ ((Xq >> 1) & 7)) + 2 * ((row >> 4) & 1)) : zadr;
        }
#pragma unroll 1
      for (int it = 0; it < 2; ++it) {
        const int g = wave + 4 * it;
        const unsigned xo = (unsigned)it << 6;
        f32x4 d[36];
#pragma unroll
        for (int q = 0; q < 36; ++q) {
          const unsigned a0 = adr[q] ^ xo;
          const f32x2 a = *(const __attribute__((address_space(3))) f32x2*)(size_t)a0;
          const f32x2 b = *(const __attribute__((address_space(3))) f32x2*)(size_t)(a0 ^ 8u);
          d[q] = (f32x4){a[0], a[1], b[0], b[1]};
        }
        const int c = cb * 16 + hh * 8 + g;               // the next layer's channel group
        float* gbase = vnext + (long)tb * W4BLOCK + row * 4;
        f32x2 vv[36][2];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          f32x2 tx[36];
#pragma unroll
          for (int v = 0; v < 6; ++v) {
            f32x2 r[6], cc[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) cc[u] = (f32x2){d[u * 6 + v][2 * hp], d[u * 6 + v][2 * hp + 1]};
            bt6<f32x2>(cc[0], cc[1], cc[2], cc[3], cc[4], cc[5], r);
#pragma unroll
            for (int i = 0; i < 6; ++i) tx[i * 6 + v] = r[i];
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            f32x2 r[6];
            bt6<f32x2>(tx[i * 6 + 0], tx[i * 6 + 1], tx[i * 6 + 2], tx[i * 6 + 3], tx[i * 6 + 4], tx[i * 6 + 5], r);
#pragma unroll
            for (int j = 0; j < 6; ++j) vv[i * 6 + j][hp] = r[j];
          }
        }
        // (the channel group's stage / unit bases: rows 1..4 live in stages c >> 1 (+ 32), rows 0 and 5 in 64 / 80 + (c >> 2))
        const int base12 = (c >> 1) * W4HALF + (c & 1) * 12 * W4UNIT, base6 = (c >> 2) * W4HALF + (c & 3) * 6 * W4UNIT;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int o = (i == 1 || i == 2) ? base12 + ((i - 1) * 6 + j) * W4UNIT
                        : (i == 3 || i == 4) ? 32 * W4HALF + base12 + ((i - 3) * 6 + j) * W4UNIT
                        : i == 0 ? 64 * W4HALF + base6 + j * W4UNIT : 80 * W4HALF + base6 + j * W4UNIT;
            const f32x2 p0 = vv[i * 6 + j][0], p1 = vv[i * 6 + j][1];
            const f32x4 v4 = {p0[0], p0[1], p1[0], p1[1]};      // (already in the row's pair order: see the reads above)
            f32x4* gp = reinterpret_cast<f32x4*>(gbase + o);
            if (emit) __builtin_nontemporal_store(v4, gp);
          }
      }
    }
  };
  half(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (stores and loads share vmcnt: nothing of the first half is counted into the second half's residual wait)
  __syncthreads();                                   // every wave has left the image: the second half may overwrite it
  half(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------ host side

// Flux [kw,kh,cin,cout] column-major -> U stage images [cout block 4][stage 96][unit 24][row 64][4], U = G k G^T in
// float64.  k is the CORRELATION kernel (NNlib's conv is a true convolution: tap (a', b') carries w[2 - a', 2 - b']).
void wino4_pack_weights(const ConvHost& c, float* out) {
  static const double G[6][3] = {{0.25, 0.0, 0.0},         {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  const int cin = c.cin, cout = c.cout;
  AGZ_REQUIRE(cin == kC && cout == kC, AGZ_BAD_ARGUMENT, "F(4x4,3x3) pack: tower layers only (%d -> %d)", cin, cout);
  std::memset(out, 0, sizeof(float) * wino4_weight_floats());
  std::vector<int> row_of(W4C);
  for (int r = 0; r < W4C; ++r) row_of[w4_cout_of_urow(r)] = r;
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci) {
      double k[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) k[a][b] = c.w[(2 - a) + 3 * ((2 - b) + 3 * (ci + (size_t)cin * o))];
      const int cb = o / W4C, r = row_of[o % W4C], cg = ci / 4, cl = ci % 4;
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          double u = 0.0;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) u += G[i][a] * k[a][b] * G[j][b];
          int stage, unit;
          w4_slot(i, j, cg, stage, unit);
          out[(size_t)cb * W4BLOCK + (size_t)stage * W4HALF + (size_t)unit * W4UNIT + w4_off(r, cl >> 1) + (cl & 1)] = (float)u;
        }
    }
}

size_t wino4_weight_floats() { return (size_t)(kC / W4C) * W4BLOCK; }
static long wino4_blocks(int bcap, int T) {
  const long rpb = w4_rows_per_block(T);
  return ((long)bcap * T * T + rpb - 1) / rpb;
}
size_t wino4_v_floats(int bcap, int N) { return (size_t)wino4_blocks(bcap, (N + 3) / 4) * W4BLOCK; }
bool wino4_whole_boards(int N) { return w4_whole_boards((N + 3) / 4); }
// multiplies per output point: F(4x4,3x3) against F(3x3,3x3); the larger tile pays from 13x13 up
bool wino4_applies(int N) { return N >= 13; }

static void wino4_check(int bcap, int N) {
  const int T = (N + 3) / 4;
  AGZ_REQUIRE((long)(wino4_blocks(bcap, T) + 1) * W4T < (1L << 31) && (long)bcap * N * N * kC * 4 < (1L << 32), AGZ_BAD_ARGUMENT,
              "batch of %d positions at %dx%d: tile index / activation byte offset exceeds 32 bits", bcap, N, N);
}

void launch_wino4_in(const float* x, float* vimg, const int* d_count, int bcap, int N, hipStream_t s, bool fixup) {
  const int T = (N + 3) / 4;
  const int blocks = (int)wino4_blocks(bcap, T);
  wino4_check(bcap, N);
  if (fixup) {
    AGZ_REQUIRE(!w4_whole_boards(T), AGZ_BAD_ARGUMENT, "fix-up transform: dense tile blocks only");
    hipLaunchKernelGGL((k_wino4_in<true>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
  } else {
    hipLaunchKernelGGL((k_wino4_in<false>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
  }
}

// y == nullptr: the activations are not needed in HBM; vnext == nullptr: no next Winograd layer
void launch_wino4_gemm(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                       float* y, float* vnext, const int* d_count, int bcap, int N, int relu, hipStream_t s) {
  const int T = (N + 3) / 4;
  const int blocks = (int)wino4_blocks(bcap, T);
  const int per_xcd = 4 * ((blocks + 7) / 8);
  const dim3 grid(8 * per_xcd), block(256);
  wino4_check(bcap, N);
#ifdef AGZ_TIMING_EXPERIMENTS
  static const int xp = getenv("AGZ_WINO4_X") ? atoi(getenv("AGZ_WINO4_X")) : 0;
  if (xp && y && vnext) {
    auto kern = xp == 1 ? k_wino4_gemm<3, 1> : xp == 2 ? k_wino4_gemm<3, 2> : xp == 4 ? k_wino4_gemm<3, 4> : xp == 5 ? k_wino4_gemm<3, 5> : k_wino4_gemm<3, 0>;
    hipLaunchKernelGGL(kern, grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
    return;
  }
#endif
  if (y && vnext)
    hipLaunchKernelGGL((k_wino4_gemm<3, 0>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
  else if (vnext)
    hipLaunchKernelGGL((k_wino4_gemm<2, 0>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
  else
    hipLaunchKernelGGL((k_wino4_gemm<1, 0>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu);
}

}  // namespace agz
