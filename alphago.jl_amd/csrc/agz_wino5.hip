// agz_wino5.hip -- the F(3x3,3x3) tower layer of small boards (N <= 12: the 9x9 headline) with a workgroup tile of
// 64 tiles x 128 couts, multiplied in FIVE ONE-ROW PASSES that are folded into the inverse transform (round 6).
//
// Why.  k_wino_gemm4 (agz_wino.hip) holds all 25 planes of a 64 x 64 tile in registers (400 accumulators per lane) and
// moves 52 KB of operands through LDS-DMA per 4 input channels: 16 flop per DMA byte, two ds_read_b64 per two MFMAs.  The
// layer sits on the board's power limit, and what it spends outside the MFMAs is mostly operand movement (DESIGN.md 4,
// HISTORY.md 4f / 12).  Twice the couts per workgroup -- 64 x 128 -- needs 25 % fewer operand bytes per flop (21.3 flop
// per DMA byte, three ds_read_b64 per four MFMAs, V served to two workgroups per tile block instead of four), but 25
// planes x 2 x 16 accumulators do not exist.  The pass structure of agz_wino4.hip makes it fit:
//   Y = A^T M A = sum_i A^T[:, i] (x) (A^T M[i][:])        M[i][j] = plane of transform row i, column j
// one pass over the input channels per transform row i (its five planes: 2 x 5 x 16 = 160 accumulators), and when the
// pass's K loop ends  t = A^T M[i][:]  (three values per element) and  Y[i'][:] += A^T[i'][i] t  into the 2 x 9 x 16 = 288
// running outputs.  Every MFMA of the one-pass form is kept; V is read in the SAME stage images agz_wino.hip's kernels
// write (a pass moves its own five 1 KB chunks of every stage image), so the stem, k_wino_in and the fused input
// transform of the epilogue are untouched; U gets its own image ([cout block 2][pass 5][super-stage 32][unit 10][row 128][4]).
//
// K loop.  Unit = one plane x 4 input channels: 1 KB of V (64 tile rows x 16 B) + 2 KB of U (128 cout rows x 16 B), four
// MFMAs per wave.  Super-stage = 2 channel groups x 5 planes = 10 units = 30 KB, triple-buffered, filled by LDS-DMA (30
// pieces of 1 KB: waves 0, 1 move eight, waves 2, 3 seven), barrier in the read stream as in k_wino_gemm4.  160
// super-stages per layer ([pass][32]) are one flat list: the DMA stream runs across pass boundaries.
// Accumulators are TRANSPOSED (D = U rows x V rows: lane = tile row, register = cout, agz_wino4.hip): wave (wm, wn) holds
// tile rows 32 wm .. and couts 64 hh + 32 wn .. of BOTH halves hh of the 128 couts, so the epilogue runs twice on the
// 64-cout tile image of k_wino_gemm4 (147 KB: 64 tiles x 9 outputs x 64 couts) with all four waves at work in either
// half -- residual by LDS-DMA, affine + ReLU into the image with 16-byte LDS accesses, image -> y, and the next layer's
// input transform for the half's 16 stages (the last two are k_wino_gemm4's phases 1b / 2, operation for operation: bt5p
// is THE arithmetic of B^T d B, so a tile's V does not depend on which kernel emitted it).
#include "agz_nn.h"
#include "agz_glds.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace agz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W5T = 64;                      // tile rows per workgroup = a tile block of agz_wino.hip (whole boards: 63 rows at 9x9)
constexpr int W5C = 128;                     // couts per workgroup
constexpr int W5H = 64;                      // couts per epilogue half
constexpr int W5G = 2;                       // 4-channel groups per super-stage
constexpr int W5UNITS = 5 * W5G;             // 10
constexpr int W5VU = W5T * 4;                // floats of a V unit (a 1 KB chunk of agz_wino.hip's stage image)
constexpr int W5UU = W5C * 4;                // floats of a U unit (2 KB)
constexpr int W5SV = W5UNITS * W5VU;         // 2560
constexpr int W5SU = W5UNITS * W5UU;         // 5120
constexpr int W5STAGE = W5SV + W5SU;         // 7680 floats = 30 KB
constexpr int W5SSP = (kC / 4) / W5G;        // 32 super-stages per pass
constexpr int W5NSS = 5 * W5SSP;             // 160
constexpr int W5UBLOCK = W5NSS * W5SU;       // floats of U per block of 128 couts: 819,200
constexpr int W5VSTAGE = 13 * W5T * 8;       // floats of one stage image of V (agz_wino.hip: A_STAGE = 26 chunks of 1 KB)
constexpr int W5IMG = W5T * 9 * W5H;         // floats of the half tile image: 147,456 B
constexpr int W5PIECES = (W5SV + W5SU) / 256;      // 30 DMA pieces of 1 KB per super-stage
static_assert(kWinoStages == 64 && kC == 256 && W5PIECES == 30, "stage structure of agz_wino.hip");
// (W5_DIST = 3 -- four stage buffers, pieces requested three super-stages ahead -- and W5_LA = 4 -- operands read four units
// ahead through a ring of five register sets -- were built and measured on the bench command, same box, alternating: +0.7 %
// and +2.7 % per step; profiles/r06_ab_wino5_prefetch_distance_2_vs_3.txt, r06_ab_wino5_lookahead_1_vs_4.txt)
#ifndef W5_DIST
#define W5_DIST 2
#endif
constexpr int W5DIST = W5_DIST;              // super-stages a DMA piece is requested ahead of its first read
constexpr int W5RING = W5DIST + 1;           // stage buffers
constexpr int W5RN0 = (W5RING * W5STAGE * 4 + 4095) / 4096;      // first residual instruction n whose image bytes lie beyond the ring
static_assert(W5DIST == 2 || W5DIST == 3, "vmcnt immediates in the kernel");
static_assert(W5RING * W5STAGE <= W5IMG, "the stage ring lies under the tile image");

// (agz_wino.hip: wino_rows_per_block / wino_whole_boards -- rows of a 64-row tile block that carry tiles)
__host__ __device__ inline int w5_rows_per_block(int T) {
  const int tt = T * T;
  const int whole = (W5T / tt) * tt;
  return (tt <= W5T && whole * 10 >= W5T * 9) ? whole : W5T;
}
__host__ __device__ inline bool w5_whole_boards(int T) { return w5_rows_per_block(T) % (T * T) == 0 && T * T <= W5T; }
// a unit image row (either operand): the unit's 4 channels as two pairs, pair h at slot (h + (row >> 4)) & 1
// (agz_wino.hip: wino_v_off -- the layout V is stored in)
__host__ __device__ __forceinline__ int w5_off(int row, int h) { return row * 4 + 2 * ((h + (row >> 4)) & 1); }

#ifdef AGZ_TIMING_EXPERIMENTS
// per workgroup, on the 100 MHz wall clock: [0] hw id | xcc id << 32, [1] start, [2] prologue done (first super-stage published),
// [3 + 2 p] end of pass p's K loop, [4 + 2 p] end of its fold, [13 + 4 hh + {0, 1, 2, 3}] half hh: residual landed, image
// written, y stored, next V stored; tools/trace_wino5.py reads it
__device__ unsigned long long w5_trace[4096][24];
#define W5_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) w5_trace[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define W5_STAMP(k) do { } while (0)
#endif

template <int MODE>      // bit 0: write y (affine, residual, ReLU applied); bit 1: emit the next layer's V stage images; bit 2: add the residual
__global__ __launch_bounds__(256, 1) void k_wino5_gemm(
    const float* __restrict__ vimg, const float* __restrict__ uimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ vnext, const int* __restrict__ d_count, int N, int T, int relu, int tb0, int tb1) {
  __shared__ __attribute__((aligned(256))) float lds[W5IMG + 64];      // three stage buffers, then the half image + 64 zeros
  __shared__ int ptab[W5T * 9];      // element offset of output point X = k * 64 + row (cout 128 cb of it) in y / res, or -1
  constexpr bool RES = (MODE & 4) != 0;
  const int P = N * N, TT = T * T;
  const long Mt = (long)(*d_count) * TT;
  const int RPB = w5_rows_per_block(T);
  // workgroup -> (tile block, cout block): the two cout blocks of a tile block are consecutive workgroups of one XCD (block
  // b runs on XCD b % 8): the V slab comes out of HBM once and the second reader finds it in that XCD's L2
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jb = bid >> 3;
  const int cb = jb & 1;
  const int tb = tb0 + xcd + 8 * (jb >> 1);
  if (tb >= tb1 || (long)tb * RPB >= Mt) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, hi = lane >> 5;

#ifdef AGZ_TIMING_EXPERIMENTS
  if (tid == 0 && blockIdx.x < 4096) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    w5_trace[blockIdx.x][0] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  }
  W5_STAMP(1);
#endif
  const float* vsrc = vimg + (long)tb * kWinoStages * W5VSTAGE;
  const float* usrc = uimg + (long)cb * W5UBLOCK;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0];

  // piece n of this wave = piece p = wave + 4 n of the super-stage: p < 10 the V unit p (channel group p / 5, plane p % 5
  // of the pass's five: chunk 5 pass + p % 5 of stage 2 ss + p / 5), else the 1 KB half (p - 10) & 1 of U unit (p - 10) >> 1
  auto dma = [&](int g, int buf, int n) {
    const int p = wave + 4 * n;
    if (n == 7 && wave >= 2) return;                   // (wave-uniform: waves 2, 3 have seven pieces)
    const int pass = g >> 5, ss = g & 31;
    // n < 2: p < 8, a V unit of channel group 0 / 1; n > 2: p >= 12, U; n == 2: p = 8, 9 (V, group 1) for waves 0, 1, else U
    const bool isv = n < 2 || (n == 2 && wave < 2);
    const float* src;
    if (isv) {
      const int cg = p >= 5 ? 1 : 0, j = p - 5 * cg;
      src = vsrc + (long)(2 * ss + cg) * W5VSTAGE + (5 * pass + j) * W5VU;
    } else {
      src = usrc + (long)g * W5SU + (p - W5UNITS) * 256;
    }
    glds16s(src, (unsigned)lane * 16u, lds0 + (unsigned)(buf * W5STAGE + p * 256) * 4u);
  };
#pragma unroll
  for (int n = 0; n < 8; ++n) dma(0, 0, n);

  for (int idx = tid; idx < W5T * 9; idx += 256) {     // (published by the barrier in front of the first operand reads)
    const int row = idx & (W5T - 1), k = idx >> 6;
    const long tile = (long)tb * RPB + row;
    int off = -1;
    if (row < RPB && tile < Mt) {
      const unsigned tile32 = (unsigned)tile, b = tile32 / (unsigned)TT, t = tile32 - b * (unsigned)TT;
      const unsigned ti = t / (unsigned)T, k3 = (unsigned)k / 3u;
      const int pi = (int)(3 * ti + k3), pj = (int)(3 * (t - ti * T) + ((unsigned)k - 3 * k3));
      if (pi < N && pj < N) off = ((int)b * P + pi + N * pj) * kC + cb * W5C;      // < 2^31 (checked by the launcher)
    }
    ptab[idx] = off;
  }
#pragma unroll
  for (int n = 0; n < 8; ++n) dma(1, 1, n);
  if (W5DIST == 3) {
#pragma unroll
    for (int n = 0; n < 8; ++n) dma(2, 2, n);
  }

  f32x16 acc[5][2];
  const int arow = wm * 32 + l31, brow = wn * 32 + l31;
  const int aoff = w5_off(arow, hi), boff = W5SV + w5_off(brow, hi);      // (row + 64: the same slot, 256 floats on)
#ifndef W5_LA
#define W5_LA 1
#endif
  constexpr int LA = W5_LA, RING = LA + 1;
  static_assert(W5UNITS % RING == 0, "ring slots must be compile-time within a super-stage");
  float2 ra[RING], rb0[RING], rb1[RING];
  auto load = [&](const float* L, int u, int s) {
    ra[s] = *reinterpret_cast<const float2*>(L + aoff + u * W5VU);
    rb0[s] = *reinterpret_cast<const float2*>(L + boff + u * W5UU);
    __builtin_amdgcn_sched_barrier(0);      // (two ds_read_b64, not one ds_read2st64_b64: 16-lane groups, 2-way conflicts on this layout)
    rb1[s] = *reinterpret_cast<const float2*>(L + boff + u * W5UU + 256);
  };
  // transposed: srcA = U (its rows become D's rows = registers: couts), srcB = V (D's columns = lanes: tile rows)
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // (zero: a pass's first MFMA of every accumulator takes C = 0 as an inline constant instead of 160 register writes per pass)
  auto mma = [&](int j, int s, bool zero) {
    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb0[s].x, ra[s].x, zero ? zero16 : acc[j][0], 0, 0, 0);
    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb1[s].x, ra[s].x, zero ? zero16 : acc[j][1], 0, 0, 0);
    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb0[s].y, ra[s].y, acc[j][0], 0, 0, 0);
    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb1[s].y, ra[s].y, acc[j][1], 0, 0, 0);
  };

  // super-stage 0 has landed (this wave's pieces of super-stage 1 may be in flight: eight or seven of them)
  if (W5DIST == 2) {
    if (wave < 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  } else {
    if (wave < 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  }
  __syncthreads();
  W5_STAMP(2);
#pragma unroll
  for (int u = 0; u < LA; ++u) load(lds, u, u);

  // One super-stage: ten units (unit u = channel group u / 5, plane u % 5).  It fetches super-stage g + 2 (the last two
  // wrap around to 0 and 1: valid memory, never read) into the buffer super-stage g - 1 was read from, and hands over to
  // g + 1 through the barrier in its read stream.  Pieces go out one per unit slot from slot D0 on; the barrier sits in
  // front of slot UB = 10 - LA, by when BAR = UB - D0 <= 7 pieces of g + 2 are out for every wave alike: vmcnt(BAR).
  constexpr int UB = W5UNITS - LA, D0 = UB >= 7 ? UB - 7 : 0, BAR = UB - D0;
  int buf = 0;
  auto sstage = [&](int g, auto first_c) {
    constexpr bool first = decltype(first_c)::value;      // the pass's first super-stage
    const int nbuf = buf == W5RING - 1 ? 0 : buf + 1;
    const int dbuf = buf == 0 ? W5RING - 1 : buf - 1;      // the buffer super-stage g - 1 was read from
    const int g2 = g + W5DIST >= W5NSS ? g + W5DIST - W5NSS : g + W5DIST;
    const float* L = lds + buf * W5STAGE;
    const float* Ln = lds + nbuf * W5STAGE;
#pragma unroll
    for (int u = 0; u < W5UNITS; ++u) {
      const int t = u + LA;
      if (t == W5UNITS) {
        // everything this wave owes to super-stage g + 1 has landed; still in flight may be its BAR pieces of the newest
        // super-stage and, with a distance of three, all eight (waves 2, 3: seven) of the one before
        static_assert(BAR == 7, "vmcnt immediates below");
        if (W5DIST == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (wave < 2) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        __syncthreads();
      }
      if (t < W5UNITS) load(L, t, t % RING);
      else load(Ln, t - W5UNITS, t % RING);
      __builtin_amdgcn_sched_barrier(0);      // (one unit's reads at a time: merged ds_read2st64_b64 conflict on this layout, agz_wino4.hip)
      if (u >= D0 && u < D0 + 8) {
        dma(g2, dbuf, u - D0);
        __builtin_amdgcn_sched_barrier(0);
      }
      mma(u % 5, u % RING, first && u < 5);
    }
    buf = nbuf;
  };

  // Running outputs: Y[hh][3 i' + j'][q] = elements 2 q, 2 q + 1 (the C/D map's registers = couts) of output point (i', j'),
  // half hh -- 288 registers beside 160 accumulators.  VALU instructions address the 256 architectural VGPRs only, and the
  // other half of a lone wave's file (the AGPRs) holds the accumulators, so 96 of the running outputs LIVE in AGPRs, by hand:
  // half 1's points 0..5 (Y1a), read / fma / written back once per pass.  Left to hipcc the same 96 become "spill slots" and
  // whatever exceeds them goes to scratch (the first build: 452 bytes, reloads behind vmcnt(0) = behind the DMA in flight).
  f32x2 Y0[9][8], Y1v[3][8];
  float Y1a[6][8][2];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int q = 0; q < 8; ++q) Y0[k][q] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int q = 0; q < 8; ++q) Y1v[k][q] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(Y1a[k][q][0]));
      asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(Y1a[k][q][1]));
    }
  // Residual half-tile -> image by LDS-DMA: instruction i = wave + 4 n (n < 36) fills image bytes 1024 i .. = points 4 i .. 4 i + 3;
  // lane = (point, unit u) fetches channel group u ^ (X & 15).  Half 0's is on its way before the K loops are through: the image
  // lies over the stage buffers (W5RING x 30 KB: 90 KB) and 55 KB beyond them, so instructions n >= W5RN0 (image bytes beyond the ring) go out
  // behind pass 3 and land during pass 4, the others when pass 4's K loop has ended -- in front of its fold, which covers part of
  // their flight (the first build waited 6 us per workgroup for a residual requested after the last fold: trace).
  auto rdma = [&](int n, int hh) {
    const int i = wave + 4 * n;
    const int Xp = 4 * i + (lane >> 4), u = lane & 15;
    const int off = ptab[Xp];
    const unsigned boff = off >= 0 ? 4u * (unsigned)(off + hh * W5H + 4 * (u ^ (Xp & 15))) : 0u;
    glds16s(res, boff, lds0 + (unsigned)(i * 256) * 4u);
  };
  int g = 0;
#pragma unroll 1
  for (int pass = 0; pass < 5; ++pass) {
    sstage(g++, std::true_type{});
#pragma unroll 1
    for (int ss = 1; ss < W5SSP; ++ss) sstage(g++, std::false_type{});
    // the fold reads the accumulators through asm, which the hazard recogniser does not see: MFMA D -> VALU read needs 18
    // wait states after the last MFMA (cdna_hip_programming.md 5.7)
    __builtin_amdgcn_sched_barrier(0);
    W5_STAMP(3 + 2 * pass);
    if (RES && pass == 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the wrapped-around DMA of the last two super-stages included)
      __syncthreads();                                       // every wave has left the K loop: the stage buffers are dead
#pragma unroll 1
      for (int n = 0; n < W5RN0; ++n) rdma(n, 0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // fold: t = A^T M[pass][:], Y[i'][:] += A^T[i'][pass] t       A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]
    const float w0 = pass == 4 ? 0.f : 1.f;
    const float w1 = pass == 1 ? 1.f : pass == 2 ? -1.f : pass == 3 ? 2.f : 0.f;
    const float w2 = pass == 0 ? 0.f : pass == 3 ? 4.f : 1.f;
    const f32x2 wv[3] = {{w0, w0}, {w1, w1}, {w2, w2}};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        f32x2 m[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          // (read where they are used: left to hipcc, all 160 accumulators are copied to VGPRs in one go ahead of the fold)
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[j][0]) : "a"(acc[j][hh][2 * q]));
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[j][1]) : "a"(acc[j][hh][2 * q + 1]));
        }
        const f32x2 s12 = m[1] + m[2];
        f32x2 t[3];
        t[0] = (m[0] + s12) + m[3];
        t[1] = (m[1] - m[2]) + 2.f * m[3];
        t[2] = (s12 + 4.f * m[3]) + m[4];
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            const int k = 3 * ii + jj;
            if (hh == 0) {
              Y0[k][q] = __builtin_elementwise_fma(wv[ii], t[jj], Y0[k][q]);
              asm volatile("" : "+v"(Y0[k][q]));      // pin the fold here (agz_wino4.hip: hipcc sinks it otherwise)
            } else if (k >= 6) {
              Y1v[k - 6][q] = __builtin_elementwise_fma(wv[ii], t[jj], Y1v[k - 6][q]);
              asm volatile("" : "+v"(Y1v[k - 6][q]));
            } else {
              f32x2 yv;
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(yv[0]) : "a"(Y1a[k][q][0]));
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(yv[1]) : "a"(Y1a[k][q][1]));
              yv = __builtin_elementwise_fma(wv[ii], t[jj], yv);
              asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Y1a[k][q][0]) : "v"(yv[0]));
              asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Y1a[k][q][1]) : "v"(yv[1]));
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    if (RES && pass == 3) {
#pragma unroll 1
      for (int n = W5RN0; n < 36; ++n) rdma(n, 0);
    }
    W5_STAMP(4 + 2 * pass);
  }
  // the accumulators are dead: the rest of half 1 joins its first six points in the AGPR half until half 0's epilogue is through
  float Y1b[3][8][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Y1b[k][q][0]) : "v"(Y1v[k][q][0]));
      asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Y1b[k][q][1]) : "v"(Y1v[k][q][1]));
    }

  // Everything the epilogue derives from the thread id or its pointer arguments is derived HERE, from opaque copies (hipcc
  // otherwise hoists it above the K loops and parks running outputs in scratch to make room: agz_wino4.hip)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const float *scale_e = scale, *shift_e = shift, *res_e = res;
  float *y_e = y, *vnext_e = vnext;
  asm volatile("" : "+s"(scale_e), "+s"(shift_e), "+s"(res_e), "+s"(y_e), "+s"(vnext_e));
  const int lane_e = tid_e & 63;
  const int wave_e = __builtin_amdgcn_readfirstlane(tid_e >> 6);
  const int wm_e = wave_e & 1, wn_e = wave_e >> 1;
  const int l31_e = lane_e & 31, hi_e = lane_e >> 5;
  float* img = lds;
  const float relu_lo = relu ? 0.f : -3.0e38f;
  if (!RES) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the wrapped-around DMA of the last two super-stages included)
    __syncthreads();                                     // every wave has left the K loop: the stage buffers are dead
  }
  if (tid_e < 64) img[W5IMG + tid_e] = 0.f;              // what phase 2 reads for a patch point off the board

  // ---- epilogue, once per half hh of the 128 couts (cout block cbe = 2 cb + hh of 64: k_wino_gemm4's).  Half image
  // img[X][16 units of 16 B], X = k * 64 + tile row; unit u of row X holds channel group u ^ (X & 15) (agz_wino.hip).
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int cbe = 2 * cb + hh;
    if (RES && hh == 1) {
      // (half 0's residual was requested around the last fold: rdma above)
#pragma unroll 4
      for (int n = 0; n < 36; ++n) {
        const int i = wave_e + 4 * n;
        const int Xp = 4 * i + (lane_e >> 4), u = lane_e & 15;
        const int off = ptab[Xp];
        const unsigned boff = off >= 0 ? 4u * (unsigned)(off + hh * W5H + 4 * (u ^ (Xp & 15))) : 0u;
        glds16s(res_e, boff, lds0 + (unsigned)(i * 256) * 4u);
      }
    }
    // phase 1: img = ReLU(img (the residual) + scale * value + shift).  The lane's tile row, and per output point its four
    // register quads = couts 32 wn + 8 qd + 4 hi .. + 3 of this half = unit 8 wn + 2 qd + hi
    {
      const int trow = wm_e * 32 + l31_e;
      float sc[16], sh[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = cbe * W5H + wn_e * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi_e;
        sc[e] = scale_e[co];
        sh[e] = shift_e[co];
      }
      unsigned a[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        a[qd] = lds0 + 4u * (unsigned)(trow * W5H) + 16u * (unsigned)((8 * wn_e + 2 * qd + hi_e) ^ (trow & 15));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the residual half-tile (and the affine) have landed
      if (RES) __syncthreads();
      W5_STAMP(13 + 4 * hh);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const unsigned kb = 4u * (unsigned)(k * W5T * W5H);
        f32x4 rr[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          if (RES) rr[qd] = *(const __attribute__((address_space(3))) f32x4*)(size_t)(a[qd] + kb);
          else rr[qd] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f32x4 v;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int e = 4 * qd + c;
            v[c] = fmaxf(__builtin_fmaf(Y0[k][e >> 1][e & 1], sc[e], sh[e]) + rr[qd][c], relu_lo);
          }
          *(__attribute__((address_space(3))) f32x4*)(size_t)(a[qd] + kb) = v;
        }
      }
    }
    __syncthreads();
    W5_STAMP(14 + 4 * hh);

    if (MODE & 1) {
      // phase 1b (k_wino_gemm4's): element = (row, k, 4 channels); 16 consecutive lanes cover the 256 contiguous bytes of one point
      constexpr int PER = W5T * 9 * (W5H / 4) / 256;      // 36 per thread
      const int cg4 = hh * W5H + 4 * ((tid_e ^ (tid_e >> 4)) & 15);
      const f32x4* ip0 = reinterpret_cast<const f32x4*>(img) + tid_e;
      const int* pt0 = ptab + (tid_e >> 4);
#pragma unroll 1
      for (int i0 = 0; i0 < PER; i0 += 12) {
        f32x4 v[12];
        int offs[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          v[j] = ip0[256 * (i0 + j)];
          offs[j] = pt0[16 * (i0 + j)];
        }
#pragma unroll
        for (int j = 0; j < 12; ++j)
          if (offs[j] >= 0) *reinterpret_cast<f32x4*>(y_e + offs[j] + cg4) = v[j];
      }
    }

    W5_STAMP(15 + 4 * hh);
    if (MODE & 2) {
      // phase 2 (k_wino_gemm4's): the next layer's input transform for this half's 64 channels = stages 16 cbe .. 16 cbe + 15 of
      // the next layer's K loop.  Task = (tile row, stage): lane = row, wave w takes stages w, w + 4, w + 8, w + 12.
      const int row = lane_e;
      const long tile = (long)tb * RPB + row;
      const bool live = row < RPB && tile < Mt;
      const int t = live ? (int)(tile % TT) : 0;
      const int ti = t / T, tj = t % T;
      unsigned adr[25];
#pragma unroll
      for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int v = 0; v < 5; ++v) {
          const int du = u == 0 ? -1 : (u == 4 ? 1 : 0), ku = u == 0 ? 2 : (u == 4 ? 0 : u - 1);
          const int dv = v == 0 ? -1 : (v == 4 ? 1 : 0), kv = v == 0 ? 2 : (v == 4 ? 0 : v - 1);
          const int pi = 3 * ti - 1 + u, pj = 3 * tj - 1 + v;
          const bool ok = live && pi >= 0 && pi < N && pj >= 0 && pj < N;
          const int Xq = (ku * 3 + kv) * W5T + row + du * T + dv;
          const int pb = ok ? Xq * W5H : W5IMG, xm = ok ? (Xq & 15) : 0;
          adr[u * 5 + v] = lds0 + 4u * (unsigned)(pb + 4 * (wave_e ^ xm) + 2 * ((row >> 4) & 1));
        }
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        const int sl = wave_e + 4 * it;
        const unsigned xo = (unsigned)it << 6;
        f32x4 d[25];
#pragma unroll
        for (int q = 0; q < 25; ++q) {
          const unsigned a0 = adr[q] ^ xo;
          const f32x2 a = *(const __attribute__((address_space(3))) f32x2*)(size_t)a0;
          const f32x2 b = *(const __attribute__((address_space(3))) f32x2*)(size_t)(a0 ^ 8u);
          d[q] = (f32x4){a[0], a[1], b[0], b[1]};
        }
        float* gdst = vnext_e + ((long)tb * kWinoStages + (cbe * 16 + sl)) * W5VSTAGE + row * 4;
        // bt5 of agz_wino.hip on channel pairs, operation for operation (every multiply-add an explicit fma)
        auto bt5p = [](f32x2 x0, f32x2 x1, f32x2 x2, f32x2 x3, f32x2 x4, f32x2* r) {
          const f32x2 two = {2.f, 2.f}, mtwo = {-2.f, -2.f}, three = {3.f, 3.f};
          r[3] = x3 - x1;
          r[0] = __builtin_elementwise_fma(two, x0 - x2, r[3]);
          r[4] = __builtin_elementwise_fma(mtwo, r[3], x4 - x2);
          r[1] = __builtin_elementwise_fma(two, x1, x2 - x3);
          r[2] = __builtin_elementwise_fma(mtwo, x1, __builtin_elementwise_fma(three, x2, -x3));
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
        for (int xi = 0; xi < 25; ++xi) {
          const f32x2 p0 = vv[xi][0], p1 = vv[xi][1];
          const f32x4 v4 = {p0[0], p0[1], p1[0], p1[1]};        // (already in the row's pair order: see the reads above)
          __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(gdst + xi * W5VU));
        }
      }
    }

    W5_STAMP(16 + 4 * hh);
    if (hh == 0) {
      // the second half runs this same code on its own outputs: out of the AGPR half, into half 0's registers
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (k < 6) {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(Y0[k][q][0]) : "a"(Y1a[k][q][0]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(Y0[k][q][1]) : "a"(Y1a[k][q][1]));
          } else {
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(Y0[k][q][0]) : "a"(Y1b[k - 6][q][0]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(Y0[k][q][1]) : "a"(Y1b[k - 6][q][1]));
          }
        }
      __syncthreads();      // every wave has left the image: the second half may overwrite it
    }
  }
}

// ------------------------------------------------------------------ host side

// Flux [kw,kh,cin,cout] column-major -> U images [cout block 2][pass 5][super-stage 32][unit 10][row 128][4], U = G k G^T in
// float64 (agz_wino.hip's G; k is the CORRELATION kernel: NNlib's conv is a true convolution).  One (cout, cin) pair per
// call, the same source on the host (test reference, agz_debug_pack_diff) and in the device kernel (the product).
__host__ __device__ inline void wino5_pack_pair(const float* w, int o, int ci, float* out) {
#pragma clang fp contract(off)
  constexpr double G[5][3] = {{0.5, 0.0, 0.0}, {0.5, 0.5, 0.5}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                              {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
  double k[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) k[a][b] = w[(2 - a) + 3 * ((2 - b) + 3 * (ci + (size_t)kC * o))];
  const int cb = o / W5C, r = o % W5C, c4 = ci / 4, cl = ci % 4;
  const int ss = c4 / W5G, cg = c4 % W5G;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      double u = 0.0;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) u += G[i][a] * k[a][b] * G[j][b];
      out[(size_t)cb * W5UBLOCK + (size_t)(i * W5SSP + ss) * W5SU + (size_t)(cg * 5 + j) * W5UU + w5_off(r, cl >> 1) + (cl & 1)] = (float)u;
    }
}
void wino5_pack_weights(const ConvHost& c, float* out) {
  AGZ_REQUIRE(c.cin == kC && c.cout == kC, AGZ_BAD_ARGUMENT, "five-pass F(3x3,3x3) pack: tower layers only (%d -> %d)", c.cin, c.cout);
  for (int o = 0; o < kC; ++o)
    for (int ci = 0; ci < kC; ++ci) wino5_pack_pair(c.w.data(), o, ci, out);
}
__global__ __launch_bounds__(256) void k_wino5_pack(const float* __restrict__ w, long wstride, int layers, float* __restrict__ out,
                                                    long per) {
  const long n = (long)layers * kC * kC;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
    const int ci = (int)(t % kC), o = (int)((t / kC) % kC), l = (int)(t / ((long)kC * kC));
    wino5_pack_pair(w + l * wstride, o, ci, out + l * per);
  }
}
void launch_wino5_pack(const float* d_w, long wstride, int layers, float* d_out, hipStream_t s) {
  const long per = (long)wino5_weight_floats();
  const int grid = (int)std::min<long>(((long)layers * kC * kC + 255) / 256, 65536);
  hipLaunchKernelGGL(k_wino5_pack, dim3(grid), dim3(256), 0, s, d_w, wstride, layers, d_out, per);     // (every word of an image is written)
}
size_t wino5_weight_floats() { return (size_t)(kC / W5C) * W5UBLOCK; }
bool wino5_applies(int N) { return w5_whole_boards((N + 2) / 3); }

// the arguments of launch_wino_gemm (agz_wino.hip) for a tower layer in exact f32; uimg: launch_wino5_pack's image
void launch_wino5_gemm(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                       float* y, float* vnext, const int* d_count, int bcap, int N, int relu, hipStream_t s, int part, int parts) {
  const int T = (N + 2) / 3;
  AGZ_REQUIRE(w5_whole_boards(T), AGZ_BAD_ARGUMENT, "five-pass F(3x3,3x3): whole-board tile blocks only (N <= 12), got %d", N);
  const long rpb = w5_rows_per_block(T);
  const int all_blocks = (int)(((long)bcap * T * T + rpb - 1) / rpb);
  AGZ_REQUIRE(parts >= 1 && part >= 0 && part < parts, AGZ_BAD_ARGUMENT, "tile-block range %d of %d", part, parts);
  AGZ_REQUIRE(y || vnext, AGZ_BAD_ARGUMENT, "five-pass F(3x3,3x3) GEMM: nothing to write");
  const int per_part = (all_blocks + parts - 1) / parts;
  const int tb0 = std::min(all_blocks, part * per_part), tb1 = part + 1 == parts ? all_blocks : std::min(all_blocks, tb0 + per_part);
  if (tb1 <= tb0) return;
  const int blocks = tb1 - tb0;
  const dim3 grid(8 * 2 * ((blocks + 7) / 8)), block(256);
  AGZ_REQUIRE((long)(all_blocks + 1) * W5T < (1L << 31) && (long)bcap * N * N * kC * 4 < (1L << 32), AGZ_BAD_ARGUMENT,
              "batch of %d positions at %dx%d: tile index / activation byte offset exceeds 32 bits", bcap, N, N);
#define W5_LAUNCH(MODE_) hipLaunchKernelGGL((k_wino5_gemm<MODE_>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1)
#ifdef AGZ_TIMING_EXPERIMENTS
  static int traced = 0;
  if (getenv("AGZ_WINO5_TRACE") && y && vnext && res && ++traced == 3) {      // third steady-state conv2-form launch
    W5_LAUNCH(7);
    (void)hipStreamSynchronize(s);
    static unsigned long long host[4096][24];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(w5_trace), sizeof(host));
    if (FILE* f = fopen(getenv("AGZ_WINO5_TRACE"), "wb")) {
      fwrite(host, 1, sizeof(host), f);
      fclose(f);
    }
    return;
  }
#endif
  const int mode = (y ? 1 : 0) | (vnext ? 2 : 0) | (res ? 4 : 0);
  switch (mode) {
    case 1: W5_LAUNCH(1); break;
    case 2: W5_LAUNCH(2); break;
    case 3: W5_LAUNCH(3); break;
    case 5: W5_LAUNCH(5); break;
    case 6: W5_LAUNCH(6); break;
    default: W5_LAUNCH(7); break;
  }
#undef W5_LAUNCH
}

}  // namespace agz
