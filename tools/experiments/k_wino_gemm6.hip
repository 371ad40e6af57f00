// agz_wino6.hip -- the exact-f32 Winograd tower layer with TWO workgroups per compute unit (round 3).
//
// k_wino_gemm4 (agz_wino.hip) holds a 64 tile x 64 cout x 25 plane tile in the 400 accumulator registers of one wave per
// SIMD.  Its K loop runs within 10 % of the MFMA floor, but every non-MFMA phase of a workgroup -- the inverse transform,
// the copy of y, the next layer's input transform: 0.40-0.45 ms of a 2.35 ms layer -- is executed by that lone wave at
// VALU / LDS latency with the matrix pipe idle, because nothing else fits on the CU (19 % of the
// launch).  Two resident workgroups per CU would run one's epilogue under the other's K loop, but need <= 256 registers
// per wave.  Round 3 measured the obvious way there and why it fails (HISTORY.md 4d, "k_wino_gemm5"): halving the tile to
// 64 x 32 on v_mfma_f32_16x16x4_f32 (200 accumulators) makes a workgroup's own K loop too weak to use the pipe alone
// (32-cycle MFMAs do not cover an LDS-DMA issue, 1.5x the DMA bytes per flop, double buffering): 2.62 ms per layer.
//
// This kernel keeps the 64 x 64 tile, the 32x32x2 MFMA, the operand traffic and the DMA count per MFMA of k_wino_gemm4,
// and halves the registers by TIME-SLICING THE TRANSFORM COLUMNS instead: Y = A^T M A = sum_j (A^T M)[.][j] A^T[.][j],
// so the 25 planes are computed five at a time -- pass j runs the whole K loop (all 256 input channels) for the planes
// (i, j), i = 0..4 (80 accumulators), reduces them to the three rows of A^T M (in place) and adds their column-j
// contribution to the nine running outputs (144 registers): 224 live registers instead of 400.  The planes of a pass
// are disjoint data, so no operand byte is fetched twice: same bytes, same MFMAs, five K loops of 5 planes instead of
// one of 25.  The sums are formed in the order of k_wino_gemm4 (x2, x4 are exact), so the two kernels agree bit for bit.
//
//   workgroup  = 64 tile rows x 64 couts, four waves = 2 x 2 quadrants of 32 x 32 (as k_wino_gemm4)
//   unit       = 8 input channels x 5 planes x (64 V rows + 64 U rows) = 20 pieces of 1 KB; three unit buffers (60 KB),
//                filled by LDS-DMA two units ahead; 160 units per tile (5 passes x 32), one barrier per unit
//   CU         = two workgroups (2 x 76 KB of LDS, 2 waves per SIMD)
//   epilogue   = the tile image of k_wino_gemm4 does not fit beside a second workgroup (147 KB), so the 64 couts leave
//                in two chunks of 32 through a 72 KB image: the two waves that own the chunk add the residual, write
//                y and emit the next layer's V while the other two wait -- slower than four waves at once, and free:
//                the other workgroup's K loop has the matrix pipe meanwhile.
// V (activations) keeps the stage-image layout of agz_wino.hip (k_wino_in and both kernels' fused transforms are
// interchangeable producers); U has its own layout: [cout block][pass j][stage][plane i][cout 64][4 channels], a plane
// image in the V format (pairs swapped by bit 4 of the row), so that A and B operands are read the same way.
#include "agz_wino_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace agz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int G6C = 64;                          // couts per workgroup
constexpr int G6_G = 4;                          // planes per pass (64 accumulator registers); plane 24 has a pass of its own
constexpr int G6_UNIT = 16 * 256;                // floats per LDS unit: 8 (channel group, plane) steps x (V piece | U piece)
constexpr int G6_NBUF = 4;                       // unit buffers: the DMA runs up to four units ahead
constexpr int G6_CH = 32;                        // couts per epilogue chunk
constexpr int G6_IMG = WT * 9 * G6_CH;           // epilogue tile image: 18,432 floats = 73,728 B
constexpr int G6_LDS = G6_IMG > G6_NBUF * G6_UNIT ? G6_IMG : G6_NBUF * G6_UNIT;

// epilogue tile image img[X][32 couts], X = output k * 64 + tile row; the 16-byte unit of channel group g (4 couts)
// sits at position g ^ ((X >> 1) & 7) of the 128-byte row: two consecutive X share a 256-byte bank window and eight
// consecutive pairs rotate through the eight unit positions -- a wave whose lanes are consecutive X reads one channel
// group with a conflict-free ds_read_b128; the accumulator layout (lane = cout, rows 4 apart in the two lane halves)
// writes with 2-way conflicts, which a ds_write_b32 hides.
__device__ __forceinline__ int g6_img_off(int X, int c) { return X * G6_CH + 4 * ((c >> 2) ^ ((X >> 1) & 7)) + (c & 3); }

// fold weights: plane xi = (i, j) adds g6_fold[xi][3 i' + j'] = A^T[i'][i] A^T[j'][j] times its accumulator to output (i', j')
// (A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]; 121 of the 225 entries are non-zero, all of them exact in f32)
struct G6Fold { float w[WXI][9]; };
static constexpr G6Fold g6_make_fold() {
  G6Fold f{};
  const int AT[3][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 0}, {0, 1, 1, 4, 1}};
  for (int xi = 0; xi < WXI; ++xi)
    for (int i2 = 0; i2 < 3; ++i2)
      for (int j2 = 0; j2 < 3; ++j2) f.w[xi][i2 * 3 + j2] = (float)(AT[i2][xi / 5] * AT[j2][xi % 5]);
  return f;
}
__constant__ G6Fold g6_fold = g6_make_fold();

#ifdef AGZ_TIMING_EXPERIMENTS
// per workgroup: {hw id | xcc id << 32, start, K loops done, end} on the 100 MHz wall clock (who shares a CU, and when)
__device__ unsigned long long g6_trace[16384][16];
__device__ __forceinline__ unsigned long long g6_now() { return __builtin_readcyclecounter() * 0 + wall_clock64(); }
#endif

// Phases 1b and 2 of a chunk, executed by the chunk's two owner waves (wm = 0, 1): img -> y (MODE & 1) and the next
// layer's input transform V = B^T d B from img -> HBM stage images (MODE & 2).  See k_wino_gemm6.
template <int MODE>
__device__ __attribute__((noinline)) void g6_emit(const float* img, const int* ptab, float* __restrict__ y, float* __restrict__ vnext,
                                                  int tb, int cb, int c, int wm, int lane, int RPB, long Mt, int TT, int T, int N) {
        const int t2 = wm * 64 + lane;                  // the chunk's two owner waves as 128 threads
        if (MODE & 1) {
          // phase 1b: element = (point X, 16-byte unit); eight consecutive lanes cover the 128 contiguous bytes of one
          // point.  Element i of thread t2: X = (t2 >> 3) + 16 i, unit position t2 & 7 -> channel group
          // (t2 ^ (t2 >> 4)) & 7 for every i ((X >> 1) & 7 = (t2 >> 4) & 7): one LDS address and one channel offset per thread
          constexpr int PER = WT * 9 * (G6_CH / 4) / 128;     // 36 per thread
          const int cg4 = c * G6_CH + 4 * ((t2 ^ (t2 >> 4)) & 7);
          const f32x4* ip0 = reinterpret_cast<const f32x4*>(img) + t2;
          const int* pt0 = ptab + (t2 >> 3);
  #pragma unroll
          for (int i0 = 0; i0 < PER; i0 += 12) {
            f32x4 v[12];
            int offs[12];
  #pragma unroll
            for (int q = 0; q < 12; ++q) {
              v[q] = ip0[128 * (i0 + q)];
              offs[q] = pt0[16 * (i0 + q)];
            }
  #pragma unroll
            for (int q = 0; q < 12; ++q)
              if (offs[q] >= 0) *reinterpret_cast<f32x4*>(y + offs[q] + cg4) = v[q];
          }
        }
        if (MODE & 2) {
          // ---- phase 2: the next layer's input transform for this chunk's 32 channels (= stages 16 cb + 8 c .. + 7 of
          // the next layer's K loop).  Task = (tile row, stage): lane = row, so that the 64 lanes of a wave fill 64
          // consecutive 16-byte rows of a stage image plane (1 KB per store instruction); owner wave wm takes stages wm, wm + 2, ...
          const int row = lane;
          const long tile = (long)tb * RPB + row;
          const bool live = row < RPB && tile < Mt;
          const int t = live ? (int)(tile % TT) : 0, lb = row / TT;
          const int ti = t / T, tj = t % T;
          const bool swap = (row >> 4) & 1;
  #pragma unroll 1
          for (int sl = wm; sl < G6_CH / WK; sl += 2) {
            f32x4 d[25];
  #pragma unroll
            for (int u5 = 0; u5 < 5; ++u5)
  #pragma unroll
              for (int v5 = 0; v5 < 5; ++v5) {
                const int pi = 3 * ti - 1 + u5, pj = 3 * tj - 1 + v5;
                const bool ok = live && pi >= 0 && pi < N && pj >= 0 && pj < N;
                // the point lives in tile (pi / 3, pj / 3) of the same board, output k = (pi % 3) * 3 + pj % 3
                const int Xq = ((pi + 3) % 3 * 3 + (pj + 3) % 3) * WT + lb * TT + ((pi + 3) / 3 - 1) * T + (pj + 3) / 3 - 1;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                d[u5 * 5 + v5] = ok ? *reinterpret_cast<const f32x4*>(img + Xq * G6_CH + 4 * (sl ^ ((Xq >> 1) & 7))) : z;
              }
            float* gq = vnext + ((long)tb * WNS + (cb * (G6C / WK) + c * (G6_CH / WK) + sl)) * A_STAGE + row * 4;
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
              for (int v5 = 0; v5 < 5; ++v5) {
                f32x2 r[5], cc[5];
  #pragma unroll
                for (int u5 = 0; u5 < 5; ++u5) cc[u5] = (f32x2){d[u5 * 5 + v5][2 * h], d[u5 * 5 + v5][2 * h + 1]};
                bt5p(cc[0], cc[1], cc[2], cc[3], cc[4], r);
  #pragma unroll
                for (int i = 0; i < 5; ++i) tx[i * 5 + v5] = r[i];
              }
  #pragma unroll
              for (int i = 0; i < 5; ++i) {
                f32x2 r[5];
                bt5p(tx[i * 5 + 0], tx[i * 5 + 1], tx[i * 5 + 2], tx[i * 5 + 3], tx[i * 5 + 4], r);
  #pragma unroll
                for (int jj = 0; jj < 5; ++jj) vv[i * 5 + jj][h] = r[jj];
              }
            }
  #pragma unroll
            for (int xi = 0; xi < 26; ++xi) {
              const f32x2 z2 = {0.f, 0.f};
              const f32x2 p0 = xi < 25 ? vv[xi < 25 ? xi : 0][0] : z2, p1 = xi < 25 ? vv[xi < 25 ? xi : 0][1] : z2;
              const f32x2 lo = swap ? p1 : p0, hi2 = swap ? p0 : p1;
              const f32x4 v4 = {lo[0], lo[1], hi2[0], hi2[1]};
              __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(gq + xi * 256));
            }
          }
        }
}

// MODE bit 0: write y (affine, residual, ReLU applied); bit 1: emit the next layer's V stage images; bit 2: a residual is added.
// NS: stages of the K loop (input channels / 4): 64, or 8 for the stem
// X: timing experiments, instantiated only under -DAGZ_TIMING_EXPERIMENTS (results are WRONG for X != 0):
//   1 = K loops only; 4 = no DMA after the prologue; 5 = no MFMA
template <int MODE, int NS, int X = 0>
__global__ __launch_bounds__(256, 2) void k_wino_gemm6(
    const float* __restrict__ vimg, const float* __restrict__ uimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ vnext, const int* __restrict__ d_count, int N, int T, int relu) {
  constexpr bool HASRES = (MODE & 4) != 0;       // a residual is added (res != NULL)
  constexpr int UN = NS / 2;                     // units of a four-plane pass (8 channels each)
  constexpr int FULL = 6 * UN;                   // units of the six four-plane passes; then NS / 8 units of plane 24 (32 channels each)
  constexpr int TOTAL = FULL + NS / 8;           // units per tile: 200 (25 for the stem)
  __shared__ __attribute__((aligned(16))) float lds[G6_LDS];
  __shared__ int ptab[WT * 9];     // element offset of output point X in y / res (cout 0 of the block), or -1
  const int P = N * N, TT = T * T;
  const int RPB = wino_rows_per_block(T);
  const long Mt = (long)(*d_count) * TT;
  // workgroup -> (tile block, cout block): as k_wino_gemm4 (U of two cout blocks stays L2-resident per XCD, the two
  // cout blocks of a tile block are neighbours on one XCD and share its V slab)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jb = bid >> 3;
  const int cb = 2 * (xcd & 1) + (jb & 1);
  const int tb = (xcd >> 1) + 4 * (jb >> 1);
  if ((long)tb * RPB >= Mt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, hi = lane >> 5;
#ifdef AGZ_TIMING_EXPERIMENTS
  if (tid == 0 && bid < 16384) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g6_trace[bid][0] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    g6_trace[bid][1] = g6_now();
  }
#endif
  const float* asrc = vimg + (long)tb * NS * A_STAGE;
  const float* bsrc = uimg + (long)cb * WXI * NS * 256;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0];

  // A unit is 8 steps of one plane x 4 channels, V piece of step s at s KB, U piece at (8 + s) KB.  Units 0 .. FULL - 1:
  // pass v / UN (planes 4 pass .. 4 pass + 3), channels 8 g .. 8 g + 7 (g = v % UN), step = (channel group, plane);
  // units FULL ..: plane 24 alone, channels 32 g .. 32 g + 31, step = channel group.  Wave w moves pieces w, w + 4, w + 8,
  // w + 12 of a unit (two of V, two of U) -- four LDS-DMA instructions per wave and 16 MFMAs.
  auto dma = [&](int v, int q) {
    const int p = wave + 4 * q, sp = p & 7;
    int st, xi;
    if (v < FULL) {
      st = 2 * (v % UN) + (sp >> 2);
      xi = G6_G * (v / UN) + (sp & 3);
    } else {
      st = 8 * (v - FULL) + sp;
      xi = WXI - 1;
    }
    const float* src = p < 8 ? asrc + (long)st * A_STAGE + xi * 256 : bsrc + ((long)xi * NS + st) * 256;
    glds16s(src, (unsigned)lane * 16u, lds0 + (unsigned)((v % G6_NBUF) * G6_UNIT + p * 256) * 4u);
  };
#pragma unroll
  for (int v = 0; v < G6_NBUF; ++v)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (v < TOTAL) dma(v, q);

  for (int idx = tid; idx < WT * 9; idx += 256) {     // (published by the barrier in front of the first operand reads)
    const int row = idx & (WT - 1), k = idx >> 6;       // X = k * 64 + row
    const long tile = (long)tb * RPB + row;
    int off = -1;
    if (row < RPB && tile < Mt) {
      const int b = (int)(tile / TT), t = (int)(tile % TT);
      const int pi = 3 * (t / T) + k / 3, pj = 3 * (t % T) + k % 3;
      if (pi < N && pj < N) off = (b * P + pi + N * pj) * kC + cb * G6C;      // < 2^31: 8192 x 361 x 256 = 7.6e8
    }
    ptab[idx] = off;
  }

  // the nine running outputs as 144 scalars, not nine 16-register tuples: they never feed an MFMA, and 32-bit live
  // ranges leave the register allocator the room that nine more 512-bit tuples beside the accumulators do not
  float o[9][16];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;

  // operand offsets inside a unit (floats): within a piece the lane's channel pair h = hi of row r sits at
  // r * 4 + 2 * ((hi + (r >> 4)) & 1) (wino_v_off)
  const int arow = wm * 32 + l31, brow = wn * 32 + l31;
  const int aoff = arow * 4 + 2 * ((hi + (arow >> 4)) & 1);
  const int boff = 8 * 256 + brow * 4 + 2 * ((hi + (brow >> 4)) & 1);
  // Operands are read LA = 2 steps ahead of their MFMAs through a ring of four register pairs (8 steps per unit: the ring
  // slots are compile-time across units).  The unit barrier sits in the READ stream, two steps before a unit's last MFMA
  // (k_wino_gemm4's scheme): the MFMAs of steps 6 and 7 cover the barrier and the first LDS latencies of the next unit,
  // and once a wave is past the barrier of unit w nobody reads w's buffer any more -- it takes the DMA of unit w + 4.
  constexpr int LA = 2, RING = 4;
  float2 ra[RING], rb[RING];
  static_assert(TOTAL >= G6_NBUF, "prologue");
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // unit 0 of this wave; units 1..3 (12 pieces) may be in flight
  __syncthreads();
#pragma unroll
  for (int s0 = 0; s0 < LA; ++s0) {
    ra[s0] = *reinterpret_cast<const float2*>(lds + aoff + s0 * 256);
    rb[s0] = *reinterpret_cast<const float2*>(lds + boff + s0 * 256);
  }

  int u = 0;
  // One pass: NPL planes P0 .. P0 + NPL - 1 through all the input channels (units of 8 / NPL channel groups), then their
  // share of A^T M A into the running outputs: plane (i, j) adds A^T[i'][i] A^T[j'][j] M_ij to output (i', j')
  // (A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]: 121 multiply-adds per element over the 25 planes, all weights exact).
  // (ONE copy of the code for the six four-plane passes -- the plane index is run-time, the fold weights come from
  // g6_fold through SGPRs -- and one for plane 24: unrolled per pass the kernel was 95 KB of code, and a CU pair's 64 KB
  // instruction cache turned every straight-line section, executed once per tile, into ~60 cycles per instruction.)
  auto pass = [&](int P0, auto npl_c) {
    constexpr int NPL = decltype(npl_c)::value;
    constexpr int UNP = NS * NPL / 8;               // units of this pass
    f32x16 acc[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll 1
    for (int g = 0; g < UNP; ++g, ++u) {
      const float* L = lds + (u % G6_NBUF) * G6_UNIT;
      const float* Ln = lds + ((u + 1) % G6_NBUF) * G6_UNIT;
      const bool next = u + 1 < TOTAL;
      const bool more = u + G6_NBUF < TOTAL && X != 4;
#pragma unroll
      for (int s = 0; s < 8; ++s) {              // step s = (channel group s / NPL, plane s % NPL)
        if (s == 8 - LA && next) {
          // unit u + 1 has landed for this wave (units u + 2, u + 3 may be in flight), then for everybody
          if (X == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (u + 3 < TOTAL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else if (u + 2 < TOTAL) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
        if (s + LA < 8) {
          ra[(s + LA) % RING] = *reinterpret_cast<const float2*>(L + aoff + (s + LA) * 256);
          rb[(s + LA) % RING] = *reinterpret_cast<const float2*>(L + boff + (s + LA) * 256);
        } else if (next) {
          ra[(s + LA) % RING] = *reinterpret_cast<const float2*>(Ln + aoff + (s + LA - 8) * 256);
          rb[(s + LA) % RING] = *reinterpret_cast<const float2*>(Ln + boff + (s + LA - 8) * 256);
        }
        // the four pieces of unit u + 4 go into this unit's own buffer, behind its barrier: two here, two at the head of u + 1
        if (s >= 8 - LA || s < 2) {
          __builtin_amdgcn_sched_barrier(0);
          if (s >= 8 - LA) { if (more) dma(u + G6_NBUF, s - (8 - LA)); }
          else if (u > 0 && u - 1 + G6_NBUF < TOTAL && X != 4) dma(u - 1 + G6_NBUF, 2 + s);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (X != 5) {
          acc[s % NPL] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % RING].x, rb[s % RING].x, acc[s % NPL], 0, 0, 0);
          acc[s % NPL] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % RING].y, rb[s % RING].y, acc[s % NPL], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(ra[s % RING]), "v"(rb[s % RING]));
        }
      }
    }
    // the fold: every plane's accumulator times its nine weights into the nine running outputs
    float w[NPL][9];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int q = 0; q < 9; ++q) w[pl][q] = g6_fold.w[P0 + pl][q];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const float m = acc[pl][e];
        if (NPL == 1) {
          o[8][e] += m;                           // plane 24 = (4, 4): weight 1 on output (2, 2) only
        } else {
#pragma unroll
          for (int q = 0; q < 9; ++q) o[q][e] = __builtin_fmaf(w[pl][q], m, o[q][e]);
        }
      }
      // (an element pair at a time: left to interleave all sixteen, the scheduler keeps their temporaries alive beside the
      // 208 registers that must survive, and the allocator answers by parking outputs in scratch for the whole tile)
      if (e & 1) __builtin_amdgcn_sched_barrier(0);
    }
    // The fold happens HERE: an empty asm that takes the outputs pins their computation in front of the next pass's
    // (volatile) DMA and wait statements.  Without it the compiler sinks the fold arithmetic towards the outputs' first
    // use in the epilogue and keeps every pass's ACCUMULATORS alive in scratch instead (1.2 KB per lane).
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(o[q][e]));
  };
#pragma unroll 1
  for (int p4 = 0; p4 < 6; ++p4) pass(G6_G * p4, std::integral_constant<int, G6_G>{});
  pass(WXI - 1, std::integral_constant<int, 1>{});
#ifdef AGZ_TIMING_EXPERIMENTS
  if (tid == 0 && bid < 16384) g6_trace[bid][2] = g6_now();
#endif
  if (X == 1) {
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) keep += o[k][e];
    if (keep == 123.456f) y[0] = keep;
    return;
  }

  // ---- epilogue, one 32-cout chunk at a time (chunk c = the couts of the waves with wn == c).  Per chunk:
  //   0   residual -> img by LDS-DMA (72 KB in 72 instructions, all four waves issue)
  //   1   the owners: img = ReLU(img (the residual) + scale * o + shift)
  //   1b  the owners: img -> y, 128-byte runs per output point (MODE & 1)
  //   2   the owners: next layer's input transform V = B^T d B from img -> HBM stage images (MODE & 2)
  // From here on this workgroup's waves are VALU / LDS / store work beside the OTHER workgroup's K loop on the same SIMDs.
  // The SIMD arbitrates issue by priority, then age; left at priority 0 the epilogue gets the slots the MFMA stream
  // leaves over and takes twice as long (118 us against 59 us alone, per workgroup: wall-clock trace, HISTORY.md 4e), while
  // the K loop -- one 64-cycle MFMA per ~16 issue slots -- loses next to nothing by waiting a slot.
  if (X != 33) __builtin_amdgcn_s_setprio(3);
  float* img = lds;
  const int col = l31;                              // cout inside the chunk
  const float sc = scale[cb * G6C + wn * G6_CH + col], sh = shift[cb * G6C + wn * G6_CH + col];
  const float relu_lo = relu ? 0.f : -3.0e38f;      // ReLU as one v_max either way
  // One chunk at a time; the same code for both (the instruction cache decides: see the pass loop).  The owners' phases 1b
  // and 2 are a function call: they want ~250 registers, and inlined here they push running outputs into scratch.
#pragma unroll 1
  for (int c = 0; c < 2; ++c) {
    const bool owner = wn == c;
#ifdef AGZ_TIMING_EXPERIMENTS
    auto stamp = [&](int k) { if (owner && wm == 0 && lane == 0 && bid < 16384) g6_trace[bid][4 + 5 * c + k] = g6_now(); };
#else
    auto stamp = [&](int) {};
#endif
    __syncthreads();                                // the K loop / the previous chunk has left the buffers
    stamp(0);
    if (HASRES) {
      // instruction i fills points 8 i .. 8 i + 7: lane = (point, unit u) fetches channel group u ^ ((X >> 1) & 7)
      for (int i = wave; i < WT * 9 / 8; i += 4) {
        const int Xp = 8 * i + (lane >> 3), un = lane & 7;
        const int off = ptab[Xp];
        const float* gp = res + (off >= 0 ? off + c * G6_CH + 4 * (un ^ ((Xp >> 1) & 7)) : 0);      // dead points: any valid address
        glds16(gp, lds0 + (unsigned)(i * 256) * 4u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                              // the residual tile has landed, for every wave
    }
    stamp(1);
    if (owner) {
      // phase 1.  C/D map of the 32x32 MFMA: col (cout) = lane & 31, row (tile) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 4) {          // four tile rows at a time: 36 LDS reads behind one wait
#ifdef AGZ_TIMING_EXPERIMENTS
        if (c == 0 && wm == 0 && lane == 0 && bid < 16384) g6_trace[bid][12 + e0 / 4] = g6_now();
#endif
        float* p0[4];
        float rr[4][9];
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int e = e0 + ee, row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
          p0[ee] = img + g6_img_off(row, col);      // output k at p0 + k * 64 * 32 (64 k is a multiple of 16: same swizzle)
#pragma unroll
          for (int k = 0; k < 9; ++k) rr[ee][k] = HASRES ? p0[ee][k * (WT * G6_CH)] : 0.f;
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee)
#pragma unroll
          for (int k = 0; k < 9; ++k)
            p0[ee][k * (WT * G6_CH)] = fmaxf(o[k][e0 + ee] * sc + sh + rr[ee][k], relu_lo);
      }
    }
    __syncthreads();
    stamp(2);
    if (owner) {
      g6_emit<MODE>(img, ptab, y, vnext, tb, cb, c, wm, lane, RPB, Mt, TT, T, N);
      stamp(4);
    }
  }
#ifdef AGZ_TIMING_EXPERIMENTS
  if (tid == 128 && bid < 16384) g6_trace[bid][3] = g6_now();
#endif
}

// ------------------------------------------------------------------ host side

// Flux [kw,kh,cin,cout] column-major -> U6[cout block 4][plane xi 25][stage ns][cout 64][4 channels]: one 1-KB piece per
// (plane, stage), a plane image in the V format (wino_v_off with xi = 0); U_xi = G k G^T in float64 (the arithmetic of
// wino_pack_weights)
void wino6_pack_weights(const ConvHost& c, float* out, int ns) {
  static const double G[5][3] = {{0.5, 0.0, 0.0}, {0.5, 0.5, 0.5}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                                 {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
  const int cin = c.cin, cout = c.cout;
  std::memset(out, 0, sizeof(float) * wino6_weight_floats(ns));
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci) {
      double k[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) k[a][b] = c.w[(2 - a) + 3 * ((2 - b) + 3 * (ci + (size_t)cin * o))];
      const int cb = o / G6C, ol = o % G6C, st = ci / WK, cl = ci % WK;
      for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
          double u = 0.0;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) u += G[i][a] * k[a][b] * G[j][b];
          out[(((size_t)cb * WXI + (i * 5 + j)) * ns + st) * 256 + wino_v_off(0, ol, cl >> 1) + (cl & 1)] = (float)u;
        }
    }
}

size_t wino6_weight_floats(int ns) { return (size_t)(kC / G6C) * WXI * ns * 256; }

void launch_wino_gemm6(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                       float* y, float* vnext, const int* d_count, int bcap, int N, int relu, hipStream_t s, int ns) {
  const int T = (N + 2) / 3;
  const long rpb = wino_rows_per_block(T);
  const int blocks = (int)(((long)bcap * T * T + rpb - 1) / rpb);
  const int per_xcd = 2 * ((blocks + 3) / 4);   // see the placement comment in k_wino_gemm6
  const dim3 grid(8 * per_xcd), block(256);
#define G6_LAUNCH(M, S, XX, PAD) hipLaunchKernelGGL((k_wino_gemm6<M, S, XX>), grid, block, PAD, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu)
  if (ns == kWinoStemStages) {                   // the stem: its output is always wanted in HBM (block 0's residual)
    constexpr int S = kWinoStemStages;
    if (vnext) G6_LAUNCH(3, S, 0, 0);
    else G6_LAUNCH(1, S, 0, 0);
    return;
  }
#ifdef AGZ_TIMING_EXPERIMENTS
  static const int xp = getenv("AGZ_WINO_X") ? atoi(getenv("AGZ_WINO_X")) : 0;
  static int traced = 0;
  static const int pad = getenv("AGZ_WINO_ONE") ? 20000 : 0;      // dynamic LDS on top: one workgroup per CU
  if (getenv("AGZ_WINO_TRACE") && y && vnext && res && ++traced == 3) {      // the third steady-state layer launch of the process
    if (xp == 33) G6_LAUNCH(7, WNS, 33, pad);
    else G6_LAUNCH(7, WNS, 0, pad);
    (void)hipStreamSynchronize(s);
    static unsigned long long host[16384][16];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g6_trace), sizeof(host));
    if (FILE* f = fopen(getenv("AGZ_WINO_TRACE"), "wb")) {
      fwrite(host, 1, sizeof(host), f);
      fclose(f);
    }
    return;
  }
  if (xp && y && vnext && res) {
    if (xp == 1) G6_LAUNCH(7, WNS, 1, 0);
    else if (xp == 4) G6_LAUNCH(7, WNS, 4, 0);
    else if (xp == 5) G6_LAUNCH(7, WNS, 5, 0);
    else if (xp == 33) G6_LAUNCH(7, WNS, 33, 0);
    else G6_LAUNCH(7, WNS, 0, 0);
    return;
  }
#endif
  if (y && vnext) { if (res) G6_LAUNCH(7, WNS, 0, 0); else G6_LAUNCH(3, WNS, 0, 0); }
  else if (vnext) { if (res) G6_LAUNCH(6, WNS, 0, 0); else G6_LAUNCH(2, WNS, 0, 0); }
  else { if (res) G6_LAUNCH(5, WNS, 0, 0); else G6_LAUNCH(1, WNS, 0, 0); }
#undef G6_LAUNCH
}

}  // namespace agz
