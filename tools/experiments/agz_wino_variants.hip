// FROZEN COPY of alphago.jl_amd/csrc/agz_wino.hip as it was in round 6 BEFORE its U operand moved to V's one-plane-per-chunk
// layout (commit cabb700): the product file with every timing variant (template parameter X), wall-clock stamp, the RESPF and
// PROLOGUE_STAMP builds and the trace dumps.  tools/build_timing_lib.sh compiles it in place of the product file (ALSO=agz_wino);
// pack and kernels are one translation unit, so the old layout is self-consistent here.  Results of X != 0 are WRONG by design.
// agz_wino.hip -- the 3x3 256->256 tower convolution as Winograd F(3x3, 3x3) on the f32 MFMA.
//
// Why: the direct implicit GEMM (agz_nn.hip) is bound by the exact-f32 MFMA rate (157 TFLOP/s,
// 1/16 of the bf16 rate on gfx950); rocprofv3 puts it at 77 % MFMA-pipe utilisation, so tuning
// has < 1.3x left.  F(3x3,3x3) computes each 3x3 output tile from a 5x5 input patch with 25
// multiplies per (cin, cout) instead of 81 -- 3.24x fewer MFMA cycles, all arithmetic still f32 --
// and a 9x9 board is exactly 3x3 tiles (19x19: 7x7 tiles over a 21x21 padded board).  Measured
// against the float64 oracle a 10-block tower stays at |d pi| ~ 1e-8, |d v| ~ 5e-7 (tolerance 1e-4).
//
//   Y = A^T [ (G k G^T) .* (B^T d B) ] A          interpolation points {0, 1, -1, 2, inf}
//   B^T = [ 2 -1 -2  1  0 ]   A^T = [ 1  1  1  1  0 ]   G = [ 1/2   0    0  ]
//         [ 0  2  1 -1  0 ]         [ 0  1 -1  2  0 ]       [ 1/2  1/2  1/2 ]
//         [ 0 -2  3 -1  0 ]         [ 0  1  1  4  1 ]       [ 1/6 -1/6  1/6 ]
//         [ 0 -1  0  1  0 ]                                 [ 1/6  1/3  2/3 ]
//         [ 0  2 -1 -2  1 ]                                 [  0    0    1  ]
//   (rows of B^T scaled to integers, the inverse scales folded into G, which is applied on the
//   host in float64 at weight-pack time.)
//
// Kernels:
//   k_wino_in     X[M][256] -> V, the 25 transformed planes, written directly in the LDS image order of the
//                 GEMM stages (HBM-bound: reads 1 KB, writes 2.9 KB per board point).  Needed in front of the
//                 FIRST tower layer only when tile blocks hold whole boards (N <= 12); every later layer's V
//                 is emitted by the previous layer's GEMM epilogue.
//   k_wino_gemm4  25 GEMMs  M_xi[tile][cout] = sum_cin V_xi[tile][cin] * U_xi[cin][cout]  on
//                 v_mfma_f32_32x32x2_f32, 64 tiles x 64 couts x 25 planes per workgroup: 410 KB of accumulators,
//                 i.e. the CU's whole 512 KB register file is the tile -- four waves, one per SIMD, 400
//                 accumulator registers each.  Inverse transform, BatchNorm affine, residual, ReLU and the NEXT
//                 layer's input transform follow in the epilogue: M is never written to memory.
// Stage = 4 input channels x {64 tiles + 64 couts} x 25 planes = 52 KB, triple-buffered in LDS and filled by
// direct global->LDS DMA (global_load_lds_dwordx4), which is why V and U are stored in HBM as ready-made,
// bank-swizzled stage images.  64x64 per workgroup gives 16 flop per DMA byte.
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

constexpr int WT = 64;            // tiles per workgroup
constexpr int WC = 64;            // output channels per workgroup
constexpr int WK = 4;             // input channels per stage
constexpr int WNS = kC / WK;      // 64 stages
static_assert(WNS == kWinoStages && kCinStemPad == kWinoStemStages * WK, "stage counts of agz_nn.h");
constexpr int WXI = 25;
constexpr int WPL = 13;           // LDS row planes: a row holds the 4 channels of TWO transform planes
constexpr int A_STAGE = WPL * WT * 8;    // floats (26,624 B)
constexpr int B_STAGE = WPL * WC * 8;
constexpr int STAGE = A_STAGE + B_STAGE;  // 13,312 floats = 52 KB

// U stage image (weights): [plane pair q 13][cout 64][8 dwords]; a row = [plane 2q: 4 cins | plane 2q+1: 4 cins]
// as four 2-dword pairs, logical pair = 2*(xi & 1) + h (h = which half of the 4 channels), rotated by f(row) so
// that the 32 rows a wave reads with one ds_read_b64 (stride 8 dwords: only 8 distinct start banks) hit 64
// distinct banks: rows r, r+8, r+16, r+24 get distinct pair slots.
__host__ __device__ __forceinline__ int wino_rot(int row) { return ((row >> 2) + (row >> 4)) & 3; }
__host__ __device__ __forceinline__ int wino_pair_pos(int row, int xi, int h) {
  return (2 * (xi & 1) + h + wino_rot(row)) & 3;
}
// V stage image (activations): [plane pair q 13][plane parity 2][tile row 64][4 dwords]; the 4 dwords of a row
// are the plane's 4 channels as two pairs, pair h at slot (h + (row >> 4)) & 1.  Row stride 4 dwords: rows r and
// r + 16 start on the same bank and take different slots, so a 32-row ds_read_b64 is conflict-free -- and a
// producer whose lane = tile row writes each 16-byte row whole, 64 lanes = 1 KB contiguous per store
// instruction (the fused input transform of k_wino_gemm4; with 32-byte rows rotated by f(row) its stores were
// 16-byte halves at 32-byte stride plus 24 selects per row, or 8-byte scatters 3.5x slower).
__host__ __device__ __forceinline__ int wino_v_off(int xi, int row, int h) {      // dword offset inside a stage image
  return (xi >> 1) * (WT * 8) + (xi & 1) * (WT * 4) + row * 4 + 2 * ((h + (row >> 4)) & 1);
}

// ---- the split-operand form (AGZ_PRECISION_F32S): every f32 operand x travels as two IEEE halves, hi = half(s x),
// lo = half(s x - hi) (s a power of two: 2^3 for V, 2^10 for U, taken out again exactly in the epilogue's scale), and
// one v_mfma_f32_32x32x16_f16 per plane and 4-channel stage forms all four cross products in f32:
//   A lanes 0-31 (k 0..7) = lanes 32-63 (k 8..15) = [a_hi c0..c3 | a_lo c0..c3]
//   B lanes 0-31          = [b_hi | b_hi],   B lanes 32-63 = [b_lo | b_lo]
//   => sum_k A_k B_k = sum_c (a_hi + a_lo)(b_hi + b_lo): products of halves are exact in f32, operands carry 22
// mantissa bits instead of 24.  A stage image keeps its size (16 B per plane and row = 4 channels x 2 halves), so
// the LDS-DMA stream, the buffers and the epilogue are those of the f32 form; the matrix pipe does 32 cycles per
// plane-stage instead of 128, and the layer becomes bound by moving V (7 GB per layer) instead of by MFMA.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr float kSplitV = 8.f, kSplitU = 1024.f;
__host__ __device__ __forceinline__ void split_half(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}
// V (split): [plane pair][parity][tile row][hi c0..c3 | lo c0..c3], read whole (ds_read_b128): dword offset of a row
__host__ __device__ __forceinline__ int wino_vs_off(int xi, int row) { return (xi >> 1) * (WT * 8) + (xi & 1) * (WT * 4) + row * 4; }
// U (split): the same rows with the hi / lo blocks swapped when bit 4 of the cout row is set (wino_v_off with
// h = 0 for hi, 1 for lo): lanes 0-31 read hi and lanes 32-63 lo with one conflict-free ds_read_b64

// rows of a 64-row tile block that carry tiles: whole boards when a board's tiles pack into 64 rows with
// <= 10 % waste (N <= 12: T*T = 1, 4, 9, 16 -> 64, 64, 63, 64 rows), else dense packing
__host__ __device__ inline int wino_rows_per_block(int T) {
  const int tt = T * T;
  const int whole = (WT / tt) * tt;
  return (tt <= WT && whole * 10 >= WT * 9) ? whole : WT;
}
__host__ __device__ inline bool wino_whole_boards(int T) { return wino_rows_per_block(T) % (T * T) == 0 && T * T <= WT; }

// Densely packed tile blocks (19x19: 49 tiles per board do not fill 64 rows in whole boards): is the 5x5 input patch of
// tile (ti, tj), sitting in row `row` of its block, made of tiles of the SAME block?  Its neighbours (ti + di, tj + dj)
// are rows row + di T + dj.  If so the GEMM epilogue of the layer before can emit this tile's V like it does for
// whole-board blocks (k_wino_gemm4 phase 2); the tiles at the two ends of a block -- about a quarter at 19x19 -- are
// left to k_wino_in in its FIXUP form.  Both kernels decide with this one function.
__host__ __device__ __forceinline__ bool wino_tile_fused(int T, int row, int ti, int tj) {
  const int lo = (ti > 0 ? T : 0) + (tj > 0 ? 1 : 0), hi = (ti < T - 1 ? T : 0) + (tj < T - 1 ? 1 : 0);
  return row - lo >= 0 && row + hi < WT;
}

// ------------------------------------------------------------------ input transform

// B^T x, five values.  ONE arithmetic for both producers of V -- k_wino_in (scalars) and the GEMM epilogue's fused
// transform (channel pairs, bt5p below): with dense tile blocks the same tile is transformed by one or the other
// depending on where its batch row puts it in a block, and a network output must not depend on the batch row (tree
// parity rests on it).  Every multiply-add is an EXPLICIT fma, so that the compiler's contraction choices cannot differ
// between the two kernels (3 x2 - x3 rounds differently fused and unfused).
__device__ __forceinline__ void bt5(float x0, float x1, float x2, float x3, float x4, float* r) {
  r[3] = x3 - x1;
  r[0] = __builtin_fmaf(2.f, x0 - x2, r[3]);
  r[4] = __builtin_fmaf(-2.f, r[3], x4 - x2);
  r[1] = __builtin_fmaf(2.f, x1, x2 - x3);
  r[2] = __builtin_fmaf(-2.f, x1, __builtin_fmaf(3.f, x2, -x3));
}

// grid = 2 x tile blocks (32-tile halves); 256 threads = 32 tiles x 8 channel pairs (= 4 consecutive stages x 2
// halves): the eight lanes of a tile read one 64-byte run of every patch point.  A pass covers 16 channels = 4
// stages; the transformed values go to an LDS copy of this half-block's part of the four stage images (4 x 26
// chunks of 32 rows x 16 B) and leave for HBM as whole 512-byte runs, 16 B per lane.  The LDS copies of the four
// stages are skewed by 8 dwords each: a ds_write_b64 is served 16 lanes (2 tiles x 8 lanes) at a time over 32
// banks, and with that skew the 16 lanes cover all 32 banks exactly once.
// NS = stages = input channels / 4: 64 for a tower layer, 8 for the stem (17 feature planes padded to 32)
// FIXUP (dense tile blocks only): only the tiles the previous layer's GEMM epilogue could not emit (!wino_tile_fused:
// the ends of every block) and the rows past the batch (zeros) are transformed and stored; the other rows of the
// image are already in place and are not touched.
template <int TPB, bool NT, bool SPLIT = false, int NS = WNS, bool FIXUP = false>
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ x, float* __restrict__ vimg,
                                                  const int* __restrict__ d_count, int N, int T) {
  static_assert(TPB == 32, "the copy-out below moves 32-row chunks");
  constexpr int LPT = 256 / TPB;             // lanes per tile
  constexpr int SP = LPT / 2;                // stages per pass
  constexpr int CH = TPB * 4;                // dwords per chunk (TPB rows of one plane)
  constexpr int IMG = 26 * CH + 8;           // LDS stride between the stage images of a pass (+8: the bank skew)
  __shared__ __attribute__((aligned(16))) float img[SP * IMG];
  const int P = N * N, TT = T * T;
  const int RPB = wino_rows_per_block(T);          // rows of a tile block that carry tiles (whole boards, N <= 12)
  const long Mt = (long)(*d_count) * TT;
  constexpr int PARTS = WT / TPB;
  const int tb = blockIdx.x / PARTS, part = blockIdx.x % PARTS;
  if ((long)tb * RPB + part * TPB >= Mt) return;
  const int hs = threadIdx.x % LPT, h = hs & 1, sl = hs >> 1;
  const int tl = threadIdx.x / LPT;                 // tile within this part
  const int row = part * TPB + tl;                  // row of the 64-row stage image
  const long tile = (long)tb * RPB + row;
  bool live = row < RPB && tile < Mt;
  const int b = live ? (int)(tile / TT) : 0, t = live ? (int)(tile % TT) : 0;
  const int ti = t / T, tj = t % T;
  if (FIXUP && live && wino_tile_fused(T, row, ti, tj)) live = false;       // in place already: nothing to read
  int off[25];                                      // element offsets < 2^31 (8192 x 361 x 256 = 7.6e8)
#pragma unroll
  for (int u = 0; u < 5; ++u)
#pragma unroll
    for (int v = 0; v < 5; ++v) {
      const int pi = 3 * ti - 1 + u, pj = 3 * tj - 1 + v;
      const bool ok = live && pi >= 0 && pi < N && pj >= 0 && pj < N;
      off[u * 5 + v] = ok ? (b * P + pi + N * pj) * (NS * WK) : -1;
    }
  // f32 form: this thread's channel pair is an 8-byte slot of the row; split form: two 4-byte slots (hi pair, lo pair)
  float* mine = img + sl * IMG + tl * 4 + (SPLIT ? 0 : 2 * ((h + (row >> 4)) & 1));
  // the unused plane slot 25 (chunk 25) is copied out with the rest: keep it finite
  if (SPLIT) { mine[25 * CH + h] = 0.f; mine[25 * CH + 2 + h] = 0.f; }
  else *reinterpret_cast<float2*>(mine + 25 * CH) = make_float2(0.f, 0.f);
  float* gdst = vimg + (long)tb * NS * A_STAGE + part * CH;
  const int cq = threadIdx.x / TPB, cl = threadIdx.x % TPB;      // copy-out: 8 chunks per round, 32 lanes each
  bool copy_row = true;                                         // FIXUP: only the rows this kernel computed leave
  if (FIXUP) {
    const int crow = part * TPB + cl;
    const long ctile = (long)tb * RPB + crow;
    const int ct = (int)(ctile % TT);
    copy_row = !(crow < RPB && ctile < Mt && wino_tile_fused(T, crow, ct / T, ct % T));
  }
  for (int sg = 0; sg < NS / SP; ++sg) {
    const int st = sg * SP + sl;
    float2 d[25];
#pragma unroll
    for (int q = 0; q < 25; ++q)
      d[q] = off[q] >= 0 ? *reinterpret_cast<const float2*>(x + off[q] + st * WK + 2 * h) : make_float2(0.f, 0.f);
    // V = B^T d B, one channel component at a time
    float tx[25], ty[25];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
      float r[5];
      bt5(d[0 * 5 + v].x, d[1 * 5 + v].x, d[2 * 5 + v].x, d[3 * 5 + v].x, d[4 * 5 + v].x, r);
#pragma unroll
      for (int i = 0; i < 5; ++i) tx[i * 5 + v] = r[i];
      bt5(d[0 * 5 + v].y, d[1 * 5 + v].y, d[2 * 5 + v].y, d[3 * 5 + v].y, d[4 * 5 + v].y, r);
#pragma unroll
      for (int i = 0; i < 5; ++i) ty[i * 5 + v] = r[i];
    }
    if (sg) __syncthreads();                        // the previous pass has left the LDS image
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float rx[5], ry[5];
      bt5(tx[i * 5 + 0], tx[i * 5 + 1], tx[i * 5 + 2], tx[i * 5 + 3], tx[i * 5 + 4], rx);
      bt5(ty[i * 5 + 0], ty[i * 5 + 1], ty[i * 5 + 2], ty[i * 5 + 3], ty[i * 5 + 4], ry);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (SPLIT) {
          _Float16 xh, xl, yh, yl;
          split_half(kSplitV * rx[j], xh, xl);
          split_half(kSplitV * ry[j], yh, yl);
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<h2*>(mine + (i * 5 + j) * CH + h) = (h2){xh, yh};
          *reinterpret_cast<h2*>(mine + (i * 5 + j) * CH + 2 + h) = (h2){xl, yl};
        } else {
          *reinterpret_cast<float2*>(mine + (i * 5 + j) * CH) = make_float2(rx[j], ry[j]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 13; ++r) {
      const int c = r * 8 + cq, s4 = c / 26, xi = c % 26;     // chunk c = (stage s4 of this pass, plane xi)
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 v = *reinterpret_cast<const f32x4*>(img + s4 * IMG + xi * CH + cl * 4);
      f32x4* gp = reinterpret_cast<f32x4*>(gdst + (long)(sg * SP + s4) * A_STAGE + (xi >> 1) * (WT * 8) + (xi & 1) * (WT * 4) + cl * 4);
      if (FIXUP && !copy_row) continue;
      if (NT) __builtin_nontemporal_store(v, gp);     // V is 2.8x the activations and is read back much later
      else *gp = v;
    }
  }
}

// ------------------------------------------------------------------ GEMM + output transform + next input transform
//
// k_wino_gemm4: the same 64 tiles x 64 couts x 25 planes workgroup tile, held by FOUR waves -- one per SIMD,
// each with all 25 planes of its 32 x 32 quadrant: 400 accumulator registers of the 512 a lone wave owns on
// gfx950 (VGPR + AGPR are one file).  What that buys over round 1's 12-wave form (4 quadrants x 3 plane groups,
// partial inverse transforms meeting through LDS in three barrier-separated rounds):
//   * the inverse transform A^T M A happens entirely in one wave's registers -- no partial sums crossing LDS,
//     no epilogue barriers -- so the stage buffers are free for something else the moment the K loop ends;
//   * that something else is the NEXT layer's input transform: a workgroup owns WHOLE boards (63 tile rows =
//     7 boards at 9x9), so after its epilogue every 5x5 input patch of its 64 output channels, halo included,
//     is at hand.  The activations go to an LDS board image, and the same workgroup emits the next layer's V
//     stage images (16 of the 64 stages: its 64 couts are the next layer's cins 64 cb .. 64 cb + 63) straight
//     to HBM.  k_wino_in -- a separate, HBM-bound pass with the MFMA pipe idle, 20 % of a step -- disappears
//     from every layer but the first, and y itself is only written where a residual or the heads need it;
//   * a v_mfma_f32_32x32x2_f32 keeps its SIMD's matrix pipe busy for 64 cycles, i.e. ~16 issue slots: the two
//     ds_read_b64 of a plane and the occasional LDS-DMA instruction fit between two MFMAs of ONE wave, so
//     there is nothing for co-resident waves to hide and nothing for them to contend for.
// K loop: operands are read 4 planes (>= 256 MFMA cycles) ahead through a ring of 5 register pairs; the stage
// barrier sits in the READ stream (before the first read of the next stage, i.e. 4 planes before the stage's
// last MFMA), so the MFMAs of planes 21..24 cover the barrier and the first LDS latencies of the next stage.
// The epilogue's tile image: img[X][16 units of 16 B], X = output k * 64 + tile row, this workgroup's 64 channels
// of output point X.  Unit u holds channel group u ^ (X & 15) = u ^ (row & 15): rows are exactly 256 B (an LDS-DMA
// instruction fills four of them, each lane choosing the global 16 B that belongs in its slot), a wave whose lanes are
// tile rows reads one channel group of 16 different rows per LDS cycle from 16 different units, and a wave whose
// lanes are channels touches every bank once.  k-major, so that the swizzle of a lane does not depend on k: the nine
// outputs of a (row, channel) pair sit at nine constant offsets from one address, and a thread of the flat pass keeps
// one channel group for all its elements -- the epilogue is VALU-issue bound (a lone wave per SIMD), address
// arithmetic per access is what it can least afford.
constexpr int IMG_FLOATS = WT * 9 * WC;          // 147,456 B
#ifdef AGZ_TIMING_EXPERIMENTS
// per workgroup: {hw id | xcc id << 32, start, K loop done, phase 1 done, phase 1b done, end} on the 100 MHz wall clock
__device__ unsigned long long g4_trace[16384][8];
#define G4_STAMP(k) do { if (tid == 0 && blockIdx.x < 16384) g4_trace[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define G4_STAMP(k) do { } while (0)
#endif

// MODE bit 0: write y (affine, residual, ReLU applied); bit 1: emit the next layer's V stage images.
// X: timing experiments, instantiated only under -DAGZ_TIMING_EXPERIMENTS (results are WRONG for X != 0):
//   1 = K loop only; 2 = no epilogue 2; 3 = epilogue 2 without its global stores; 4 = no DMA after the prologue;
//   5 = no MFMA; 6 = no LDS operand reads; 7..10, 16..18 = projections (see the K loop); 13 / 14 / 15 = every stage's DMA (of both operands / U / V) from the same two (L2-resident) stage images
// NS: stages of the K loop (input channels / 4): 64, or 8 for the stem
// COH: every LDS-DMA load carries sc1 (served by L2, not by this CU's L1): the persistent tower kernel below reads what
// other workgroups of its XCD wrote earlier in the same launch.  Costs nothing (same-box A/B +-0).
// One workgroup's work: tile block tb x cout block cb of one layer.  lds / ptab: the workgroup's shared memory.
template <int X, bool SPLIT, int NS, bool COH>
__device__ __forceinline__ void wino_wg(
    float* __restrict__ lds, int* __restrict__ ptab, const float* __restrict__ vimg, const float* __restrict__ uimg,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ vnext, const long Mt, const int N, const int T, const int relu, const int tb, const int cb, const int MODE,
    const int tid) {
  const int P = N * N, TT = T * T;
  const int RPB = wino_rows_per_block(T);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, hi = lane >> 5;
#ifdef AGZ_TIMING_EXPERIMENTS
  if (tid == 0 && blockIdx.x < 16384) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g4_trace[blockIdx.x][0] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  }
  G4_STAMP(1);
#endif
  // (timing 16 / 17 / 18: what a quad-private, cache-resident V scratch would buy -- 64 slabs instead of one per tile
  // block: 17 the V loads come from slab tb % 64, 16 the V stores go there, 18 both)
  const float* asrc = vimg + (long)((X == 17 || X == 18) ? (tb & 63) : tb) * NS * A_STAGE;
  const float* bsrc = uimg + (long)cb * NS * B_STAGE;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0];

  // a stage is 52 chunks of 1 KB: 0..25 from V, 26..51 from U; wave w moves the 13 consecutive chunks 13 w ..
  // (waves 0, 1: V; waves 2, 3: U) with a wave-uniform base in SGPRs and a 32-bit lane offset: per-lane 64-bit
  // address arithmetic in front of every piece cost 8 % of the K loop (2.10 -> 1.95 ms per layer)
  auto dma = [&](int st, int buf, int j) {
    const int c = 13 * wave + j;
    // (timing: every stage from the same two L2-resident stage images: 13 both operands, 14 only U, 15 only V)
    const int sa = (X == 13 || X == 15) ? (st & 1) : st, sb = (X == 13 || X == 14) ? (st & 1) : st;
    const float* g = wave < 2 ? asrc + (long)sa * A_STAGE + c * 256 : bsrc + (long)sb * B_STAGE + (c - 26) * 256;
    if constexpr (COH) glds16s_l2(g, (unsigned)lane * 16u, lds0 + (unsigned)(buf * STAGE + c * 256) * 4u);
    else glds16s(g, (unsigned)lane * 16u, lds0 + (unsigned)(buf * STAGE + c * 256) * 4u);
  };

  // the first stage is on its way before anything else
#pragma unroll
  for (int j = 0; j < 13; ++j) dma(0, 0, j);

  for (int idx = tid; idx < WT * 9; idx += 256) {     // (published by the barrier in front of the first operand reads)
    const int row = idx & (WT - 1), k = idx >> 6;       // X = k * 64 + row
    const long tile = (long)tb * RPB + row;
    int off = -1;
    if (row < RPB && tile < Mt) {
      // 32-bit divisions (tile < Mt < 2^31, checked by the launchers): the 64-bit ones were a third of this set-up, which
      // is NOT hidden behind the first stages' flight -- they land before it is done (wall-clock trace: 3.5 us of set-up,
      // 0.1 us of waiting)
      const unsigned tile32 = (unsigned)tile, b = tile32 / (unsigned)TT, t = tile32 - b * (unsigned)TT;
      const unsigned ti = t / (unsigned)T, k3 = (unsigned)k / 3u;
      const int pi = (int)(3 * ti + k3), pj = (int)(3 * (t - ti * T) + ((unsigned)k - 3 * k3));
      if (pi < N && pj < N) off = ((int)b * P + pi + N * pj) * kC + cb * WC;      // < 2^31: 8192 x 361 x 256 = 7.6e8
    }
    ptab[idx] = off;
  }
  // stage 1 behind the point table rather than right behind stage 0: 26 pieces back to back stall at issue when a
  // whole round of workgroups starts at once (set-up 1.9 us at best, 3.8 on average); -0.1 % per forward, same box
#pragma unroll
  for (int j = 0; j < 13; ++j) dma(1, 1, j);

  // (not zeroed: the first MFMA of every plane takes C = 0 as an inline constant -- 400 register writes per lane that
  // sat, exposed, between the workgroup's start and its first MFMA)
  f32x16 acc[WXI];
  if (X == 5) {      // (the no-MFMA timing variant never writes them)
#pragma unroll
    for (int i = 0; i < WXI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  }

  const int arow = wm * 32 + l31, brow = wn * 32 + l31;
  const int brot = wino_rot(brow);
  // per-lane float offsets of the even / odd planes' pair slots; plane xi adds (xi >> 1) * 512 floats
  const int aoff[2] = {wino_v_off(0, arow, hi), wino_v_off(1, arow, hi)};
  const int boff[2] = {A_STAGE + brow * 8 + 2 * ((hi + brot) & 3), A_STAGE + brow * 8 + 2 * ((2 + hi + brot) & 3)};
  constexpr int LA = 4, RING = LA + 1;     // 25 % RING == 0: ring slots are compile-time within a stage
  // operands of a plane: f32 form a float2 of A and of B (the lane's two channels); split form the whole 16-byte
  // A row (hi and lo halves of 4 channels) and the lane's 8-byte hi or lo block of B
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef typename std::conditional<SPLIT, f32x4, float2>::type a_t;
  a_t ra[RING];
  float2 rb[RING];
  const int asoff[2] = {wino_vs_off(0, arow), wino_vs_off(1, arow)};
  const int bsoff[2] = {A_STAGE + wino_v_off(0, brow, hi), A_STAGE + wino_v_off(1, brow, hi)};
  auto load = [&](const float* L, int xi, a_t& a, float2& b) {
    if constexpr (SPLIT) {
      a = *reinterpret_cast<const f32x4*>(L + asoff[xi & 1] + (xi >> 1) * (WT * 8));
      b = *reinterpret_cast<const float2*>(L + bsoff[xi & 1] + (xi >> 1) * (WC * 8));
    } else {
      if (X == 6) { a = make_float2(1.f, (float)lane); b = make_float2(2.f, (float)xi); return; }
      a = *reinterpret_cast<const float2*>(L + aoff[xi & 1] + (xi >> 1) * (WT * 8));
      b = *reinterpret_cast<const float2*>(L + boff[xi & 1] + (xi >> 1) * (WC * 8));
    }
  };

  // 400 accumulators: hipcc gives EVERY MFMA of a kernel the same accumulator register class, so beyond 256 it
  // shuttles tuples between the two halves of the file around every MFMA (308 v_accvgpr moves and 24 scratch
  // accesses per stage when left alone).  The planes of transform columns 0..2 (xi % 5 < 3: 15 planes) therefore
  // use the builtin (accumulators in AGPRs) and the 10 planes of columns 3, 4 an asm MFMA whose "+v" constraint
  // keeps their 160 registers in the VGPR half -- whole columns, so that the inverse transform can start with the
  // two columns that are already in VGPRs and free 160 registers before the AGPR columns are read (no spill).
  // Wait states (cdna_hip_programming.md 5.7): an accumulate chain needs none; the leading s_nop 1 covers a
  // compiler v_mov into an A/B operand; the D -> VALU distance is padded once, after the K loop.
  auto in_agpr = [](int k) { return k % 5 < 3; };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto mma0 = [&](int k, const a_t& a, const float2& b) {      // the plane's first k-step: D = A B + 0
    if constexpr (SPLIT) {
      const h8 ah = __builtin_bit_cast(h8, a);
      const f32x4 bb = {b.x, b.y, b.x, b.y};
      const h8 bh = __builtin_bit_cast(h8, bb);
      if (in_agpr(k)) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, zero16, 0, 0, 0);
      else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[k]) : "v"(ah), "v"(bh));
    } else if (in_agpr(k)) {
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, zero16, 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[k], 0, 0, 0);
    } else {
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0\n\tv_mfma_f32_32x32x2_f32 %0, %3, %4, %0"
                   : "=&v"(acc[k])
                   : "v"(a.x), "v"(b.x), "v"(a.y), "v"(b.y));
    }
  };
  auto mma = [&](int k, const a_t& a, const float2& b) {
    if constexpr (SPLIT) {
      const h8 ah = __builtin_bit_cast(h8, a);
      const f32x4 bb = {b.x, b.y, b.x, b.y};                   // [block | block]: hi for lanes 0-31, lo for 32-63
      const h8 bh = __builtin_bit_cast(h8, bb);
      if (in_agpr(k)) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[k], 0, 0, 0);
      else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(ah), "v"(bh));
    } else if (in_agpr(k)) {
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[k], 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[k], 0, 0, 0);
    } else {
      asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x2_f32 %0, %3, %4, %0"
                   : "+v"(acc[k])
                   : "v"(a.x), "v"(b.x), "v"(a.y), "v"(b.y));
    }
  };

  // (projection experiments, DESIGN 4e: X = 8 moves 11 instead of 13 pieces per wave and stage -- the DMA volume of a
  // half-transformed A operand; X = 9 adds the 50 packed operations per stage its second pass would cost the wave;
  // X = 10 stores 15 of the 25 planes in phase 2 after one transform pass; X = 7 all three.  Results are WRONG.)
  constexpr int PW = (X == 7 || X == 8) ? 11 : 13;
  auto wait_stage = [&]() {
    if constexpr (PW == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  };
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t dum0 = {1.f, 2.f}, dum1 = {3.f, 4.f}, dumc = {0.5f, 0.25f};
#ifdef AGZ_X_PROLOGUE_STAMP
  G4_STAMP(6);      // (this build only: slot 6 = set-up done, slot 7 = first stage landed and published)
#endif
  wait_stage();
  __syncthreads();
#ifdef AGZ_X_PROLOGUE_STAMP
  G4_STAMP(7);
#endif
#pragma unroll
  for (int k = 0; k < LA; ++k) load(lds, k, ra[k], rb[k]);

  // One stage.  MORE: stage st+2 exists and is fetched during this stage; NEXT: stage st+1 exists.  Both are
  // compile-time so that the 62 full stages run a branch-free body (the two tail stages are separate code).
  // Residual prefetch.  The tile image of the epilogue lies over the three stage buffers; the buffer of stage NS-3 is
  // dead during stage NS-2 and that of stage NS-2 during stage NS-1, when the K loop has no pieces of its own left to
  // fetch: the residual's pieces for those two thirds of the image (instruction i = wave + 4 n fills image bytes
  // 1024 i ..: n 13..25 lie in buffer 1, n 26..35 in buffer 2, NS % 3 == 1 makes those the dead ones) go out in the DMA
  // slots of the last two stages instead of behind the loop, where the inverse transform (1.5 us) is too short to cover
  // 144 KB arriving from HBM at 16 B/clk/CU (3.9 us).  n 0..12 (buffer 0, the last stage's) follow behind the loop.
  static_assert(NS % 3 == 1 || NS == kWinoStemStages, "residual prefetch assumes the last stage is read from buffer 0");
#ifdef AGZ_X_RESPF         // measured: phase 1 -2.9 us, phase 2 +2.1 us (the epilogues stay in lockstep), layer +-0
  constexpr bool RESPF = NS % 3 == 1;
#else
  constexpr bool RESPF = false;
#endif
  auto rdma = [&](int n) {
    // instruction i fills points 4i .. 4i+3: lane = (point, unit u) fetches channel group u ^ (X & 15)
    const int i = wave + 4 * n;
    const int Xp = 4 * i + (lane >> 4), u = lane & 15;
    const int off = ptab[Xp];
    // uniform base + 32-bit byte offset (< 2^32: checked by the launchers); dead points: any valid address
    const unsigned boff = off >= 0 ? 4u * (unsigned)(off + 4 * (u ^ (Xp & 15))) : 0u;
    if constexpr (COH) glds16s_l2(res, boff, lds0 + (unsigned)(i * 256) * 4u);
    else glds16s(res, boff, lds0 + (unsigned)(i * 256) * 4u);
  };
  int buf = 0;
  auto stage = [&](int st, auto more_c, auto next_c, auto first_c) {
    constexpr bool more = decltype(more_c)::value, next = decltype(next_c)::value, first = decltype(first_c)::value;
    const int nbuf = buf == 2 ? 0 : buf + 1;          // stage st+1
    const int dbuf = buf == 0 ? 2 : buf - 1;          // stage st+2 (= the buffer stage st-1 was read from)
    const float* L = lds + buf * STAGE;
    const float* Ln = lds + nbuf * STAGE;
#pragma unroll
    for (int k = 0; k < WXI; ++k) {
      const int t = k + LA;
      if (t == WXI && next) {
        // everything this wave owes to stage st+1 has landed (its share of stage st+2, all 13 pieces issued by
        // now, may still be in flight); hipcc adds lgkmcnt(0) in front of the barrier: all reads of stage st are back
        // (stage NS-2 with a residual: the 13 youngest are this stage's residual pieces)
        if (more && X != 4) wait_stage();
        else if (!more && RESPF && res && X != 4) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (t < WXI) load(L, t, ra[t % RING], rb[t % RING]);
      else if (next) load(Ln, t - WXI, ra[t % RING], rb[t % RING]);
      // one piece per plane slot: in the middle of the stage in the f32 form (vs every other slot: -2 %), at its very
      // top in the split form, whose matrix work is short and whose loop waits for the stream anyway
      constexpr int D0 = SPLIT ? 0 : 6;      // (f32, same-box: D0 = 3 and 9 the same, D0 = 0 +1 % per forward)
      if (more && k >= D0 && k < D0 + PW) {
        __builtin_amdgcn_sched_barrier(0);
        if (X != 4) dma(st + 2, dbuf, k - D0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (X == 7 || X == 9) {
        asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n\tv_pk_fma_f32 %1, %1, %2, %1" : "+v"(dum0), "+v"(dum1) : "v"(dumc));
      }
      if (!more && RESPF && k >= D0 && k < D0 + (next ? 13 : 10)) {
        __builtin_amdgcn_sched_barrier(0);
        if (res) rdma((next ? 13 : 26) + k - D0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (X != 5) {
        if (first) mma0(k, ra[k % RING], rb[k % RING]);
        else mma(k, ra[k % RING], rb[k % RING]);
      }
    }
    buf = nbuf;
  };
  static_assert(NS >= 4, "stage 0 is peeled off the loop");
  stage(0, std::true_type{}, std::true_type{}, std::true_type{});
  for (int st = 1; st < NS - 2; ++st) stage(st, std::true_type{}, std::true_type{}, std::false_type{});
  stage(NS - 2, std::false_type{}, std::true_type{}, std::false_type{});
  stage(NS - 1, std::false_type{}, std::false_type{}, std::false_type{});

  // the asm MFMAs' D registers -> VALU readers below: 18 wait states.  The pad must NAME those registers, or the
  // scheduler is free to lift a register-only VALU read of them above it (seen: 4e-4 errors on some lanes)
  static_assert(WXI == 25, "the pad below lists the accumulators of transform columns 3 and 4");
  asm volatile("s_nop 15\n\ts_nop 15"
               : "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[18]),
                 "+v"(acc[19]), "+v"(acc[23]), "+v"(acc[24]));
  if (X == 1) {
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < WXI; ++k) keep += acc[k][0] + acc[k][15];
    if (keep == 123.456f) y[0] = keep;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // residual pieces may still be landing in LDS
    return;
  }

  // ---- epilogue.  The stage buffers are dead once every wave has left the K loop; they become the tile image
  // img (layout above).  Phases, each with every memory operation of the phase in flight at once -- a lone wave
  // per SIMD has nobody to hide a load -> use round trip behind:
  //   0   residual -> img by LDS-DMA (144 KB in 144 instructions), in flight during the register work of phase 1
  //   1   inverse transform A^T M A + BatchNorm affine in registers, then img = ReLU(img (the residual) + value)
  //   1b  img -> y: 256-byte runs per output point, 16 B per lane; only where y is wanted (MODE & 1)
  //   2   next layer's input transform V = B^T d B from img -> HBM stage images   (MODE & 2)
  __syncthreads();
  G4_STAMP(2);
  float* img = lds;
  static_assert(IMG_FLOATS + WC <= 3 * STAGE, "tile image (+ the block of zeros phase 2 reads for off-board points) exceeds the stage buffers");
  if (tid < WC) img[IMG_FLOATS + tid] = 0.f;        // (published by the barrier behind phase 1)
  const bool pass1b = (MODE & 1) != 0;
  if (res) {
    static_assert(WT * 9 / 4 == 4 * 36, "36 residual pieces per wave");
#pragma unroll
    for (int n = 0; n < (RESPF ? 13 : 36); ++n) rdma(n);
  }
  {
    // phase 1.  C/D map of the 32x32 MFMA: col (cout) = lane & 31, row (tile) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const int col = wn * 32 + l31;
    const float sc = scale[cb * WC + col], sh = shift[cb * WC + col];
    const float relu_lo = relu ? 0.f : -3.0e38f;      // ReLU here, on the way into the image (one v_max either way): phase 1b only has to copy
    // whole 16-register tuples at a time (element e of a tuple = tile row e of the C/D map): extracting single
    // elements of AGPR-resident tuples made hipcc copy entire tuples back and forth (330 instructions per e)
    f32x16 o[9];
    {
      f32x16 tmp[3][5];
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const int j = (jj + 3) % 5;          // columns 3, 4 (VGPR-resident) first
        const f32x16 m0 = acc[0 * 5 + j], m1 = acc[1 * 5 + j], m2 = acc[2 * 5 + j], m3 = acc[3 * 5 + j], m4 = acc[4 * 5 + j];
        tmp[0][j] = ((m0 + m1) + m2) + m3;
        tmp[1][j] = (m1 - m2) + 2.f * m3;
        tmp[2][j] = ((m1 + m2) + 4.f * m3) + m4;
        __builtin_amdgcn_sched_barrier(0);       // a column at a time: its five accumulators die here
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        o[i * 3 + 0] = ((tmp[i][0] + tmp[i][1]) + tmp[i][2]) + tmp[i][3];
        o[i * 3 + 1] = (tmp[i][1] - tmp[i][2]) + 2.f * tmp[i][3];
        o[i * 3 + 2] = ((tmp[i][1] + tmp[i][2]) + 4.f * tmp[i][3]) + tmp[i][4];
      }
    }
    // hipcc would otherwise sink the whole transform below the wait and the barrier (it is register arithmetic, nothing
    // orders it against them) and the wave would sit out the residual's flight before starting on it
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if (RESPF) asm volatile("" : "+v"(o[k]));
#ifndef AGZ_X_PROLOGUE_STAMP
    G4_STAMP(6);
#endif
    if (res) {                                    // the residual tile has landed, for every wave
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
#ifndef AGZ_X_PROLOGUE_STAMP
    G4_STAMP(7);
#endif
    // img = (residual +) value.  ds_add_f32 would do the sum in one instruction, but LDS float atomics run at a
    // fraction of the ds_write rate (+0.75 ms per layer measured): read the nine residuals of a row, add, write
    auto rows = [&](auto with_res) {
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 4) {          // four tile rows at a time: 36 LDS reads behind one wait
        // output k of a (row, cout) at byte k * 16 KB from the row's address: three bases 64 KB apart, so that every
        // access is base + a 16-bit immediate (the compiler otherwise adds a 32-bit constant in front of each access
        // with k >= 4) and neighbouring k pair up in ds_read2st64 / ds_write2st64
        typedef __attribute__((address_space(3))) float lds_f;
        unsigned pa[4][3];
        float rr[4][9];
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int e = e0 + ee, row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
          pa[ee][0] = lds0 + 4u * (unsigned)(row * WC + 4 * ((col >> 2) ^ (row & 15)) + (col & 3));
          pa[ee][1] = pa[ee][0] + 4u * 4 * (WT * WC);
          pa[ee][2] = pa[ee][0] + 4u * 8 * (WT * WC);
          asm volatile("" : "+v"(pa[ee][1]), "+v"(pa[ee][2]));
#pragma unroll
          for (int k = 0; k < 9; ++k)
            rr[ee][k] = decltype(with_res)::value ? *((lds_f*)(size_t)pa[ee][k >> 2] + (k & 3) * (WT * WC)) : 0.f;
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee)
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            float v = o[k][e0 + ee] * sc + sh + rr[ee][k];
            v = fmaxf(v, relu_lo);
            *((lds_f*)(size_t)pa[ee][k >> 2] + (k & 3) * (WT * WC)) = v;
          }
      }
    };
    if (res) rows(std::true_type{});
    else rows(std::false_type{});
  }
  __syncthreads();
  G4_STAMP(3);
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  if (pass1b) {
    // phase 1b: element = (row, k, 4 channels); 16 consecutive lanes cover the 256 contiguous bytes of one point
    constexpr int PER = WT * 9 * (WC / 4) / 256;      // 36 per thread
    // element i of thread tid: point X = (tid >> 4) + 16 i, unit tid & 15 -> channel group (tid ^ (tid >> 4)) & 15 for
    // every i: one LDS address and one channel offset per thread, the rest are immediate offsets
    const int cg4 = 4 * ((tid ^ (tid >> 4)) & 15);
    f32x4* ip0 = reinterpret_cast<f32x4*>(img) + tid;
    const int* pt0 = ptab + (tid >> 4);
    int offs[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) offs[i] = pt0[16 * i];
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += 12) {            // twelve elements at a time: no branch between a load and its use
      f32x4 v[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) v[j] = ip0[256 * (i0 + j)];
#pragma unroll
      for (int j = 0; j < 12; ++j)
        if ((MODE & 1) && offs[i0 + j] >= 0) *reinterpret_cast<f32x4*>(y + offs[i0 + j] + cg4) = v[j];
    }
  }
  G4_STAMP(4);
  if (!(MODE & 2) || X == 2) return;

  // ---- phase 2: the next layer's input transform for this workgroup's 64 channels (= stages 16 cb .. 16 cb + 15
  // of the next layer's K loop).  Task = (tile row, stage): lane = row, so that the 64 lanes of a wave fill 64
  // consecutive 32-byte rows of a stage image; wave w takes stages w, w+4, w+8, w+12.
  {
    const int row = lane;
    const long tile = (long)tb * RPB + row;
    const bool live = row < RPB && tile < Mt;
    const int t = live ? (int)(tile % TT) : 0;
    const int ti = t / T, tj = t % T;
    // Whole-board blocks (N <= 12): every tile's patch is in the block, all 64 rows are written (dead rows as zeros).
    // Dense blocks (19x19): only the tiles whose neighbours are rows of this block (wino_tile_fused); the others, and
    // the dead rows, are k_wino_in<FIXUP>'s.
    const bool emit = wino_whole_boards(T) || (live && wino_tile_fused(T, row, ti, tj));
    // Patch point (u, v) of tile (ti, tj) is board point (3 ti - 1 + u, 3 tj - 1 + v): output k = (ku, kv) of the tile
    // du rows of tiles / dv tiles further on, with (du, ku) = (-1, 2), (0, 0), (0, 1), (0, 2), (1, 0) for u = 0..4 -- written
    // out, because the compiler cannot see that and spends fifty run-time divisions by 3 on `pi / 3`, `pi % 3`.  A point
    // off the board (or a lane that emits nothing) reads a block of zeros behind the image instead of being masked out:
    // no exec juggling around 25 (50) conditional loads per task.
    // adr[q]: LDS byte address of patch point q's unit for stage `wave` (this wave's first): row Xq of the image, unit
    // wave ^ (Xq & 15), and in the f32 form the lane's half of it.  Stage wave + 4 it is the same address with bits 6-7
    // XORed by it (the image is 256-byte aligned and a row is 256 bytes): one v_xor per read in the loop below instead
    // of xor + shift-add + add (-200 instructions per wave and workgroup)
    unsigned adr[25];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
      for (int v = 0; v < 5; ++v) {
        const int du = u == 0 ? -1 : (u == 4 ? 1 : 0), ku = u == 0 ? 2 : (u == 4 ? 0 : u - 1);
        const int dv = v == 0 ? -1 : (v == 4 ? 1 : 0), kv = v == 0 ? 2 : (v == 4 ? 0 : v - 1);
        const int pi = 3 * ti - 1 + u, pj = 3 * tj - 1 + v;
        const bool ok = live && emit && pi >= 0 && pi < N && pj >= 0 && pj < N;
        const int Xq = (ku * 3 + kv) * WT + row + du * T + dv;
        const int pb = ok ? Xq * WC : IMG_FLOATS, xm = ok ? (Xq & 15) : 0;
        // a V image row is one plane's 4 channels as two pairs, pair h at slot (h + (row >> 4)) & 1 (wino_v_off)
        adr[u * 5 + v] = lds0 + 4u * (unsigned)(pb + 4 * (wave ^ xm) + (SPLIT ? 0 : 2 * ((row >> 4) & 1)));
      }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(WC / WK == 16, "four stages per wave");
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {      // (unrolling by 2 for ILP: no change)
      const int sl = wave + 4 * it;
      const unsigned xo = (unsigned)it << 6;
      f32x4 d[25];
#pragma unroll
      for (int q = 0; q < 25; ++q) {
        const unsigned a0 = adr[q] ^ xo;
        if constexpr (SPLIT) {
          d[q] = *(const __attribute__((address_space(3))) f32x4*)(size_t)a0;
        } else {
          // the lane's two channel pairs in the order its V row wants them (rows with bit 4 set store pair 1 first): two
          // 8-byte reads at lane-dependent halves of the unit instead of 104 selects per task on the way out
          const f32x2 a = *(const __attribute__((address_space(3))) f32x2*)(size_t)a0;
          const f32x2 b = *(const __attribute__((address_space(3))) f32x2*)(size_t)(a0 ^ 8u);
          d[q] = (f32x4){a[0], a[1], b[0], b[1]};
        }
      }
      float* g = vnext + ((long)((X == 16 || X == 18) ? (tb & 63) : tb) * WNS + (cb * (WC / WK) + sl)) * A_STAGE + row * 4;
      // B^T d B on channel PAIRS (v_pk_*_f32: two channels per VALU instruction; no MFMA runs beside this): the
      // arithmetic of bt5 above, operation for operation (9 packed operations per five values)
      auto bt5p = [](f32x2 x0, f32x2 x1, f32x2 x2, f32x2 x3, f32x2 x4, f32x2* r) {      // bt5, two channels at a time
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
          if (X == 7 || X == 10) {
#pragma unroll
            for (int j = 0; j < 5; ++j) r[j] = tx[i * 5 + j];
          } else {
            bt5p(tx[i * 5 + 0], tx[i * 5 + 1], tx[i * 5 + 2], tx[i * 5 + 3], tx[i * 5 + 4], r);
          }
#pragma unroll
          for (int j = 0; j < 5; ++j) vv[i * 5 + j][h] = r[j];
        }
      }
#pragma unroll
      for (int xi = 0; xi < ((X == 7 || X == 10) ? 15 : WXI); ++xi) {      // (the 26th plane slot of an image is padding nobody multiplies with)
        const f32x2 p0 = vv[xi][0], p1 = vv[xi][1];
        f32x4 v4;
        if constexpr (SPLIT) {
          _Float16 ah[4], al[4];
          split_half(kSplitV * p0[0], ah[0], al[0]);
          split_half(kSplitV * p0[1], ah[1], al[1]);
          split_half(kSplitV * p1[0], ah[2], al[2]);
          split_half(kSplitV * p1[1], ah[3], al[3]);
          const h8 pk = {ah[0], ah[1], ah[2], ah[3], al[0], al[1], al[2], al[3]};
          v4 = __builtin_bit_cast(f32x4, pk);
        } else {
          v4 = (f32x4){p0[0], p0[1], p1[0], p1[1]};        // (already in the row's pair order: see the reads above)
        }
        f32x4* gp = reinterpret_cast<f32x4*>(g + (xi >> 1) * (WT * 8) + (xi & 1) * (WT * 4));
        if (X == 3) {
          if (v4[0] + v4[3] == 123.456f) *gp = v4;
          continue;
        }
        if (emit) __builtin_nontemporal_store(v4, gp);      // 1.9 GB per layer, read back a whole layer later: keep it out of L2
      }
    }
  }
  G4_STAMP(5);
}

template <int MODE, int X = 0, bool SPLIT = false, int NS = WNS>
__global__ __launch_bounds__(256, 1) void k_wino_gemm4(
    const float* __restrict__ vimg, const float* __restrict__ uimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ vnext, const int* __restrict__ d_count, int N, int T, int relu, int tb0, int tb1) {
  __shared__ __attribute__((aligned(256))) float lds[3 * STAGE];
  __shared__ int ptab[WT * 9];     // element offset of output point X in y / res, or -1 (off the board / dead row)
  const long Mt = (long)(*d_count) * (T * T);
  // workgroup -> (tile block, cout block): block b runs on XCD b % 8.  The four cout blocks of a tile block are four
  // CONSECUTIVE workgroups of one XCD, so its V slab (1.7 MB) comes out of HBM once and is served to the other three from
  // that XCD's L2.  Rounds 1-2 split a tile block over an XCD pair (two cout blocks each, so that the 3.3 MB of U an XCD
  // needs stay in its 4 MB L2): V then crossed HBM twice -- 4.0 of the layer's 7.2 GB.  With all of U (6.5 MB) wanted by
  // every XCD the U misses go to the Infinity Cache instead; measured -0.65 % per forward (A/B, same box) and V read once.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jb = bid >> 3;
  const int cb = jb & 3;
  const int tb = tb0 + xcd + 8 * (jb >> 2);      // this launch covers tile blocks [tb0, tb1) (launch_wino_gemm: part / parts)
  if (tb >= tb1 || (long)tb * wino_rows_per_block(T) >= Mt) return;
  wino_wg<X, SPLIT, NS, false>(lds, ptab, vimg, uimg, scale, shift, res, y, vnext, Mt, N, T, relu, tb, cb, MODE, (int)threadIdx.x);
}

// ------------------------------------------------------------------ the whole tower in one launch
//
// k_wino_tower: one persistent workgroup per CU runs wino_wg for every (layer, tile block, cout block) of the tower.
// Why: a layer is 4684 workgroups at 8192 positions of 9x9 -- 18.3 rounds of 256, and every per-layer launch pays for 19
// (3.7 %).  With whole-board tile blocks a tile block of layer l + 1 needs nothing but the SAME tile block of layer l
// (its four cout blocks), so the layers' (tile block) items form independent chains and one list of all of them,
// layer-major, dealt out round-robin, is uneven by at most one item per forward instead of one round per layer.
//
// Who runs what.  A workgroup reads the XCD it runs on from the hardware (XCC_ID) and takes a slot 0..31 on that XCD's
// roster: quad = slot / 4, cout block = slot % 4.  XCD x owns the tile blocks tb = x (mod 8) -- in every layer, so the
// V slab and the y tile a workgroup reads were written by workgroups of its own XCD and are in that XCD's L2 (or behind
// it): no cross-XCD coherence is involved, and the LDS-DMA loads carry sc1 so that they are served by L2 and never by
// a line the CU's own vector L1 kept from two layers ago.  Quad q of XCD x runs items q, q + 8, q + 16 ... of the list
// item = l * nbx + i  ->  (layer l, tile block x + 8 i); its four workgroups run the same item at the same time, which
// is what lets three of them take V from L2 (as four consecutive blocks of a per-layer launch do).
//
// Ordering.  Producer: every wave waits for its own stores (s_waitcnt vmcnt(0): acknowledged by L2), barrier, lane 0
// adds 1 to done[l][tb] (agent-scope atomic).  Consumer: lane 0 polls done[l-1][tb] (agent-scope load) until it is 4,
// barrier, go.  A dependency is an item ~18 places back in every quad's own sequence, so the poll normally passes at
// once; it gives up after kTowerSpinLimit polls and raises sched[kTowerErr] rather than hang the GPU.
typedef WinoTowerLayer TowerLayer;
static_assert(kWinoTowerErrWord == 8, "");
constexpr int kTowerErr = kWinoTowerErrWord, kTowerDone = 64;      // int offsets in sched: [0..7] roster, [8] error, [64..] done[l][tb]
constexpr int kTowerSpinLimit = 1 << 22;           // x (s_sleep + an L2 round trip, ~1 us): seconds

template <bool SPLIT>
__global__ __launch_bounds__(256, 1) void k_wino_tower(const TowerLayer* __restrict__ layers, int nl, int* __restrict__ sched,
                                                       int blocks_cap, const int* __restrict__ d_count, int N, int T, int order) {
  __shared__ __attribute__((aligned(256))) float lds[3 * STAGE];
  __shared__ int ptab[WT * 9];
  __shared__ int s_bc;
  const int tid = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int xcd = (int)(xcc & 7u);
  if (tid == 0) s_bc = atomicAdd(&sched[xcd], 1);
  __syncthreads();
  const int slot = s_bc;
  if (slot >= 32) {                                  // more than 32 workgroups on one XCD: another XCD is short of them
    if (tid == 0) atomicOr(&sched[kTowerErr], 1);
    return;
  }
  const int quad = slot >> 2, cb = slot & 3;
  const int RPB = wino_rows_per_block(T);
  const long Mt = (long)(*d_count) * (T * T);
  const int blocks = (int)((Mt + RPB - 1) / RPB);    // tile blocks with a live row
  const int nbx = blocks > xcd ? (blocks - xcd + 7) >> 3 : 0;
  const int total = nl * nbx;
  int* done = sched + kTowerDone;
#ifdef AGZ_TIMING_EXPERIMENTS
  long long tw_wait = 0, tw_body = 0, tw_fin = 0, tw_items = 0;
  const long long tw_t0 = wall_clock64();
#define TW_NOW() ((long long)wall_clock64())
#endif
  // order 0: the XCD's list layer-major (every quad on the same layer, a tile block's next layer ~18 items later);
  // order 1: chain-major -- a quad takes a tile block through ALL layers before its next one, so the V it reads was
  // written one item earlier (by itself) and is still in the Infinity Cache
  const int nitems = order == 0 ? (total > quad ? (total - quad + 7) >> 3 : 0) : (nbx > quad ? ((nbx - quad + 7) >> 3) * nl : 0);
  for (int k = 0; k < nitems; ++k) {
#ifdef AGZ_TIMING_EXPERIMENTS
    const long long tw_a = TW_NOW();
#endif
    int l, tb;
    if (order == 0) {
      const int item = quad + 8 * k;
      l = item / nbx;
      tb = xcd + 8 * (item - l * nbx);
    } else {
      const int j = k / nl;
      l = k - j * nl;
      tb = xcd + 8 * (quad + 8 * j);
    }
    if (tid == 0) s_bc = 0;
    if (l > 0 && tid == 0) {
      const int* flag = done + (size_t)(l - 1) * blocks_cap + tb;
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4) {
        __builtin_amdgcn_s_sleep(8);
        // give up after kTowerSpinLimit polls -- or as soon as anybody else has: one stuck chain must not cost every
        // waiter its own time-out
        if (++spins > kTowerSpinLimit || ((spins & 1023) == 0 && __hip_atomic_load(&sched[kTowerErr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          atomicOr(&sched[kTowerErr], 2);
          s_bc = 1;
          break;
        }
      }
    }
    __syncthreads();      // the dependency is in; and every wave has left the previous item's epilogue (image, ptab)
    if (s_bc) return;     // (uniform: the forward's output is garbage and Net::forward reports the error word)
#ifdef AGZ_TIMING_EXPERIMENTS
    const long long tw_b = TW_NOW();
#endif
    const TowerLayer L = layers[l];
    // the thread id through an opaque asm: everything the body derives from it (lane offsets, LDS addresses, tables) is
    // recomputed per item -- a handful of VALU instructions -- instead of being hoisted out of this loop and spilled
    // (41 VGPRs in scratch when left to the compiler: the body owns all 512 registers)
    int t = tid;
    asm volatile("" : "+v"(t));
    wino_wg<0, SPLIT, WNS, true>(lds, ptab, L.v, L.u, L.scale, L.shift, L.res, L.y, L.vnext, Mt, N, T, L.relu, tb, cb, L.mode, t);
#ifdef AGZ_TIMING_EXPERIMENTS
    const long long tw_c = TW_NOW();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have reached L2
    __syncthreads();
    if (tid == 0) {
      // ... and the quad leaves an item together: its four workgroups share the V slab through L2 only while they
      // read the same K-loop stage within a few microseconds of each other (the XCD's L2 turns over every ~2.4 stages)
      int* flag = done + (size_t)l * blocks_cap + tb;
      if (__hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 3) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4 && ++spins < kTowerSpinLimit) __builtin_amdgcn_s_sleep(2);
      }
    }
#ifdef AGZ_TIMING_EXPERIMENTS
    const long long tw_d = TW_NOW();
    if (tid == 0 && slot < 4 && tw_items < 384) {      // the first quad of every XCD: every item's {start, end, layer, tile block}
      unsigned long long* r = g4_trace[256 + (xcd * 4 + slot) * 384 + tw_items];
      r[0] = tw_b; r[1] = tw_c; r[2] = l; r[3] = tb;
    }
    tw_wait += tw_b - tw_a; tw_body += tw_c - tw_b; tw_fin += tw_d - tw_c; ++tw_items;
#endif
  }
#ifdef AGZ_TIMING_EXPERIMENTS
  if (tid == 0) {      // per persistent workgroup: {xcd | slot << 8, items, wait, body, finish, start, end} (10 ns ticks)
    unsigned long long* r = g4_trace[blockIdx.x];
    r[0] = (unsigned long long)xcd | ((unsigned long long)slot << 8);
    r[1] = tw_items; r[2] = tw_wait; r[3] = tw_body; r[4] = tw_fin; r[5] = tw_t0; r[6] = TW_NOW();
  }
#endif
}

// one workgroup per CU: which XCD does block b run on?  (the tower kernel does not depend on the answer being b % 8,
// only on every XCD getting 32 of 256 resident workgroups: checked once per process, on the device it will run on)
__global__ void k_xcd_census(int* out) {
  __shared__ float hog[150 * 1024 / 4];      // one workgroup per CU, like the tower kernel
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) {
    hog[0] = 1.f;
    atomicAdd(&out[xcc & 7u], 1);
    const long long t0 = wall_clock64();      // stay resident until the whole grid is (100 us)
    while ((long long)wall_clock64() - t0 < 10000) __builtin_amdgcn_s_sleep(32);
    if (hog[0] == 2.f) out[8] = 1;
  }
}

// ------------------------------------------------------------------ host side

// Flux [kw,kh,cin,cout] column-major -> stage images U[cout block][stage][plane][cout 64][8] with
// U_xi = G k G^T computed in float64.  k is the CORRELATION kernel: NNlib's conv is a true
// convolution, so tap (a', b') reading x[i + a' - 1, j + b' - 1] carries w[2 - a', 2 - b'].
// One (cout, cin) pair = one call of wino_pack_pair: the host loop below (test reference, agz_debug_pack_diff) and the
// device kernel (the product: weights never leave the GPU between a training step / a broadcast and the next forward)
// run the same source with FP contraction off, and produce the same bits.
template <bool SPLIT>
__host__ __device__ inline void wino_pack_pair(const float* w, int cin, int o, int ci, int ns, float* out) {
#pragma clang fp contract(off)
  constexpr double G[5][3] = {{0.5, 0.0, 0.0}, {0.5, 0.5, 0.5}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                              {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
  double k[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) k[a][b] = w[(2 - a) + 3 * ((2 - b) + 3 * (ci + (size_t)cin * o))];
  const int cb = o / WC, ol = o % WC, st = ci / WK, cl = ci % WK;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      double u = 0.0;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) u += G[i][a] * k[a][b] * G[j][b];
      const int xi = i * 5 + j;
      if (SPLIT) {          // u' = 2^10 u as (hi, lo) halves, rows laid out by wino_v_off (block h = 0: hi, 1: lo)
        _Float16* o16 = reinterpret_cast<_Float16*>(out);
        _Float16 hi, lo;
        split_half((float)(u * (double)kSplitU), hi, lo);
        const size_t base = ((size_t)cb * ns + st) * B_STAGE;       // floats
        o16[2 * (base + wino_v_off(xi, ol, 0)) + cl] = hi;
        o16[2 * (base + wino_v_off(xi, ol, 1)) + cl] = lo;
      } else {
        const int pos = 2 * wino_pair_pos(ol, xi, cl >> 1) + (cl & 1);
        out[(((size_t)cb * ns + st) * WPL + (xi >> 1)) * WC * 8 + (size_t)ol * 8 + pos] = (float)u;
      }
    }
}

void wino_pack_weights(const ConvHost& c, float* out, int ns) {
  std::memset(out, 0, sizeof(float) * wino_weight_floats(ns));
  for (int o = 0; o < c.cout; ++o)
    for (int ci = 0; ci < c.cin; ++ci) wino_pack_pair<false>(c.w.data(), c.cin, o, ci, ns, out);
}
void wino_pack_weights_split(const ConvHost& c, float* out, int ns) {
  std::memset(out, 0, sizeof(float) * wino_weight_floats(ns));
  for (int o = 0; o < c.cout; ++o)
    for (int ci = 0; ci < c.cin; ++ci) wino_pack_pair<true>(c.w.data(), c.cin, o, ci, ns, out);
}

// the same images from Flux-layout weights that are already on the device (Net's master copy): `layers` consecutive
// [3][3][cin][256] tensors, `wstride` floats apart, into `layers` images.  One thread per (layer, cout, cin).
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_wino_pack(const float* __restrict__ w, long wstride, int cin, int layers, int ns,
                                                   float* __restrict__ out, long per) {
  const long n = (long)layers * kC * cin;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
    const int ci = (int)(t % cin), o = (int)((t / cin) % kC), l = (int)(t / ((long)cin * kC));
    wino_pack_pair<SPLIT>(w + l * wstride, cin, o, ci, ns, out + l * per);
  }
}
void launch_wino_pack(const float* d_w, long wstride, int cin, int layers, float* d_out, int ns, bool split, hipStream_t s) {
  const long per = (long)wino_weight_floats(ns);
  AGZ_HIP(hipMemsetAsync(d_out, 0, sizeof(float) * (size_t)per * layers, s));      // (the stem's 17 channels fill 5 of 8 stages' rows)
  const int grid = (int)std::min<long>(((long)layers * kC * cin + 255) / 256, 65536);
  if (split) hipLaunchKernelGGL(k_wino_pack<true>, dim3(grid), dim3(256), 0, s, d_w, wstride, cin, layers, ns, d_out, per);
  else hipLaunchKernelGGL(k_wino_pack<false>, dim3(grid), dim3(256), 0, s, d_w, wstride, cin, layers, ns, d_out, per);
}
float wino_split_descale() { return 1.f / (kSplitV * kSplitU); }

size_t wino_weight_floats(int ns) { return (size_t)(kC / WC) * ns * B_STAGE; }
static long wino_blocks(int bcap, int T) {
  const long rpb = wino_rows_per_block(T);
  return ((long)bcap * T * T + rpb - 1) / rpb;
}
size_t wino_v_floats(int bcap, int T) { return (size_t)wino_blocks(bcap, T) * WNS * A_STAGE; }
bool wino_fusable(int N) { return wino_whole_boards((N + 2) / 3); }

void launch_wino_in(const float* x, float* vimg, const int* d_count, int bcap, int N, bool split, hipStream_t s, int ns,
                    bool fixup) {
  const int T = (N + 2) / 3;
  const int blocks = (int)wino_blocks(bcap, T);
  if (fixup) {      // the rows the previous GEMM's epilogue left out (dense tile blocks; a tower layer's 64 stages)
    AGZ_REQUIRE(ns == kWinoStages && !wino_whole_boards(T), AGZ_BAD_ARGUMENT, "fix-up transform: dense tile blocks, tower layers");
    if (split) hipLaunchKernelGGL((k_wino_in<32, true, true, WNS, true>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
    else hipLaunchKernelGGL((k_wino_in<32, true, false, WNS, true>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
    return;
  }
  if (ns == kWinoStemStages) {
    if (split) hipLaunchKernelGGL((k_wino_in<32, true, true, kWinoStemStages>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
    else hipLaunchKernelGGL((k_wino_in<32, true, false, kWinoStemStages>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
    return;
  }
  if (split) hipLaunchKernelGGL((k_wino_in<32, true, true>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
  else hipLaunchKernelGGL((k_wino_in<32, true, false>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
}

// y == nullptr: the activations are not needed in HBM (only their transform is); vnext == nullptr: no next
// Winograd layer (or a board size whose tile blocks do not hold whole boards: wino_fusable(N) is false)
void launch_wino_gemm(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                      float* y, float* vnext, const int* d_count, int bcap, int N, int relu, bool split, hipStream_t s, int ns, int part,
                      int parts) {
  const int T = (N + 2) / 3;
  // part / parts: the part-th of `parts` equal ranges of tile blocks.  With whole-board blocks a range's layers depend on
  // nothing outside it, so ranges can run as independent layer chains on different streams (Net::forward)
  const int all_blocks = (int)wino_blocks(bcap, T);
  AGZ_REQUIRE(parts >= 1 && part >= 0 && part < parts && (parts == 1 || wino_whole_boards(T)), AGZ_BAD_ARGUMENT,
              "tile-block range %d of %d", part, parts);
  const int per_part = (all_blocks + parts - 1) / parts;
  const int tb0 = std::min(all_blocks, part * per_part), tb1 = part + 1 == parts ? all_blocks : std::min(all_blocks, tb0 + per_part);
  if (tb1 <= tb0) return;
  const int blocks = tb1 - tb0;
  const int per_xcd = 4 * ((blocks + 7) / 8);   // see the placement comment in k_wino_gemm4
  const dim3 grid(8 * per_xcd), block(256);
  AGZ_REQUIRE((long)(all_blocks + 1) * WT < (1L << 31) && (long)bcap * N * N * kC * 4 < (1L << 32), AGZ_BAD_ARGUMENT,
              "batch of %d positions at %dx%d: tile index / activation byte offset exceeds 32 bits", bcap, N, N);
  if (ns == kWinoStemStages) {                   // the stem: its output is always wanted in HBM (block 0's residual)
    constexpr int S = kWinoStemStages;
    if (split) {
      if (vnext) hipLaunchKernelGGL((k_wino_gemm4<3, 0, true, S>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
      else hipLaunchKernelGGL((k_wino_gemm4<1, 0, true, S>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
    } else {
      if (vnext) hipLaunchKernelGGL((k_wino_gemm4<3, 0, false, S>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
      else hipLaunchKernelGGL((k_wino_gemm4<1, 0, false, S>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
    }
    return;
  }
#ifdef AGZ_TIMING_EXPERIMENTS
  static const int xp = getenv("AGZ_WINO_X") ? atoi(getenv("AGZ_WINO_X")) : 0;
  static int traced = 0;
  if (getenv("AGZ_WINO_TRACE") && y && vnext && res && !split && ++traced == 3) {      // third steady-state layer launch
    hipLaunchKernelGGL((k_wino_gemm4<3>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
    (void)hipStreamSynchronize(s);
    static unsigned long long host[16384][8];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4_trace), sizeof(host));
    if (FILE* f = fopen(getenv("AGZ_WINO_TRACE"), "wb")) {
      fwrite(host, 1, sizeof(host), f);
      fclose(f);
    }
    return;
  }
  if (xp && y && vnext && res && split) {
    auto kern = xp == 1 ? k_wino_gemm4<3, 1, true> : xp == 2 ? k_wino_gemm4<3, 2, true> : xp == 3 ? k_wino_gemm4<3, 3, true>
              : xp == 5 ? k_wino_gemm4<3, 5, true> : k_wino_gemm4<3, 0, true>;
    hipLaunchKernelGGL(kern, grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
    return;
  }
  if (xp && y && vnext && res && !split) {
    auto kern = xp == 1 ? k_wino_gemm4<3, 1> : xp == 2 ? k_wino_gemm4<3, 2> : xp == 3 ? k_wino_gemm4<3, 3>
              : xp == 4 ? k_wino_gemm4<3, 4> : xp == 5 ? k_wino_gemm4<3, 5> : xp == 6 ? k_wino_gemm4<3, 6>
              : xp == 16 ? k_wino_gemm4<3, 16> : xp == 17 ? k_wino_gemm4<3, 17> : xp == 18 ? k_wino_gemm4<3, 18> : xp == 13 ? k_wino_gemm4<3, 13> : xp == 14 ? k_wino_gemm4<3, 14> : xp == 15 ? k_wino_gemm4<3, 15> : xp == 7 ? k_wino_gemm4<3, 7> : xp == 8 ? k_wino_gemm4<3, 8> : xp == 9 ? k_wino_gemm4<3, 9> : xp == 10 ? k_wino_gemm4<3, 10>
              : xp == 11 ? k_wino_gemm4<1, 0> : xp == 12 ? k_wino_gemm4<2, 0> : k_wino_gemm4<3, 0>;
    hipLaunchKernelGGL(kern, grid, block, 0, s, vimg, uimg, scale, shift, xp == 12 ? nullptr : res, y, vnext, d_count, N, T, relu, tb0, tb1);
    return;
  }
#endif
  if (split) {
    if (y && vnext)
      hipLaunchKernelGGL((k_wino_gemm4<3, 0, true>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
    else if (vnext)
      hipLaunchKernelGGL((k_wino_gemm4<2, 0, true>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
    else
      hipLaunchKernelGGL((k_wino_gemm4<1, 0, true>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
    return;
  }
  if (y && vnext)
    hipLaunchKernelGGL((k_wino_gemm4<3>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
  else if (vnext)
    hipLaunchKernelGGL((k_wino_gemm4<2>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
  else
    hipLaunchKernelGGL((k_wino_gemm4<1>), grid, block, 0, s, vimg, uimg, scale, shift, res, y, vnext, d_count, N, T, relu, tb0, tb1);
}

size_t wino_tower_sched_ints(int layers, int bcap, int N) {
  const int T = (N + 2) / 3;
  return (size_t)kTowerDone + (size_t)layers * wino_blocks(bcap, T);
}

// 256 resident workgroups, 32 on every XCD?  (cached per device)
bool wino_tower_supported(hipStream_t s) {
  static int cached[64];      // 0 unknown, 1 yes, 2 no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (cached[dev]) return cached[dev] == 1;
  hipDeviceProp_t prop;
  bool ok = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount == 256;
  if (ok) {
    int* d = nullptr;
    int h[9] = {0};
    ok = hipMalloc((void**)&d, sizeof(h)) == hipSuccess;
    if (ok) {
      (void)hipMemsetAsync(d, 0, sizeof(h), s);
      hipLaunchKernelGGL(k_xcd_census, dim3(256), dim3(64), 0, s, d);
      ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
      (void)hipFree(d);
      for (int x = 0; ok && x < 8; ++x) ok = h[x] == 32;
    }
  }
  cached[dev] = ok ? 1 : 2;
  return ok;
}

// the tower's layers (device array of TowerLayer, see agz_nn.hip) in one launch; sched: wino_tower_sched_ints() ints
void launch_wino_tower(const void* d_layers, int layers, int* d_sched, const int* d_count, int bcap, int N, bool split, hipStream_t s) {
  const int T = (N + 2) / 3;
  AGZ_REQUIRE(wino_whole_boards(T), AGZ_BAD_ARGUMENT, "the persistent tower kernel needs whole-board tile blocks");
  AGZ_REQUIRE((long)(wino_blocks(bcap, T) + 1) * WT < (1L << 31) && (long)bcap * N * N * kC * 4 < (1L << 32), AGZ_BAD_ARGUMENT,
              "batch of %d positions at %dx%d: tile index / activation byte offset exceeds 32 bits", bcap, N, N);
  const int blocks_cap = (int)wino_blocks(bcap, T);
  (void)hipMemsetAsync(d_sched, 0, sizeof(int) * wino_tower_sched_ints(layers, bcap, N), s);
#ifdef AGZ_TIMING_EXPERIMENTS
  static const int order = getenv("AGZ_TOWER_ORDER") ? atoi(getenv("AGZ_TOWER_ORDER")) : 0;
#else
  constexpr int order = 0;
#endif
  if (split)
    hipLaunchKernelGGL((k_wino_tower<true>), dim3(256), dim3(256), 0, s, (const TowerLayer*)d_layers, layers, d_sched, blocks_cap, d_count, N, T, order);
  else
    hipLaunchKernelGGL((k_wino_tower<false>), dim3(256), dim3(256), 0, s, (const TowerLayer*)d_layers, layers, d_sched, blocks_cap, d_count, N, T, order);
#ifdef AGZ_TIMING_EXPERIMENTS
  static int traced = 0;
  if (getenv("AGZ_TOWER_TRACE") && ++traced == 3) {
    (void)hipStreamSynchronize(s);
    static unsigned long long host[256 + 32 * 384][8];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4_trace), sizeof(host));
    if (FILE* f = fopen(getenv("AGZ_TOWER_TRACE"), "wb")) {
      fwrite(host, 1, sizeof(host), f);
      fclose(f);
    }
  }
#endif
}

}  // namespace agz
