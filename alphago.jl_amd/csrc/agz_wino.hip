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
// Two kernels per layer:
//   k_wino_in    X[M][256] -> V, the 25 transformed planes, written directly in the LDS image
//                order of the GEMM stages (HBM-bound: reads 1 KB, writes 2.9 KB per board point)
//   k_wino_gemm  25 GEMMs  M_xi[tile][cout] = sum_cin V_xi[tile][cin] * U_xi[cin][cout]  on
//                v_mfma_f32_32x32x2_f32, 64 tiles x 64 couts x 25 planes per workgroup: 410 KB of
//                accumulators, i.e. the CU's whole 512 KB register file is the tile.  12 waves =
//                4 quadrants (32 x 32) x 3 plane groups (9/8/8 planes, 144 accumulator VGPRs each).
//                The inverse transform A^T M A is linear in the planes: every wave reduces its own
//                planes to a partial 3x3 output, the partials meet through the (by then idle) stage
//                buffers in a FIXED order, and bias+BatchNorm affine, residual add and ReLU follow in
//                registers: M is never written to memory.
// Stage = 4 input channels x {64 tiles + 64 couts} x 25 planes = 52 KB, triple-buffered in LDS and
// filled by direct global->LDS DMA (global_load_lds_dwordx4), which is why V and U are stored in
// HBM as ready-made, bank-swizzled stage images.  64x64 per workgroup gives 16 flop per DMA byte;
// the L2->LDS path (~11 TB/s measured with the MFMAs compiled out) and the MFMA pipe are co-critical.
#include "agz_nn.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace agz {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WT = 64;            // tiles per workgroup
constexpr int WC = 64;            // output channels per workgroup
constexpr int WK = 4;             // input channels per stage
constexpr int WNS = kC / WK;      // 64 stages
constexpr int WXI = 25;
constexpr int WPL = 13;           // LDS row planes: a row holds the 4 channels of TWO transform planes
constexpr int A_STAGE = WPL * WT * 8;    // floats (26,624 B)
constexpr int B_STAGE = WPL * WC * 8;
constexpr int STAGE = A_STAGE + B_STAGE;  // 13,312 floats = 52 KB

// A row of a stage image is 8 dwords = [plane 2q: 4 channels | plane 2q+1: 4 channels], i.e. four
// 2-dword "pairs": logical pair = 2*(xi & 1) + h, h = which half of the 4 channels.  The 32 rows a
// wave reads with one ds_read_b64 start on only 8 distinct banks (8 dwords * r mod 64), so the pairs
// are rotated by f(row): rows r, r+8, r+16, r+24 (same bank mod 64, the 32-lane groups of
// ds_read_b64) and rows r, r+4, r+8, r+12 (same bank mod 32, the 16-lane groups of the fused
// ds_read2*_b64 forms hipcc may emit) all get distinct pair slots => conflict-free either way.
__host__ __device__ __forceinline__ int wino_rot(int row) { return ((row >> 2) + (row >> 4)) & 3; }
__host__ __device__ __forceinline__ int wino_pair_pos(int row, int xi, int h) {
  return (2 * (xi & 1) + h + wino_rot(row)) & 3;
}

// ------------------------------------------------------------------ input transform

__device__ __forceinline__ void bt5(float x0, float x1, float x2, float x3, float x4, float* r) {
  r[0] = 2.f * x0 - x1 - 2.f * x2 + x3;
  r[1] = 2.f * x1 + x2 - x3;
  r[2] = -2.f * x1 + 3.f * x2 - x3;
  r[3] = x3 - x1;
  r[4] = 2.f * x1 - x2 - 2.f * x3 + x4;
}

// grid = 2 x tile blocks (32-tile halves); 256 threads = 32 tiles x 8 channel pairs (= 4 consecutive
// stages x 2 halves): the eight lanes of a tile read one 64-byte run of every patch point.  A pass
// covers 16 channels = 4 stages; the transformed values go to an LDS copy of this half-block's part
// of the four stage images (4 x 13 chunks of 32 rows x 32 B = 1 KB) and leave for HBM as whole
// 1 KB runs, 16 B per lane -- written straight from registers they were 16-byte fragments spread
// over four stage images, and the kernel sat at 3.8 TB/s.  Stage images are skewed by {0,4,16,20}
// dwords in LDS: a ds_write_b64 is served 16 lanes (2 tiles x 8 lanes) at a time over 32 banks, and
// with that skew the 16 lanes cover all 32 banks exactly once (PMC: SQ_LDS_BANK_CONFLICT 0; the
// {0,4,32,36} skew that would suit a 64-bank / 32-lane model measured 40 % conflicted cycles).
template <int TPB, bool NT>   // tiles per workgroup: 32 (8 lanes = 64 B per patch point) or 16 (16 lanes = one 128 B line)
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ x, float* __restrict__ vimg,
                                                  const int* __restrict__ d_count, int N, int T) {
  constexpr int LPT = 256 / TPB;             // lanes per tile
  constexpr int SP = LPT / 2;                // stages per pass
  constexpr int CH = TPB * 8;                // dwords per chunk (TPB rows of one row pair)
  constexpr int IMG = 13 * CH + 64;          // LDS stride between the stage images of a pass
  constexpr int CPR = 256 / (TPB * 2);       // chunks copied out per round
  __shared__ __attribute__((aligned(16))) float img[SP * IMG];
  auto skew = [](int sl) { return (sl & 1) * 4 + (sl >> 1) * 16; };
  const int P = N * N, TT = T * T;
  const long Mt = (long)(*d_count) * TT;
  constexpr int PARTS = WT / TPB;
  const int tb = blockIdx.x / PARTS, part = blockIdx.x % PARTS;
  if ((long)tb * WT + part * TPB >= Mt) return;
  const int hs = threadIdx.x % LPT, h = hs & 1, sl = hs >> 1;
  const int tl = threadIdx.x / LPT;                 // tile within this part
  const int row = part * TPB + tl;                  // row of the 64-row stage image
  const long tile = (long)tb * WT + row;
  const bool live = tile < Mt;
  const int b = live ? (int)(tile / TT) : 0, t = live ? (int)(tile % TT) : 0;
  const int ti = t / T, tj = t % T;
  int off[25];                                      // element offsets < 2^31 (8192 x 361 x 256 = 7.6e8)
#pragma unroll
  for (int u = 0; u < 5; ++u)
#pragma unroll
    for (int v = 0; v < 5; ++v) {
      const int pi = 3 * ti - 1 + u, pj = 3 * tj - 1 + v;
      const bool ok = live && pi >= 0 && pi < N && pj >= 0 && pj < N;
      off[u * 5 + v] = ok ? (b * P + pi + N * pj) * kC : -1;
    }
  const int rot = wino_rot(row);
  float* mine = img + sl * IMG + skew(sl) + tl * 8;
  // the unused plane slot 25 (second half of pair 12) is copied out with the rest: keep it finite
  *reinterpret_cast<float2*>(mine + 12 * CH + 2 * ((2 + h + rot) & 3)) = make_float2(0.f, 0.f);
  float* gdst = vimg + (long)tb * WNS * A_STAGE + part * CH;
  const int cq = threadIdx.x / (TPB * 2), cl = threadIdx.x % (TPB * 2);
  for (int sg = 0; sg < WNS / SP; ++sg) {
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
        const int xi = i * 5 + j;
        *reinterpret_cast<float2*>(mine + (xi >> 1) * CH + 2 * ((2 * (xi & 1) + h + rot) & 3)) = make_float2(rx[j], ry[j]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 13; ++r) {
      const int c = r * CPR + cq, s4 = c / 13, q = c % 13;     // chunk c = (stage s4 of this pass, row pair q)
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 v = *reinterpret_cast<const f32x4*>(img + s4 * IMG + skew(s4) + q * CH + cl * 4);
      f32x4* gp = reinterpret_cast<f32x4*>(gdst + (long)(sg * SP + s4) * A_STAGE + q * (WT * 8) + cl * 4);
      if (NT) __builtin_nontemporal_store(v, gp);     // V is 2.8x the activations and is read back much later
      else *gp = v;
    }
  }
}

// ------------------------------------------------------------------ GEMM + output transform

// 16 bytes per lane straight from global memory into LDS (wave-uniform LDS base in M0 + lane*16).
// Issued through inline asm on purpose: hipcc cannot prove that the DMA target (the OTHER stage
// buffer) does not alias the ds_reads of the current stage and would put an s_waitcnt vmcnt(0) in
// front of them, serialising load and compute (measured: 38 % MFMA utilisation).  An asm statement
// is outside its vmcnt book-keeping, so the wait is placed by hand, once per stage, right before
// the barrier that hands the buffer over.
__device__ __forceinline__ void glds16(const float* g, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);   // make the SGPR operand provable
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_byte_addr)
      : "memory");
}

// Inverse-transform coefficients A^T (3x5)
__device__ __forceinline__ constexpr float wino_at(int o, int i) {
  return o == 0 ? (i < 4 ? 1.f : 0.f)
       : o == 1 ? (i == 1 ? 1.f : i == 2 ? -1.f : i == 3 ? 2.f : 0.f)
                : (i == 1 ? 1.f : i == 2 ? 1.f : i == 3 ? 4.f : i == 4 ? 1.f : 0.f);
}

// grid = tile blocks x (256 / WC); 768 threads = 12 waves = 3 per SIMD.
//   sp = wave & 3  -> which 32-tile x 32-cout quadrant of the 64 x 64 workgroup tile
//   pg = wave >> 2 -> which third of the 25 transform planes (9 / 8 / 8) this wave accumulates
// A wave therefore keeps 9 x 16 = 144 accumulator VGPRs and runs v_mfma_f32_32x32x2_f32 fed by two
// ds_read_b64 per plane; the three plane groups of a quadrant land on the same SIMD, so every SIMD
// owns 25 planes x 2 MFMAs x 64 cycles = 3200 MFMA-cycles per 4-channel stage.  The inverse
// transform is linear in the planes: each wave reduces its own planes to a partial 3x3 output, the
// partials of groups 1 and 2 cross to group 0 through the (by then idle) stage buffers in LDS.
template <int DBG, int PG>
__device__ __forceinline__ void wino_gemm_body(float (*lds)[STAGE], const float* __restrict__ asrc,
                                               const float* __restrict__ bsrc, const float* __restrict__ scale,
                                               const float* __restrict__ shift, const float* __restrict__ res,
                                               float* __restrict__ y, long Mt, int N, int T, int relu, int tb,
                                               int cb, int wave, int lane) {
  constexpr int X0 = PG == 0 ? 0 : PG == 1 ? 9 : 17;     // first plane of this group
  constexpr int NX = PG == 0 ? 9 : 8;
  const int P = N * N, TT = T * T;
  const int sp = wave & 3, wm = sp & 1, wn = sp >> 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0][0];

  // a stage is 52 chunks of 1 KB (64 lanes x 16 B): chunks 0..25 from V, 26..51 from U
  auto issue = [&](int st, int buf) {
    const float* a = asrc + (long)st * A_STAGE;
    const float* b = bsrc + (long)st * B_STAGE;
    for (int c = wave; c < 52; c += 12) {
      if ((DBG == 7 && c >= 26) || (DBG == 8 && c < 26)) continue;     // timing: only the V / only the U stream
      const float* g = c < 26 ? a + c * 256 : b + (c - 26) * 256;
      glds16(g + lane * 4, lds0 + (unsigned)(buf * STAGE + c * 256) * 4u);
    }
  };

  // the j-th chunk of this wave (c = wave + 12 j).  The five DMA instructions of a stage are spread
  // between the MFMAs of the K loop: issued in one block at the top of a stage -- by all twelve
  // waves at once, straight after the barrier -- their ~90 scalar/VALU/VMEM instructions per wave
  // kept every wave of a SIMD off the MFMA pipe at the same time (measured: +0.6 ms per layer even
  // with the loads never waited for and the MFMAs fed from constants).
  auto dma = [&](int st, int buf, int j) {
    const int c = wave + 12 * j;
    if (c >= 52) return;
    if ((DBG == 7 && c >= 26) || (DBG == 8 && c < 26)) return;       // timing: only the V / only the U stream
    const float* g = c < 26 ? asrc + (long)st * A_STAGE + c * 256 : bsrc + (long)st * B_STAGE + (c - 26) * 256;
    glds16(g + lane * 4, lds0 + (unsigned)(buf * STAGE + c * 256) * 4u);
  };

  f32x16 acc[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int arow = wm * 32 + l31, brow = wn * 32 + l31;
  const int arot = wino_rot(arow), brot = wino_rot(brow);
  const int abase = arow * 8, bbase = A_STAGE + brow * 8;

  // Three stage buffers, two stages of DMA in flight: the L2->LDS path is latency-bound at one
  // stage in flight (measured ~35 GB/s per CU), so the wait before a barrier is a COUNTED vmcnt
  // that leaves the newest stage's requests (5 per wave for waves 0-3, 4 for the rest) outstanding.
  const bool five = wave < 4;
  issue(0, 0);
  if (DBG != 1 && DBG != 3 && DBG != 4 && DBG != 5) issue(1, 1);
  if (DBG == 1 || DBG == 3 || DBG == 4 || DBG == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  // Software-pipelined across the barrier.  Reading a stage's operands only after the barrier that
  // publishes it left the LDS latency -- ~600 cycles under DMA write traffic -- exposed once per
  // 3200-cycle stage (0.6 ms of 2.6 per layer).  So the LAST planes of every stage (NH of them) are
  // "held back": their operands are read before the barrier, their MFMAs are issued after it, right
  // behind the first reads of the next stage, whose latency they cover.  The 8-plane groups hold two
  // planes (8 VGPRs); the 9-plane group has no register to spare (144 accumulators) and holds none --
  // its reads hide behind the held MFMAs of the two other waves of its SIMD.
  constexpr int NH = PG == 0 ? 0 : 2, NS = NX - NH;
  float2 aH[NH + 1], bH[NH + 1];
  auto load = [&](const float* L, int k, float2& a, float2& b) {
    const int xi = X0 + k;
    const int lp = 2 * (xi & 1) + hi;
    a = *reinterpret_cast<const float2*>(L + abase + (xi >> 1) * WT * 8 + 2 * ((lp + arot) & 3));
    b = *reinterpret_cast<const float2*>(L + bbase + (xi >> 1) * WC * 8 + 2 * ((lp + brot) & 3));
    if (DBG == 4 || DBG == 5 || DBG == 9) { a = make_float2(1.f, 1.f); b = make_float2(2.f, (float)lane); }   // timing: no LDS reads
  };
  auto mma = [&](int k, const float2& a, const float2& b) {
    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[k], 0, 0, 0);
    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[k], 0, 0, 0);
  };
  int buf = 0;
  for (int st = 0; st < WNS; ++st) {
    const int nbuf = buf == 0 ? 2 : buf - 1;            // (st + 2) % 3
    const bool more = st + 2 < WNS && DBG != 1 && DBG != 3 && DBG != 4 && DBG != 5;
    const float* L = lds[buf];
    if (DBG == 2) {
      if (more) issue(st + 2, nbuf);
    } else {
      // operands are read one plane ahead (two register pairs, alternating); the DMA instruction of
      // a slot sits between that read and the MFMAs it will feed
      constexpr int LA = 1;   // planes read ahead of the MFMAs (register ring of LA + 1; 2 was measured slower)
      float2 ra[LA + 1], rb[LA + 1];
#pragma unroll
      for (int k = 0; k < LA; ++k) load(L, k, ra[k], rb[k]);
      if (st > 0) {
#pragma unroll
        for (int k = 0; k < NH; ++k) mma(NS + k, aH[k], bH[k]);     // the previous stage's held planes
      }
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        if (k + LA < NS) load(L, k + LA, ra[(k + LA) % (LA + 1)], rb[(k + LA) % (LA + 1)]);
        if (k < 5) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) dma(st + 2, nbuf, k);
          __builtin_amdgcn_sched_barrier(0);
        }
        mma(k, ra[k % (LA + 1)], rb[k % (LA + 1)]);
      }
#pragma unroll
      for (int k = 0; k < NH; ++k) load(L, NS + k, aH[k], bH[k]);
    }
    // this wave's share of stage st+1 has landed (stage st+2 may still be in flight) ...  (hipcc puts
    // an s_waitcnt lgkmcnt(0) in front of the barrier itself: every LDS read of stage st has
    // returned before any wave can overwrite that buffer with the DMA of stage st+3)
    if (DBG == 6 || DBG == 7 || DBG == 8 || DBG == 9) { if (st == WNS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // timing: never wait for the DMA
    else if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (DBG != 3) __syncthreads();                      // ... and so has everybody else's
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (DBG != 2) {
#pragma unroll
    for (int k = 0; k < NH; ++k) mma(NS + k, aH[k], bH[k]);
  }

  // epilogue.  C/D map of the 32x32 MFMA: col (cout) = lane & 31, row (tile) = (e&3) + 8*(e>>2) + 4*(lane>>5),
  // so the planes of one (tile, cout) pair sit in the same lane and register index e of the three
  // plane-group waves of a quadrant.  Each wave reduces ITS planes to a partial 3x3 output per e;
  // the 16 register indices are owned 6 / 5 / 5 by the three groups, partials for foreign e's cross
  // through LDS (the stage buffers are idle now), and every wave finalises its own e's -- affine,
  // residual, ReLU, store -- so all 12 waves share the memory traffic.  Three rounds of 2+2+2 e's
  // keep the exchange at 110 KB.
  if (DBG == 5) {   // timing: K loop only -- keep the accumulators alive, skip the epilogue
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < NX; ++k) keep += acc[k][0] + acc[k][15];
    if (keep == 123.456f) y[0] = keep;
    return;
  }
  float* xch = &lds[0][0];   // [owner pg][source slot 0..1][sp][ei 0..1][9][64 lanes]
  const int co = cb * WC + wn * 32 + l31;
  const float sc = scale[co], sh = shift[co];
  auto partial = [&](int e, float* out) {   // A^T M A restricted to this wave's planes
#pragma unroll
    for (int oi = 0; oi < 3; ++oi)
#pragma unroll
      for (int oj = 0; oj < 3; ++oj) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
          const int xi = X0 + k, i = xi / 5, j = xi % 5;
          const float c = wino_at(oi, i) * wino_at(oj, j);
          if (c != 0.f) v += c * acc[k][e];
        }
        out[oi * 3 + oj] = v;
      }
  };
#pragma unroll
  for (int rd = 0; rd < 3; ++rd) {
    // e's of this round: owner 0 -> {2rd, 2rd+1}; owner 1 -> {6+2rd, 7+2rd} (rd<2) or {10}; owner 2 -> {11+2rd, 12+2rd} or {15}
    float own[2][9];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int ne = (q == 0 || rd < 2) ? 2 : 1;
#pragma unroll
      for (int ei = 0; ei < 2; ++ei) {
        if (ei >= ne) continue;
        const int e = (q == 0 ? 0 : q == 1 ? 6 : 11) + 2 * rd + ei;
        float part[9];
        partial(e, part);
        if (q == PG) {
#pragma unroll
          for (int k = 0; k < 9; ++k) own[ei][k] = part[k];
        } else {
          const int slot = PG < q ? PG : PG - 1;
#pragma unroll
          for (int k = 0; k < 9; ++k) xch[((((q * 2 + slot) * 4 + sp) * 2 + ei) * 9 + k) * 64 + lane] = part[k];
        }
      }
    }
    __syncthreads();
    {
      const int ne = (PG == 0 || rd < 2) ? 2 : 1;
#pragma unroll
      for (int ei = 0; ei < 2; ++ei) {
        if (ei >= ne) continue;
        const int e = (PG == 0 ? 0 : PG == 1 ? 6 : 11) + 2 * rd + ei;
        const long tile = (long)tb * WT + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (tile >= Mt) continue;
        const int b = (int)(tile / TT), t = (int)(tile % TT);
        const int ti = t / T, tj = t % T;
        long mrow[9];
        float rv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {   // nine residual loads issued back to back: one wait, not nine
          const int pi = 3 * ti + k / 3, pj = 3 * tj + k % 3;
          mrow[k] = (pi < N && pj < N) ? ((long)b * P + pi + (long)N * pj) * kC + co : -1;
          rv[k] = (res && mrow[k] >= 0) ? res[mrow[k]] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (mrow[k] < 0) continue;
          // fixed association (group0 + group1) + group2 whoever owns e: a position's result must
          // not depend on which register index -- i.e. which batch row -- it happens to land on
          const float x0 = xch[((((PG * 2 + 0) * 4 + sp) * 2 + ei) * 9 + k) * 64 + lane];
          const float x1 = xch[((((PG * 2 + 1) * 4 + sp) * 2 + ei) * 9 + k) * 64 + lane];
          const float yy = PG == 0 ? (own[ei][k] + x0) + x1 : PG == 1 ? (x0 + own[ei][k]) + x1 : (x0 + x1) + own[ei][k];
          float v = yy * sc + sh + rv[k];
          if (relu) v = fmaxf(v, 0.f);
          y[mrow[k]] = v;
        }
      }
    }
    __syncthreads();
  }
}

template <int DBG>   // timing experiments only: 0 = product; 1 = no DMA after stage 0; 2 = no MFMA; 3 = 1 + no barriers; 4 = 1 + no LDS reads
__global__ __launch_bounds__(768, 3) void k_wino_gemm(
    const float* __restrict__ vimg, const float* __restrict__ uimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    const int* __restrict__ d_count, int N, int T, int relu) {
  __shared__ __attribute__((aligned(16))) float lds[3][STAGE];
  const long Mt = (long)(*d_count) * T * T;
  // Workgroup -> (tile block, cout block) placement for the per-XCD L2 (4 MB, block b runs on XCD b % 8):
  //   * even XCDs work on cout blocks {0,1}, odd XCDs on {2,3}: the transformed weights an XCD streams
  //     over and over are 2 x 1.6 MB and stay L2-resident (with all four blocks per XCD the 6.5 MB
  //     cyclic stream thrashed the L2 and U came from MALL/HBM: 7.5 GB per layer);
  //   * the two cout blocks of a tile block are adjacent workgroups of one XCD and share the L2 copy
  //     of that tile block's V slab, which is fetched from HBM by two XCDs (3.9 GB instead of 2).
  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  const int cb = 2 * (xcd & 1) + (j & 1);
  const int tb = (xcd >> 1) + 4 * (j >> 1);
  if ((long)tb * WT >= Mt) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (SGPR)
  const float* asrc = vimg + (long)tb * WNS * A_STAGE;
  const float* bsrc = uimg + (long)cb * WNS * B_STAGE;
  const int pg = wave >> 2;
  if (pg == 0) wino_gemm_body<DBG, 0>(lds, asrc, bsrc, scale, shift, res, y, Mt, N, T, relu, tb, cb, wave, lane);
  else if (pg == 1) wino_gemm_body<DBG, 1>(lds, asrc, bsrc, scale, shift, res, y, Mt, N, T, relu, tb, cb, wave, lane);
  else wino_gemm_body<DBG, 2>(lds, asrc, bsrc, scale, shift, res, y, Mt, N, T, relu, tb, cb, wave, lane);
}

// ------------------------------------------------------------------ host side

// Flux [kw,kh,cin,cout] column-major -> stage images U[cout block][stage][plane][cout 64][8] with
// U_xi = G k G^T computed in float64.  k is the CORRELATION kernel: NNlib's conv is a true
// convolution, so tap (a', b') reading x[i + a' - 1, j + b' - 1] carries w[2 - a', 2 - b'].
void wino_pack_weights(const ConvHost& c, float* out) {
  static const double G[5][3] = {{0.5, 0.0, 0.0}, {0.5, 0.5, 0.5}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                                 {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
  const int cin = c.cin, cout = c.cout;
  std::memset(out, 0, sizeof(float) * wino_weight_floats());
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci) {
      double k[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) k[a][b] = c.w[(2 - a) + 3 * ((2 - b) + 3 * (ci + (size_t)cin * o))];
      const int cb = o / WC, ol = o % WC, st = ci / WK, cl = ci % WK;
      for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
          double u = 0.0;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) u += G[i][a] * k[a][b] * G[j][b];
          const int xi = i * 5 + j;
          const int pos = 2 * wino_pair_pos(ol, xi, cl >> 1) + (cl & 1);
          out[(((size_t)cb * WNS + st) * WPL + (xi >> 1)) * WC * 8 + (size_t)ol * 8 + pos] = (float)u;
        }
    }
}

size_t wino_weight_floats() { return (size_t)(kC / WC) * WNS * B_STAGE; }
size_t wino_v_floats(int bcap, int T) {
  const long blocks = ((long)bcap * T * T + WT - 1) / WT;
  return (size_t)blocks * WNS * A_STAGE;
}

void launch_wino_conv(const float* x, float* vimg, const float* uimg, const float* scale, const float* shift,
                      const float* res, float* y, const int* d_count, int bcap, int N, int relu, hipStream_t s) {
  const int T = (N + 2) / 3;
  const int blocks = (int)(((long)bcap * T * T + WT - 1) / WT);
  hipLaunchKernelGGL((k_wino_in<32, true>), dim3(2 * blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
  static const int dbg = getenv("AGZ_WINO_DEBUG") ? atoi(getenv("AGZ_WINO_DEBUG")) : 0;   // timing experiments only
  auto kern = dbg == 1 ? k_wino_gemm<1> : dbg == 2 ? k_wino_gemm<2> : dbg == 3 ? k_wino_gemm<3> : dbg == 4 ? k_wino_gemm<4> : dbg == 5 ? k_wino_gemm<5> : dbg == 6 ? k_wino_gemm<6> : dbg == 7 ? k_wino_gemm<7> : dbg == 8 ? k_wino_gemm<8> : dbg == 9 ? k_wino_gemm<9> : k_wino_gemm<0>;
  const int per_xcd = 2 * ((blocks + 3) / 4);   // see the placement comment in k_wino_gemm
  hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(768), 0, s, (const float*)vimg, uimg, scale, shift,
                     res, y, d_count, N, T, relu);
}

}  // namespace agz
