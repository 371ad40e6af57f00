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
//                order of the GEMM stages (HBM-bound: reads 1 KB, writes 2.8 KB per board point)
//   k_wino_gemm  25 GEMMs  M_xi[tile][cout] = sum_cin V_xi[tile][cin] * U_xi[cin][cout]  on
//                v_mfma_f32_16x16x4_f32, all 25 accumulators of a (tile, cout) pair in ONE lane, so
//                the inverse transform A^T M A, the bias+BatchNorm affine, the residual add and the
//                ReLU happen in registers in the epilogue: M is never written to memory.
// Stage = 8 input channels x {64 tiles + 32 couts} x 25 planes = 76.8 KB, double-buffered in LDS
// (153.6 KB of the CU's 160 KB) and filled by direct global->LDS DMA (global_load_lds_dwordx4),
// which is why V and U are stored in HBM as ready-made, bank-swizzled stage images.
#include "agz_nn.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace agz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WT = 64;            // tiles per workgroup
constexpr int WC = 32;            // output channels per workgroup
constexpr int WK = 8;             // input channels per stage
constexpr int WNS = kC / WK;      // stages
constexpr int WXI = 25;
constexpr int A_STAGE = WXI * WT * WK;   // floats
constexpr int B_STAGE = WXI * WC * WK;
constexpr int STAGE = A_STAGE + B_STAGE;  // 19200 floats = 76.8 KB

// physical position (in 2-float pairs) of logical k-pair g in the 8-float row `row`.  A row is 8
// dwords, so rows r, r+4, r+8, r+12 of a 16-row MFMA operand start on the same bank modulo 32 and
// r, r+8 modulo 64; rotating the pairs by (row >> 2) gives every lane of a read group its own
// 2-dword slot under BOTH LDS bankings -- ds_read_b64 (32-lane groups, 64 banks) and the
// ds_read2st64_b64 pairs hipcc likes to fuse neighbouring planes into (16-lane groups, 32 banks).
__host__ __device__ __forceinline__ int wino_pair_pos(int row, int g) { return (g + (row >> 2)) & 3; }

// ------------------------------------------------------------------ input transform

__device__ __forceinline__ void bt5(float x0, float x1, float x2, float x3, float x4, float* r) {
  r[0] = 2.f * x0 - x1 - 2.f * x2 + x3;
  r[1] = 2.f * x1 + x2 - x3;
  r[2] = -2.f * x1 + 3.f * x2 - x3;
  r[3] = x3 - x1;
  r[4] = 2.f * x1 - x2 - 2.f * x3 + x4;
}

// grid = tile blocks; 256 threads = 4 waves x (16 tiles x 4 channel pairs); loops over stages
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ x, float* __restrict__ vimg,
                                                  const int* __restrict__ d_count, int N, int T) {
  const int P = N * N, TT = T * T;
  const long Mt = (long)(*d_count) * TT;
  const int tb = blockIdx.x;
  if ((long)tb * WT >= Mt) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane & 3;
  const int tl = wave * 16 + (lane >> 2);           // tile within the block
  const long tile = (long)tb * WT + tl;
  const bool live = tile < Mt;
  const int b = live ? (int)(tile / TT) : 0, t = live ? (int)(tile % TT) : 0;
  const int ti = t / T, tj = t % T;
  // row offsets (in floats) of the 25 patch points, -1 where the point is off the board
  long off[25];
#pragma unroll
  for (int u = 0; u < 5; ++u)
#pragma unroll
    for (int v = 0; v < 5; ++v) {
      const int pi = 3 * ti - 1 + u, pj = 3 * tj - 1 + v;
      const bool ok = live && pi >= 0 && pi < N && pj >= 0 && pj < N;
      off[u * 5 + v] = ok ? ((long)b * P + pi + (long)N * pj) * kC : -1;
    }
  const int gp = wino_pair_pos(tl, g);
  float* dst0 = vimg + (long)tb * WNS * A_STAGE + (long)tl * WK + 2 * gp;
  for (int st = 0; st < WNS; ++st) {
    float2 d[25];
#pragma unroll
    for (int q = 0; q < 25; ++q)
      d[q] = off[q] >= 0 ? *reinterpret_cast<const float2*>(x + off[q] + st * WK + 2 * g) : make_float2(0.f, 0.f);
    // V = B^T d B: first along u (rows) for every column v, then along v
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
    float* dst = dst0 + (long)st * A_STAGE;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float rx[5], ry[5];
      bt5(tx[i * 5 + 0], tx[i * 5 + 1], tx[i * 5 + 2], tx[i * 5 + 3], tx[i * 5 + 4], rx);
      bt5(ty[i * 5 + 0], ty[i * 5 + 1], ty[i * 5 + 2], ty[i * 5 + 3], ty[i * 5 + 4], ry);
#pragma unroll
      for (int j = 0; j < 5; ++j)
        *reinterpret_cast<float2*>(dst + (long)(i * 5 + j) * WT * WK) = make_float2(rx[j], ry[j]);
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

// grid = tile blocks x (256 / WC); 512 threads = 8 waves: wm = wave & 3 -> 16 tiles, wn = wave >> 2 -> 16 couts
template <int DBG>   // 0 = product; 1 = no DMA after stage 0 (compute ceiling); 2 = no MFMA (DMA ceiling): timing only
__global__ __launch_bounds__(512, 2) void k_wino_gemm(
    const float* __restrict__ vimg, const float* __restrict__ uimg, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    const int* __restrict__ d_count, int N, int T, int relu) {
  __shared__ __attribute__((aligned(16))) float lds[2][STAGE];
  const int P = N * N, TT = T * T;
  const long Mt = (long)(*d_count) * TT;

  // XCD-aware bijective remap: the 8 cout blocks of one tile block are consecutive logical ids and
  // therefore share an XCD, i.e. one L2 copy of that tile block's 1.6 MB slab of V.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  constexpr int NCB = kC / WC;
  const int tb = lid / NCB, cb = lid % NCB;
  if ((long)tb * WT >= Mt) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (SGPR)
  const int wm = wave & 3, wn = wave >> 2;
  const int l15 = lane & 15, hi = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0][0];

  const float* asrc = vimg + (long)tb * WNS * A_STAGE;
  const float* bsrc = uimg + (long)cb * WNS * B_STAGE;
  // a stage is 75 chunks of 1 KB (64 lanes x 16 B): chunks 0..49 from V, 50..74 from U
  auto issue = [&](int st, int buf) {
    const float* a = asrc + (long)st * A_STAGE;
    const float* b = bsrc + (long)st * B_STAGE;
    for (int c = wave; c < 75; c += 8) {
      const float* g = c < 50 ? a + c * 256 : b + (c - 50) * 256;
      glds16(g + lane * 4, lds0 + (unsigned)(buf * STAGE + c * 256) * 4u);
    }
  };

  f32x4 acc[WXI];
#pragma unroll
  for (int i = 0; i < WXI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int arow = wm * 16 + l15, brow = wn * 16 + l15;
  const int aoff = arow * WK + 2 * wino_pair_pos(arow, hi);
  const int boff = A_STAGE + brow * WK + 2 * wino_pair_pos(brow, hi);

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = 0; st < WNS; ++st) {
    const int buf = st & 1;
    if (st + 1 < WNS && DBG != 1) issue(st + 1, buf ^ 1);
    const float* L = lds[buf];
    if (DBG != 2)
#pragma unroll
    for (int q5 = 0; q5 < 5; ++q5) {
      float2 a[5], b[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        a[j] = *reinterpret_cast<const float2*>(L + aoff + (q5 * 5 + j) * WT * WK);
        b[j] = *reinterpret_cast<const float2*>(L + boff + (q5 * 5 + j) * WC * WK);
      }
#pragma unroll
      for (int j = 0; j < 5; ++j)
        acc[q5 * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b[j].x, acc[q5 * 5 + j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 5; ++j)
        acc[q5 * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b[j].y, acc[q5 * 5 + j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stage st+1 has landed
    __syncthreads();                                    // ... and so has everybody else's
  }

  // epilogue: C/D map of the 16x16 MFMA -- col (cout) = lane & 15, row (tile) = 4*(lane>>4) + reg.
  // All 25 planes of a (tile, cout) pair sit in this lane: Y = A^T M A in registers.
  const int co = cb * WC + wn * 16 + l15;
  const float sc = scale[co], sh = shift[co];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const long tile = (long)tb * WT + wm * 16 + hi * 4 + rg;
    if (tile >= Mt) continue;
    const int b = (int)(tile / TT), t = (int)(tile % TT);
    const int ti = t / T, tj = t % T;
    float h[3][5];   // A^T M  (rows)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float m0 = acc[0 * 5 + j][rg], m1 = acc[1 * 5 + j][rg], m2 = acc[2 * 5 + j][rg],
                  m3 = acc[3 * 5 + j][rg], m4 = acc[4 * 5 + j][rg];
      h[0][j] = m0 + m1 + m2 + m3;
      h[1][j] = m1 - m2 + 2.f * m3;
      h[2][j] = m1 + m2 + 4.f * m3 + m4;
    }
#pragma unroll
    for (int oi = 0; oi < 3; ++oi) {
      const float y0 = h[oi][0] + h[oi][1] + h[oi][2] + h[oi][3];
      const float y1 = h[oi][1] - h[oi][2] + 2.f * h[oi][3];
      const float y2 = h[oi][1] + h[oi][2] + 4.f * h[oi][3] + h[oi][4];
      const float yy[3] = {y0, y1, y2};
      const int pi = 3 * ti + oi;
      if (pi >= N) continue;
#pragma unroll
      for (int oj = 0; oj < 3; ++oj) {
        const int pj = 3 * tj + oj;
        if (pj >= N) continue;
        const long m = (long)b * P + pi + (long)N * pj;
        float v = yy[oj] * sc + sh;
        if (res) v += res[m * kC + co];
        if (relu) v = fmaxf(v, 0.f);
        y[m * kC + co] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ host side

// Flux [kw,kh,cin,cout] column-major -> stage images U[cout block][stage][xi][cout 32][8 cin swizzled]
// with U_xi = G k G^T computed in float64.  k is the CORRELATION kernel: NNlib's conv is a true
// convolution, so tap (a', b') reading x[i + a' - 1, j + b' - 1] carries w[2 - a', 2 - b'].
void wino_pack_weights(const ConvHost& c, float* out) {
  static const double G[5][3] = {{0.5, 0.0, 0.0}, {0.5, 0.5, 0.5}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                                 {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0.0, 0.0, 1.0}};
  const int cin = c.cin, cout = c.cout;
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci) {
      double k[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) k[a][b] = c.w[(2 - a) + 3 * ((2 - b) + 3 * (ci + (size_t)cin * o))];
      const int cb = o / WC, ol = o % WC, st = ci / WK, cl = ci % WK;
      const int pos = 2 * wino_pair_pos(ol, cl >> 1) + (cl & 1);
      for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
          double u = 0.0;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) u += G[i][a] * k[a][b] * G[j][b];
          out[(((size_t)cb * WNS + st) * WXI + (i * 5 + j)) * WC * WK + (size_t)ol * WK + pos] = (float)u;
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
  hipLaunchKernelGGL(k_wino_in, dim3(blocks), dim3(256), 0, s, x, vimg, d_count, N, T);
  static const int dbg = getenv("AGZ_WINO_DEBUG") ? atoi(getenv("AGZ_WINO_DEBUG")) : 0;   // timing experiments only
  auto kern = dbg == 1 ? k_wino_gemm<1> : dbg == 2 ? k_wino_gemm<2> : k_wino_gemm<0>;
  hipLaunchKernelGGL(kern, dim3(blocks * (kC / WC)), dim3(512), 0, s, (const float*)vimg, uimg, scale, shift,
                     res, y, d_count, N, T, relu);
}

}  // namespace agz
