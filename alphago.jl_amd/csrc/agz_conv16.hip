// agz_conv16.hip -- the 3x3 256->256 tower convolution with fp16 operands on
// v_mfma_f32_32x32x16_f16 (f32 accumulate): the "fp16 MFMA path" of BASELINE.json configs[4].
// Opt-in (agz_net_set_precision); the default network is exact f32 (agz_wino.hip).
//
// Implicit GEMM, M = B*N^2 board points, N = 256 couts, K = 9*256 ordered (cin chunk, tap, cin).
// What makes it different from the f32 direct kernel (agz_nn.hip) is where the nine taps come from:
// a workgroup owns 256 consecutive rows and ALL 256 output channels, brings the 32-channel slab of
// its rows plus a halo of N+1 rows on either side into LDS ONCE per channel chunk, and the nine taps
// read that slab at nine row offsets (a lane whose neighbour is off the board selects zeros).
// Activations are therefore fetched once instead of nine times, and the weights -- 20 KB per
// (chunk, tap) -- are the only per-stage stream.  fp16 MFMA is 16x the f32 rate, so everything
// around it has to move that much less.
//
//   tower activations  [M][256] half in HBM (the f32 stem output is converted once per forward; the
//                      last conv writes f32 for the heads; residuals in either type)
//   workgroup          256 rows x CH couts; CH = 128 in the product: 4 waves (one per 64-row group), two
//                      workgroups per CU; a wave holds 2 x 4 accumulator tiles of 32x32 (128 VGPRs)
//   weights            stored in HBM as ready-made padded LDS tile images Wi[stage 72][256 rows][40 halves]
//                      (80 B rows: the 16 rows of a ds_read_b128 phase fall on 16 distinct 16-B bank
//                      groups), true-convolution flip applied at pack time; brought in by LDS-DMA
//                      (global_load_lds_dwordx4), triple-buffered, two stages in flight, counted vmcnt
//   slab               also by DMA: a lane may fetch from any global address but always writes LDS slot
//                      chunk*64 + lane, so the slab is stored unpadded (64 B rows) and bank conflicts are
//                      avoided by a swizzle applied on the SOURCE side: slot (row, p) holds piece
//                      p ^ ((row >> 2) & 3).  Rows outside [0, M) are fetched from a clamped address:
//                      every use of them is masked.
//   a stage            one (chunk, tap): 2 k-steps x 8 MFMAs per wave, both k-steps' operands read up
//                      front, the DMA instructions interleaved between the MFMAs (see agz_wino.hip for
//                      why), one barrier per stage
//   epilogue           half-in/half-out layers stage their tile through LDS so that residual and result
//                      move as 16-byte pieces of whole rows (2-byte scatters cost 0.27 ms per layer)
//   history            a first version that moved the weights global -> VGPR -> LDS topped out at
//                      ~10 B/clk/CU on that stream (1.19 ms per layer; now 0.80)
// The reduction order of an output is fixed (chunk, tap, k) and does not depend on where its row
// sits in the batch: tree parity with the oracle (which calls this network) stays bit-exact.
#include "agz_nn.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace agz {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HM = 256;            // output rows per workgroup
constexpr int HK = 32;             // channels per chunk
constexpr int HS = HK + 8;         // LDS row stride (halves)
constexpr int HCH = kC / HK;       // 8 chunks
constexpr int HSLAB_MAX = HM + 2 * (19 + 1);   // 296 rows

static size_t conv16_weight_halves() { return (size_t)HCH * 9 * kC * HK; }

// Wh[chunk][tap][cout][32]; tap reads x[row + da, col + db] with da = tap % 3 - 1, db = tap / 3 - 1 and
// multiplies Flux's w[a = 1 - da, b = 1 - db] (NNlib true convolution), as in pack_conv3 (agz_nn.hip)
static void conv16_pack_weights(const ConvHost& c, uint16_t* out) {
  for (int cc = 0; cc < HCH; ++cc)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        const int tap = (2 - a) + 3 * (2 - b);
        for (int o = 0; o < kC; ++o)
          for (int k = 0; k < HK; ++k) {
            const int ci = cc * HK + k;
            const __half h = __float2half_rn(c.w[a + 3 * (b + 3 * (ci + (size_t)kC * o))]);
            out[(((size_t)cc * 9 + tap) * kC + o) * HK + k] = *reinterpret_cast<const uint16_t*>(&h);
          }
      }
}

constexpr int H2_BIMG = kC * HS;                      // halves per weight tile image (20,480 B)
constexpr int H2_SLABCH = (HSLAB_MAX * 4 + 63) / 64;   // 19 chunks of 64 pieces
size_t conv16_image_halves() { return (size_t)HCH * 9 * H2_BIMG; }

void conv16_pack_images(const ConvHost& c, uint16_t* out) {
  std::vector<uint16_t> w(conv16_weight_halves());
  conv16_pack_weights(c, w.data());
  for (int st = 0; st < HCH * 9; ++st)
    for (int o = 0; o < kC; ++o)
      for (int k = 0; k < HS; ++k)
        out[((size_t)st * kC + o) * HS + k] = k < HK ? w[((size_t)st * kC + o) * HK + k] : 0;
}

__device__ __forceinline__ void glds16h(const void* g, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_byte_addr)
      : "memory");
}

// CH = output channels per workgroup: 256 (8 waves, one workgroup per CU) or 128 (4 waves, TWO workgroups
// per CU: one's barriers, prologue and epilogue hide behind the other's MFMAs).
// DBG: timing experiments, instantiated only under -DAGZ_TIMING_EXPERIMENTS (wrong results): 0 = product; 1 = no MFMA;
// 2 = no DMA in the loop; 3 = no LDS reads; 4 = no main loop; 5 = no epilogue.  What they showed (round 2): the
// 0.68 ms main loop is the SUM of its three phases (MFMA 0.21 + DMA issue 0.27 + LDS reads 0.23 ms as additive shares),
// i.e. the waves of a SIMD do not overlap them; reading a stage's operands one stage ahead (second register set, a
// fourth weight image, three stages in flight) and scalar-base DMA addressing were each built, bit-identical, and did
// not move it (0.79 / 0.81 ms against 0.77): what serialises is the per-stage barrier of a 512-cycle stage.
template <int DBG, int CH>
__global__ __launch_bounds__(CH / 32 * 64, CH == 128 ? 2 : 1) void k_conv3x3_f16_dma(const _Float16* __restrict__ x, const uint16_t* __restrict__ wi,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const void* __restrict__ res, int res_f32, void* __restrict__ y,
                                                          int out_f32, const int* __restrict__ d_count, int N, int relu) {
  constexpr int NWAVE = CH / 32, BIMG = CH * HS, NCHW = BIMG / 512;   // waves; halves / 1 KB chunks per weight tile
  // one LDS block: 3 weight tile images + 2 slab buffers (100,352 B / 69,632 B); the epilogue reuses it
  __shared__ __attribute__((aligned(16))) _Float16 smem[3 * BIMG + 2 * H2_SLABCH * 512];
  _Float16(*sb)[BIMG] = reinterpret_cast<_Float16(*)[BIMG]>(smem);
  _Float16(*sa)[H2_SLABCH * 512] = reinterpret_cast<_Float16(*)[H2_SLABCH * 512]>(smem + 3 * BIMG);
  const int P = N * N;
  const long M = (long)(*d_count) * P;
  // consecutive logical workgroups (the cout halves of one row tile, then the next row tile) run on the
  // same XCD and share its L2 copy of the slab: block b runs on XCD b % 8
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int NCB = kC / CH;
  const long m0 = (long)(lid / NCB) * HM;
  const int n0 = (lid % NCB) * CH;
  if (m0 >= M) return;
  const int halo = N + 1, slab = HM + 2 * halo;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, wc = wave >> 2;
  const int l31 = lane & 31, hi = lane >> 5;
  const unsigned sb0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)&smem[0];
  const unsigned sa0 = sb0 + 3u * BIMG * 2u;
  const int nslabch = (slab * 4 + 63) / 64;

  // j-th weight chunk of this wave for `stage` into buffer `buf` (chunks wave, wave+NWAVE, ... of NCHW)
  auto dma_w = [&](int stage, int buf, int j) {
    const int c = wave + NWAVE * j;
    if (c >= NCHW) return;
    glds16h(wi + (size_t)stage * H2_BIMG + n0 * HS + c * 512 + lane * 8, sb0 + (unsigned)(buf * BIMG + c * 512) * 2u);
  };
  // j-th slab chunk of this wave for channel chunk cc into slab buffer `buf`
  auto dma_a = [&](int cc, int buf, int j) {
    const int c = wave + NWAVE * j;
    if (c >= nslabch) return;
    const int slot = c * 64 + lane, s = slot >> 2, q = (slot & 3) ^ ((s >> 2) & 3);
    long g = m0 - halo + s;
    g = g < 0 ? 0 : (g >= M ? M - 1 : g);                 // out-of-range rows are only ever read masked
    glds16h(x + g * kC + cc * HK + q * 8, sa0 + (unsigned)(buf * (H2_SLABCH * 512) + c * 512) * 2u);
  };
  const int nb = wave < NCHW - 2 * NWAVE ? 3 : 2;          // weight chunks this wave issues per stage
  constexpr int JS = (H2_SLABCH + NWAVE - 1) / NWAVE;      // slab chunks per wave, at most

  int srow[2];
  unsigned vmask[2];
#pragma unroll
  for (int rbk = 0; rbk < 2; ++rbk) {
    const int lr = wr * 64 + rbk * 32 + l31;
    const long m = m0 + lr;
    srow[rbk] = lr + halo;
    unsigned mk = 0;
    if (m < M) {
      const int p = (int)(m % P), i = p % N, j = p / N;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int da = tap % 3 - 1, db = tap / 3 - 1;
        if ((unsigned)(i + da) < (unsigned)N && (unsigned)(j + db) < (unsigned)N) mk |= 1u << tap;
      }
    }
    vmask[rbk] = mk;
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  constexpr int NST = HCH * 9;
#pragma unroll
  for (int j = 0; j < JS; ++j) dma_a(0, 0, j);
#pragma unroll
  for (int j = 0; j < 3; ++j) dma_w(0, 0, j);
#pragma unroll
  for (int j = 0; j < 3; ++j) dma_w(1, 1, j);
  if (nb == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __syncthreads();

  int buf = 0;
  for (int st = 0; st < (DBG == 4 ? 0 : NST); ++st) {
    const int cc = st / 9, tap = st - cc * 9;
    const int nbuf = buf == 0 ? 2 : buf - 1;               // (st + 2) % 3
    const bool more = st + 2 < NST && DBG != 2;
    const bool slab_now = tap == 7 && cc + 1 < HCH && DBG != 2;   // the next chunk's slab: issued BEFORE this stage's weights
    const _Float16* A = sa[cc & 1];
    const _Float16* B = sb[buf];
    const int off = (tap % 3 - 1) + N * (tap / 3 - 1);
    h8 af[2][2], bf[2][4];                                 // [k-step][...]: both k-steps' operands up front
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk) {
        const int R = srow[rbk] + off;
        const h8 v = *reinterpret_cast<const h8*>(A + (R * 4 + ((ks * 2 + hi) ^ ((R >> 2) & 3))) * 8);
        const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        af[ks][rbk] = ((vmask[rbk] >> tap) & 1u) ? v : z;
        if (DBG == 3) af[ks][rbk] = h8{(_Float16)1, (_Float16)2, (_Float16)st, 0, 0, 0, 0, (_Float16)lane};
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        bf[ks][cb] = DBG == 3 ? h8{(_Float16)1, (_Float16)cb, (_Float16)st, 0, 0, 0, 0, (_Float16)lane}
                              : *reinterpret_cast<const h8*>(B + (wc * 128 + cb * 32 + l31) * HS + ks * 16 + hi * 8);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          if (DBG != 1) acc[rbk][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][rbk], bf[ks][cb], acc[rbk][cb], 0, 0, 0);
          else acc[rbk][cb][0] += (float)af[ks][rbk][0] + (float)bf[ks][cb][1];
        // a DMA slot after every group of four MFMAs; all slab chunks are issued before the weights
        // of st+2, so that the counted wait below covers them
        const int slot = ks * 2 + rbk;
        __builtin_amdgcn_sched_barrier(0);
        if (slab_now && slot == 0) dma_a(cc + 1, (cc + 1) & 1, 0);
        if (slab_now && slot == 1) {
#pragma unroll
          for (int j = 1; j < JS; ++j) dma_a(cc + 1, (cc + 1) & 1, j);
        }
        if (more && slot == 2) dma_w(st + 2, nbuf, 0);
        if (more && slot == 3) { dma_w(st + 2, nbuf, 1); dma_w(st + 2, nbuf, 2); }
        __builtin_amdgcn_sched_barrier(0);
      }
    // this wave's chunks of stage st+1 (and any slab issued so far) have landed; st+2's may be in flight
    if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nb == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __syncthreads();
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (DBG == 5) {
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) keep += acc[a][b][0] + acc[a][b][15];
    if (keep == 123.456f) reinterpret_cast<float*>(y)[0] = keep;
    return;
  }

  // epilogue: y = act(scale*acc + shift (+ residual)).  C/D map of the 32x32 MFMA: col = lane & 31,
  // row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5) -- a lane holds single couts of 16 rows, which as
  // half stores would be 2-byte scatters (0.27 ms per layer).  Half-in/half-out layers therefore go
  // through a wave-private LDS tile [32 rows][128 couts]: the residual arrives and the result leaves
  // as 16-byte pieces of whole rows.  (The f32 residual of block 0 and the f32 output of the last
  // layer keep the direct path: two layers of twenty.)
  float sc[4], sh[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    sc[cb] = scale[n0 + wc * 128 + cb * 32 + l31];
    sh[cb] = shift[n0 + wc * 128 + cb * 32 + l31];
  }
  if (!out_f32 && !(res && res_f32)) {
    constexpr int TS = 128 + 8;                                  // tile row stride (halves)
    _Float16* T = smem + wave * (32 * TS);
    const _Float16* rh = reinterpret_cast<const _Float16*>(res);
    _Float16* yh = reinterpret_cast<_Float16*>(y);
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
      const long mrow0 = m0 + wr * 64 + rbk * 32;
      if (res) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                            // 32 rows x 16 pieces of 8 halves
          const int pc = lane + 64 * i, r = pc >> 4, c8 = (pc & 15) * 8;
          const long m = mrow0 + r;
          const uint4 v = m < M ? *reinterpret_cast<const uint4*>(rh + m * kC + n0 + wc * 128 + c8) : make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(T + r * TS + c8) = v;
        }
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = (e & 3) + 8 * (e >> 2) + 4 * hi;
          _Float16* t = T + r * TS + cb * 32 + l31;
          float v = acc[rbk][cb][e] * sc[cb] + sh[cb];
          if (res) v += (float)*t;
          if (relu) v = fmaxf(v, 0.f);
          *t = (_Float16)v;
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pc = lane + 64 * i, r = pc >> 4, c8 = (pc & 15) * 8;
        const long m = mrow0 + r;
        if (m < M) *reinterpret_cast<uint4*>(yh + m * kC + n0 + wc * 128 + c8) = *reinterpret_cast<const uint4*>(T + r * TS + c8);
      }
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int n = n0 + wc * 128 + cb * 32 + l31;
#pragma unroll
    for (int rbk = 0; rbk < 2; ++rbk) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long m = m0 + wr * 64 + rbk * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (m < M) {
          float v = acc[rbk][cb][e] * sc[cb] + sh[cb];
          if (res) v += res_f32 ? reinterpret_cast<const float*>(res)[m * kC + n]
                                : (float)reinterpret_cast<const _Float16*>(res)[m * kC + n];
          if (relu) v = fmaxf(v, 0.f);
          if (out_f32) reinterpret_cast<float*>(y)[m * kC + n] = v;
          else reinterpret_cast<_Float16*>(y)[m * kC + n] = (_Float16)v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_f32_to_f16(const float* __restrict__ x, _Float16* __restrict__ y,
                                                     const int* __restrict__ d_count, long per_position) {
  const long n = (long)(*d_count) * per_position;        // multiple of 8
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 256 * 8) {
    const float4 u = *reinterpret_cast<const float4*>(x + i), v = *reinterpret_cast<const float4*>(x + i + 4);
    h8 h = {(_Float16)u.x, (_Float16)u.y, (_Float16)u.z, (_Float16)u.w, (_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    *reinterpret_cast<h8*>(y + i) = h;
  }
}

void launch_f32_to_f16(const float* x, uint16_t* y, const int* d_count, int bcap, int N, hipStream_t s) {
  const long n = (long)bcap * N * N * kC;
  const int grid = (int)std::min<long>((n / 8 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(k_f32_to_f16, dim3(grid), dim3(256), 0, s, x, (_Float16*)y, d_count, (long)N * N * kC);
}

void launch_conv16_dma(const uint16_t* x, const uint16_t* wi, const float* scale, const float* shift, const void* res,
                       int res_f32, void* y, int out_f32, const int* d_count, int bcap, int N, int relu, hipStream_t s) {
  const long rows = (long)bcap * N * N;
  const int tiles = (int)((rows + HM - 1) / HM);
  const _Float16* xh = (const _Float16*)x;
#define AGZ_C16_LAUNCH(D, C)                                                                                       \
  hipLaunchKernelGGL((k_conv3x3_f16_dma<D, C>), dim3(tiles * (kC / C)), dim3(C / 32 * 64), 0, s, xh, wi, scale, shift, \
                     res, res_f32, y, out_f32, d_count, N, relu)
#ifdef AGZ_TIMING_EXPERIMENTS
  // timing experiments only (make EXTRA=-DAGZ_TIMING_EXPERIMENTS): kernel variants with a phase compiled out
  // (wrong results) and the one-workgroup-per-CU form, selected by environment; tools/c16_x.sh
  static const int dbg = getenv("AGZ_C16_DEBUG") ? atoi(getenv("AGZ_C16_DEBUG")) : 0;
  static const int ch = getenv("AGZ_C16_CH") ? atoi(getenv("AGZ_C16_CH")) : 128;
  if (ch == 256) {
    switch (dbg) {
      case 1: AGZ_C16_LAUNCH(1, 256); break;
      case 2: AGZ_C16_LAUNCH(2, 256); break;
      case 3: AGZ_C16_LAUNCH(3, 256); break;
      case 4: AGZ_C16_LAUNCH(4, 256); break;
      case 5: AGZ_C16_LAUNCH(5, 256); break;
      default: AGZ_C16_LAUNCH(0, 256);
    }
    return;
  }
  switch (dbg) {
    case 1: AGZ_C16_LAUNCH(1, 128); return;
    case 2: AGZ_C16_LAUNCH(2, 128); return;
    case 3: AGZ_C16_LAUNCH(3, 128); return;
    case 4: AGZ_C16_LAUNCH(4, 128); return;
    case 5: AGZ_C16_LAUNCH(5, 128); return;
    default: break;
  }
#endif
  AGZ_C16_LAUNCH(0, 128);
#undef AGZ_C16_LAUNCH
}

}  // namespace agz
