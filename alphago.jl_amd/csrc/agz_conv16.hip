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

void conv16_pack_groups(const ConvHost& c, uint16_t* out);
void conv16_pack_frags(const ConvHost& c, uint16_t* out);
constexpr int H2_BIMG = kC * HS;                      // halves per weight tile image (20,480 B)
constexpr int H2_SLABCH = (HSLAB_MAX * 4 + 63) / 64;   // 19 chunks of 64 pieces
static size_t conv16_old_halves() { return (size_t)HCH * 9 * H2_BIMG; }
size_t conv16_image_halves() { return conv16_old_halves() + 2 * (size_t)HCH * 9 * kC * HK; }

void conv16_pack_images(const ConvHost& c, uint16_t* out) {
  std::vector<uint16_t> w(conv16_weight_halves());
  conv16_pack_weights(c, w.data());
  for (int st = 0; st < HCH * 9; ++st)
    for (int o = 0; o < kC; ++o)
      for (int k = 0; k < HS; ++k)
        out[((size_t)st * kC + o) * HS + k] = k < HK ? w[((size_t)st * kC + o) * HK + k] : 0;
  conv16_pack_groups(c, out + conv16_old_halves());
  conv16_pack_frags(c, out + conv16_old_halves() + (size_t)HCH * 9 * kC * HK);
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

// ---------------------------------------------------------------------------------------------------------
// One wave per SIMD form: a workgroup = 4 waves = 256 rows x ALL 256 couts, a wave = 128 rows x 128 couts
// (4 x 4 accumulator tiles: 256 AGPRs).  Why: the 64 x 128 wave tile above reads 0.75 KB of LDS operands per
// MFMA; at 32 cycles per MFMA and 128 B/clk of LDS per CU that alone is 73 % of the MFMA time, and the weight
// DMA writes into LDS on top of it -- the loop is LDS-bound.  128 x 128 reads 0.5 KB per MFMA and the weights
// arrive once per CU instead of twice.
//   weights   HBM images Wg[(chunk, tap column) 24][tap row 3][256 couts][32 halves], 16-byte pieces swizzled at
//             pack time (piece p of cout o sits in slot p ^ ((o >> 2) & 3): unpadded 64 B rows, conflict-free
//             ds_read_b128); one group = 48 KB contiguous = 48 DMA pieces, 12 per wave; two group buffers
//   slab      as above (source-side swizzle), two buffers; lanes whose neighbour is off the board read a 64-byte
//             row of zeros instead (one v_cndmask on the address, not eight on the data)
//   loop      k-step = 16 MFMAs on one register set while the 8 ds_read_b128 of the next k-step fill the other;
//             ONE barrier per group of 6 k-steps (3072 MFMA cycles), placed between k-steps 4 and 5: by then
//             every read of the group's buffer has returned, so k-step 5 already reads the next group's buffer
//             and issues the DMA of the group after that into the one just released -- no LDS latency is exposed
//             at the barrier.
constexpr int W1_WG = 3 * kC * HK;                  // halves per weight group image (49,152 B)
constexpr int W1_SLAB = H2_SLABCH * 512;            // halves per slab buffer (19,456 B) ...
constexpr int W1_SLABS = W1_SLAB + 32;              // ... followed by its 64-byte row of zeros
constexpr int W1_SMEM = 2 * W1_WG + 2 * W1_SLABS;   // 137,344 B
constexpr int W1_NG = HCH * 3;                      // 24 groups

void conv16_pack_groups(const ConvHost& c, uint16_t* out) {
  std::vector<uint16_t> w(conv16_weight_halves());
  conv16_pack_weights(c, w.data());
  for (int st = 0; st < HCH * 9; ++st)
    for (int o = 0; o < kC; ++o)
      for (int p = 0; p < 4; ++p)
        for (int k = 0; k < 8; ++k)
          out[(((size_t)st * kC + o) * 4 + (p ^ ((o >> 2) & 3))) * 8 + k] = w[((size_t)st * kC + o) * HK + p * 8 + k];
}
size_t conv16_group_halves() { return (size_t)W1_NG * W1_WG; }

// uniform base in SGPRs + per-lane 32-bit byte offset
__device__ __forceinline__ void glds16hs(const void* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_off), "s"(gbase_uniform), "s"(lds_byte_addr) : "memory");
}

template <int DBG>
__global__ __launch_bounds__(256, 1) void k_conv3x3_f16_w1(const _Float16* __restrict__ x, const uint16_t* __restrict__ wg,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const void* __restrict__ res, int res_f32, void* __restrict__ y,
                                                         int out_f32, const int* __restrict__ d_count, int N, int relu) {
  __shared__ __attribute__((aligned(128))) _Float16 smem[W1_SMEM];
  const int P = N * N;
  const long M = (long)(*d_count) * P;
  const long m0 = (long)blockIdx.x * HM;
  if (m0 >= M) return;
  const int halo = N + 1, slab = HM + 2 * halo;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = wave >> 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const unsigned s0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)&smem[0];
  const int nslabch = (slab * 4 + 63) / 64;
  const char* sm = reinterpret_cast<const char*>(smem);
  if (tid < 8) reinterpret_cast<uint4*>(smem + 2 * W1_WG + (tid >> 2) * W1_SLABS + W1_SLAB)[tid & 3] = make_uint4(0, 0, 0, 0);

  const unsigned wlane = (unsigned)lane * 16u;
  auto dma_w = [&](int g, int buf, int j) {               // piece j (0..11) of this wave's twelve of group g
    const int c = wave * 12 + j;
    glds16hs(wg + (size_t)g * W1_WG + c * 512, wlane, s0 + (unsigned)(buf * W1_WG + c * 512) * 2u);
  };
  unsigned aoff[5];                                       // slab piece j of this wave: byte offset of its source row piece
#pragma unroll
  for (int j = 0; j < 5; ++j) {                           // the spare slots repeat the last piece
    int c = wave + 4 * j;
    c = c < nslabch ? c : nslabch - 1;
    const int slot = c * 64 + lane, s = slot >> 2, q = (slot & 3) ^ ((s >> 2) & 3);
    long g = m0 - halo + s;
    g = g < 0 ? 0 : (g >= M ? M - 1 : g);                 // < 2^32 bytes: 8192 x 361 rows x 512 B
    aoff[j] = (unsigned)(g * (kC * 2) + q * 16);
  }
  auto dma_a = [&](int cc, int buf, int j) {
    int c = wave + 4 * j;
    c = c < nslabch ? c : nslabch - 1;
    glds16hs(x + cc * HK, aoff[j], s0 + (unsigned)(2 * W1_WG + buf * W1_SLABS + c * 512) * 2u);
  };

  // slab-relative byte address of (tap, row block)'s k-step-0 piece; lanes whose neighbour is off the board (or whose
  // row is past the batch) point at the zero row behind the slab
  int pre[9][4];
#pragma unroll
  for (int rbk = 0; rbk < 4; ++rbk) {
    const int lr = wr * 128 + rbk * 32 + l31;
    const long m = m0 + lr;
    const int p = (int)(m % P), bi = p % N, bj = p / N;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int da = tap % 3 - 1, db = tap / 3 - 1;
      const int R = lr + halo + da + N * db;
      const bool ok = m < M && (unsigned)(bi + da) < (unsigned)N && (unsigned)(bj + db) < (unsigned)N;
      pre[tap][rbk] = ok ? R * 64 + ((hi ^ ((R >> 2) & 3)) << 4) : W1_SLAB * 2;
    }
  }
  // B operand byte offset inside a group image for k-step 0 of a tap (tap row and cout block are immediates)
  const int bsw = (l31 >> 2) & 3;
  const int bo0 = (wc * 128 + l31) * 64 + (((bsw & 2) | (hi ^ (bsw & 1))) << 4);

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  h8 A[2][4], B[2][4];
  int aaddr[4];

  auto tap_addr = [&](int sbuf, int tapi) {
    const int base = (2 * W1_WG + sbuf * W1_SLABS) * 2;
#pragma unroll
    for (int rbk = 0; rbk < 4; ++rbk) aaddr[rbk] = pre[tapi][rbk] + base;
  };
  auto read_a = [&](int set, int ks, int rbk) {
    A[set][rbk] = *reinterpret_cast<const h8*>(sm + (aaddr[rbk] ^ (ks << 5)));
  };
  auto read_b = [&](int set, int buf, int t, int ks, int cb) {
    B[set][cb] = *reinterpret_cast<const h8*>(sm + buf * (W1_WG * 2) + (bo0 ^ (ks << 5)) + t * (kC * HK * 2) + cb * (32 * 64));
  };

  // prologue
#pragma unroll
  for (int j = 0; j < 5; ++j) dma_a(0, 0, j);
#pragma unroll
  for (int j = 0; j < 12; ++j) dma_w(0, 0, j);
#pragma unroll
  for (int j = 0; j < 12; ++j) dma_w(1, 1, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  tap_addr(0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) { read_a(0, 0, i); read_b(0, 0, 0, 0, i); }

  // group j (tap column) of channel chunk cc.  FIRST: cc == 0, LAST: cc == HCH - 1 (compile-time, so that the six
  // middle chunks run a branch-free body)
  auto group = [&](int cc, auto jc, auto firstc, auto lastc) {
    constexpr int j = decltype(jc)::value;
    constexpr bool first = decltype(firstc)::value, last = decltype(lastc)::value;
    const int g = cc * 3 + j, buf = g & 1;
    constexpr bool w2 = !(last && j >= 1), gn = !(last && j == 2), sl = j == 0 && !last, g0 = first && j == 0;
    const int ncc = j == 2 ? cc + 1 : cc;
#pragma unroll
    for (int ksi = 0; ksi < 6; ++ksi) {
      const int set = ksi & 1, nset = set ^ 1;
      const int nt = (ksi + 1) >> 1, nks = (ksi + 1) & 1;      // the k-step being prefetched (nt == 3: next group's first)
      if (nks == 0) {                                           // new tap: addresses first
        if (ksi < 5) tap_addr(cc & 1, j * 3 + nt);
        else if (gn) tap_addr(ncc & 1, ((j + 1) % 3) * 3);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rbk = i >> 2, cb = i & 3;
        if (DBG != 1) acc[rbk][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[set][rbk], B[set][cb], acc[rbk][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 8 && (ksi < 5 || gn)) {
          if (i < 4) read_a(nset, nks, i);
          else if (ksi < 5) read_b(nset, buf, nt, nks, i - 4);
          else read_b(nset, buf ^ 1, 0, 0, i - 4);
        }
        if (DBG != 2) {
          // weights of group g+2 go into the buffer this group releases at its barrier: 4 pieces in k-step 5, the
          // other 8 in k-steps 0 and 1 of the next group (which sees them as "group g+1")
          if (DBG == 7) {
            if (ksi == 5 && w2 && i >= 4) dma_w(g + 2, buf, i - 4);
          } else {
            if (ksi == 5 && w2 && i >= 8 && i < 12) dma_w(g + 2, buf, i - 8);
            if (ksi < 2 && gn && !g0 && i >= 8 && i < 12) dma_w(g + 1, buf ^ 1, 4 + ksi * 4 + (i - 8));
          }
          if (ksi == 2 && sl && i >= 8 && i < 13) dma_a(cc + 1, (cc + 1) & 1, i - 8);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ksi == 4 && gn) {
        if (DBG == 6) {
        } else if (sl) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  if (DBG != 4) {
    group(0, I0{}, std::true_type{}, std::false_type{});
    group(0, I1{}, std::true_type{}, std::false_type{});
    group(0, I2{}, std::true_type{}, std::false_type{});
    for (int cc = 1; cc < HCH - 1; ++cc) {
      group(cc, I0{}, std::false_type{}, std::false_type{});
      group(cc, I1{}, std::false_type{}, std::false_type{});
      group(cc, I2{}, std::false_type{}, std::false_type{});
    }
    group(HCH - 1, I0{}, std::false_type{}, std::true_type{});
    group(HCH - 1, I1{}, std::false_type{}, std::true_type{});
    group(HCH - 1, I2{}, std::false_type{}, std::true_type{});
  }
  if (DBG == 5) {
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) keep += acc[a][b][0] + acc[a][b][15];
    if (keep == 123.456f) reinterpret_cast<float*>(y)[0] = keep;
    return;
  }

  // epilogue (see the two-workgroup form above for the layout reasoning)
  float sc[4], sh[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    sc[cb] = scale[wc * 128 + cb * 32 + l31];
    sh[cb] = shift[wc * 128 + cb * 32 + l31];
  }
  if (!out_f32 && !(res && res_f32)) {
    constexpr int TS = 128 + 8;
    __syncthreads();                                             // every wave is out of the operand buffers
    _Float16* T = smem + wave * (32 * TS);
    const _Float16* rh = reinterpret_cast<const _Float16*>(res);
    _Float16* yh = reinterpret_cast<_Float16*>(y);
#pragma unroll
    for (int rbk = 0; rbk < 4; ++rbk) {
      const long mrow0 = m0 + wr * 128 + rbk * 32;
      if (res) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int pc = lane + 64 * i, r = pc >> 4, c8 = (pc & 15) * 8;
          const long m = mrow0 + r;
          const uint4 v = m < M ? *reinterpret_cast<const uint4*>(rh + m * kC + wc * 128 + c8) : make_uint4(0, 0, 0, 0);
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
        if (m < M) *reinterpret_cast<uint4*>(yh + m * kC + wc * 128 + c8) = *reinterpret_cast<const uint4*>(T + r * TS + c8);
      }
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int n = wc * 128 + cb * 32 + l31;
#pragma unroll
    for (int rbk = 0; rbk < 4; ++rbk) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long m = m0 + wr * 128 + rbk * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
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

// ---------------------------------------------------------------------------------------------------------
// Persistent one-wave-per-SIMD form (the product).  What the timing variants of the form above and of this one
// showed (tools/c16_w1.sh): (i) with the weights in LDS the loop runs at 0.38 ms when their DMA is compiled out
// and 0.48 ms with it -- the LDS-DMA writes compete with the operand reads for the LDS port; (ii) a loop of
// nothing but its MFMAs takes 0.44 ms, not the 0.34 of 2.4 GHz: the chip clocks matrix-bound code to ~1.9 GHz
// (MI355X_MICROARCH.md, DVFS); (iii) result stores are issue-bound per CU (~5 B/clk) and vmcnt returns in
// order, so a wave that stores its tile and then starts the next one finds its weight loads queued behind the
// store tail: 0.15 ms per layer, untouched by staggering the workgroups or by deeper prefetch.  Hence:
//   * a wave owns 224 rows x 64 couts (7 x 2 accumulator tiles in AGPRs).  Its weight fragments never touch
//     LDS: they are stored in HBM in MFMA operand order, Wf[k-step 144][cout block 8][lane 64][8 halves], so a
//     wave's two fragments of a k-step are one contiguous 2 KB that global_load_dwordx4 brings straight into
//     registers, W2_D k-steps ahead, through a ring of W2_RING register sets.  No wave reads another wave's
//     weights, and LDS carries the activation slab only (two 17 KB buffers, one barrier per channel chunk).
//   * the accumulators are computed transposed (D = W . X^T: lane = board point, register = cout), so a lane
//     holds four consecutive couts of one row per register quad and the epilogue works on 8-byte groups.
//   * workgroups are persistent (one per CU, tiles blockIdx.x, +gridDim.x, ...).  The epilogue of a half-in /
//     half-out layer only computes: its results stay in a wave-private LDS image (7 passes x 4 KB per wave, dense
//     128-byte rows, 16-byte pieces swizzled by row) and leave for HBM one 1 KB piece every fourth k-step of the
//     NEXT tile's loop -- 28 store instructions spread over 126 k-steps instead of a tail.  (224 rows, not 256:
//     the image of an eighth pass does not fit beside the slabs.)  The last tile's image is flushed at the end.
//     Rows past the batch are stored too: the half activation buffers are padded by one tile (Net::reserve).
//   * the next tile's first slab and weight fragments are in flight while the epilogue runs, and residual pieces
//     are fetched four passes ahead, the first three during the last channel chunk.
// The f32-residual layer (first block) and the f32-output layer (last) keep a direct epilogue through two small
// tiles per wave: two layers of twenty.
constexpr int W2_RB = 7, W2_HM = 32 * W2_RB;        // row blocks / rows per tile
constexpr int W2_SLABCH = ((W2_HM + 2 * 20) * 4 + 63) / 64;   // 17 pieces of 1 KB
constexpr int W2_SLAB = W2_SLABCH * 512;            // halves per slab buffer (17,408 B) ...
constexpr int W2_SLABS = W2_SLAB + 32;              // ... followed by its 64-byte row of zeros
constexpr int W2_OFF_SC = 2 * W2_SLABS;             // scale[256], shift[256] as floats
constexpr int W2_OFF_OUT = W2_OFF_SC + 1024;        // result images [wave 4][pass 7][32 rows][128 B]
constexpr int W2_OUTB = 4 * W2_RB * 4096;           // bytes
constexpr int W2_TB = 32 * 272;                     // direct epilogue: bytes per tile (f32 rows of 64 + 4 pad), 8 of them
constexpr int W2_SMEM = W2_OFF_OUT + W2_OUTB / 2;   // 151,680 B
constexpr int W2_KS = HCH * 18;                     // k-steps per tile
constexpr int W2_D = 17, W2_RING = 18;
constexpr int W2_RR = 2;                            // epilogue passes of residual in flight (f32 residual: 1)
static_assert(W2_OFF_OUT % 64 == 0 && 8 * W2_TB <= W2_OUTB && W2_KS % W2_RING == 0 && 18 % W2_RING == 0, "layout");

void conv16_pack_frags(const ConvHost& c, uint16_t* out) {
  std::vector<uint16_t> w(conv16_weight_halves());
  conv16_pack_weights(c, w.data());
  for (int st = 0; st < HCH * 9; ++st)
    for (int ks = 0; ks < 2; ++ks)
      for (int cb = 0; cb < 8; ++cb)
        for (int lane = 0; lane < 64; ++lane)
          for (int k = 0; k < 8; ++k)
            out[((((size_t)st * 2 + ks) * 8 + cb) * 64 + lane) * 8 + k] =
                w[((size_t)st * kC + cb * 32 + (lane & 31)) * HK + (ks * 2 + (lane >> 5)) * 8 + k];
}

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}

// RES: 0 = no residual, 1 = half, 2 = f32 (first block's skip); OUTF: the output is f32 (last layer), else half.
// DBG (timing variants, -DAGZ_TIMING_EXPERIMENTS, wrong results): bit mask of what is compiled out -- 1 epilogue,
// 2 weight loads, 4 LDS operand reads, 8 slab DMA, 16 MFMA, 32 result stores.
template <int DBG, int RES, bool OUTF>
__global__ __launch_bounds__(256, 1) void k_conv3x3_f16_w2(const _Float16* __restrict__ x, const uint16_t* __restrict__ wf,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const void* __restrict__ res, void* __restrict__ y,
                                                         const int* __restrict__ d_count, int N, int relu) {
  constexpr bool RESF = RES == 2;
  constexpr bool TRICKLE = !RESF && !OUTF;
  // direct epilogue tiles [32 rows][64 couts]: row stride / 16-byte pieces per row / pieces per lane, per element type
  constexpr int RSB = RESF ? 272 : 144, RPR = RESF ? 16 : 8, RNP = RESF ? 8 : 4;
  constexpr int OSB = OUTF ? 272 : 144, OPR = OUTF ? 16 : 8, ONP = OUTF ? 8 : 4;
  constexpr int NRR = RESF ? 1 : W2_RR;               // 8 uint4 of ring either way
  __shared__ __attribute__((aligned(128))) _Float16 smem[W2_SMEM];
  const int P = N * N;
  const int M = (*d_count) * P;                          // < 2^31: 8192 x 361 rows
  const int ntiles = (M + W2_HM - 1) / W2_HM;
  if ((int)blockIdx.x >= ntiles) return;
  const int halo = N + 1, slab = W2_HM + 2 * halo;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const unsigned s0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)&smem[0];
  const int nslabch = (slab * 4 + 63) / 64;
  char* sm = reinterpret_cast<char*>(smem);
  if (tid < 8) reinterpret_cast<uint4*>(smem + (tid >> 2) * W2_SLABS + W2_SLAB)[tid & 3] = make_uint4(0, 0, 0, 0);
  {
    float* tab = reinterpret_cast<float*>(smem + W2_OFF_SC);
    tab[tid] = scale[tid];
    tab[256 + tid] = shift[tid];
  }
  const float invP = 1.f / (float)P, invN = 1.f / (float)N;
  const unsigned wlane = (unsigned)lane * 16u;

  unsigned aoff[5];                                       // slab piece j of this wave: byte offset of its source row piece
  auto slab_src = [&](int m0) __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));                          // (keeps hipcc from hoisting the lane terms into spilled registers)
#pragma unroll
    for (int j = 0; j < 5; ++j) {                         // the spare slots repeat the last piece
      int c = wave + 4 * j;
      c = c < nslabch ? c : nslabch - 1;
      const int slot = c * 64 + ln, s = slot >> 2, q = (slot & 3) ^ ((s >> 2) & 3);
      int g = m0 - halo + s;
      g = g < 0 ? 0 : (g >= M ? M - 1 : g);               // out-of-range rows are only ever read masked
      aoff[j] = (unsigned)g * (unsigned)(kC * 2) + (unsigned)(q * 16);
    }
  };
  auto dma_a = [&](int cc, int buf, int j) __attribute__((always_inline)) {
    int c = wave + 4 * j;
    c = c < nslabch ? c : nslabch - 1;
    glds16hs(x + cc * HK, aoff[j], s0 + (unsigned)(buf * W2_SLABS + c * 512) * 2u);
  };

  unsigned vm[W2_RB];                                     // per row block: bit `tap` = that neighbour is on the board
  auto tile_masks = [&](int m0) __attribute__((always_inline)) {
    const int p0 = m0 % P;
#pragma unroll
    for (int rbk = 0; rbk < W2_RB; ++rbk) {
      const int lr = rbk * 32 + l31, v = p0 + lr;         // < P + 224: the float quotients below are exact
      const int p = v - (int)(((float)v + 0.5f) * invP) * P;
      const int bj = (int)(((float)p + 0.5f) * invN), bi = p - bj * N;
      const unsigned cm = (bi > 0 ? 1u : 0u) | 2u | (bi < N - 1 ? 4u : 0u);
      unsigned mk = (bj > 0 ? cm : 0u) | (cm << 3) | (bj < N - 1 ? cm << 6 : 0u);
      vm[rbk] = m0 + lr < M ? mk : 0u;
    }
  };

  f32x16 acc[W2_RB][2];
  h8 A[W2_RB], Bf[W2_RING][2];
  int aaddr[W2_RB];
  auto tap_addr = [&](int sbuf, int tapi) __attribute__((always_inline)) {
    const int off = (tapi % 3 - 1) + N * (tapi / 3 - 1);
    const int base = sbuf * (W2_SLABS * 2);
    const int R0 = l31 + halo + off;
    const int a0 = base + (R0 << 6) + ((((R0 >> 2) ^ hi) & 3) << 4);      // row block rbk: + rbk * 2048, same swizzle
#pragma unroll
    for (int rbk = 0; rbk < W2_RB; ++rbk) aaddr[rbk] = ((vm[rbk] >> tapi) & 1u) ? a0 + rbk * 2048 : base + W2_SLAB * 2;
  };
  // (one register set: a row block's fragment of the next k-step is fetched right after the block's two MFMAs)
  auto read_a = [&](int ks, int rbk) __attribute__((always_inline)) {
    A[rbk] = *reinterpret_cast<const h8*>(sm + (aaddr[rbk] ^ (ks << 5)));
  };
  const char* wfw = reinterpret_cast<const char*>(wf) + wave * 2048;
  auto load_b = [&](int slot, const char* wbase, int kk) __attribute__((always_inline)) {   // this wave's two fragments of k-step kk
    const char* p = wbase + (size_t)kk * 8192;
    Bf[slot][0] = *reinterpret_cast<const h8*>(p + wlane);
    Bf[slot][1] = *reinterpret_cast<const h8*>(p + 1024 + wlane);
  };

  // result image of this wave; a lane's piece i (0..3) of a pass is row (lane >> 3) + 8 i, 16-byte column lane & 7
  char* outw = sm + W2_OFF_OUT * 2 + wave * (W2_RB * 4096);
  int tl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (lane >> 3) + 8 * i;
    tl[i] = row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  const unsigned tg = (unsigned)((lane >> 3) * (kC * 2) + (lane & 7) * 16 + wave * 128);   // + i * 4096 + pass * 16384 + tile base
  const int lx = (l31 * 128 + 8 * hi) | (((l31 >> 1) & 7) << 4);                             // lane's 8-byte group: lx ^ (piece << 4)
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 treg = {0, 0, 0, 0};
  char* yprev = nullptr;                                  // tile whose image is leaving: base of its rows in y

  int tile = blockIdx.x;
  int m0 = tile * W2_HM;
  slab_src(m0);
#pragma unroll
  for (int j = 0; j < 5; ++j) dma_a(0, 0, j);
#pragma unroll
  for (int k = 0; k < W2_D; ++k) load_b(k, wfw, k);
  tile_masks(m0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  tap_addr(0, 0);
  static_for<0, W2_RB>([&](auto ic) __attribute__((always_inline)) { read_a(0, decltype(ic)::value); });
  // (the first tile has no predecessor: it sends its own rows' stale image ahead of the real one, same wave, same
  // addresses, in order -- cheaper than a branch around every piece)
  yprev = reinterpret_cast<char*>(y) + (size_t)m0 * (kC * 2);

  // Residual pieces of the next NRR epilogue passes (a pass is ~1 us of work, an HBM read under load is more).
  // Scalars, not an array: hipcc moves a by-reference captured array into LDS.
  uint4 rr0, rr1, rr2, rr3, rr4, rr5, rr6, rr7, rr8, rr9, rr10, rr11, rr12, rr13, rr14, rr15;
  auto rr = [&](auto ic) __attribute__((always_inline)) -> uint4& {
    constexpr int i = decltype(ic)::value;
    if constexpr (i == 0) return rr0;
    else if constexpr (i == 1) return rr1;
    else if constexpr (i == 2) return rr2;
    else if constexpr (i == 3) return rr3;
    else if constexpr (i == 4) return rr4;
    else if constexpr (i == 5) return rr5;
    else if constexpr (i == 6) return rr6;
    else if constexpr (i == 7) return rr7;
    else if constexpr (i == 8) return rr8;
    else if constexpr (i == 9) return rr9;
    else if constexpr (i == 10) return rr10;
    else if constexpr (i == 11) return rr11;
    else if constexpr (i == 12) return rr12;
    else if constexpr (i == 13) return rr13;
    else if constexpr (i == 14) return rr14;
    else return rr15;
  };
  auto load_res = [&](auto rc) __attribute__((always_inline)) {      // pass r -> ring slot r % NRR
    constexpr int r = decltype(rc)::value;
    static_for<0, RNP>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int pc = lane + 64 * i, row = pc / RPR, c16 = pc % RPR;
      int m = m0 + r * 32 + row;
      m = m < M ? m : M - 1;                              // rows past the batch: any value, never used
      rr(std::integral_constant<int, (r % NRR) * RNP + i>{}) =
          *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(res) + ((size_t)m * kC + wave * 64) * (RESF ? 4 : 2) + c16 * 16);
    });
  };

  // one channel chunk: 9 taps x 2 k-steps of 14 MFMAs.  FIRST / LAST chunk of a tile are compile-time.
  auto chunk = [&](int cc, auto firstc, auto lastc) __attribute__((always_inline)) {
    constexpr bool first = decltype(firstc)::value, last = decltype(lastc)::value;
    const int sbuf = cc & 1;
    // (laundered once per chunk: hipcc otherwise hoists every address and every mask test of the 18 k-steps out of the
    // chunk loop and keeps them all live in SGPRs -- which it then spills to VGPR lanes)
    unsigned wboff = 0;
    asm volatile("" : "+s"(wboff));
    const char* wb = wfw + wboff;
#pragma unroll
    for (int rbk = 0; rbk < W2_RB; ++rbk) asm volatile("" : "+v"(vm[rbk]));
    static_for<0, 18>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int slot = i % W2_RING;
      constexpr int nks = (i + 1) & 1;
      if (nks == 0) {                                     // the k-step being prefetched opens a new tap
        if (i < 17) tap_addr(sbuf, (i + 1) >> 1);
        else if (!last) tap_addr(sbuf ^ 1, 0);
      }
      int kn = cc * 18 + i + W2_D;
      kn = kn >= W2_KS ? kn - W2_KS : kn;
#pragma unroll
      for (int mi = 0; mi < 2 * W2_RB; ++mi) {
        const int rbk = mi >> 1, cb = mi & 1;
        if (DBG & 16) {
          acc[rbk][cb][mi] += (float)Bf[slot][cb][0] * (float)A[rbk][1];
        } else if (first && i == 0) {
          const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          acc[rbk][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bf[slot][cb], A[rbk], z, 0, 0, 0);
        } else {
          acc[rbk][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bf[slot][cb], A[rbk], acc[rbk][cb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (cb == 1 && (i < 17 || !last) && !(DBG & 4)) read_a(nks, rbk);
        if (mi == 6 && !(DBG & 2)) load_b((i + W2_D) % W2_RING, wb, kn);
        // the next chunk's slab (the next tile's first, in the last chunk) goes out in k-steps 0 and 1
        // (after the last tile the spare buffer just receives the first slab once more: no branch in the loop)
        if (!(DBG & 8)) {
          if (i == 0 && mi >= 9 && mi < 12) dma_a(last ? 0 : cc + 1, sbuf ^ 1, mi - 9);
          if (i == 1 && mi >= 9 && mi < 11) dma_a(last ? 0 : cc + 1, sbuf ^ 1, 3 + mi - 9);
        }
        // chunk cc sends pass cc of the previous tile's image on its way: piece i / 4, read one k-step before it is stored
        if (TRICKLE && !last && !(DBG & 33) && mi == 12) {
          if (i % 4 == 2) treg = *reinterpret_cast<const u4*>(outw + cc * 4096 + tl[i / 4]);
          if (i % 4 == 3) {
            if (DBG & 64) __builtin_nontemporal_store(treg, reinterpret_cast<u4*>(yprev + (size_t)cc * (32 * kC * 2) + (i / 4) * 4096 + tg));
            else if (DBG & 128) *reinterpret_cast<u4*>(reinterpret_cast<char*>(y) + (size_t)blockIdx.x * (32 * kC * 2) + (i / 4) * 4096 + tg) = treg;
            else *reinterpret_cast<u4*>(yprev + (size_t)cc * (32 * kC * 2) + (i / 4) * 4096 + tg) = treg;
          }
        }
        if (last && RES != 0 && !(DBG & 1) && mi == 12) {  // the first passes' residual, spread over the last chunk
          if (i == 4) load_res(std::integral_constant<int, 0>{});
          if (i == 10 && NRR > 1) load_res(std::integral_constant<int, 1>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 16) {
        // the slab pieces issued in k-steps 0 and 1 are older than the 2 W2_D weight fragments that may be in flight
        asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
        __syncthreads();
      }
    });
  };

  for (;;) {
    const int next_tile = tile + gridDim.x;
    const bool more = next_tile < ntiles;
    chunk(0, std::true_type{}, std::false_type{});
    for (int cc = 1; cc < HCH - 1; ++cc) chunk(cc, std::false_type{}, std::false_type{});
    if (more) slab_src(next_tile * W2_HM);
    chunk(HCH - 1, std::false_type{}, std::true_type{});

    // ---- epilogue: value = act(scale * acc + shift (+ residual)); acc[rbk][cb][4q + k] is row rbk*32 + l31,
    // cout wave*64 + cb*32 + 8q + 4hi + k
    const float* tab = reinterpret_cast<const float*>(smem + W2_OFF_SC);
    const float lo = relu ? 0.f : -3.0e38f;
    if (DBG & 1) {
      float keep = 0.f;
#pragma unroll
      for (int a = 0; a < W2_RB; ++a) keep += acc[a][0][0] + acc[a][1][15];
      if (keep == 123.456f) reinterpret_cast<float*>(y)[0] = keep;
    } else if (TRICKLE) {
      // compute only: results (and before them the residual) live in this wave's image, the stores ride on the next tile
      auto pass = [&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        char* img = outw + r * 4096;
        if (RES != 0) {
          static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            *reinterpret_cast<uint4*>(img + tl[i]) = rr(std::integral_constant<int, (r % NRR) * 4 + i>{});
          });
          if constexpr (r + NRR < W2_RB) load_res(std::integral_constant<int, r + NRR>{});
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = cb * 32 + 8 * q + 4 * hi;
            const float4 sc = *reinterpret_cast<const float4*>(tab + wave * 64 + n);
            const float4 sh = *reinterpret_cast<const float4*>(tab + 256 + wave * 64 + n);
            h4* t = reinterpret_cast<h4*>(img + (lx ^ ((cb * 4 + q) << 4)));
            float v0 = acc[r][cb][4 * q + 0] * sc.x + sh.x, v1 = acc[r][cb][4 * q + 1] * sc.y + sh.y;
            float v2 = acc[r][cb][4 * q + 2] * sc.z + sh.z, v3 = acc[r][cb][4 * q + 3] * sc.w + sh.w;
            if (RES != 0) {
              const h4 rv = *t;
              v0 += (float)rv[0]; v1 += (float)rv[1]; v2 += (float)rv[2]; v3 += (float)rv[3];
            }
            v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
            *t = h4{(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
          }
        asm volatile("" ::: "memory");
      };
      static_for<0, W2_RB>(pass);
      yprev = reinterpret_cast<char*>(y) + (size_t)m0 * (kC * 2);
    } else {
      // direct: residual and result cross a wave-private LDS tile each, so that HBM sees 16-byte pieces of whole rows
      char* Tin = sm + W2_OFF_OUT * 2 + wave * W2_TB;
      char* Tout = Tin + 4 * W2_TB;
      auto pass = [&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        if (RES != 0) {
          static_for<0, RNP>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const int pc = lane + 64 * i, row = pc / RPR, c16 = pc % RPR;
            *reinterpret_cast<uint4*>(Tin + row * RSB + c16 * 16) = rr(std::integral_constant<int, (r % NRR) * RNP + i>{});
          });
          if constexpr (r + NRR < W2_RB) load_res(std::integral_constant<int, r + NRR>{});
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = cb * 32 + 8 * q + 4 * hi;
            const float4 sc = *reinterpret_cast<const float4*>(tab + wave * 64 + n);
            const float4 sh = *reinterpret_cast<const float4*>(tab + 256 + wave * 64 + n);
            float v0 = acc[r][cb][4 * q + 0] * sc.x + sh.x, v1 = acc[r][cb][4 * q + 1] * sc.y + sh.y;
            float v2 = acc[r][cb][4 * q + 2] * sc.z + sh.z, v3 = acc[r][cb][4 * q + 3] * sc.w + sh.w;
            if (RES != 0) {
              if (RESF) {
                const float4 rv = *reinterpret_cast<const float4*>(Tin + l31 * RSB + n * 4);
                v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
              } else {
                const h4 rv = *reinterpret_cast<const h4*>(Tin + l31 * RSB + n * 2);
                v0 += (float)rv[0]; v1 += (float)rv[1]; v2 += (float)rv[2]; v3 += (float)rv[3];
              }
            }
            v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
            if (OUTF) *reinterpret_cast<float4*>(Tout + l31 * OSB + n * 4) = make_float4(v0, v1, v2, v3);
            else *reinterpret_cast<h4*>(Tout + l31 * OSB + n * 2) = h4{(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
          }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < ONP; ++i) {
          const int pc = lane + 64 * i, row = pc / OPR, c16 = pc % OPR;
          const int m = m0 + r * 32 + row;
          if (m < M && !(DBG & 32))
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + ((size_t)m * kC + wave * 64) * (OUTF ? 4 : 2) + c16 * 16) =
                *reinterpret_cast<const uint4*>(Tout + row * OSB + c16 * 16);
        }
        asm volatile("" ::: "memory");
      };
      static_for<0, W2_RB>(pass);
    }
    if (!more) break;
    tile = next_tile;
    m0 = tile * W2_HM;
    tile_masks(m0);
    tap_addr(0, 0);
    static_for<0, W2_RB>([&](auto ic) __attribute__((always_inline)) { read_a(0, decltype(ic)::value); });
  }
  if (TRICKLE && !(DBG & 33)) {                            // the last tile's image
#pragma unroll
    for (int r = 0; r < W2_RB; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<u4*>(yprev + (size_t)r * (32 * kC * 2) + i * 4096 + tg) = *reinterpret_cast<const u4*>(outw + r * 4096 + tl[i]);
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
  static const int w1 = getenv("AGZ_C16_W1") ? atoi(getenv("AGZ_C16_W1")) : 10;
  if (w1 >= 10) {
    static int ncu = 0;
    if (!ncu) AGZ_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const uint16_t* wfp = wi + conv16_old_halves() + (size_t)HCH * 9 * kC * HK;
    const int grid = std::min((int)((rows + W2_HM - 1) / W2_HM), ncu);
#define AGZ_C16_W2(D, R, OF) hipLaunchKernelGGL((k_conv3x3_f16_w2<D, R, OF>), dim3(grid), dim3(256), 0, s, xh, wfp, scale, shift, res, y, d_count, N, relu)
#define AGZ_C16_W2D(D)                               \
  do {                                               \
    if (out_f32) {                                   \
      if (rk == 0) AGZ_C16_W2(D, 0, true);           \
      else if (rk == 1) AGZ_C16_W2(D, 1, true);      \
      else AGZ_C16_W2(D, 2, true);                   \
    } else {                                         \
      if (rk == 0) AGZ_C16_W2(D, 0, false);          \
      else if (rk == 1) AGZ_C16_W2(D, 1, false);     \
      else AGZ_C16_W2(D, 2, false);                  \
    }                                                \
  } while (0)
    const int rk = !res ? 0 : (res_f32 ? 2 : 1);
    switch (w1 - 10) {              // timing variants: bit mask of what is compiled out (see the kernel)
      case 1: AGZ_C16_W2D(1); break;
      case 3: AGZ_C16_W2D(3); break;
      case 5: AGZ_C16_W2D(5); break;
      case 9: AGZ_C16_W2D(9); break;
      case 7: AGZ_C16_W2D(7); break;
      case 15: AGZ_C16_W2D(15); break;
      case 17: AGZ_C16_W2D(17); break;
      case 32: AGZ_C16_W2D(32); break;
      case 64: AGZ_C16_W2D(64); break;
      case 128: AGZ_C16_W2D(128); break;
      case 2: AGZ_C16_W2D(2); break;
      case 34: AGZ_C16_W2D(34); break;
      default: AGZ_C16_W2D(0);
    }
#undef AGZ_C16_W2D
#undef AGZ_C16_W2
    return;
  }
  if (w1) {
    const uint16_t* wgp = wi + conv16_old_halves();
#define AGZ_C16_W1(D) hipLaunchKernelGGL((k_conv3x3_f16_w1<D>), dim3(tiles), dim3(256), 0, s, xh, wgp, scale, shift, res, res_f32, y, out_f32, d_count, N, relu)
    switch (w1) {
      case 2: AGZ_C16_W1(1); break;
      case 3: AGZ_C16_W1(2); break;
      case 5: AGZ_C16_W1(4); break;
      case 6: AGZ_C16_W1(5); break;
      case 7: AGZ_C16_W1(6); break;
      case 8: AGZ_C16_W1(7); break;
      default: AGZ_C16_W1(0);
    }
#undef AGZ_C16_W1
    return;
  }
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
