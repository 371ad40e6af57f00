// agz_nn.hip -- hand-written gfx950 kernels for the AlphaGo.jl policy/value network.
//
//   k_conv3x3_mfma   3x3 convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32 with the
//                    bias + inference-BatchNorm affine, residual add and ReLU fused into the
//                    epilogue (neural_net.jl:19-21, resnet.jl:26-32).  Exact-f32 MFMA: the
//                    north_star asks for 1e-4 agreement with the fp32/fp64 Flux CPU path.
//   k_head_conv      the two 1x1 head convolutions (256->1, 256->2) + BN + ReLU, one wavefront
//                    per board point with a DPP/shuffle reduction (neural_net.jl:23,28).
//   k_head_fc        Dense(2N^2->A)+softmax and Dense(N^2->256,relu)+Dense(256->1,tanh),
//                    one workgroup per position (neural_net.jl:24-26,29-30).
//   k_feats_*        board-plane feature extraction (features.jl:3-26).
//
// GEMM view of a tower conv: M = B*N^2 rows (positions x points, row = b*P + p,
// p = row + N*col), N = 256 output channels, K = 9*256 ordered (tap, cin).  Activations are
// [M][256] f32 (channel fastest) so that an A-tile row is one 128-byte line per 32-channel
// chunk; weights are pre-packed once to Wt[cout][tap][cin] with the true-convolution kernel
// flip of NNlib applied at pack time, so the B-tile has the same shape as the A-tile.
#include "agz_nn.h"

#include <cmath>
#include <algorithm>
#include <cstring>

#include "../../include/agz_draws.h"

namespace agz {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ conv3x3 implicit GEMM

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_STRIDE = BK + 4;   // 36 floats: r*36 mod 64 hits 16 distinct 16-B slots for
                                     // the 16 rows of every ds_read_b128 lane group

// TAPSPLIT (the training step's small batches, launch_conv3x3_direct_taps): blockIdx.y = one of the nine taps, the
// block reduces over that tap's channels only and writes its partial tile to y + blockIdx.y * ystride
template <int CIN, bool TAPSPLIT = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_mfma(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y,
    const int* __restrict__ d_count, int N, int relu, long ystride = 0) {
  constexpr int NCHUNK = 9 * CIN / BK;
  constexpr int CPT = CIN / BK;  // chunks per tap
  const int c_begin = TAPSPLIT ? (int)blockIdx.y * CPT : 0, c_end = TAPSPLIT ? c_begin + CPT : NCHUNK;
  if (TAPSPLIT) y += (long)blockIdx.y * ystride;
  __shared__ __attribute__((aligned(16))) float lds[2][2][BM * LDS_STRIDE];

  const int P = N * N;
  const long M = (long)(*d_count) * P;

  // XCD-aware, bijective remap: consecutive logical tiles (which share the A rows / halo)
  // run on the same XCD and therefore hit the same L2.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const long m0 = (long)(swz >> 1) * BM;
  const int n0 = (swz & 1) * BN;
  if (m0 >= M) return;

  const int tid = threadIdx.x;
  const int lrow = tid >> 3, c4 = tid & 7;

  // per-thread staging rows (4 rows of the A tile, 4 rows of the B tile); everything is kept
  // in named scalars so that nothing is ever indexed dynamically (no scratch).
#define AGZ_ROWSETUP(i)                                            \
  const long mrow##i = m0 + lrow + 32 * i;                         \
  const bool rvalid##i = mrow##i < M;                              \
  const int p##i = (int)(mrow##i % P);                             \
  const int ri##i = p##i % N, cj##i = p##i / N;                    \
  const float* wrow##i = wt + (long)(n0 + lrow + 32 * i) * (9 * CIN) + c4 * 4;
  AGZ_ROWSETUP(0) AGZ_ROWSETUP(1) AGZ_ROWSETUP(2) AGZ_ROWSETUP(3)
#undef AGZ_ROWSETUP

  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define AGZ_LOADROW(i, da, db, aoff, boff)                                                             \
  {                                                                                                     \
    const bool ok = rvalid##i && (unsigned)(ri##i + da) < (unsigned)N && (unsigned)(cj##i + db) < (unsigned)N; \
    ra##i = ok ? *reinterpret_cast<const float4*>(x + (mrow##i + da + N * db) * CIN + aoff)             \
               : make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
    rb##i = *reinterpret_cast<const float4*>(wrow##i + boff);                                           \
  }
#define AGZ_LOAD_CHUNK(c)                                         \
  {                                                               \
    const int tap_ = (c) / CPT, ci0_ = ((c) % CPT) * BK;          \
    const int da_ = tap_ % 3 - 1, db_ = tap_ / 3 - 1;             \
    const int aoff_ = ci0_ + c4 * 4, boff_ = (c) * BK;            \
    AGZ_LOADROW(0, da_, db_, aoff_, boff_)                        \
    AGZ_LOADROW(1, da_, db_, aoff_, boff_)                        \
    AGZ_LOADROW(2, da_, db_, aoff_, boff_)                        \
    AGZ_LOADROW(3, da_, db_, aoff_, boff_)                        \
  }
#define AGZ_STOREROW(buf, i)                                                                           \
  *reinterpret_cast<float4*>(&lds[buf][0][(lrow + 32 * i) * LDS_STRIDE + c4 * 4]) = ra##i;             \
  *reinterpret_cast<float4*>(&lds[buf][1][(lrow + 32 * i) * LDS_STRIDE + c4 * 4]) = rb##i;
#define AGZ_STORE_CHUNK(buf) { AGZ_STOREROW(buf, 0) AGZ_STOREROW(buf, 1) AGZ_STOREROW(buf, 2) AGZ_STOREROW(buf, 3) }

  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  AGZ_LOAD_CHUNK(c_begin)
  AGZ_STORE_CHUNK(0)
  __syncthreads();

  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) AGZ_LOAD_CHUNK(c + 1)   // global loads fly under the MFMAs below
    const float* As = &lds[buf][0][(wr * 64 + l31) * LDS_STRIDE + hi * 4];
    const float* Bs = &lds[buf][1][(wc * 64 + l31) * LDS_STRIDE + hi * 4];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      // K is consumed in a permuted order (lanes 0-31 take k = 8kk+s, lanes 32-63 take
      // k = 8kk+4+s): the reduction is order-free and both operands use the same map, which
      // lets every lane fetch its four k-steps with one ds_read_b128.
      const float4 a0 = *reinterpret_cast<const float4*>(As + kk * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(As + 32 * LDS_STRIDE + kk * 8);
      const float4 b0 = *reinterpret_cast<const float4*>(Bs + kk * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(Bs + 32 * LDS_STRIDE + kk * 8);
#define AGZ_MFMA4(f)                                                                          \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.f, b0.f, acc[0][0], 0, 0, 0);             \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.f, b1.f, acc[0][1], 0, 0, 0);             \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.f, b0.f, acc[1][0], 0, 0, 0);             \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.f, b1.f, acc[1][1], 0, 0, 0);
      AGZ_MFMA4(x) AGZ_MFMA4(y) AGZ_MFMA4(z) AGZ_MFMA4(w)
#undef AGZ_MFMA4
    }
    if (c + 1 < c_end) AGZ_STORE_CHUNK(buf ^ 1)
    __syncthreads();
  }

#undef AGZ_LOADROW
#undef AGZ_LOAD_CHUNK
#undef AGZ_STOREROW
#undef AGZ_STORE_CHUNK
  // epilogue: y = act(scale*acc + shift (+ residual)); C/D map of the 32x32 MFMA:
  // col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + wc * 64 + tn * 32 + l31;
    const float sc = scale[n], sh = shift[n];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long m = m0 + wr * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
        if (m < M) {
          float v = acc[tm][tn][e] * sc + sh;
          if (res) v += res[m * kC + n];
          if (relu) v = fmaxf(v, 0.f);
          y[m * kC + n] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ heads

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// hp: wv[256] | wp0[256] | wp1[256] | sv tv sp0 tp0 sp1 tp1
__global__ __launch_bounds__(256) void k_head_conv(const float* __restrict__ x, const float* __restrict__ hp,
                                                    float* __restrict__ vh, float* __restrict__ ph,
                                                    const int* __restrict__ d_count, int P) {
  const long M = (long)(*d_count) * P;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 wv = *reinterpret_cast<const float4*>(hp + lane * 4);
  const float4 w0 = *reinterpret_cast<const float4*>(hp + 256 + lane * 4);
  const float4 w1 = *reinterpret_cast<const float4*>(hp + 512 + lane * 4);
  const float sv = hp[768], tv = hp[769], s0 = hp[770], t0 = hp[771], s1 = hp[772], t1 = hp[773];
  for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
    const float4 xv = *reinterpret_cast<const float4*>(x + m * kC + lane * 4);
    float dv = xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
    float d0 = xv.x * w0.x + xv.y * w0.y + xv.z * w0.z + xv.w * w0.w;
    float d1 = xv.x * w1.x + xv.y * w1.y + xv.z * w1.z + xv.w * w1.w;
    dv = wave_sum(dv);
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    if (lane == 0) {
      vh[m] = fmaxf(dv * sv + tv, 0.f);
      ph[m * 2 + 0] = fmaxf(d0 * s0 + t0, 0.f);
      ph[m * 2 + 1] = fmaxf(d1 * s1 + t1, 0.f);
    }
  }
}

__device__ float block_reduce(float v, bool is_max, float* red /*[4]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, t) : v + t;
  }
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

__global__ __launch_bounds__(256) void k_head_fc(
    const float* __restrict__ vh, const float* __restrict__ ph, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    const float* __restrict__ wp, const float* __restrict__ bp, float* __restrict__ pi,
    float* __restrict__ v, const int* __restrict__ d_count, int P, int A) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* vin = sm;             // [P]
  float* pin = sm + P;         // [2P]  index p + P*c (Julia reshape of W x H x C)
  float* logit = pin + 2 * P;  // [A]
  float* red = logit + A;      // [4]
  const int b = blockIdx.x;
  if (b >= *d_count) return;
  const int tid = threadIdx.x;
  for (int i = tid; i < P; i += 256) {
    vin[i] = vh[(long)b * P + i];
    pin[i] = ph[((long)b * P + i) * 2 + 0];
    pin[P + i] = ph[((long)b * P + i) * 2 + 1];
  }
  __syncthreads();
  // policy logits: Dense(2P -> A), W [out,in] column-major
  float lmax = -INFINITY;
  for (int o = tid; o < A; o += 256) {
    float z = bp[o];
    for (int i = 0; i < 2 * P; ++i) z += wp[o + (long)A * i] * pin[i];
    logit[o] = z;
    lmax = fmaxf(lmax, z);
  }
  const float mx = block_reduce(lmax, true, red);
  float lsum = 0.f;
  for (int o = tid; o < A; o += 256) {
    const float e = expf(logit[o] - mx);
    logit[o] = e;
    lsum += e;
  }
  const float sum = block_reduce(lsum, false, red);
  for (int o = tid; o < A; o += 256) pi[(long)b * A + o] = logit[o] / sum;
  // value: Dense(P -> 256, relu) -> Dense(256 -> 1, tanh)
  float h = b1[tid];
  for (int i = 0; i < P; ++i) h += w1[tid + 256 * i] * vin[i];
  h = fmaxf(h, 0.f);
  const float s = block_reduce(h * w2[tid], false, red);
  if (tid == 0) v[b] = tanhf(s + b2[0]);
}

// ------------------------------------------------------------------ features

// One workgroup per position; threads stride over the board.  The eight history boards are
// rebuilt exactly as stone_features does (features.jl:8-14): B_k = B_{k-1} - delta_{k-1}
// while deltas last, then the oldest board is repeated.
__global__ __launch_bounds__(128) void k_feats_from_deltas(
    const int8_t* __restrict__ boards, const int8_t* __restrict__ deltas, const int32_t* __restrict__ ndeltas,
    const int8_t* __restrict__ to_play, int B, int N, float* __restrict__ x32, float* __restrict__ whcn) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int P = N * N;
  const int nd = ndeltas[b];
  const int tp = to_play[b];
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    int cur = boards[(long)b * P + p];
    float f[32];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k >= 1 && k <= nd) cur -= deltas[((long)b * 7 + (k - 1)) * P + p];
      f[2 * k] = cur == tp ? 1.f : 0.f;
      f[2 * k + 1] = cur == -tp ? 1.f : 0.f;
    }
    f[16] = (float)tp;   // colour plane is +1 / -1, features.jl:22
#pragma unroll
    for (int c = 17; c < 32; ++c) f[c] = 0.f;
    if (x32) {
      float4* dst = reinterpret_cast<float4*>(x32 + ((long)b * P + p) * 32);
#pragma unroll
      for (int c = 0; c < 8; ++c) dst[c] = make_float4(f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]);
    }
    if (whcn) {
#pragma unroll
      for (int c = 0; c < 17; ++c) whcn[(long)P * (c + 17L * b) + p] = f[c];
    }
  }
}

__global__ __launch_bounds__(128) void k_whcn_to_x32(const float* __restrict__ whcn, int B, int P,
                                                      float* __restrict__ x32) {
  const int b = blockIdx.x;
  if (b >= B) return;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float* dst = x32 + ((long)b * P + p) * 32;
    for (int c = 0; c < 17; ++c) dst[c] = whcn[(long)P * (c + 17L * b) + p];
    for (int c = 17; c < 32; ++c) dst[c] = 0.f;
  }
}

void launch_features_from_deltas(const int8_t* d_boards, const int8_t* d_deltas, const int32_t* d_ndeltas,
                                 const int8_t* d_to_play, int B, int N, float* d_x32, float* d_whcn,
                                 hipStream_t stream) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_feats_from_deltas, dim3(B), dim3(128), 0, stream, d_boards, d_deltas, d_ndeltas,
                     d_to_play, B, N, d_x32, d_whcn);
  AGZ_HIP(hipGetLastError());
}

void launch_whcn_to_x32(const float* d_whcn, int B, int N, float* d_x32, hipStream_t stream) {
  if (B <= 0) return;
  hipLaunchKernelGGL(k_whcn_to_x32, dim3(B), dim3(128), 0, stream, d_whcn, B, N * N, d_x32);
  AGZ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ host side

static void conv_init(ConvHost& c, int k, int cin, int cout) {
  c.k = k; c.cin = cin; c.cout = cout;
  c.w.assign((size_t)k * k * cin * cout, 0.f);
  c.b.assign(cout, 0.f);
  c.beta.assign(cout, 0.f);
  c.gamma.assign(cout, 1.f);
  c.mean.assign(cout, 0.f);
  c.var.assign(cout, 1.f);
  c.eps = 1e-5f;
}
static void dense_init(DenseHost& d, int in, int out) {
  d.in = in; d.out = out;
  d.w.assign((size_t)in * out, 0.f);
  d.b.assign(out, 0.f);
}

static std::vector<float>* conv_field(ConvHost* c, int kind);

Net::Net(int N, int tower, hipStream_t stream) : N_(N), P_(N * N), A_(N * N + 1), tower_(tower), stream_(stream) {
  conv_init(stem_, 3, kCinStem, kC);
  tconv_.resize(2 * tower);
  for (auto& c : tconv_) conv_init(c, 3, kC, kC);
  conv_init(vconv_, 1, kC, 1);
  conv_init(pconv_, 1, kC, 2);
  dense_init(vfc1_, P_, 256);
  dense_init(vfc2_, 256, 1);
  dense_init(pfc_, 2 * P_, A_);
  std::vector<std::pair<int, int>> keys;
  weight_keys(tower_, keys);
  for (auto& lk : keys) {
    const size_t n = (size_t)param_count(lk.first, lk.second);
    slots_.push_back({lk.first, lk.second, flux_n_, n});
    flux_n_ += n;
  }
  d_flux_.alloc(flux_n_);
  upload_host_all();
}

// every parameter in a fixed (layer, kind) order: the layout of the device master and of agz_broadcast_weights
void Net::weight_keys(int tower, std::vector<std::pair<int, int>>& keys) {
  for (int l = 0; l <= 2 * tower; ++l)
    for (int k = 0; k < 7; ++k) keys.push_back({l, k});
  for (int l : {AGZ_L_VALUE_CONV, AGZ_L_POLICY_CONV})
    for (int k = 0; k < 7; ++k) keys.push_back({l, k});
  for (int l : {AGZ_L_VALUE_FC1, AGZ_L_VALUE_FC2, AGZ_L_POLICY_FC})
    for (int k = 0; k < 2; ++k) keys.push_back({l, k});
}

size_t Net::flux_offset(int layer, int kind) const {
  for (const auto& sl : slots_)
    if (sl.layer == layer && sl.kind == kind) return sl.off;
  AGZ_REQUIRE(false, AGZ_BAD_ARGUMENT, "no parameter (layer %d, kind %d)", layer, kind);
  return 0;
}

const float* Net::host_slot(const FluxSlot& sl) const {
  if (const ConvHost* c = conv(sl.layer))
    return sl.kind == AGZ_K_BN_EPS ? &c->eps : conv_field(const_cast<ConvHost*>(c), sl.kind)->data();
  const DenseHost* d = dense(sl.layer);
  return (sl.kind == AGZ_K_WEIGHT ? d->w : d->b).data();
}

// host vector -> its place in the device master (the host vectors never move: an enqueued copy may read them later)
void Net::upload_slot(const FluxSlot& sl) {
  AGZ_HIP(hipMemcpyAsync(d_flux_.p + sl.off, host_slot(sl), sizeof(float) * sl.n, hipMemcpyHostToDevice, stream_));
}
void Net::upload_host_all() {
  std::vector<float> flat(flux_n_);
  for (const auto& sl : slots_) std::memcpy(flat.data() + sl.off, host_slot(sl), sizeof(float) * sl.n);
  AGZ_HIP(hipMemcpyAsync(d_flux_.p, flat.data(), sizeof(float) * flux_n_, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  derived_dirty_ = true;
  host_stale_ = false;
}

// device master -> host vectors, when somebody other than agz_net_set_weights wrote the master
void Net::sync_host() const {
  if (!host_stale_) return;
  std::vector<float> flat(flux_n_);
  AGZ_HIP(hipMemcpyAsync(flat.data(), d_flux_.p, sizeof(float) * flux_n_, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  for (const auto& sl : slots_) std::memcpy(const_cast<float*>(host_slot(sl)), flat.data() + sl.off, sizeof(float) * sl.n);
  host_stale_ = false;
}

ConvHost* Net::conv(int layer) {
  if (layer == 0) return &stem_;
  if (layer >= 1 && layer <= 2 * tower_) return &tconv_[layer - 1];
  if (layer == AGZ_L_VALUE_CONV) return &vconv_;
  if (layer == AGZ_L_POLICY_CONV) return &pconv_;
  return nullptr;
}
const ConvHost* Net::conv(int layer) const { return const_cast<Net*>(this)->conv(layer); }
DenseHost* Net::dense(int layer) {
  if (layer == AGZ_L_VALUE_FC1) return &vfc1_;
  if (layer == AGZ_L_VALUE_FC2) return &vfc2_;
  if (layer == AGZ_L_POLICY_FC) return &pfc_;
  return nullptr;
}
const DenseHost* Net::dense(int layer) const { return const_cast<Net*>(this)->dense(layer); }

static std::vector<float>* conv_field(ConvHost* c, int kind) {
  switch (kind) {
    case AGZ_K_WEIGHT: return &c->w;
    case AGZ_K_BIAS: return &c->b;
    case AGZ_K_BN_BETA: return &c->beta;
    case AGZ_K_BN_GAMMA: return &c->gamma;
    case AGZ_K_BN_MEAN: return &c->mean;
    case AGZ_K_BN_VAR: return &c->var;
    default: return nullptr;
  }
}

int64_t Net::param_count(int layer, int kind) const {
  if (const ConvHost* c = conv(layer)) {
    if (kind == AGZ_K_BN_EPS) return 1;
    auto* f = conv_field(const_cast<ConvHost*>(c), kind);
    return f ? (int64_t)f->size() : -1;
  }
  if (const DenseHost* d = dense(layer)) {
    if (kind == AGZ_K_WEIGHT) return (int64_t)d->w.size();
    if (kind == AGZ_K_BIAS) return (int64_t)d->b.size();
  }
  return -1;
}

void Net::set(int layer, int kind, const float* data, int64_t count) {
  AGZ_REQUIRE(data != nullptr, AGZ_BAD_ARGUMENT, "agz_net_set_weights: null data");
  const int64_t want = param_count(layer, kind);
  AGZ_REQUIRE(want >= 0, AGZ_BAD_ARGUMENT, "agz_net_set_weights: no parameter (layer %d, kind %d)", layer, kind);
  AGZ_REQUIRE(want == count, AGZ_BAD_SHAPE, "agz_net_set_weights: layer %d kind %d expects %lld floats, got %lld",
              layer, kind, (long long)want, (long long)count);
  sync_host();
  ++param_version_;
  if (ConvHost* c = conv(layer)) {
    if (kind == AGZ_K_BN_EPS) c->eps = data[0];
    else std::memcpy(conv_field(c, kind)->data(), data, sizeof(float) * (size_t)count);
  } else {
    DenseHost* d = dense(layer);
    std::memcpy((kind == AGZ_K_WEIGHT ? d->w : d->b).data(), data, sizeof(float) * (size_t)count);
  }
  for (const auto& sl : slots_)
    if (sl.layer == layer && sl.kind == kind) upload_slot(sl);
  // the copy reads a pageable host vector: drain it before anything (init_synthetic, conv_init, ~Net) may reassign or
  // free that vector.  ~20 us per parameter; bulk device-side paths (train step, broadcast) do not come here.
  AGZ_HIP(hipStreamSynchronize(stream_));
  derived_dirty_ = true;
}

void Net::get(int layer, int kind, float* out, int64_t count) const {
  AGZ_REQUIRE(out != nullptr, AGZ_BAD_ARGUMENT, "agz_net_get_weights: null output");
  const int64_t want = param_count(layer, kind);
  AGZ_REQUIRE(want >= 0, AGZ_BAD_ARGUMENT, "agz_net_get_weights: no parameter (layer %d, kind %d)", layer, kind);
  AGZ_REQUIRE(want == count, AGZ_BAD_SHAPE, "agz_net_get_weights: layer %d kind %d holds %lld floats, asked %lld",
              layer, kind, (long long)want, (long long)count);
  sync_host();
  if (const ConvHost* c = conv(layer)) {
    if (kind == AGZ_K_BN_EPS) out[0] = c->eps;
    else std::memcpy(out, conv_field(const_cast<ConvHost*>(c), kind)->data(), sizeof(float) * (size_t)count);
  } else {
    const DenseHost* d = dense(layer);
    std::memcpy(out, (kind == AGZ_K_WEIGHT ? d->w : d->b).data(), sizeof(float) * (size_t)count);
  }
}

// glorot_uniform over nfan (Flux utils): limit = sqrt(6/(fan_in+fan_out)); the element stream
// is the AGZ_SITE_WEIGHTS site of the draw header, so any consumer of that header (the test
// oracle included) generates the same tensors from the same seed.
static void glorot(std::vector<float>& w, double fan_in, double fan_out, uint64_t seed, int layer) {
  const double limit = std::sqrt(6.0 / (fan_in + fan_out));
  const uint64_t key = (uint64_t)(int64_t)(layer + 4096);
  for (size_t i = 0; i < w.size(); ++i) {
    const double u = agz_u01(agz_draw_u64(seed, key, 0, AGZ_SITE_WEIGHTS, (uint64_t)i));
    w[i] = (float)((2.0 * u - 1.0) * limit);
  }
}

void Net::init_synthetic(uint64_t seed) {
  host_stale_ = false;          // everything is about to be overwritten
  ++param_version_;
  for (int l = 0; l <= 2 * tower_; ++l) {
    ConvHost* c = conv(l);
    const int cin = c->cin, cout = c->cout, k = c->k;
    conv_init(*c, k, cin, cout);
    glorot(c->w, 9.0 * cin, 9.0 * cout, seed, l);
  }
  conv_init(vconv_, 1, kC, 1);
  conv_init(pconv_, 1, kC, 2);
  glorot(vconv_.w, kC, 1, seed, AGZ_L_VALUE_CONV);
  glorot(pconv_.w, kC, 2, seed, AGZ_L_POLICY_CONV);
  dense_init(vfc1_, P_, 256);
  dense_init(vfc2_, 256, 1);
  dense_init(pfc_, 2 * P_, A_);
  glorot(vfc1_.w, vfc1_.in, vfc1_.out, seed, AGZ_L_VALUE_FC1);
  glorot(vfc2_.w, vfc2_.in, vfc2_.out, seed, AGZ_L_VALUE_FC2);
  glorot(pfc_.w, pfc_.in, pfc_.out, seed, AGZ_L_POLICY_FC);
  upload_host_all();
}

// Flux [kw,kh,cin,cout] column-major -> Wt[cout][tap][cin_pad].  NNlib's conv is a TRUE
// convolution: Flux index (a,b) multiplies x[i + 1 - a, j + 1 - b], i.e. tap offset
// (da,db) = (1-a, 1-b); tap = (da+1) + 3*(db+1).  Element idx of the image (0 in the channel padding):
__host__ __device__ inline float direct_image_element(const float* w, int cin, int cin_pad, size_t idx) {
  const int ci = (int)(idx % cin_pad), tap = (int)((idx / cin_pad) % 9), o = (int)(idx / ((size_t)9 * cin_pad));
  const int a = 2 - tap % 3, b = 2 - tap / 3;
  return ci < cin ? w[a + 3 * (b + 3 * (ci + (size_t)cin * o))] : 0.f;
}
static void pack_conv3(const ConvHost& c, int cin_pad, float* out) {          // host restatement (test reference)
  const size_t n = (size_t)c.cout * 9 * cin_pad;
  for (size_t i = 0; i < n; ++i) out[i] = direct_image_element(c.w.data(), c.cin, cin_pad, i);
}
__global__ __launch_bounds__(256) void k_pack_direct(const float* __restrict__ w, long wstride, int cin, int cin_pad, int layers,
                                                     float* __restrict__ out) {
  const long per = (long)kC * 9 * cin_pad, n = per * layers;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
    const long l = t / per;
    out[t] = direct_image_element(w + l * wstride, cin, cin_pad, (size_t)(t - l * per));
  }
}

void launch_pack_direct(const float* d_w, long wstride, int cin, int cin_pad, int layers, float* d_out, hipStream_t s) {
  const long n = (long)kC * 9 * cin_pad * layers;
  hipLaunchKernelGGL(k_pack_direct, dim3((int)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s, d_w, wstride, cin, cin_pad,
                     layers, d_out);
}

// inference BatchNorm folded into the convolution: y = scale * conv + shift, in float64, rounded once
__host__ __device__ inline void bn_affine_one(float gamma, float var, float eps, float beta, float b, float mean, float* scale,
                                              float* shift) {
#pragma clang fp contract(off)
  const double s = (double)gamma / sqrt((double)var + (double)eps);
  *scale = (float)s;
  *shift = (float)((double)beta + s * ((double)b - (double)mean));
}
static void bn_affine(const ConvHost& c, float* scale, float* shift) {        // host restatement (test reference)
  for (int o = 0; o < c.cout; ++o) bn_affine_one(c.gamma[o], c.var[o], c.eps, c.beta[o], c.b[o], c.mean[o], scale + o, shift + o);
}
// conv layer l of a run of `layers` whose small parameters sit `stride` floats apart behind `base` (= the layer's bias: the
// master's order is w, b, beta, gamma, mean, var, eps): scale / shift rows l, and optionally scale * mul into scale2
__global__ __launch_bounds__(256) void k_bn_affine(const float* __restrict__ base, long stride, int layers, int cout,
                                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ scale2,
                                                   float mul) {
  const int n = layers * cout;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
    const int l = t / cout, o = t % cout;
    const float* q = base + l * stride;
    float sc, sh;
    bn_affine_one(q[2 * cout + o], q[4 * cout + o], q[5 * cout], q[cout + o], q[o], q[3 * cout + o], &sc, &sh);      // gamma, var, eps, beta, b, mean
    scale[t] = sc;
    shift[t] = sh;
    if (scale2) scale2[t] = sc * mul;        // (a power of two: exact)
  }
}
// d_head_[776]: value conv w [256], policy conv w [2][256] (Flux [1,1,cin,2] column-major: ci + cin * o), the two folded affines
__global__ void k_pack_head(const float* __restrict__ vw, const float* __restrict__ pw, float* __restrict__ hp) {
  const int c = threadIdx.x;
  hp[c] = vw[c];
  hp[256 + c] = pw[c];
  hp[512 + c] = pw[kC + c];
  if (c < 8) hp[768 + c] = 0.f;
  __syncthreads();
  if (c == 0) {
    const float* q = vw + kC;            // b, beta, gamma, mean, var, eps of the 1-channel BN
    bn_affine_one(q[2], q[4], q[5], q[1], q[0], q[3], hp + 768, hp + 769);
    const float* r = pw + 2 * kC;        // b[2], beta[2], gamma[2], mean[2], var[2], eps
    bn_affine_one(r[4], r[8], r[10], r[2], r[0], r[6], hp + 770, hp + 771);
    bn_affine_one(r[5], r[9], r[10], r[3], r[1], r[7], hp + 772, hp + 773);
  }
}

// Every inference image from the device master, by kernels on stream_: no host work, no synchronisation.  The optional
// images (F(4x4,3x3), fp16, split) are built when the mode that reads them is first used, and again whenever the master changes.
void Net::pack() {
  const int L = 1 + 2 * tower_;
  const float* F = d_flux_.p;
  const size_t w0 = flux_offset(0, AGZ_K_WEIGHT);
  const size_t w1 = tower_ > 0 ? flux_offset(1, AGZ_K_WEIGHT) : 0;
  const long tstride = tower_ > 0 ? (long)(flux_offset(2, AGZ_K_WEIGHT) - w1) : 0;      // tower layers (2 per block) are equally spaced
  if (derived_dirty_) { packed4_ = packed5_ = packed16_ = packed_split_ = false; }
  if (use_wino5() && !packed5_) {
    d_uwino5_.ensure(wino5_weight_floats() * 2 * tower_);
    launch_wino5_pack(F + w1, tstride, 2 * tower_, d_uwino5_.p, stream_);
    packed5_ = true;
  }
  if (precision_ == 2 && tower_ > 0 && !packed_split_) {
    d_uwino_s_.ensure(wino_weight_floats() * 2 * tower_);
    launch_wino_pack(F + w1, tstride, kC, 2 * tower_, d_uwino_s_.p, kWinoStages, true, stream_);
    d_ustem_s_.ensure(wino_weight_floats(kWinoStemStages));
    launch_wino_pack(F + w0, 0, kCinStem, 1, d_ustem_s_.p, kWinoStemStages, true, stream_);
    packed_split_ = true;
  }
  if (use_wino4() && !packed4_) {
    d_uwino4_.ensure(wino4_weight_floats() * 2 * tower_);
    launch_wino4_pack(F + w1, tstride, 2 * tower_, d_uwino4_.p, stream_);
    packed4_ = true;
  }
  if (precision_ == 1 && tower_ > 0 && !packed16_) {
    d_wi16_.ensure(conv16_image_halves() * 2 * tower_);
    launch_conv16_pack(F + w1, tstride, 2 * tower_, d_wi16_.p, stream_);
    packed16_ = true;
  }
  if (!derived_dirty_) return;
  d_wstem_.ensure((size_t)kC * 9 * kCinStemPad);
  launch_pack_direct(F + w0, 0, kCinStem, kCinStemPad, 1, d_wstem_.p, stream_);
  d_scale_.ensure((size_t)L * kC);
  d_shift_.ensure((size_t)L * kC);
  d_scale_s_.ensure((size_t)L * kC);
  // affines: stem row 0, tower layer l row l; the split form's scales (x 1 / (operand scales)) tower first, the stem's last
  hipLaunchKernelGGL(k_bn_affine, dim3(1), dim3(256), 0, stream_, F + flux_offset(0, AGZ_K_BIAS), 0L, 1, kC, d_scale_.p, d_shift_.p,
                     d_scale_s_.p + (size_t)2 * tower_ * kC, wino_split_descale());
  if (tower_ > 0) {
    hipLaunchKernelGGL(k_bn_affine, dim3(2 * tower_), dim3(256), 0, stream_, F + flux_offset(1, AGZ_K_BIAS), tstride, 2 * tower_, kC,
                       d_scale_.p + kC, d_shift_.p + kC, d_scale_s_.p, wino_split_descale());
    d_wtower_.ensure((size_t)kC * 9 * kC * 2 * tower_);
    launch_pack_direct(F + w1, tstride, kC, kC, 2 * tower_, d_wtower_.p, stream_);
    d_uwino_.ensure(wino_weight_floats() * 2 * tower_);
    launch_wino_pack(F + w1, tstride, kC, 2 * tower_, d_uwino_.p, kWinoStages, false, stream_);
    d_ustem_.ensure(wino_weight_floats(kWinoStemStages));
    launch_wino_pack(F + w0, 0, kCinStem, 1, d_ustem_.p, kWinoStemStages, false, stream_);
  }
  d_head_.ensure(776);
  hipLaunchKernelGGL(k_pack_head, dim3(1), dim3(256), 0, stream_, F + flux_offset(AGZ_L_VALUE_CONV, AGZ_K_WEIGHT),
                     F + flux_offset(AGZ_L_POLICY_CONV, AGZ_K_WEIGHT), d_head_.p);
  // the dense layers are read where they are
  d_vfc1w_ = F + flux_offset(AGZ_L_VALUE_FC1, AGZ_K_WEIGHT); d_vfc1b_ = F + flux_offset(AGZ_L_VALUE_FC1, AGZ_K_BIAS);
  d_vfc2w_ = F + flux_offset(AGZ_L_VALUE_FC2, AGZ_K_WEIGHT); d_vfc2b_ = F + flux_offset(AGZ_L_VALUE_FC2, AGZ_K_BIAS);
  d_pfcw_ = F + flux_offset(AGZ_L_POLICY_FC, AGZ_K_WEIGHT); d_pfcb_ = F + flux_offset(AGZ_L_POLICY_FC, AGZ_K_BIAS);
  AGZ_HIP(hipGetLastError());
  derived_dirty_ = false;
}

// (test hook, agz_debug_pack_diff) the device images against the host restatement of the same packs, word for word
long Net::debug_pack_diff(int which) {
  pack();
  sync_host();
  const int L = 1 + 2 * tower_;
  auto diff = [&](const void* dev, const void* host, size_t bytes) {
    std::vector<uint32_t> got((bytes + 3) / 4, 0);
    AGZ_HIP(hipMemcpyAsync(got.data(), dev, bytes, hipMemcpyDeviceToHost, stream_));
    AGZ_HIP(hipStreamSynchronize(stream_));
    const uint32_t* want = static_cast<const uint32_t*>(host);
    long bad = 0;
    for (size_t i = 0; i < bytes / 4; ++i) bad += got[i] != want[i];
    return bad;
  };
  switch (which) {
    case 0: {      // direct images (stem + tower)
      std::vector<float> w((size_t)kC * 9 * kCinStemPad);
      pack_conv3(stem_, kCinStemPad, w.data());
      long bad = diff(d_wstem_.p, w.data(), w.size() * 4);
      const size_t per = (size_t)kC * 9 * kC;
      w.resize(per);
      for (int l = 0; l < 2 * tower_; ++l) { pack_conv3(tconv_[l], kC, w.data()); bad += diff(d_wtower_.p + per * l, w.data(), per * 4); }
      return bad;
    }
    case 1: {      // F(3x3,3x3) images (tower + stem)
      long bad = 0;
      std::vector<float> u(wino_weight_floats());
      for (int l = 0; l < 2 * tower_; ++l) { wino_pack_weights(tconv_[l], u.data()); bad += diff(d_uwino_.p + u.size() * l, u.data(), u.size() * 4); }
      if (tower_ > 0) {
        std::vector<float> us(wino_weight_floats(kWinoStemStages));
        wino_pack_weights(stem_, us.data(), kWinoStemStages);
        bad += diff(d_ustem_.p, us.data(), us.size() * 4);
      }
      return bad;
    }
    case 2: {      // F(4x4,3x3) images
      if (!use_wino4()) return 0;
      long bad = 0;
      std::vector<float> u(wino4_weight_floats());
      for (int l = 0; l < 2 * tower_; ++l) { wino4_pack_weights(tconv_[l], u.data()); bad += diff(d_uwino4_.p + u.size() * l, u.data(), u.size() * 4); }
      return bad;
    }
    case 3: {      // fp16 images (precision f16 selected)
      if (precision_ != 1 || tower_ == 0) return 0;
      long bad = 0;
      std::vector<uint16_t> wi(conv16_image_halves());
      for (int l = 0; l < 2 * tower_; ++l) { conv16_pack_images(tconv_[l], wi.data()); bad += diff(d_wi16_.p + wi.size() * l, wi.data(), wi.size() * 2); }
      return bad;
    }
    case 4: {      // split images + scales (precision f32s selected)
      if (precision_ != 2 || tower_ == 0) return 0;
      long bad = 0;
      std::vector<float> u(wino_weight_floats());
      for (int l = 0; l < 2 * tower_; ++l) { wino_pack_weights_split(tconv_[l], u.data()); bad += diff(d_uwino_s_.p + u.size() * l, u.data(), u.size() * 4); }
      std::vector<float> us(wino_weight_floats(kWinoStemStages));
      wino_pack_weights_split(stem_, us.data(), kWinoStemStages);
      bad += diff(d_ustem_s_.p, us.data(), us.size() * 4);
      std::vector<float> sc((size_t)L * kC), tmp(kC);
      for (int l = 0; l <= 2 * tower_; ++l) {
        bn_affine(l < 2 * tower_ ? tconv_[l] : stem_, sc.data() + (size_t)l * kC, tmp.data());
        for (int o = 0; o < kC; ++o) sc[(size_t)l * kC + o] *= wino_split_descale();
      }
      return bad + diff(d_scale_s_.p, sc.data(), sc.size() * 4);
    }
    case 6: {      // five-pass F(3x3,3x3) images (agz_net_set_winograd(3))
      if (!use_wino5()) return 0;
      long bad = 0;
      std::vector<float> u(wino5_weight_floats());
      for (int l = 0; l < 2 * tower_; ++l) { wino5_pack_weights(tconv_[l], u.data()); bad += diff(d_uwino5_.p + u.size() * l, u.data(), u.size() * 4); }
      return bad;
    }
    case 5: {      // folded affines + head block
      std::vector<float> scale((size_t)L * kC), shift((size_t)L * kC);
      bn_affine(stem_, scale.data(), shift.data());
      for (int l = 0; l < 2 * tower_; ++l) bn_affine(tconv_[l], scale.data() + (size_t)(l + 1) * kC, shift.data() + (size_t)(l + 1) * kC);
      long bad = diff(d_scale_.p, scale.data(), scale.size() * 4) + diff(d_shift_.p, shift.data(), shift.size() * 4);
      std::vector<float> hp(776, 0.f);
      for (int c = 0; c < kC; ++c) { hp[c] = vconv_.w[c]; hp[256 + c] = pconv_.w[c]; hp[512 + c] = pconv_.w[kC + c]; }
      float s[2], t[2];
      bn_affine(vconv_, s, t);
      hp[768] = s[0]; hp[769] = t[0];
      bn_affine(pconv_, s, t);
      hp[770] = s[0]; hp[771] = t[0]; hp[772] = s[1]; hp[773] = t[1];
      return bad + diff(d_head_.p, hp.data(), hp.size() * 4);
    }
    default: return -1;
  }
}

void Net::reserve(int bcap) {
  if (precision_ == 1 && tower_ > 0 && d_ha_.n < ((size_t)std::max(bcap, bcap_) * P_ + 256) * kC) {
    // + one tile of rows: the persistent fp16 convolution stores whole 224-row tiles (agz_conv16.hip)
    const size_t n = ((size_t)std::max(bcap, bcap_) * P_ + 256) * kC;
    d_ha_.alloc(n);
    d_hb_.alloc(n);
    d_ht_.alloc(n);
  }
  if (bcap <= bcap_) return;
  const size_t rows = (size_t)bcap * P_;
  d_a_.alloc(rows * kC);
  d_b_.alloc(rows * kC);
  d_t_.alloc(rows * kC);
  d_vh_.alloc(rows);
  d_ph_.alloc(rows * 2);
  if (tower_ > 0) {
    const size_t vf = std::max(wino_v_floats(bcap, (N_ + 2) / 3), wino4_applies(N_) ? wino4_v_floats(bcap, N_) : (size_t)0);
    d_vimg_.alloc(vf);
    d_vimg2_.alloc(vf);
  }
  bcap_ = bcap;
}

static inline int conv_grid(int bcap, int P) { return ceil_div((long)bcap * P, BM) * (kC / BN); }
int conv3x3_direct_blocks(int bcap, int N) { return conv_grid(bcap, N * N); }

void launch_conv3x3_direct(const float* x, const float* wt, const float* scale, const float* shift, const float* res,
                           float* y, const int* d_count, int bcap, int N, int relu, int cin_pad, hipStream_t s) {
  const int grid = conv_grid(bcap, N * N);
  if (cin_pad == kCinStemPad)
    hipLaunchKernelGGL((k_conv3x3_mfma<kCinStemPad>), dim3(grid), dim3(256), 0, s, x, wt, scale, shift, res, y, d_count, N, relu);
  else
    hipLaunchKernelGGL((k_conv3x3_mfma<kC>), dim3(grid), dim3(256), 0, s, x, wt, scale, shift, res, y, d_count, N, relu);
}

// y[m][n] = sum over the nine tap partials of launch_conv3x3_direct_taps, in tap order, + shift[n]
__global__ __launch_bounds__(256) void k_sum_taps(const float* __restrict__ part, long stride, const float* __restrict__ shift,
                                                   float* __restrict__ y, const int* __restrict__ d_count, int P) {
  const long n4 = (long)(*d_count) * P * (kC / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 s = *reinterpret_cast<const float4*>(part + i * 4);
#pragma unroll
    for (int z = 1; z < 9; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(part + z * stride + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float4 b = *reinterpret_cast<const float4*>(shift + (i * 4) % kC);
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(s.x + b.x, s.y + b.y, s.z + b.z, s.w + b.w);
  }
}

// The direct convolution for batches too small to fill the chip with 128-row tiles (the training step at the
// reference's batch of 32: 42 workgroups): nine times the workgroups, one tap each, partial sums in `part`
// ([9][bcap N^2][256] floats), then a fixed-order sum + shift.  No scale / residual / ReLU.
void launch_conv3x3_direct_taps(const float* x, const float* wt, const float* ones, const float* zeros, const float* shift,
                                float* y, float* part, const int* d_count, int bcap, int N, int cin_pad, hipStream_t s) {
  const int grid = conv_grid(bcap, N * N);
  const long stride = (long)bcap * N * N * kC;
  if (cin_pad == kCinStemPad)
    hipLaunchKernelGGL((k_conv3x3_mfma<kCinStemPad, true>), dim3(grid, 9), dim3(256), 0, s, x, wt, ones, zeros,
                       (const float*)nullptr, part, d_count, N, 0, stride);
  else
    hipLaunchKernelGGL((k_conv3x3_mfma<kC, true>), dim3(grid, 9), dim3(256), 0, s, x, wt, ones, zeros, (const float*)nullptr,
                       part, d_count, N, 0, stride);
  const int g = (int)std::min<long>(((long)bcap * N * N * (kC / 4) + 255) / 256, 2048);
  hipLaunchKernelGGL(k_sum_taps, dim3(g), dim3(256), 0, s, (const float*)part, stride, shift, y, d_count, N * N);
}

// Tower layer chains (agz_net_set_tower_streams): chain 0 runs on stream_, chains 1.. on streams of their own that wait for
// everything enqueued so far (fork) and that stream_ waits for afterwards (join).  With more than one chain the layers
// overlap, so the profile keeps ONE event pair around the whole tower (2 * tower layers).  Returns whether this tower is
// being timed.
static int tower_chunks() {
#ifdef AGZ_TIMING_EXPERIMENTS
  static const int n = getenv("AGZ_TOWER_CHUNKS") ? atoi(getenv("AGZ_TOWER_CHUNKS")) : 0;     // tools/chunks_sweep.sh
  return n;
#else
  return 0;
#endif
}

bool Net::fork_chains(int parts) {
  const bool pt = prof_on_ && prof_n_ < kProfMax;
  if (parts <= 1) return pt;
  if (!ev_fork_) AGZ_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
  for (int i = 0; i + 1 < parts; ++i)
    if (!streamx_[i]) {
      AGZ_HIP(hipStreamCreateWithFlags(&streamx_[i], hipStreamNonBlocking));
      AGZ_HIP(hipEventCreateWithFlags(&ev_join_[i], hipEventDisableTiming));
    }
  if (pt) (void)hipEventRecord(prof_ev_[2 * prof_n_], stream_);
  AGZ_HIP(hipEventRecord(ev_fork_, stream_));
  for (int i = 0; i + 1 < parts; ++i) AGZ_HIP(hipStreamWaitEvent(streamx_[i], ev_fork_, 0));
  return pt;
}

void Net::join_chains(int parts, bool pt) {
  if (parts <= 1) return;
  for (int i = 0; i + 1 < parts; ++i) {          // (never throws: it also runs while an exception is in flight)
    (void)hipEventRecord(ev_join_[i], streamx_[i]);
    (void)hipStreamWaitEvent(stream_, ev_join_[i], 0);
  }
  if (pt) {
    (void)hipEventRecord(prof_ev_[2 * prof_n_ + 1], stream_);
    prof_mult_[prof_n_] = 2 * tower_;
    prof_fwd_of_[prof_n_++] = prof_fwd_;
  }
}

void Net::check_async_error() {
  if (tower_err_ && *tower_err_) {
    const int err = *tower_err_;
    *tower_err_ = 0;
    AGZ_REQUIRE(false, AGZ_HIP_ERROR,
                "the persistent tower kernel of an earlier forward gave up (scheduler error word %d: 1 = more than 32 "
                "workgroups on one XCD, 2 = a tile block's producer never arrived); its outputs were garbage", err);
  }
}

void Net::forward(const float* d_x32, const int* d_count, int bcap, float* d_pi, float* d_v) {
  check_async_error();      // (whichever launch form this forward takes)
  if (use_wino4()) wino4_validate(bcap, N_);      // before anything is enqueued
  pack();
  reserve(bcap);
  const int grid = conv_grid(bcap, P_);
  float *a = d_a_.p, *b = d_b_.p, *t = d_t_.p;
  if (prof_on_ && prof_fwd_ < kProfMax)
    (void)hipMemcpyAsync(&prof_counts_[prof_fwd_], d_count, sizeof(int32_t), hipMemcpyDeviceToHost, stream_);
  // The stem goes through the Winograd GEMM too when the tower does (8 K-loop stages for its 17 -> 32 padded input
  // planes instead of a direct convolution over K = 9 x 32): its epilogue then also emits the first tower layer's V,
  // and the one k_wino_in per forward shrinks to the 32-channel feature planes.
  const bool stem_wino = winograd_ && tower_ > 0;
  if (!stem_wino)
    hipLaunchKernelGGL((k_conv3x3_mfma<kCinStemPad>), dim3(grid), dim3(256), 0, stream_, d_x32, d_wstem_.p,
                       d_scale_.p, d_shift_.p, (const float*)nullptr, a, d_count, N_, 1);
  // every tower-conv launch goes through here so that bench.py's HIP-event roofline leg sees it
  auto timed = [&](auto&& launch) {
    const bool p = prof_on_ && prof_n_ < kProfMax;
    if (p) (void)hipEventRecord(prof_ev_[2 * prof_n_], stream_);
    launch();
    if (p) {
      (void)hipEventRecord(prof_ev_[2 * prof_n_ + 1], stream_);
      prof_mult_[prof_n_] = 1;
      prof_fwd_of_[prof_n_++] = prof_fwd_;
    }
  };
  const float* sc = d_scale_.p + kC;     // affine of tower layer l at + l * kC
  const float* sh = d_shift_.p + kC;
  if (precision_ == 1 && tower_ > 0) {
    // fp16 tower (agz_conv16.hip): half activations between the f32 stem output and the f32 input of
    // the heads; the residual of block 0 is the unrounded f32 stem output
    const size_t iper = conv16_image_halves();
    uint16_t *cur = d_hb_.p, *nxt = d_ha_.p;
    if (stem_wino) {                                 // the f32 stem as an 8-stage Winograd GEMM (y only)
      launch_wino_in(d_x32, d_vimg_.p, d_count, bcap, N_, false, stream_, kWinoStemStages);
      launch_wino_gemm(d_vimg_.p, d_ustem_.p, d_scale_.p, d_shift_.p, nullptr, a, nullptr, d_count, bcap, N_, 1, false, stream_,
                       kWinoStemStages);
    }
    launch_f32_to_f16(a, cur, d_count, bcap, N_, stream_);
    for (int blk = 0; blk < tower_; ++blk) {
      const int l1 = 2 * blk, l2 = 2 * blk + 1;
      const bool first = blk == 0, last = blk + 1 == tower_;
      timed([&] {
        launch_conv16_dma(cur, d_wi16_.p + iper * l1, sc + (size_t)l1 * kC, sh + (size_t)l1 * kC, nullptr, 0, d_ht_.p, 0,
                          d_count, bcap, N_, 1, stream_);
      });
      timed([&] {
        launch_conv16_dma(d_ht_.p, d_wi16_.p + iper * l2, sc + (size_t)l2 * kC, sh + (size_t)l2 * kC,
                          first ? (const void*)a : (const void*)cur, first ? 1 : 0, last ? (void*)b : (void*)nxt,
                          last ? 1 : 0, d_count, bcap, N_, 1, stream_);
      });
      std::swap(cur, nxt);
    }
    a = b;     // the heads read the f32 output of the last block
  } else {
    const size_t per = (size_t)kC * 9 * kC, uper = wino_weight_floats();
    const bool split = precision_ == 2 && winograd_;
    const float* usrc = split ? d_uwino_s_.p : d_uwino_.p;
    if (split) sc = d_scale_s_.p;
    if (winograd_) {
      // Winograd with the input transform of layer l+1 fused into the GEMM of layer l: only the first layer needs a full
      // k_wino_in; conv1 of a block leaves nothing but V in HBM, conv2 leaves the block output (the next residual,
      // and the heads' input) and the next block's V.  Whole-board tile blocks (N <= 12) emit every tile's V from the
      // epilogue.  Dense blocks (19x19) emit the tiles whose 5x5 patch lies inside their block -- three quarters of
      // them -- and a fix-up pass of k_wino_in transforms the rest from y, which every layer therefore writes
      // (round 2 ran the full k_wino_in in front of every layer there: 20 % of a step, MFMA pipe idle).
      const bool dense = !wino_fusable(N_);
      float *vcur = d_vimg_.p, *vnxt = d_vimg2_.p;
      if (use_wino4() && stem_wino) {
        // Boards of 13x13 and larger, exact f32: the tower on F(4x4,3x3) (agz_wino4.hip: 2.49 multiplies per output point
        // at 19x19 against 3.39).  The stem stays an 8-stage F(3x3,3x3) GEMM (y only); one full input transform in front
        // of the first tower layer, then every layer's epilogue emits the next V -- all of it when tile blocks hold
        // whole boards (N = 13..16), the block ends through the fix-up transform otherwise.
        const size_t per4 = wino4_weight_floats();
        const bool dense4 = !wino4_whole_boards(N_);
        launch_wino_in(d_x32, vnxt, d_count, bcap, N_, false, stream_, kWinoStemStages);
        launch_wino_gemm(vnxt, d_ustem_.p, d_scale_.p, d_shift_.p, nullptr, a, nullptr, d_count, bcap, N_, 1, false, stream_,
                         kWinoStemStages);
        launch_wino4_in(a, vcur, d_count, bcap, N_, stream_, false);
        // Two chains on two streams when the batch is large enough to fill the chip twice over (>= 4 workgroup rounds
        // per chain); the chains touch disjoint rows of the same buffers.
        // (a chain should still fill the chip a few times over: at least ~4 workgroup rounds = 256 tile blocks each)
        const long tblocks = ((long)bcap * ((N_ + 3) / 4) * ((N_ + 3) / 4) + 63) / 64;
        const int nst = (int)std::max<long>(1, std::min<long>(tower_streams_, tblocks / 256));
        // (experiment, AGZ_TOWER_CHUNKS: more ranges than streams -- a stream runs its ranges one after the other, each through
        // all layers, so that a range's V might stay in the Infinity Cache between consecutive layers; HISTORY.md 12)
        const int parts = nst <= 1 ? 1 : (int)std::max<long>(nst, std::min<long>(tower_chunks(), tblocks / 64));
        const bool pt = fork_chains(nst);
        try {                          // (a launcher that throws must not leave the side streams un-joined: ADVICE r4)
        for (int part = 0; part < parts; ++part) {
          hipStream_t st = part % nst == 0 ? stream_ : streamx_[part % nst - 1];
          float *pa = a, *pb = b, *vc = vcur, *vn = vnxt;
          for (int blk = 0; blk < tower_; ++blk) {
            const int l1 = 2 * blk, l2 = 2 * blk + 1;
            const bool last = blk + 1 == tower_;
            auto layer1 = [&] {
              // (conv1's y is read by the fix-up transform only: with paired packing it is stored only there)
              launch_wino4_gemm(vc, d_uwino4_.p + per4 * l1, sc + (size_t)l1 * kC, sh + (size_t)l1 * kC, nullptr, dense4 ? t : nullptr,
                                vn, d_count, bcap, N_, 1, st, part, parts, dense4 && wino4_paired(N_));
              if (dense4) launch_wino4_in(t, vn, d_count, bcap, N_, st, true, part, parts);
            };
            auto layer2 = [&] {
              launch_wino4_gemm(vn, d_uwino4_.p + per4 * l2, sc + (size_t)l2 * kC, sh + (size_t)l2 * kC, pa, pb, last ? nullptr : vc,
                                d_count, bcap, N_, 1, st, part, parts);
              if (dense4 && !last) launch_wino4_in(pb, vc, d_count, bcap, N_, st, true, part, parts);
            };
            if (parts == 1) { timed(layer1); timed(layer2); }
            else { layer1(); layer2(); }
            std::swap(pa, pb);
          }
        }
        } catch (...) {
          join_chains(nst, pt);
          throw;
        }
        join_chains(nst, pt);
        if (tower_ % 2) std::swap(a, b);               // the block outputs alternate between a and b
      } else {
      if (stem_wino) {
        launch_wino_in(d_x32, vnxt, d_count, bcap, N_, split, stream_, kWinoStemStages);
        launch_wino_gemm(vnxt, split ? d_ustem_s_.p : d_ustem_.p, split ? d_scale_s_.p + (size_t)2 * tower_ * kC : d_scale_.p,
                         d_shift_.p, nullptr, a, vcur, d_count, bcap, N_, 1, split, stream_, kWinoStemStages);
        if (dense) launch_wino_in(a, vcur, d_count, bcap, N_, split, stream_, kWinoStages, true);
      }
      const bool five = use_wino5() && !dense && stem_wino;      // tower layers on the five-pass 64 x 128 form (agz_wino5.hip)
      const size_t per5 = wino5_weight_floats();
      const bool persistent = tower_persistent_ && !five && !dense && stem_wino && wino_tower_supported(stream_);
      if (persistent) {
        // the same layers as the loop below, as a table for ONE persistent launch (k_wino_tower)
        const int nl = 2 * tower_;
        std::vector<WinoTowerLayer> tab(nl);
        float *pa = a, *pb = b, *vc = vcur, *vn = vnxt;
        for (int blk = 0; blk < tower_; ++blk) {
          const int l1 = 2 * blk, l2 = 2 * blk + 1;
          const bool last = blk + 1 == tower_;
          tab[l1] = {vc, usrc + uper * l1, sc + (size_t)l1 * kC, sh + (size_t)l1 * kC, nullptr, nullptr, vn, 2, 1};
          tab[l2] = {vn, usrc + uper * l2, sc + (size_t)l2 * kC, sh + (size_t)l2 * kC, pa, pb, last ? nullptr : vc, last ? 1 : 3, 1};
          std::swap(pa, pb);
        }
        const size_t tbytes = sizeof(WinoTowerLayer) * tab.size();
        if (tower_layers_host_.size() != tbytes || std::memcmp(tower_layers_host_.data(), tab.data(), tbytes) != 0) {
          AGZ_HIP(hipStreamSynchronize(stream_));      // (rare: first forward, or the workspace moved)
          tower_layers_host_.assign((const char*)tab.data(), (const char*)tab.data() + tbytes);
          d_tower_layers_.ensure(tbytes);
          AGZ_HIP(hipMemcpy(d_tower_layers_.p, tower_layers_host_.data(), tbytes, hipMemcpyHostToDevice));
        }
        d_tower_sched_.ensure(wino_tower_sched_ints(nl, bcap, N_));
        if (!tower_err_) {
          AGZ_HIP(hipHostMalloc((void**)&tower_err_, sizeof(int32_t), hipHostMallocDefault));
          *tower_err_ = 0;
        }
        const bool p = prof_on_ && prof_n_ < kProfMax;
        if (p) (void)hipEventRecord(prof_ev_[2 * prof_n_], stream_);
        launch_wino_tower(d_tower_layers_.p, nl, d_tower_sched_.p, d_count, bcap, N_, split, stream_);
        if (p) {
          (void)hipEventRecord(prof_ev_[2 * prof_n_ + 1], stream_);
          prof_mult_[prof_n_] = nl;
          prof_fwd_of_[prof_n_++] = prof_fwd_;
        }
        (void)hipMemcpyAsync(tower_err_, d_tower_sched_.p + kWinoTowerErrWord, sizeof(int32_t), hipMemcpyDeviceToHost, stream_);
        if (tower_ % 2) std::swap(a, b);               // the block outputs alternate between a and b
      } else {
      // relu(BN2(conv2(relu(BN1(conv1(x))))) + x), resnet.jl:26-32.  With whole-board tile blocks the batch's tile blocks can
      // run as independent layer chains on several streams, like the F(4x4,3x3) tower's (above): agz_net_set_tower_streams,
      // default 2.  Measured on the 9x9 headline (8192 positions, the layer on the board's power limit), alternating runs
      // on one box: 46.80 / 47.00 ms per step with one chain, 46.20 / 46.08 with two (-1.6 %), bit-identical outputs; a later
      // sweep (tools/chains_sweep.sh): 46.8 / 47.7 with one, 46.4 / 46.6 with two, 46.6 / 46.8 with three, 47.0 / 47.0 with four.
      const int chains33 = tower_streams_;
      const long tblocks3 = ((long)bcap * ((N_ + 2) / 3) * ((N_ + 2) / 3) + 62) / 63;
      const int nst = dense ? 1 : (int)std::max<long>(1, std::min<long>(std::min(chains33, (int)kMaxTowerStreams), tblocks3 / 256));
      const int parts = nst <= 1 ? 1 : (int)std::max<long>(nst, std::min<long>(tower_chunks(), tblocks3 / 64));
      const bool pt = fork_chains(nst);
      try {
      for (int part = 0; part < parts; ++part) {
        hipStream_t st = part % nst == 0 ? stream_ : streamx_[part % nst - 1];
        float *pa = a, *pb = b;
        for (int blk = 0; blk < tower_; ++blk) {
          const int l1 = 2 * blk, l2 = 2 * blk + 1;
          const bool last = blk + 1 == tower_;
          auto layer1 = [&] {
            if (five) {
              launch_wino5_gemm(vcur, d_uwino5_.p + per5 * l1, sc + (size_t)l1 * kC, sh + (size_t)l1 * kC, nullptr, nullptr, vnxt,
                                d_count, bcap, N_, 1, st, part, parts);
              return;
            }
            launch_wino_gemm(vcur, usrc + uper * l1, sc + (size_t)l1 * kC, sh + (size_t)l1 * kC, nullptr, dense ? t : nullptr, vnxt,
                             d_count, bcap, N_, 1, split, st, kWinoStages, part, parts);
            if (dense) launch_wino_in(t, vnxt, d_count, bcap, N_, split, st, kWinoStages, true);
          };
          auto layer2 = [&] {
            if (five) {
              launch_wino5_gemm(vnxt, d_uwino5_.p + per5 * l2, sc + (size_t)l2 * kC, sh + (size_t)l2 * kC, pa, pb,
                                last ? nullptr : vcur, d_count, bcap, N_, 1, st, part, parts);
              return;
            }
            launch_wino_gemm(vnxt, usrc + uper * l2, sc + (size_t)l2 * kC, sh + (size_t)l2 * kC, pa, pb,
                             last ? nullptr : vcur, d_count, bcap, N_, 1, split, st, kWinoStages, part, parts);
            if (dense && !last) launch_wino_in(pb, vcur, d_count, bcap, N_, split, st, kWinoStages, true);
          };
          if (parts == 1) { timed(layer1); timed(layer2); }
          else { layer1(); layer2(); }
          std::swap(pa, pb);
        }
      }
      } catch (...) {
        join_chains(nst, pt);
        throw;
      }
      join_chains(nst, pt);
      if (tower_ % 2) std::swap(a, b);               // the block outputs alternate between a and b
      }
      }
    } else {                                         // the direct implicit GEMM (agz_net_set_winograd(0)); the stem ran above
      for (int blk = 0; blk < tower_; ++blk) {
        const int l1 = 2 * blk, l2 = 2 * blk + 1;
        timed([&] {
          hipLaunchKernelGGL((k_conv3x3_mfma<kC>), dim3(grid), dim3(256), 0, stream_, (const float*)a, d_wtower_.p + per * l1,
                             sc + (size_t)l1 * kC, sh + (size_t)l1 * kC, (const float*)nullptr, t, d_count, N_, 1);
        });
        timed([&] {
          hipLaunchKernelGGL((k_conv3x3_mfma<kC>), dim3(grid), dim3(256), 0, stream_, (const float*)t, d_wtower_.p + per * l2,
                             sc + (size_t)l2 * kC, sh + (size_t)l2 * kC, (const float*)a, b, d_count, N_, 1);
        });
        std::swap(a, b);
      }
    }
  }
  const int hgrid = std::min(ceil_div((long)bcap * P_, 4), 256 * 16);
  hipLaunchKernelGGL(k_head_conv, dim3(hgrid), dim3(256), 0, stream_, (const float*)a, d_head_.p, d_vh_.p,
                     d_ph_.p, d_count, P_);
  const size_t smem = sizeof(float) * (size_t)(3 * P_ + A_ + 4);
  hipLaunchKernelGGL(k_head_fc, dim3(bcap), dim3(256), smem, stream_, (const float*)d_vh_.p,
                     (const float*)d_ph_.p, d_vfc1w_, d_vfc1b_, d_vfc2w_, d_vfc2b_, d_pfcw_,
                     d_pfcb_, d_pi, d_v, d_count, P_, A_);
  if (prof_on_ && prof_fwd_ < kProfMax) prof_fwd_++;
  AGZ_HIP(hipGetLastError());
}

// registers only: no memory, no LDS -- the matrix pipes of every SIMD busy, and the clock wherever the power limit puts it
__global__ __launch_bounds__(256) void k_mfma_f32_sustained(int iters, float* out) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const float a = 1.f + threadIdx.x * 1e-6f, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  if (s == 123.456f) out[0] = s;
}

// The same with operands that CHANGE like a layer's do: 16 pseudo-random A and B values per lane, a different pair in front
// of every MFMA.  A matrix pipe's power follows the toggling of its operands (MI355X_MICROARCH.md: zero-filled inputs ran
// +19 % TF/s at the same counters; round 5 measured the fp16 tower's MFMAs alone at 0.74 ms on constants and 0.92 ms on real
// data), so the constant-operand rate above flatters the ceiling.  F16: v_mfma_f32_32x32x16_f16 instead of the f32 MFMA.
template <bool F16, bool DENSE>
__global__ __launch_bounds__(256) void k_mfma_sustained_data(int iters, float* out) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  unsigned st = 0x9e3779b9u * (threadIdx.x + 256u * blockIdx.x + 1u);
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f; };   // [-1, 1)
  float a32[16], b32[16];
  h8 a16[16], b16[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // B like a post-ReLU activation (half of it zero: the fp16 tower's operand), or DENSE (a Winograd-transformed one)
    a32[k] = rnd(); b32[k] = DENSE ? rnd() : fmaxf(rnd(), 0.f) * 2.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a16[k][e] = (_Float16)(0.06f * rnd());
      b16[k][e] = (_Float16)(DENSE ? rnd() : fmaxf(rnd(), 0.f) * 2.f);
    }
  }
  for (int it = 0; it < iters; it += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[(i + u) & 15], b16[(3 * i + 5 * u) & 15], acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[(i + u) & 15], b32[(3 * i + 5 * u) & 15], acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  if (s == 123.456f) out[0] = s;
}

// mode 0: constant f32 operands (agz_debug_mfma_sustained); 1 / 2: changing f32 / fp16 (32x32x16) operands, B half zero;
// 3 / 4: the same with a dense B
float Net::mfma_sustained_tflops(int millis, int mode) {
  AGZ_REQUIRE(millis >= 50 && millis <= 5000, AGZ_BAD_ARGUMENT, "mfma_sustained_tflops: %d ms (50..5000)", millis);
  AGZ_REQUIRE(mode >= 0 && mode <= 4, AGZ_BAD_ARGUMENT, "mfma_sustained_tflops: mode %d", mode);
  const bool f16 = mode == 2 || mode == 4;
  DevBuf<float> out;
  out.alloc(16);
  hipEvent_t e0, e1;
  AGZ_HIP(hipEventCreate(&e0));
  AGZ_HIP(hipEventCreate(&e1));
  const int iters = f16 ? 20000 : 10000, grid = 1024;      // 4 workgroups of 4 waves per CU; ~11 ms per launch
  const double flop = (double)grid * 4 * iters * 8.0 * (f16 ? 32768.0 : 4096.0);
  std::vector<double> tf;
  double spent = 0.0;
  while (spent < millis) {
    AGZ_HIP(hipEventRecord(e0, stream_));
    if (mode == 0) hipLaunchKernelGGL(k_mfma_f32_sustained, dim3(grid), dim3(256), 0, stream_, iters, out.p);
    else if (mode == 1) hipLaunchKernelGGL((k_mfma_sustained_data<false, false>), dim3(grid), dim3(256), 0, stream_, iters, out.p);
    else if (mode == 2) hipLaunchKernelGGL((k_mfma_sustained_data<true, false>), dim3(grid), dim3(256), 0, stream_, iters, out.p);
    else if (mode == 3) hipLaunchKernelGGL((k_mfma_sustained_data<false, true>), dim3(grid), dim3(256), 0, stream_, iters, out.p);
    else hipLaunchKernelGGL((k_mfma_sustained_data<true, true>), dim3(grid), dim3(256), 0, stream_, iters, out.p);
    AGZ_HIP(hipEventRecord(e1, stream_));
    AGZ_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    AGZ_HIP(hipEventElapsedTime(&ms, e0, e1));
    tf.push_back(flop / (ms * 1e-3) / 1e12);
    spent += ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  std::vector<double> tail(tf.begin() + tf.size() / 2, tf.end());
  std::sort(tail.begin(), tail.end());
  return (float)tail[tail.size() / 2];
}

Net::~Net() {
  for (auto e : prof_ev_) (void)hipEventDestroy(e);
  if (prof_counts_) (void)hipHostFree(prof_counts_);
  if (tower_err_) (void)hipHostFree(tower_err_);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  for (auto e : ev_join_)
    if (e) (void)hipEventDestroy(e);
  for (auto st : streamx_)
    if (st) (void)hipStreamDestroy(st);
}

void Net::profile_enable(bool on) {
  if (on && prof_ev_.empty()) {
    prof_ev_.resize(2 * kProfMax);
    for (auto& e : prof_ev_) AGZ_HIP(hipEventCreate(&e));
    prof_fwd_of_.assign(kProfMax, 0);
    prof_mult_.assign(kProfMax, 1);
    AGZ_HIP(hipHostMalloc((void**)&prof_counts_, sizeof(int32_t) * kProfMax, hipHostMallocDefault));
  }
  prof_on_ = on;
  prof_n_ = 0;
  prof_fwd_ = 0;
}

void Net::profile_read(double* total_ms, double* total_flop, int64_t* launches) {
  AGZ_HIP(hipStreamSynchronize(stream_));
  double ms = 0.0, fl = 0.0;
  int64_t n = 0;
  for (int i = 0; i < prof_n_; ++i) {
    float t = 0.f;
    AGZ_HIP(hipEventElapsedTime(&t, prof_ev_[2 * i], prof_ev_[2 * i + 1]));
    ms += t;
    const int f = prof_fwd_of_[i];
    fl += prof_mult_[i] * conv_flops_per_launch(f < kProfMax ? prof_counts_[f] : 0);
    n += prof_mult_[i];         // the persistent tower launch counts as its layers: `launches` stays "tower layers timed"
  }
  *total_ms = ms;
  *total_flop = fl;
  *launches = n;
}

void Net::launch_tower_conv_once(const int* d_count, int bcap) {
  pack();
  reserve(bcap);
  AGZ_REQUIRE(tower_ > 0, AGZ_BAD_ARGUMENT, "no tower conv in a tower_height=0 network");
  const int grid = conv_grid(bcap, P_);
  if (precision_ == 1)
    launch_conv16_dma(d_ha_.p, d_wi16_.p, d_scale_.p + kC, d_shift_.p + kC, nullptr, 0, d_ht_.p, 0, d_count, bcap, N_, 1,
                      stream_);
  else if (use_wino4()) {
    launch_wino4_gemm(d_vimg_.p, d_uwino4_.p, d_scale_.p + kC, d_shift_.p + kC, d_a_.p, d_t_.p, d_vimg2_.p, d_count, bcap, N_, 1, stream_);
    if (!wino4_whole_boards(N_)) launch_wino4_in(d_t_.p, d_vimg2_.p, d_count, bcap, N_, stream_, true);
  }
  else if (use_wino5())                       // (time_conv: a steady-state layer of the five-pass form)
    launch_wino5_gemm(d_vimg_.p, d_uwino5_.p, d_scale_.p + kC, d_shift_.p + kC, d_a_.p, d_t_.p, d_vimg2_.p, d_count, bcap, N_, 1, stream_);
  else if (winograd_ && wino_fusable(N_)) {   // a steady-state tower layer: residual in, y and the next V out
    const bool split = precision_ == 2;
    launch_wino_gemm(d_vimg_.p, split ? d_uwino_s_.p : d_uwino_.p, split ? d_scale_s_.p : d_scale_.p + kC, d_shift_.p + kC,
                     d_a_.p, d_t_.p, d_vimg2_.p, d_count, bcap, N_, 1, split, stream_);
  } else if (winograd_) {                       // dense tile blocks: the GEMM emits most of the next V, the fix-up pass the rest
    const bool split = precision_ == 2;
    launch_wino_gemm(d_vimg_.p, split ? d_uwino_s_.p : d_uwino_.p, split ? d_scale_s_.p : d_scale_.p + kC, d_shift_.p + kC,
                     d_a_.p, d_t_.p, d_vimg2_.p, d_count, bcap, N_, 1, split, stream_);
    launch_wino_in(d_t_.p, d_vimg2_.p, d_count, bcap, N_, split, stream_, kWinoStages, true);
  }
  else
  hipLaunchKernelGGL((k_conv3x3_mfma<kC>), dim3(grid), dim3(256), 0, stream_, (const float*)d_a_.p,
                     d_wtower_.p, d_scale_.p + kC, d_shift_.p + kC, (const float*)nullptr, d_t_.p, d_count, N_, 1);
  AGZ_HIP(hipGetLastError());
}

double Net::flops_per_eval() const {
  // F_eval(N,t)
  const double P = P_;
  return 2.0 * P * (9.0 * 17 * 256 + tower_ * 2.0 * 9 * 256 * 256) + 2.0 * P * 256 * 3 +
         2.0 * (2.0 * P * (P + 1) + P * 256 + 256);
}

}  // namespace agz
