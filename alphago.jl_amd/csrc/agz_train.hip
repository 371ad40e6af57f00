// agz_train.hip -- one optimisation step of the policy/value network on the device.
//
// Reference: `_train` and its three losses, /root/reference/src/neural_net.jl:75-101; the optimiser
// `Momentum(2f-2)`, /root/reference/src/train.jl:54; the call, train.jl:67-74.  `_train` is broken at the
// reference's HEAD (undefined `loss_avg`, un-imported `update!`, `params(nn)` of a struct that is not a functor,
// a B-vector minus a 1 x B matrix in `loss_value`), so what is restated here is the INTENDED step:
//
//   p, v  = nn(positions, train = true)            BatchNorm normalises with the batch's own statistics and
//                                                  moves its running statistics by momentum 0.1 (Flux BatchNorm)
//   loss  = 0.01 * (-sum(pi .* log(p)) / B)        loss_pi     = crossentropy(p, pi; weight = 0.01)
//         + 0.01 * mean((v - z)^2)                 loss_value  = 0.01 * mse(z, v)
//         + 1e-4 * sum over every parameter of theta^2          loss_reg
//   back!(loss); update!(Momentum(eta = 0.02, rho = 0.9), params):   vel = rho vel - eta grad;  theta += vel
//
// It is pinned by a torch float64 autograd twin of the same network (tests/test_gpu_train.py), not by the
// reference ("parity unpinned": the reference cannot run this step).
//
// Layout: activations [M = B P][C] f32, channel fastest, exactly as the inference path; the 3x3 convolutions
// (forward and input gradient) run on the same f32-MFMA implicit GEMM as inference (k_conv3x3_mfma with an
// identity affine); everything else -- batch statistics, BatchNorm forward/backward, weight gradients, the heads,
// the optimiser -- is plain HIP: at the reference's batch of 32 positions a step is a few GFLOP, and this row is
// about having the step on the device with the right numbers, not about its roofline.
#include <cmath>
#include <cstring>

#include "agz_nn.h"

namespace agz {

namespace {

constexpr float kLossPiW = 0.01f, kLossVW = 0.01f, kRegW = 1e-4f, kBnMomentum = 0.1f;

// ---- column statistics: sums[c] = sum_m u[m][c], sums[C + c] = sum_m u[m][c]^2 (double, atomics over row slices)
__global__ __launch_bounds__(256) void k_colsums(const float* __restrict__ u, long M, int C, double* __restrict__ sums) {
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0, s2 = 0.0;
  if (c < C)
    for (long m = (long)blockIdx.y * 4 + rl; m < M; m += (long)gridDim.y * 4) {
      const double v = u[m * C + c];
      s += v;
      s2 += v * v;
    }
  red[0][rl][cl] = s;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    atomicAdd(&sums[c], red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
    atomicAdd(&sums[C + c], red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
  }
}

// BatchNorm forward with batch statistics: out = act(gamma (u - mean) rstd + beta (+ res)); stats = {mean, var, rstd}[C]
__global__ void k_bn_fwd(const float* __restrict__ u, long M, int C, const double* __restrict__ sums,
                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                         const float* __restrict__ res, float* __restrict__ out, int relu, float* __restrict__ stats) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const int c = (int)(i % C);
  const double mean = sums[c] / (double)M;
  const double var = fmax(sums[C + c] / (double)M - mean * mean, 0.0);
  const double rstd = 1.0 / sqrt(var + (double)eps);
  if (i < C) { stats[c] = (float)mean; stats[C + c] = (float)var; stats[2 * C + c] = (float)rstd; }
  float v = (float)(((double)u[i] - mean) * rstd) * gamma[c] + beta[c];
  if (res) v += res[i];
  if (relu) v = fmaxf(v, 0.f);
  out[i] = v;
}

// BatchNorm backward, reduction: with dy = dout (.* [out > 0] if relu): sums[c] = sum dy, sums[C + c] = sum dy xhat
__global__ __launch_bounds__(256) void k_bn_bwd_sums(const float* __restrict__ dout, const float* __restrict__ out,
                                                      const float* __restrict__ u, const float* __restrict__ stats,
                                                      long M, int C, int relu, double* __restrict__ sums) {
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0, s2 = 0.0;
  if (c < C) {
    const double mean = stats[c], rstd = stats[2 * C + c];
    for (long m = (long)blockIdx.y * 4 + rl; m < M; m += (long)gridDim.y * 4) {
      const long i = m * C + c;
      const double dy = (relu && !(out[i] > 0.f)) ? 0.0 : (double)dout[i];
      s += dy;
      s2 += dy * ((double)u[i] - mean) * rstd;
    }
  }
  red[0][rl][cl] = s;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    atomicAdd(&sums[c], red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
    atomicAdd(&sums[C + c], red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
  }
}

// du = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)); dres (optional) = dy: the shortcut's share of a block output
__global__ void k_bn_bwd_apply(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ u,
                               const float* __restrict__ stats, const float* __restrict__ gamma,
                               const double* __restrict__ sums, long M, int C, int relu, float* __restrict__ du,
                               float* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const int c = (int)(i % C);
  const double mean = stats[c], rstd = stats[2 * C + c];
  const double dy = (relu && !(out[i] > 0.f)) ? 0.0 : (double)dout[i];
  const double xhat = ((double)u[i] - mean) * rstd;
  du[i] = (float)((double)gamma[c] * rstd * (dy - sums[c] / (double)M - xhat * sums[C + c] / (double)M));
  if (dres) dres[i] = (float)dy;
  if (i < C) { dgamma[c] = (float)sums[C + c]; dbeta[c] = (float)sums[c]; }
}

// out[c] = (float) sums[c]  (bias gradients = column sums of du)
__global__ void k_take_sums(const double* __restrict__ sums, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = (float)sums[c];
}

// 3x3 weight gradient in the packed layout of the forward kernel: dWt[co][tap][ci] = sum_m du[m][co] x[m + shift(tap)][ci]
// as a GEMM whose reduction runs over the rows m, on v_mfma_f32_32x32x2_f32 with both operands straight from global
// memory: for one row pair an A operand is du[m + k][co0 + lane % 32] and a B operand x[m + k + shift][ci0 + lane % 32]
// (k = lane / 32) -- two 128-byte runs each.  grid (256/32, 9, CIN/32), 256 threads: a 32 x 32 (co x ci) tile per block,
// its four waves take the row pairs 4 i + wave and add their partial tiles through LDS in wave order (deterministic).
// (Round 2's first version was a scalar outer product with double accumulators: 70 % of the training step.  The f32
// MFMA accumulates B N^2 <= 21 k products per output in f32: relative error ~1e-5 against the 2e-3 bar of
// tests/test_gpu_train.py.)
// Large batches split the rows over blockIdx.x / 8 (chunks of `mchunk` rows, a multiple of 8): each split writes its own
// partial gradient at dwt + split * 256 * 9 * CIN and k_sum_parts adds them in order.
__global__ __launch_bounds__(256) void k_wgrad3x3(const float* __restrict__ x, const float* __restrict__ du, int B, int N,
                                                   int CIN, float* __restrict__ dwt, int mchunk) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  __shared__ float part[3][16][64];
  const int P = N * N;
  const int split = blockIdx.x >> 3;
  const int mlo = split * mchunk;
  const int M = min(B * P, mlo + mchunk);                  // < 2^31; this block's rows are [mlo, M)
  dwt += (size_t)split * kC * 9 * CIN;
  const int co0 = (blockIdx.x & 7) * 32, tap = blockIdx.y, ci0 = blockIdx.z * 32;
  const int da = tap % 3 - 1, db = tap / 3 - 1, sh = da + N * db;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  int m = mlo + 2 * wave + kh;                             // this lane's first row; then every eighth
  int p = m % P, ri = p % N, cj = p / N;
  const int dr = 8 % N, dc = 8 / N;
  const float* dcol = du + co0 + l31;
  const float* xcol = x + ci0 + l31;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int iters = (M - mlo + 7) / 8;                     // wave-uniform: MFMA needs the whole wave
  // Branch-free (rows past the end and off-board neighbours load row 0 and are zeroed by a select) and in groups of
  // 16 row pairs with all 32 loads issued before the first MFMA: left to itself hipcc waits for each pair's loads in
  // front of its MFMA -- one L2 round trip per 64-cycle MFMA, 0.35 ms per layer at batch 32.
  constexpr int U = 16;
  for (int it = 0; it < iters; it += U) {
    float av[U], bv[U];
    bool ina[U], okb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ina[u] = m < M;
      okb[u] = ina[u] && (unsigned)(ri + da) < (unsigned)N && (unsigned)(cj + db) < (unsigned)N;
      av[u] = dcol[(size_t)(ina[u] ? m : 0) * kC];
      bv[u] = xcol[(size_t)(okb[u] ? m + sh : 0) * CIN];
      m += 8;
      ri += dr;
      cj += dc;
      if (ri >= N) { ri -= N; ++cj; }
      if (cj >= N) cj -= N;
      if (cj >= N) cj -= N;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ina[u] ? av[u] : 0.f, okb[u] ? bv[u] : 0.f, acc, 0, 0, 0);
  }
  // C/D map: col = lane & 31 (ci), row = (e & 3) + 8 (e >> 2) + 4 kh (co)
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) part[wave - 1][e][lane] = acc[e];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = ((acc[e] + part[0][e][lane]) + part[1][e][lane]) + part[2][e][lane];
      const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
      dwt[((long)(co0 + row) * 9 + tap) * CIN + ci0 + l31] = v;
    }
  }
}

// out[i] = part[0][i] + part[1][i] + ... in order (the row splits of k_wgrad3x3)
__global__ __launch_bounds__(256) void k_sum_parts(const float* __restrict__ part, int nparts, long stride, float* __restrict__ out,
                                                    long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float s = part[i];
    for (int z = 1; z < nparts; ++z) s += part[z * stride + i];
    out[i] = s;
  }
}

// ---- heads, forward (raw, before BatchNorm): cv[m] = x[m] . wv + bv;  cp[m][j] = x[m] . wp[j] + bp[j]
__global__ __launch_bounds__(256) void k_head1x1_fwd(const float* __restrict__ x, long M, const float* __restrict__ wv,
                                                      const float* __restrict__ bv, const float* __restrict__ wp,
                                                      const float* __restrict__ bp, float* __restrict__ cv,
                                                      float* __restrict__ cp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
    double dv = 0.0, d0 = 0.0, d1 = 0.0;
    for (int c = lane; c < kC; c += 64) {
      const double xv = x[m * kC + c];
      dv += xv * wv[c];
      d0 += xv * wp[c];
      d1 += xv * wp[kC + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      dv += __shfl_xor(dv, o, 64);
      d0 += __shfl_xor(d0, o, 64);
      d1 += __shfl_xor(d1, o, 64);
    }
    if (lane == 0) {
      cv[m] = (float)(dv + bv[0]);
      cp[m * 2 + 0] = (float)(d0 + bp[0]);
      cp[m * 2 + 1] = (float)(d1 + bp[1]);
    }
  }
}

// Dense forward: out[b][o] = act(sum_i W[o + O i] in[b][i] + bias[o]); act 0 none, 1 relu, 2 tanh.  in index map:
// in_mode 0: in[b*I + i]; 1: value head hv[(b*P + i)]; 2: policy head hp[((b*P + p)*2 + c)] with i = p + P c
__device__ __forceinline__ long in_index(int mode, int b, int i, int I, int P) {
  if (mode == 2) return ((long)b * P + (i % P)) * 2 + (i / P);
  return (long)b * I + i;
}
__global__ void k_dense_fwd(const float* __restrict__ in, int in_mode, int P, const float* __restrict__ W,
                            const float* __restrict__ bias, int B, int I, int O, int act, float* __restrict__ out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (o >= O) return;
  double s = bias[o];
  for (int i = 0; i < I; ++i) s += (double)W[o + (long)O * i] * (double)in[in_index(in_mode, b, i, I, P)];
  float v = (float)s;
  if (act == 1) v = fmaxf(v, 0.f);
  if (act == 2) v = tanhf(v);
  out[(long)b * O + o] = v;
}

// softmax + the two data losses + the gradients at the network outputs.
//   dlogit[b][a] = w_pi / B * (p_a * sum(pi) - pi_a);  ds[b] = 2 w_v / B * (v - z) * (1 - v^2)   (through tanh)
// losses[0] += -w_pi / B * sum pi log p;  losses[1] += w_v / B * (v - z)^2
__global__ __launch_bounds__(256) void k_outputs(const float* __restrict__ logits, const float* __restrict__ v,
                                                  const float* __restrict__ pi, const float* __restrict__ z, int B, int A,
                                                  float* __restrict__ p_out, float* __restrict__ dlogit,
                                                  float* __restrict__ ds, double* __restrict__ losses) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double mx = -1e300;
  for (int a = tid; a < A; a += 256) mx = fmax(mx, (double)logits[(long)b * A + a]);
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmax(red[tid], red[tid + o]); __syncthreads(); }
  mx = red[0];
  __syncthreads();
  double se = 0.0, spi = 0.0;
  for (int a = tid; a < A; a += 256) { se += exp((double)logits[(long)b * A + a] - mx); spi += pi[(long)b * A + a]; }
  red[tid] = se;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  se = red[0];
  __syncthreads();
  red[tid] = spi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  spi = red[0];
  __syncthreads();
  double lp = 0.0;
  for (int a = tid; a < A; a += 256) {
    const double lg = (double)logits[(long)b * A + a] - mx - log(se);      // log p
    const double p = exp(lg), t = pi[(long)b * A + a];
    p_out[(long)b * A + a] = (float)p;
    dlogit[(long)b * A + a] = (float)((double)kLossPiW / B * (p * spi - t));
    if (t != 0.0) lp += t * lg;
  }
  red[tid] = lp;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) {
    const double dv = (double)v[b] - (double)z[b];
    atomicAdd(&losses[0], -(double)kLossPiW / B * red[0]);
    atomicAdd(&losses[1], (double)kLossVW / B * dv * dv);
    ds[b] = (float)(2.0 * kLossVW / B * dv * (1.0 - (double)v[b] * (double)v[b]));
  }
}

// Dense backward.  dW[o + O i] = sum_b dout[b][o] in[b][i];  db[o] = sum_b dout[b][o]
__global__ void k_dense_wgrad(const float* __restrict__ in, int in_mode, int P, const float* __restrict__ dout, int B, int I,
                              int O, float* __restrict__ dW, float* __restrict__ dbias) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)O * I) return;
  const int o = (int)(idx % O), i = (int)(idx / O);
  double s = 0.0, sb = 0.0;
  for (int b = 0; b < B; ++b) {
    const double d = dout[(long)b * O + o];
    s += d * (double)in[in_index(in_mode, b, i, I, P)];
    sb += d;
  }
  dW[idx] = (float)s;
  if (i == 0) dbias[o] = (float)sb;
}
// din[b][i] = sum_o W[o + O i] dout[b][o]  (* [out_prev > 0] if the producing layer ended in a ReLU)
__global__ void k_dense_dgrad(const float* __restrict__ W, const float* __restrict__ dout, int B, int I, int O,
                              const float* __restrict__ act_out, int in_mode, int P, float* __restrict__ din) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= I) return;
  double s = 0.0;
  for (int o = 0; o < O; ++o) s += (double)W[o + (long)O * i] * (double)dout[(long)b * O + o];
  const long j = in_index(in_mode, b, i, I, P);
  if (act_out && !(act_out[in_mode == 0 ? (long)b * I + i : j] > 0.f)) s = 0.0;
  din[j] = (float)s;
}

// heads, backward of the 1x1 convolutions: dx[m][c] = dcv[m] wv[c] + dcp[m][0] wp[c] + dcp[m][1] wp[256 + c]
__global__ void k_head1x1_dgrad(const float* __restrict__ dcv, const float* __restrict__ dcp, const float* __restrict__ wv,
                                const float* __restrict__ wp, long M, float* __restrict__ dx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * kC) return;
  const long m = i / kC;
  const int c = (int)(i % kC);
  dx[i] = dcv[m] * wv[c] + dcp[m * 2] * wp[c] + dcp[m * 2 + 1] * wp[kC + c];
}
// dwv[c] = sum_m dcv[m] x[m][c]; dwp[j][c]; dbv, dbp.  One thread per channel, the rows in gridDim.x slices whose double
// partial sums meet in sums[0..3 kC + 3) (zeroed by the caller; one block over all rows was 10 % of a batch-32 step)
__global__ __launch_bounds__(256) void k_head1x1_wgrad(const float* __restrict__ x, const float* __restrict__ dcv,
                                                        const float* __restrict__ dcp, long M, double* __restrict__ sums) {
  const int c = threadIdx.x;
  const long per = (M + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < M ? lo + per : M;
  double sv = 0.0, s0 = 0.0, s1 = 0.0, bv = 0.0, b0 = 0.0, b1 = 0.0;
  for (long m = lo; m < hi; ++m) {
    const double xv = x[m * kC + c];
    sv += xv * dcv[m];
    s0 += xv * dcp[m * 2];
    s1 += xv * dcp[m * 2 + 1];
    if (c == 0) { bv += dcv[m]; b0 += dcp[m * 2]; b1 += dcp[m * 2 + 1]; }
  }
  atomicAdd(&sums[c], sv);
  atomicAdd(&sums[kC + c], s0);
  atomicAdd(&sums[2 * kC + c], s1);
  if (c == 0) { atomicAdd(&sums[3 * kC], bv); atomicAdd(&sums[3 * kC + 1], b0); atomicAdd(&sums[3 * kC + 2], b1); }
}
__global__ __launch_bounds__(256) void k_head1x1_wgrad_take(const double* __restrict__ sums, float* __restrict__ dwv,
                                                             float* __restrict__ dwp, float* __restrict__ dbv, float* __restrict__ dbp) {
  const int c = threadIdx.x;
  dwv[c] = (float)sums[c];
  dwp[c] = (float)sums[kC + c];
  dwp[kC + c] = (float)sums[2 * kC + c];
  if (c == 0) { dbv[0] = (float)sums[3 * kC]; dbp[0] = (float)sums[3 * kC + 1]; dbp[1] = (float)sums[3 * kC + 2]; }
}

__global__ void k_add(float* __restrict__ a, const float* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}

// Momentum (Flux): g = grad + 2 * 1e-4 * theta; vel = rho vel - eta g; theta += vel -- every parameter array in one
// launch: blockIdx.x -> (array, chunk of kOptChunk elements) through a work list built once per parameter set; sum theta^2 of the values BEFORE the update (loss_reg) goes to losses[2] on the way
struct OptArray { float* theta; const float* grad; float* vel; long n; };
struct OptWork { int array; int chunk; };
constexpr int kOptChunk = 16384;
__global__ __launch_bounds__(256) void k_momentum_all(const OptArray* __restrict__ arrays, const OptWork* __restrict__ work,
                                                       float eta, float rho, double* __restrict__ sumsq) {
  __shared__ double red[256];
  const OptWork wk = work[blockIdx.x];
  const OptArray a = arrays[wk.array];
  const long lo = (long)wk.chunk * kOptChunk, hi = lo + kOptChunk < a.n ? lo + kOptChunk : a.n;
  double s = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float th = a.theta[i];
    s += (double)th * (double)th;
    const float g = a.grad[i] + 2.f * kRegW * th;
    const float v = rho * a.vel[i] - eta * g;
    a.vel[i] = v;
    a.theta[i] = th + v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(sumsq, red[0]);
}

// input-gradient weights of a 256 -> 256 layer: Wd[ci][tap'][co] = Wt[co][8 - tap'][ci]  (shift(8 - tap) = -shift(tap))
__global__ void k_make_wd(const float* __restrict__ wt, float* __restrict__ wd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)kC * 9 * kC) return;
  const int co = (int)(i % kC), tp = (int)((i / kC) % 9), ci = (int)(i / (9 * kC));
  wd[i] = wt[((long)co * 9 + (8 - tp)) * kC + ci];
}

inline dim3 g1(long n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

}  // namespace

// ------------------------------------------------------------------ the trainer

struct Trainer::Param {
  DevBuf<float> theta, grad, vel;
  size_t n = 0;
};

Trainer::Trainer(Net& net, hipStream_t s) : net_(net), stream_(s) {}
Trainer::~Trainer() {}

// The epsilon of TRAINING-mode BatchNorm.  A checkpoint's epsilon is the one its inference statistics were folded
// with (bson_weights stores 0 for the Flux <= 0.7 dumps the reference ships: their sigma already contains it), and
// 1 / sqrt(batch variance + 0) is inf on a dead channel -- 0 * inf = NaN, which Momentum then spreads to every
// parameter.  Training never goes below Flux's default 1e-5 (BatchNorm(...; eps = 1f-5)).
static inline float train_eps(float eps) { return eps > 1e-5f ? eps : 1e-5f; }

void Trainer::reset() { have_vel_ = false; }

// Flux [3][3][cin][256] <- Wt[cout][tap][cin_pad]: the inverse of the direct image (agz_nn.hip: direct_image_element)
__global__ __launch_bounds__(256) void k_unpack_direct(const float* __restrict__ wt, int cin, int cin_pad, float* __restrict__ w) {
  const long n = (long)9 * cin * kC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int a = (int)(i % 3), b = (int)((i / 3) % 3), ci = (int)((i / 9) % cin), o = (int)(i / ((long)9 * cin));
    w[i] = wt[((long)o * 9 + (2 - a) + 3 * (2 - b)) * cin_pad + ci];
  }
}
// Flux BatchNorm's running statistics: (1 - 0.1) old + 0.1 batch, the variance with the m / (m - 1) correction
__global__ void k_running_stats(float* __restrict__ mean, float* __restrict__ var, const float* __restrict__ bmean,
                                const float* __restrict__ bvar, int n, float corr) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  mean[o] = (1.f - kBnMomentum) * mean[o] + kBnMomentum * bmean[o];
  var[o] = (1.f - kBnMomentum) * var[o] + kBnMomentum * bvar[o] * corr;
}

// The network's device master (Flux layouts, Net::flux_device) -> training layouts, device to device.  Convolutions:
// Wt[cout][tap][cin_pad], the layout of the forward kernel (true-convolution flip applied); everything else as it is.
void Trainer::upload() {
  // the training copies are current unless somebody else wrote the master since the last step
  if (!params_.empty() && uploaded_version_ == net_.param_version()) {
    if (!have_vel_)
      for (auto& p : params_) AGZ_HIP(hipMemsetAsync(p->vel.p, 0, sizeof(float) * p->n, stream_));
    have_vel_ = true;
    return;
  }
  net_.sync_host();             // (the BatchNorm epsilons are read from the host copies)
  uploaded_version_ = net_.param_version();
  const int t = net_.tower(), L = 1 + 2 * t;
  if (params_.empty())
    for (size_t i = 0; i < (size_t)4 * L + 8 + 6; ++i) params_.emplace_back(new Param);
  const float* F = net_.flux_device();
  auto size = [&](Param& p, size_t n) {
    p.n = n;
    p.theta.ensure(n);
    p.grad.ensure(n);
    if (p.vel.n < n) { p.vel.ensure(n); have_vel_ = false; }
  };
  auto put = [&](Param& p, int layer, int kind) {
    size(p, (size_t)net_.param_count(layer, kind));
    AGZ_HIP(hipMemcpyAsync(p.theta.p, F + net_.flux_offset(layer, kind), sizeof(float) * p.n, hipMemcpyDeviceToDevice, stream_));
  };
  for (int l = 0; l < L; ++l) {
    const int cin = l == 0 ? kCinStem : kC, cinp = l == 0 ? kCinStemPad : kC;
    size(*params_[4 * l + 0], (size_t)kC * 9 * cinp);
    launch_pack_direct(F + net_.flux_offset(l, AGZ_K_WEIGHT), 0, cin, cinp, 1, params_[4 * l + 0]->theta.p, stream_);
    put(*params_[4 * l + 1], l, AGZ_K_BIAS);
    put(*params_[4 * l + 2], l, AGZ_K_BN_GAMMA);
    put(*params_[4 * l + 3], l, AGZ_K_BN_BETA);
  }
  size_t k = (size_t)4 * L;
  for (int l : {AGZ_L_VALUE_CONV, AGZ_L_POLICY_CONV}) {
    put(*params_[k++], l, AGZ_K_WEIGHT);            // [1,1,256,cout] column-major = w[ci + 256 o]
    put(*params_[k++], l, AGZ_K_BIAS);
    put(*params_[k++], l, AGZ_K_BN_GAMMA);
    put(*params_[k++], l, AGZ_K_BN_BETA);
  }
  for (int l : {AGZ_L_VALUE_FC1, AGZ_L_VALUE_FC2, AGZ_L_POLICY_FC}) {
    put(*params_[k++], l, AGZ_K_WEIGHT);
    put(*params_[k++], l, AGZ_K_BIAS);
  }
  if (!have_vel_)
    for (auto& p : params_) AGZ_HIP(hipMemsetAsync(p->vel.p, 0, sizeof(float) * p->n, stream_));
  have_vel_ = true;
}

// The inverse of upload() after a step, plus the running BatchNorm statistics (d_stats_: the step's batch statistics):
// the updated parameters go back into the network's device master -- device to device, nothing visits the host -- and
// the inference images are rebuilt from there before the next forward (Net::pack).
void Trainer::publish(long M) {
  const int t = net_.tower(), L = 1 + 2 * t;
  float* F = net_.flux_device();
  auto back = [&](Param& p, int layer, int kind) {
    AGZ_HIP(hipMemcpyAsync(F + net_.flux_offset(layer, kind), p.theta.p, sizeof(float) * p.n, hipMemcpyDeviceToDevice, stream_));
  };
  const float corr = (float)((double)M / (double)(M - 1));
  for (int l = 0; l < L; ++l) {
    const int cin = l == 0 ? kCinStem : kC, cinp = l == 0 ? kCinStemPad : kC;
    hipLaunchKernelGGL(k_unpack_direct, dim3(l == 0 ? 160 : 2304), dim3(256), 0, stream_, (const float*)params_[4 * l + 0]->theta.p, cin,
                       cinp, F + net_.flux_offset(l, AGZ_K_WEIGHT));
    back(*params_[4 * l + 1], l, AGZ_K_BIAS);
    back(*params_[4 * l + 2], l, AGZ_K_BN_GAMMA);
    back(*params_[4 * l + 3], l, AGZ_K_BN_BETA);
    const float* st = d_stats_.p + (size_t)3 * kC * l;
    hipLaunchKernelGGL(k_running_stats, dim3(1), dim3(256), 0, stream_, F + net_.flux_offset(l, AGZ_K_BN_MEAN),
                       F + net_.flux_offset(l, AGZ_K_BN_VAR), st, st + kC, kC, corr);
  }
  size_t k = (size_t)4 * L;
  const float* sh = d_stats_.p + (size_t)3 * kC * L;         // value head {mean, var, rstd}; policy head {mean[2], var[2], rstd[2]}
  int hc = 0;
  for (int l : {AGZ_L_VALUE_CONV, AGZ_L_POLICY_CONV}) {
    back(*params_[k++], l, AGZ_K_WEIGHT);
    back(*params_[k++], l, AGZ_K_BIAS);
    back(*params_[k++], l, AGZ_K_BN_GAMMA);
    back(*params_[k++], l, AGZ_K_BN_BETA);
    const int n = hc == 0 ? 1 : 2;
    hipLaunchKernelGGL(k_running_stats, dim3(1), dim3(64), 0, stream_, F + net_.flux_offset(l, AGZ_K_BN_MEAN),
                       F + net_.flux_offset(l, AGZ_K_BN_VAR), sh + (hc == 0 ? 0 : 3), sh + (hc == 0 ? 1 : 5), n, corr);
    ++hc;
  }
  for (int l : {AGZ_L_VALUE_FC1, AGZ_L_VALUE_FC2, AGZ_L_POLICY_FC}) {
    back(*params_[k++], l, AGZ_K_WEIGHT);
    back(*params_[k++], l, AGZ_K_BIAS);
  }
  AGZ_HIP(hipGetLastError());
  net_.device_master_written();                 // host copies stale, inference images stale
  uploaded_version_ = net_.param_version();     // ... and the training copies are what was just published
}

void Trainer::step(const float* feats, const float* pi, const float* z, int B, bool is_device, float eta, float rho,
                   float* losses_out) {
  AGZ_REQUIRE(B >= 2, AGZ_BAD_ARGUMENT, "a training batch needs at least 2 positions (BatchNorm batch statistics)");
  AGZ_REQUIRE(feats && pi && z, AGZ_BAD_ARGUMENT, "null pointer");
  const int N = net_.N(), P = net_.P(), A = net_.A(), t = net_.tower(), L = 1 + 2 * t;
  const long M = (long)B * P;
  hipStream_t s = stream_;
  upload();
  // ---- workspace
  const size_t act = (size_t)M * kC;
  d_x32_.ensure((size_t)M * kCinStemPad);
  d_u_.ensure(act * L);
  d_o_.ensure(act * L);
  d_ga_.ensure(act);
  d_gb_.ensure(act);
  d_gc_.ensure(act);
  d_stats_.ensure((size_t)3 * kC * L + 16);
  d_sums_.ensure((size_t)4 * kC + 8);
  d_ones_.ensure(kC);
  d_wd_.ensure((size_t)kC * 9 * kC);
  d_small_.ensure((size_t)M * 9 + (size_t)B * (A * 3 + 256 * 2 + 8) + 64);
  d_in_.ensure((size_t)B * 17 * P + (size_t)B * A + B);
  d_cnt_.ensure(1);
  {
    std::vector<float> ones(kC, 1.f);
    AGZ_HIP(hipMemcpyAsync(d_ones_.p, ones.data(), sizeof(float) * kC, hipMemcpyHostToDevice, s));
    AGZ_HIP(hipMemcpyAsync(d_cnt_.p, &B, sizeof(int), hipMemcpyHostToDevice, s));
    AGZ_HIP(hipStreamSynchronize(s));
  }
  const hipMemcpyKind kin = is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  float* d_feats = d_in_.p;
  float* d_pi = d_feats + (size_t)B * 17 * P;
  float* d_z = d_pi + (size_t)B * A;
  AGZ_HIP(hipMemcpyAsync(d_feats, feats, sizeof(float) * (size_t)B * 17 * P, kin, s));
  AGZ_HIP(hipMemcpyAsync(d_pi, pi, sizeof(float) * (size_t)B * A, kin, s));
  AGZ_HIP(hipMemcpyAsync(d_z, z, sizeof(float) * (size_t)B, kin, s));
  launch_whcn_to_x32(d_feats, B, N, d_x32_.p, s);
  // small buffers
  float* cv = d_small_.p;                    // [M]
  float* cp = cv + M;                        // [M][2]
  float* hv = cp + 2 * M;                    // [M]     relu(BN(cv))
  float* hp = hv + M;                        // [M][2]  relu(BN(cp))
  float* dcv = hp + 2 * M;                   // [M]
  float* dcp = dcv + M;                      // [M][2]  (8 M floats so far)
  float* d1 = dcp + 2 * M;                   // [B][256] relu(Dense1)
  float* vout = d1 + (size_t)B * 256;        // [B]
  float* logits = vout + B;                  // [B][A]
  float* pout = logits + (size_t)B * A;      // [B][A]
  float* dlogit = pout + (size_t)B * A;      // [B][A]
  float* dsv = dlogit + (size_t)B * A;       // [B]
  float* dd1 = dsv + B;                      // [B][256]
  double* d_losses = reinterpret_cast<double*>(d_sums_.p) + 4 * kC;       // [4] behind the column sums
  double* sums = reinterpret_cast<double*>(d_sums_.p);
  auto zero_sums = [&](int C) { AGZ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s)); };
  AGZ_HIP(hipMemsetAsync(d_losses, 0, sizeof(double) * 4, s));
  const int RS = 64;                          // row slices of the column reductions

  auto P4 = [&](int l, int k) -> Param& { return *params_[(size_t)4 * l + k]; };
  float* stats0 = d_stats_.p;
  auto U = [&](int l) { return d_u_.p + act * l; };
  auto O = [&](int l) { return d_o_.p + act * l; };
  auto ST = [&](int l) { return stats0 + (size_t)3 * kC * l; };

  // ---- forward, training mode
  // small batches: nine tap-split workgroups per tile and a fixed-order sum (42 workgroups of 128 rows do not fill 256 CUs)
  const bool taps = conv3x3_direct_blocks(B, N) < 192;
  d_zero_.ensure(kC);
  AGZ_HIP(hipMemsetAsync(d_zero_.p, 0, sizeof(float) * kC, s));
  if (taps) d_part_.ensure((size_t)9 * B * N * N * kC);
  auto conv_bn = [&](int l, const float* in, const float* res) {
    if (taps)
      launch_conv3x3_direct_taps(in, P4(l, 0).theta.p, d_ones_.p, d_zero_.p, P4(l, 1).theta.p, U(l), d_part_.p, d_cnt_.p, B, N,
                                 l == 0 ? kCinStemPad : kC, s);
    else
      launch_conv3x3_direct(in, P4(l, 0).theta.p, d_ones_.p, P4(l, 1).theta.p, nullptr, U(l), d_cnt_.p, B, N, 0,
                            l == 0 ? kCinStemPad : kC, s);
    zero_sums(kC);
    hipLaunchKernelGGL(k_colsums, dim3(kC / 64, RS), dim3(256), 0, s, (const float*)U(l), M, (int)kC, sums);
    hipLaunchKernelGGL(k_bn_fwd, g1(M * kC), dim3(256), 0, s, (const float*)U(l), M, (int)kC, (const double*)sums,
                       (const float*)P4(l, 2).theta.p, (const float*)P4(l, 3).theta.p, train_eps(net_.conv(l)->eps), res, O(l), 1, ST(l));
  };
  conv_bn(0, d_x32_.p, nullptr);
  for (int blk = 0; blk < t; ++blk) {          // relu(BN2(conv2(relu(BN1(conv1(x))))) + x), resnet.jl:26-32
    conv_bn(1 + 2 * blk, O(2 * blk), nullptr);
    conv_bn(2 + 2 * blk, O(1 + 2 * blk), O(2 * blk));
  }
  const float* xL = O(L - 1);
  const size_t hk = (size_t)4 * L;             // head parameter slots: vconv w b g be | pconv w b g be | fc...
  Param &Wv = *params_[hk + 0], &Bv = *params_[hk + 1], &Gv = *params_[hk + 2], &BEv = *params_[hk + 3];
  Param &Wp = *params_[hk + 4], &Bp = *params_[hk + 5], &Gp = *params_[hk + 6], &BEp = *params_[hk + 7];
  Param &W1 = *params_[hk + 8], &B1 = *params_[hk + 9], &W2 = *params_[hk + 10], &B2 = *params_[hk + 11];
  Param &Wf = *params_[hk + 12], &Bf = *params_[hk + 13];
  float* st_v = stats0 + (size_t)3 * kC * L;   // {mean, var, rstd} of the 1-channel BN
  float* st_p = st_v + 3;                      // 2-channel BN: {mean[2], var[2], rstd[2]}
  hipLaunchKernelGGL(k_head1x1_fwd, dim3(256), dim3(256), 0, s, xL, M, (const float*)Wv.theta.p, (const float*)Bv.theta.p,
                     (const float*)Wp.theta.p, (const float*)Bp.theta.p, cv, cp);
  zero_sums(2);
  hipLaunchKernelGGL(k_colsums, dim3(1, RS), dim3(256), 0, s, (const float*)cv, M, 1, sums);
  hipLaunchKernelGGL(k_bn_fwd, g1(M), dim3(256), 0, s, (const float*)cv, M, 1, (const double*)sums, (const float*)Gv.theta.p,
                     (const float*)BEv.theta.p, train_eps(net_.conv(AGZ_L_VALUE_CONV)->eps), (const float*)nullptr, hv, 1, st_v);
  zero_sums(2);
  hipLaunchKernelGGL(k_colsums, dim3(1, RS), dim3(256), 0, s, (const float*)cp, M, 2, sums);
  hipLaunchKernelGGL(k_bn_fwd, g1(M * 2), dim3(256), 0, s, (const float*)cp, M, 2, (const double*)sums,
                     (const float*)Gp.theta.p, (const float*)BEp.theta.p, train_eps(net_.conv(AGZ_L_POLICY_CONV)->eps),
                     (const float*)nullptr, hp, 1, st_p);
  hipLaunchKernelGGL(k_dense_fwd, dim3(1, B), dim3(256), 0, s, (const float*)hv, 0, P, (const float*)W1.theta.p,
                     (const float*)B1.theta.p, B, P, 256, 1, d1);
  hipLaunchKernelGGL(k_dense_fwd, dim3(1, B), dim3(64), 0, s, (const float*)d1, 0, P, (const float*)W2.theta.p,
                     (const float*)B2.theta.p, B, 256, 1, 2, vout);
  hipLaunchKernelGGL(k_dense_fwd, dim3((A + 255) / 256, B), dim3(256), 0, s, (const float*)hp, 2, P, (const float*)Wf.theta.p,
                     (const float*)Bf.theta.p, B, 2 * P, A, 0, logits);
  hipLaunchKernelGGL(k_outputs, dim3(B), dim3(256), 0, s, (const float*)logits, (const float*)vout, (const float*)d_pi,
                     (const float*)d_z, B, A, pout, dlogit, dsv, d_losses);

  // ---- backward: heads
  hipLaunchKernelGGL(k_dense_wgrad, g1((long)A * 2 * P), dim3(256), 0, s, (const float*)hp, 2, P, (const float*)dlogit, B, 2 * P,
                     A, Wf.grad.p, Bf.grad.p);
  float* dhp = dcp;       // grad wrt hp lands in dcp's storage first, BN backward turns it into dcp in place
  hipLaunchKernelGGL(k_dense_dgrad, dim3((2 * P + 255) / 256, B), dim3(256), 0, s, (const float*)Wf.theta.p,
                     (const float*)dlogit, B, 2 * P, A, (const float*)nullptr, 2, P, dhp);
  hipLaunchKernelGGL(k_dense_wgrad, g1(256), dim3(256), 0, s, (const float*)d1, 0, P, (const float*)dsv, B, 256, 1, W2.grad.p,
                     B2.grad.p);
  hipLaunchKernelGGL(k_dense_dgrad, dim3(1, B), dim3(256), 0, s, (const float*)W2.theta.p, (const float*)dsv, B, 256, 1,
                     (const float*)d1, 0, P, dd1);
  hipLaunchKernelGGL(k_dense_wgrad, g1((long)256 * P), dim3(256), 0, s, (const float*)hv, 0, P, (const float*)dd1, B, P, 256,
                     W1.grad.p, B1.grad.p);
  float* dhv = dcv;
  hipLaunchKernelGGL(k_dense_dgrad, dim3((P + 255) / 256, B), dim3(256), 0, s, (const float*)W1.theta.p, (const float*)dd1, B, P,
                     256, (const float*)nullptr, 0, P, dhv);
  // head BatchNorms (ReLU mask from hv / hp), in place: dhv -> dcv, dhp -> dcp
  zero_sums(2);
  hipLaunchKernelGGL(k_bn_bwd_sums, dim3(1, RS), dim3(256), 0, s, (const float*)dhv, (const float*)hv, (const float*)cv,
                     (const float*)st_v, M, 1, 1, sums);
  hipLaunchKernelGGL(k_bn_bwd_apply, g1(M), dim3(256), 0, s, (const float*)dhv, (const float*)hv, (const float*)cv,
                     (const float*)st_v, (const float*)Gv.theta.p, (const double*)sums, M, 1, 1, dcv, (float*)nullptr, Gv.grad.p,
                     BEv.grad.p);
  zero_sums(2);
  hipLaunchKernelGGL(k_bn_bwd_sums, dim3(1, RS), dim3(256), 0, s, (const float*)dhp, (const float*)hp, (const float*)cp,
                     (const float*)st_p, M, 2, 1, sums);
  hipLaunchKernelGGL(k_bn_bwd_apply, g1(M * 2), dim3(256), 0, s, (const float*)dhp, (const float*)hp, (const float*)cp,
                     (const float*)st_p, (const float*)Gp.theta.p, (const double*)sums, M, 2, 1, dcp, (float*)nullptr, Gp.grad.p,
                     BEp.grad.p);
  AGZ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * (3 * kC + 3), s));
  hipLaunchKernelGGL(k_head1x1_wgrad, dim3(RS), dim3(256), 0, s, xL, (const float*)dcv, (const float*)dcp, M, sums);
  hipLaunchKernelGGL(k_head1x1_wgrad_take, dim3(1), dim3(256), 0, s, (const double*)sums, Wv.grad.p, Wp.grad.p, Bv.grad.p,
                     Bp.grad.p);
  float* g = d_ga_.p;      // gradient wrt the current block output
  hipLaunchKernelGGL(k_head1x1_dgrad, g1(M * kC), dim3(256), 0, s, (const float*)dcv, (const float*)dcp, (const float*)Wv.theta.p,
                     (const float*)Wp.theta.p, M, g);

  // ---- backward: tower and stem
  // BN backward of layer l with upstream `dout` (ReLU mask from O(l)); du -> `du`; dres (optional) gets the masked dout
  auto bn_back = [&](int l, const float* dout, float* du, float* dres) {
    zero_sums(kC);
    hipLaunchKernelGGL(k_bn_bwd_sums, dim3(kC / 64, RS), dim3(256), 0, s, dout, (const float*)O(l), (const float*)U(l),
                       (const float*)ST(l), M, (int)kC, 1, sums);
    hipLaunchKernelGGL(k_bn_bwd_apply, g1(M * kC), dim3(256), 0, s, dout, (const float*)O(l), (const float*)U(l),
                       (const float*)ST(l), (const float*)P4(l, 2).theta.p, (const double*)sums, M, (int)kC, 1, du, dres,
                       P4(l, 2).grad.p, P4(l, 3).grad.p);
    zero_sums(kC);
    hipLaunchKernelGGL(k_colsums, dim3(kC / 64, RS), dim3(256), 0, s, (const float*)du, M, (int)kC, sums);
    hipLaunchKernelGGL(k_take_sums, dim3(1), dim3(256), 0, s, (const double*)sums, (int)kC, P4(l, 1).grad.p);
  };
  // rows per weight-gradient block: ~2600 (the reference's batch of 32 at 9x9), more blocks for more rows
  const int wsplit = (int)std::min<long>(16, std::max<long>(1, (M + 2047) / 2592));
  const int wchunk = (int)(((M + wsplit - 1) / wsplit + 7) / 8 * 8);
  if (wsplit > 1) d_wpart_.ensure((size_t)wsplit * kC * 9 * kC);
  auto wgrad = [&](int l, const float* in, const float* du) {
    const int cinp = l == 0 ? kCinStemPad : kC;
    const long n = (long)kC * 9 * cinp;
    if (wsplit == 1) {
      hipLaunchKernelGGL(k_wgrad3x3, dim3(kC / 32, 9, cinp / 32), dim3(256), 0, s, in, du, B, N, cinp, P4(l, 0).grad.p, wchunk);
    } else {
      hipLaunchKernelGGL(k_wgrad3x3, dim3(kC / 32 * wsplit, 9, cinp / 32), dim3(256), 0, s, in, du, B, N, cinp, d_wpart_.p, wchunk);
      hipLaunchKernelGGL(k_sum_parts, g1(n), dim3(256), 0, s, (const float*)d_wpart_.p, wsplit, n, P4(l, 0).grad.p, n);
    }
  };
  // dx = conv(du) with Wd[ci][tap'][co] = Wt[co][8 - tap'][ci]
  auto dgrad = [&](int l, const float* du, float* dx) {
    hipLaunchKernelGGL(k_make_wd, g1((long)kC * 9 * kC), dim3(256), 0, s, (const float*)P4(l, 0).theta.p, d_wd_.p);
    if (taps) launch_conv3x3_direct_taps(du, d_wd_.p, d_ones_.p, d_zero_.p, d_zero_.p, dx, d_part_.p, d_cnt_.p, B, N, kC, s);
    else launch_conv3x3_direct(du, d_wd_.p, d_ones_.p, d_zero_.p, nullptr, dx, d_cnt_.p, B, N, 0, kC, s);
  };
  float *du = d_gb_.p, *dsc = d_gc_.p;
  for (int blk = t - 1; blk >= 0; --blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    bn_back(l2, g, du, dsc);                       // through relu(BN2(u2) + x): du2, and the shortcut's share dsc
    wgrad(l2, O(l1), du);
    dgrad(l2, du, g);                              // g <- grad wrt relu(BN1(u1))
    bn_back(l1, g, du, nullptr);
    wgrad(l1, O(2 * blk), du);
    dgrad(l1, du, g);                              // g <- grad wrt the block input through the convolutions
    hipLaunchKernelGGL(k_add, g1(M * kC), dim3(256), 0, s, g, (const float*)dsc, M * kC);
  }
  bn_back(0, g, du, nullptr);
  wgrad(0, d_x32_.p, du);

  // ---- the optimiser (one launch over every array; it also sums theta^2 of the values before the update), then the losses
  if (d_optw_.n == 0) {
    std::vector<OptArray> arr;
    std::vector<OptWork> work;
    for (auto& p : params_) {
      for (long c = 0; c * kOptChunk < (long)p->n; ++c) work.push_back(OptWork{(int)arr.size(), (int)c});
      arr.push_back(OptArray{p->theta.p, p->grad.p, p->vel.p, (long)p->n});
    }
    d_opta_.alloc(arr.size() * sizeof(OptArray));
    d_optw_.alloc(work.size() * sizeof(OptWork));
    AGZ_HIP(hipMemcpyAsync(d_opta_.p, arr.data(), arr.size() * sizeof(OptArray), hipMemcpyHostToDevice, s));
    AGZ_HIP(hipMemcpyAsync(d_optw_.p, work.data(), work.size() * sizeof(OptWork), hipMemcpyHostToDevice, s));
    AGZ_HIP(hipStreamSynchronize(s));
    n_optw_ = (int)work.size();
  }
  hipLaunchKernelGGL(k_momentum_all, dim3(n_optw_), dim3(256), 0, s, reinterpret_cast<const OptArray*>(d_opta_.p),
                     reinterpret_cast<const OptWork*>(d_optw_.p), eta, rho, d_losses + 2);
  publish(M);       // parameters and running statistics into the network's device master (enqueued behind the update)
  double hl[4];
  AGZ_HIP(hipMemcpyAsync(hl, d_losses, sizeof(hl), hipMemcpyDeviceToHost, s));
  AGZ_HIP(hipGetLastError());
  AGZ_HIP(hipStreamSynchronize(s));
  if (losses_out) {
    losses_out[1] = (float)hl[0];
    losses_out[2] = (float)hl[1];
    losses_out[3] = (float)((double)kRegW * hl[2]);
    losses_out[0] = losses_out[1] + losses_out[2] + losses_out[3];
  }
}

}  // namespace agz
