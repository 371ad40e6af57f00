/*
 * agz_oracle_nn.c -- dual-head ResNet forward of the CPU oracle (inference only).
 * TEST INFRASTRUCTURE (see agz_oracle.h).  Topology from
 *   /root/reference/src/neural_net.jl:13-33,57-73 and /root/reference/src/resnet.jl:11-32.
 * The layer arithmetic itself lives in Flux 0.10.4 / NNlib 0.6.6 (Manifest.toml:193-197,
 * 298-302), which are not vendored in the reference tree; their documented semantics are
 * restated here:
 *   Conv   : TRUE convolution (kernel flipped in both spatial dims), zero padding, + bias
 *   BN     : inference form  y = gamma * (x - mu) / sqrt(var + eps) + beta
 *   Dense  : sigma.(W*x .+ b), W is [out,in] column-major
 *   softmax: per column with max subtraction
 * No reference test exercises the network => "parity unpinned" for this file; it is checked
 * against torch fp64 conv2d/batch_norm in tests/test_oracle_nn.py.
 *
 * Tensors cross the API in Flux layout (conv [kw,kh,cin,cout] column-major, dense
 * [out,in] column-major, inputs WHCN).  Internally activations are [b][p][c] (channel
 * fastest) and conv weights are re-packed once into [tap][cin][cout].
 */
#include "agz_oracle.h"
#include "../include/agz_draws.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define C 256

typedef struct {
  int k, cin, cout;          /* k = 3 or 1 */
  float* w;                  /* Flux layout [k,k,cin,cout] */
  float* b;                  /* [cout] */
  float *beta, *gamma, *mean, *var;  /* BN, [cout] */
  float eps;
  float* wp;                 /* packed [tap][cin][cout], flip applied */
  float* wpn;                /* the same as 16-channel panels [cout/16][tap][cin][16] (cout % 16 == 0) */
  float *wp16, *wpn16;       /* wp / wpn rounded to IEEE half (the fp16-operand mode, precision 16) */
  int dirty;
} OConv;

typedef struct {
  int in, out;
  float* w;                  /* [out,in] column-major: w[o + out*i] */
  float* b;
} ODense;

struct ONet {
  int N, P, A, tower;
  OConv stem;
  OConv* tconv;              /* 2*tower */
  OConv vconv, pconv;
  ODense vfc1, vfc2, pfc;
};

static void conv_alloc(OConv* c, int k, int cin, int cout) {
  c->k = k; c->cin = cin; c->cout = cout;
  size_t nw = (size_t)k * k * cin * cout;
  c->w = (float*)calloc(nw, sizeof(float));
  c->wp = (float*)calloc(nw, sizeof(float));
  c->wpn = (float*)calloc(nw, sizeof(float));
  c->wp16 = (float*)calloc(nw, sizeof(float));
  c->wpn16 = (float*)calloc(nw, sizeof(float));
  c->b = (float*)calloc((size_t)cout, sizeof(float));
  c->beta = (float*)calloc((size_t)cout, sizeof(float));
  c->gamma = (float*)calloc((size_t)cout, sizeof(float));
  c->mean = (float*)calloc((size_t)cout, sizeof(float));
  c->var = (float*)calloc((size_t)cout, sizeof(float));
  for (int o = 0; o < cout; ++o) { c->gamma[o] = 1.0f; c->var[o] = 1.0f; }
  c->eps = 1e-5f;
  c->dirty = 1;
}
static void conv_free(OConv* c) {
  free(c->wp16); free(c->wpn16);
  free(c->w); free(c->wp); free(c->wpn); free(c->b); free(c->beta); free(c->gamma); free(c->mean); free(c->var);
}
static void dense_alloc(ODense* d, int in, int out) {
  d->in = in; d->out = out;
  d->w = (float*)calloc((size_t)in * out, sizeof(float));
  d->b = (float*)calloc((size_t)out, sizeof(float));
}

ONet* or_net_new(int N, int tower_height) {
  ONet* n = (ONet*)calloc(1, sizeof(ONet));
  n->N = N; n->P = N * N; n->A = N * N + 1; n->tower = tower_height;
  conv_alloc(&n->stem, 3, 17, C);
  n->tconv = (OConv*)calloc((size_t)(2 * tower_height > 0 ? 2 * tower_height : 1), sizeof(OConv));
  for (int l = 0; l < 2 * tower_height; ++l) conv_alloc(&n->tconv[l], 3, C, C);
  conv_alloc(&n->vconv, 1, C, 1);
  conv_alloc(&n->pconv, 1, C, 2);
  dense_alloc(&n->vfc1, n->P, 256);
  dense_alloc(&n->vfc2, 256, 1);
  dense_alloc(&n->pfc, 2 * n->P, n->A);
  return n;
}

void or_net_free(ONet* n) {
  if (!n) return;
  conv_free(&n->stem);
  for (int l = 0; l < 2 * n->tower; ++l) conv_free(&n->tconv[l]);
  free(n->tconv);
  conv_free(&n->vconv); conv_free(&n->pconv);
  free(n->vfc1.w); free(n->vfc1.b); free(n->vfc2.w); free(n->vfc2.b); free(n->pfc.w); free(n->pfc.b);
  free(n);
}

static OConv* get_conv(const ONet* n, int layer) {
  if (layer == 0) return (OConv*)&n->stem;
  if (layer >= 1 && layer <= 2 * n->tower) return &n->tconv[layer - 1];
  if (layer == OR_L_VALUE_CONV) return (OConv*)&n->vconv;
  if (layer == OR_L_POLICY_CONV) return (OConv*)&n->pconv;
  return NULL;
}
static ODense* get_dense(const ONet* n, int layer) {
  if (layer == OR_L_VALUE_FC1) return (ODense*)&n->vfc1;
  if (layer == OR_L_VALUE_FC2) return (ODense*)&n->vfc2;
  if (layer == OR_L_POLICY_FC) return (ODense*)&n->pfc;
  return NULL;
}

static float* param_ptr(const ONet* n, int layer, int kind, int64_t* count) {
  OConv* c = get_conv(n, layer);
  if (c) {
    switch (kind) {
      case OR_K_WEIGHT: *count = (int64_t)c->k * c->k * c->cin * c->cout; c->dirty = 1; return c->w;
      case OR_K_BIAS: *count = c->cout; return c->b;
      case OR_K_BN_BETA: *count = c->cout; return c->beta;
      case OR_K_BN_GAMMA: *count = c->cout; return c->gamma;
      case OR_K_BN_MEAN: *count = c->cout; return c->mean;
      case OR_K_BN_VAR: *count = c->cout; return c->var;
      case OR_K_BN_EPS: *count = 1; return &c->eps;
      default: return NULL;
    }
  }
  ODense* d = get_dense(n, layer);
  if (d) {
    if (kind == OR_K_WEIGHT) { *count = (int64_t)d->in * d->out; return d->w; }
    if (kind == OR_K_BIAS) { *count = d->out; return d->b; }
  }
  return NULL;
}

int64_t or_net_param_count(const ONet* n, int layer, int kind) {
  int64_t c = 0;
  return param_ptr(n, layer, kind, &c) ? c : -1;
}
int or_net_set(ONet* n, int layer, int kind, const float* data, int64_t count) {
  int64_t c = 0;
  float* p = param_ptr(n, layer, kind, &c);
  if (!p || c != count) return OR_BAD_SHAPE;
  memcpy(p, data, sizeof(float) * (size_t)count);
  return OR_OK;
}
int or_net_get(const ONet* n, int layer, int kind, float* out, int64_t count) {
  int64_t c = 0;
  float* p = param_ptr(n, layer, kind, &c);
  if (!p || c != count) return OR_BAD_SHAPE;
  memcpy(out, p, sizeof(float) * (size_t)count);
  return OR_OK;
}

/* glorot_uniform over (fan_in + fan_out), Flux utils (nfan): limit = sqrt(6/(fi+fo)) */
static void glorot(float* w, int64_t count, double fan_in, double fan_out, uint64_t seed, int layer) {
  double limit = sqrt(6.0 / (fan_in + fan_out));
  uint64_t key = (uint64_t)(int64_t)(layer + 4096);
  for (int64_t i = 0; i < count; ++i) {
    double u = agz_u01(agz_draw_u64(seed, key, 0, AGZ_SITE_WEIGHTS, (uint64_t)i));
    w[i] = (float)((2.0 * u - 1.0) * limit);
  }
}

void or_net_init_synthetic(ONet* n, uint64_t seed) {
  int layers = 1 + 2 * n->tower;
  for (int l = 0; l < layers; ++l) {
    OConv* c = get_conv(n, l);
    double rf = (double)c->k * c->k;
    glorot(c->w, (int64_t)c->k * c->k * c->cin * c->cout, rf * c->cin, rf * c->cout, seed, l);
    c->dirty = 1;
  }
  glorot(n->vconv.w, C * 1, C, 1, seed, OR_L_VALUE_CONV); n->vconv.dirty = 1;
  glorot(n->pconv.w, C * 2, C, 2, seed, OR_L_POLICY_CONV); n->pconv.dirty = 1;
  glorot(n->vfc1.w, (int64_t)n->vfc1.in * n->vfc1.out, n->vfc1.in, n->vfc1.out, seed, OR_L_VALUE_FC1);
  glorot(n->vfc2.w, (int64_t)n->vfc2.in * n->vfc2.out, n->vfc2.in, n->vfc2.out, seed, OR_L_VALUE_FC2);
  glorot(n->pfc.w, (int64_t)n->pfc.in * n->pfc.out, n->pfc.in, n->pfc.out, seed, OR_L_POLICY_FC);
}

/* round a float to the nearest IEEE binary16 value (ties to even), returned as float: what a
 * float -> half -> float round trip does on the GPU (v_cvt_f16_f32, __float2half_rn) */
static float quant_h(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  uint32_t sign = v.u & 0x80000000u, a = v.u & 0x7fffffffu;
  if (a >= 0x7f800000u) return f;                       /* inf / nan */
  if (a >= 0x477ff000u) { v.u = sign | 0x7f800000u; return v.f; }   /* >= 65520 rounds to inf */
  if (a < 0x38800000u) {                                /* below 2^-14: half subnormals, quantum 2^-24 */
    float q = 5.9604644775390625e-08f;
    float r = rintf(fabsf(f) / q) * q;                  /* rintf: round-half-even in the default mode */
    return sign ? -r : r;
  }
  uint32_t lsb = (a >> 13) & 1u;
  a += 0x0fffu + lsb;
  a &= ~0x1fffu;
  v.u = sign | a;
  return v.f;
}
float or_quant_half(float f) { return quant_h(f); }

/* [kw,kh,cin,cout] -> [tap=(da,db)][cin][cout] where the tap reads x[i+da, j+db].
 * True convolution: Flux index (a,b) (0-based) multiplies x[i + 1 - a, j + 1 - b] for k=3. */
static void conv_pack(OConv* c) {
  if (!c->dirty) return;
  int k = c->k, cin = c->cin, cout = c->cout, r = k / 2;
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      int da = r - a, db = r - b;                  /* offset applied to (row i, col j) */
      int tap = (da + r) + k * (db + r);
      for (int ci = 0; ci < cin; ++ci)
        for (int o = 0; o < cout; ++o)
          c->wp[((size_t)tap * cin + ci) * cout + o] =
              c->w[a + (size_t)k * (b + (size_t)k * (ci + (size_t)cin * o))];
    }
  if (cout % 16 == 0)
    for (int nb = 0; nb < cout / 16; ++nb)
      for (int tap = 0; tap < k * k; ++tap)
        for (int ci = 0; ci < cin; ++ci)
          memcpy(c->wpn + (((size_t)nb * k * k + tap) * cin + ci) * 16,
                 c->wp + ((size_t)tap * cin + ci) * cout + nb * 16, 16 * sizeof(float));
  {
    size_t nw = (size_t)k * k * cin * cout;
    for (size_t i = 0; i < nw; ++i) { c->wp16[i] = quant_h(c->wp[i]); c->wpn16[i] = quant_h(c->wpn[i]); }
  }
  c->dirty = 0;
}

static int g_threads = 0;
void or_set_num_threads(int n) { g_threads = n; }
int or_get_num_procs(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

#define T float
#define SUF f32
#define EXPF expf
#define TANHF tanhf
#define SQRTF sqrtf
#include "agz_oracle_nn_impl.inc"
#undef T
#undef SUF
#undef EXPF
#undef TANHF
#undef SQRTF

#define T double
#define SUF f64
#define EXPF exp
#define TANHF tanh
#define SQRTF sqrt
#include "agz_oracle_nn_impl.inc"
#undef T
#undef SUF
#undef EXPF
#undef TANHF
#undef SQRTF

void or_net_forward_feats(const ONet* n, const float* x, int B, float* pi, float* v, int precision) {
  if (precision == 64) {
    size_t nx = (size_t)B * 17 * n->P;
    double* xd = (double*)malloc(sizeof(double) * nx);
    double* pid = (double*)malloc(sizeof(double) * (size_t)B * n->A);
    double* vd = (double*)malloc(sizeof(double) * (size_t)B);
    for (size_t i = 0; i < nx; ++i) xd[i] = x[i];
    forward_f64(n, xd, B, pid, vd);
    for (size_t i = 0; i < (size_t)B * n->A; ++i) pi[i] = (float)pid[i];
    for (int b = 0; b < B; ++b) v[b] = (float)vd[b];
    free(xd); free(pid); free(vd);
  } else if (precision == 16) {
    /* the fp16-operand tower (agz_net_set_precision(F16)) restated: weights and tower activations
     * rounded to half at exactly the points where the GPU stores / loads them, sums in float64 */
    size_t nx = (size_t)B * 17 * n->P;
    double* xd = (double*)malloc(sizeof(double) * nx);
    double* pid = (double*)malloc(sizeof(double) * (size_t)B * n->A);
    double* vd = (double*)malloc(sizeof(double) * (size_t)B);
    for (size_t i = 0; i < nx; ++i) xd[i] = x[i];
    forward_ex_f64(n, xd, B, pid, vd, 1);
    for (size_t i = 0; i < (size_t)B * n->A; ++i) pi[i] = (float)pid[i];
    for (int b = 0; b < B; ++b) v[b] = (float)vd[b];
    free(xd); free(pid); free(vd);
  } else {
    forward_f32(n, x, B, pi, v);
  }
}

void or_net_forward_feats_f64(const ONet* n, const double* x, int B, double* pi, double* v) {
  forward_f64(n, x, B, pi, v);
}

void or_net_callable(void* ctx, const OPos* const* positions, int B, float* pi, float* v) {
  const ONet* n = (const ONet*)ctx;
  size_t per = (size_t)17 * n->P;
  float* x = (float*)malloc(sizeof(float) * per * (size_t)B);
  for (int b = 0; b < B; ++b) or_get_feats(positions[b], x + per * b);
  forward_f32(n, x, B, pi, v);
  free(x);
}
