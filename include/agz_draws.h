/*
 * agz_draws.h -- the injected draw stream of the self-play hot path.
 *
 * The reference draws from Julia's global MersenneTwister at five call sites:
 *   - tie-break among equal PUCT scores     rand(possible_moves)     src/mcts.jl:133
 *   - tie-break among equal visit counts    sample(possible_moves)   src/mcts_play.jl:61
 *   - early-game soft pick                  rand()                   src/mcts_play.jl:66
 *   - root exploration noise                rand(Dirichlet(a*1_A))   src/mcts.jl:235
 *   - resign-disable coin                   rand() < 0.05            src/selfplay.jl:9
 * Julia's stream cannot be reproduced without Julia, so every one of those draws is
 * DEFINED here as a pure function of (seed, game, move, site, index).  The HIP kernels,
 * the CPU oracle and any future Julia-side override all include this one header, which
 * makes "identical visit counts and selected moves under a fixed RNG" a testable claim.
 *
 * Everything in this file is restricted to IEEE-754 +,-,*,/ and integer ops with
 * floating-point contraction switched off, so that gcc on the host and hipcc for gfx950
 * produce bit-identical results (checked by tests/test_draws.py and, on the GPU, by
 * tests/test_gpu_draws.py).  log/exp/pow are therefore implemented here rather than
 * taken from libm/ocml.
 *
 * Plain C99; also valid C++ and HIP device code.
 */
#ifndef AGZ_DRAWS_H
#define AGZ_DRAWS_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define AGZ_HD __host__ __device__
#else
#define AGZ_HD
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif

/* draw sites */
#define AGZ_SITE_PUCT_TIE 1u   /* idx = select attempt * 1024 + depth            */
#define AGZ_SITE_PICK_TIE 2u   /* idx = 0                                         */
#define AGZ_SITE_SOFTPICK 3u   /* idx = 0                                         */
#define AGZ_SITE_DIRICHLET 4u  /* idx = action index (0-based); a private stream  */
#define AGZ_SITE_RESIGN 5u     /* move = 0, idx = 0                               */
#define AGZ_SITE_WEIGHTS 6u    /* synthetic weight init: game = layer, idx = elem */
#define AGZ_SITE_STAGGER 7u    /* bench-only random opening prefix                */

static inline AGZ_HD uint64_t agz_mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}

/* one 64-bit draw, a pure function of its key */
static inline AGZ_HD uint64_t agz_draw_u64(uint64_t seed, uint64_t game, uint32_t move,
                                           uint32_t site, uint64_t idx) {
  uint64_t h = agz_mix64(seed + 0x9E3779B97F4A7C15ull);
  h = agz_mix64(h ^ (game + 0xD1B54A32D192ED03ull));
  h = agz_mix64(h ^ (((uint64_t)move << 8) | (uint64_t)site));
  h = agz_mix64(h ^ (idx + 0x8CB92BA72F3D8DD7ull));
  return h;
}

/* uniform double strictly inside (0,1) */
static inline AGZ_HD double agz_u01(uint64_t bits) {
  return ((double)(bits >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

/* uniform integer in [0,n) */
static inline AGZ_HD uint32_t agz_index(uint64_t bits, uint32_t n) {
  return (uint32_t)(((bits >> 32) * (uint64_t)n) >> 32);
}

/* a small sequential stream (splitmix64) keyed by one draw; used by the gamma sampler */
typedef struct { uint64_t s; } agz_stream;
static inline AGZ_HD agz_stream agz_stream_open(uint64_t seed, uint64_t game, uint32_t move,
                                                uint32_t site, uint64_t idx) {
  agz_stream st; st.s = agz_draw_u64(seed, game, move, site, idx); return st;
}
static inline AGZ_HD uint64_t agz_stream_next(agz_stream* st) {
  st->s += 0x9E3779B97F4A7C15ull;
  return agz_mix64(st->s);
}

/* ---- deterministic elementary functions (double) --------------------------------- */

static inline AGZ_HD double agz_bits2d(uint64_t u) {
  union { uint64_t u; double d; } c; c.u = u; return c.d;
}
static inline AGZ_HD uint64_t agz_d2bits(double d) {
  union { uint64_t u; double d; } c; c.d = d; return c.u;
}

/* 2^k for k in [-1022, 1023] */
static inline AGZ_HD double agz_pow2i(int k) {
  return agz_bits2d((uint64_t)(k + 1023) << 52);
}

/* natural log of a finite x > 0 (normal or subnormal) */
static inline AGZ_HD double agz_log(double x) {
  int e = 0;
  uint64_t b = agz_d2bits(x);
  if ((b >> 52) == 0) { x = x * 18014398509481984.0; /* 2^54 */ e = -54; b = agz_d2bits(x); }
  e += (int)((b >> 52) & 0x7FF) - 1023;
  double m = agz_bits2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }                         /* [0.707,1.414] */
  double s = (m - 1.0) / (m + 1.0);
  double z = s * s;
  /* 2*atanh(s) = 2*(s + s^3/3 + s^5/5 + ...), |s| <= 0.1716, 12 terms ~ 1e-19 */
  double p = 1.0 / 23.0;
  p = p * z + 1.0 / 21.0;
  p = p * z + 1.0 / 19.0;
  p = p * z + 1.0 / 17.0;
  p = p * z + 1.0 / 15.0;
  p = p * z + 1.0 / 13.0;
  p = p * z + 1.0 / 11.0;
  p = p * z + 1.0 / 9.0;
  p = p * z + 1.0 / 7.0;
  p = p * z + 1.0 / 5.0;
  p = p * z + 1.0 / 3.0;
  p = p * z + 1.0;
  double lm = 2.0 * s * p;
  return (double)e * 0.6931471805599453 + lm;
}

/* e^x; returns 0 below -745, clamps above 709 */
static inline AGZ_HD double agz_exp(double x) {
  if (x < -745.0) return 0.0;
  if (x > 709.0) x = 709.0;
  double kf = x * 1.4426950408889634;
  int k = (int)(kf < 0.0 ? kf - 0.5 : kf + 0.5);
  double r = (x - (double)k * 0.693147180369123816490e+00) - (double)k * 1.90821492927058770002e-10;
  /* Taylor to degree 14, |r| <= 0.3466 */
  double p = 1.0 / 87178291200.0;
  p = p * r + 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  /* scale by 2^k in two steps so subnormal results are produced by a multiply */
  if (k < -1000) return (p * agz_pow2i(k + 1000)) * agz_pow2i(-1000);
  return p * agz_pow2i(k);
}

/* x^y for x >= 0 (0^y = 0 for y > 0) */
static inline AGZ_HD double agz_pow(double x, double y) {
  if (x <= 0.0) return 0.0;
  return agz_exp(y * agz_log(x));
}

/* ---- sqrt by Newton from an exact-op seed is unnecessary: IEEE sqrt is correctly
 * rounded on both targets; we still avoid it in the gamma sampler by using the polar
 * method on v = (1+c*x)^3 with c precomputed through agz_rsqrt below. ---------------- */
static inline AGZ_HD double agz_sqrt(double x) {
  /* Newton iterations on y = sqrt(x) using only +,*,/ ; converges to within 1 ulp.
   * Deterministic (not necessarily correctly rounded), which is all the sampler needs. */
  if (x <= 0.0) return 0.0;
  uint64_t b = agz_d2bits(x);
  int e = (int)((b >> 52) & 0x7FF) - 1023;
  double y = agz_pow2i(e / 2);
  if (y * y > x) y = y * 0.5;
  y = y * 1.2;
  for (int i = 0; i < 8; ++i) y = 0.5 * (y + x / y);
  return y;
}

/* standard normal by Marsaglia's polar method (consumes the stream) */
static inline AGZ_HD double agz_normal(agz_stream* st) {
  for (int it = 0; it < 64; ++it) {
    double u1 = 2.0 * agz_u01(agz_stream_next(st)) - 1.0;
    double u2 = 2.0 * agz_u01(agz_stream_next(st)) - 1.0;
    double s = u1 * u1 + u2 * u2;
    if (s < 1.0 && s > 0.0) return u1 * agz_sqrt(-2.0 * agz_log(s) / s);
  }
  return 0.0;
}

/* Gamma(alpha, 1) for alpha > 0: Marsaglia-Tsang (2000) on alpha+1 with the
 * U^(1/alpha) boost when alpha < 1 -- the same family of method Distributions.jl uses
 * for rand(Dirichlet) (src/mcts.jl:235 draws A independent gammas and normalises). */
static inline AGZ_HD double agz_gamma(double alpha, agz_stream* st) {
  double a = alpha < 1.0 ? alpha + 1.0 : alpha;
  double d = a - 1.0 / 3.0;
  double c = 1.0 / agz_sqrt(9.0 * d);
  double g = d;
  for (int it = 0; it < 256; ++it) {
    double x = agz_normal(st);
    double t = 1.0 + c * x;
    if (t <= 0.0) continue;
    double v = t * t * t;
    double u = agz_u01(agz_stream_next(st));
    if (agz_log(u) < 0.5 * x * x + d - d * v + d * agz_log(v)) { g = d * v; break; }
  }
  if (alpha < 1.0) {
    double u = agz_u01(agz_stream_next(st));
    g = g * agz_exp(agz_log(u) / alpha);
  }
  return g;
}

/* the a-th un-normalised Dirichlet component for (seed, game, move) */
static inline AGZ_HD double agz_dirichlet_gamma(uint64_t seed, uint64_t game, uint32_t move,
                                                uint32_t a, double alpha) {
  agz_stream st = agz_stream_open(seed, game, move, AGZ_SITE_DIRICHLET, (uint64_t)a);
  return agz_gamma(alpha, &st);
}

#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC pop_options
#endif

#endif /* AGZ_DRAWS_H */
