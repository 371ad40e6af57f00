// agz_engine.hip -- the HIP instantiation of the search templates (one 64-lane wavefront per
// game tree) and the host engine that strings a self-play step together:
//
//   k_pre            lifecycle + per-move phase + select up to 8 leaves     (1 wave / game)
//   k_scan           prefix sum of the leaf counts -> rows of the NN batch
//   k_leaf_features  17 board planes of every leaf -> stem input            (1 wave / leaf)
//   Net::forward     stem + tower (MFMA implicit GEMM) + heads              (agz_nn.hip)
//   k_post           revert virtual loss, expand, back up                   (1 wave / game)
//
// Everything is enqueued on one stream with the batch size left in device memory, so a step
// needs no host synchronisation.  Compiled with -ffp-contract=off: the PUCT arithmetic mixes
// Float32 and Float64 exactly as the reference does.
#include "agz_engine.h"

#include <algorithm>
#include <cstring>

#include "agz_search.h"

namespace agz {

// ------------------------------------------------------------------ the wave primitives on gfx950

struct HipWave {
  static constexpr bool kRegisterRows = true;     // select_leaf_rows (agz_search.h)
  int lane;
  __device__ HipWave() : lane((int)threadIdx.x) {}
  __device__ explicit HipWave(int l) : lane(l) {}      // (a wave of a multi-wave workgroup: kernels that never call sync())
  template <class F>
  __device__ __forceinline__ void for_each(int n, F f) const {
    for (int i = lane; i < n; i += kWave) f(i);
  }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ bool leader() const { return lane == 0; }
  __device__ __forceinline__ int reduce_sum(int v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
  }
  __device__ __forceinline__ int reduce_min(int v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, kWave); v = t < v ? t : v; }
    return v;
  }
  __device__ __forceinline__ int reduce_max(int v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, kWave); v = t > v ? t : v; }
    return v;
  }
  __device__ __forceinline__ double reduce_max(double v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o, kWave); v = t > v ? t : v; }
    return v;
  }
  __device__ __forceinline__ float reduce_sum_f(float v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
  }
  __device__ __forceinline__ float reduce_max_f(float v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o, kWave); v = t > v ? t : v; }
    return v;
  }
  __device__ __forceinline__ bool any(bool v) const { return __any(v) != 0; }
  __device__ __forceinline__ float shfl(float v, int src) const { return __shfl(v, src, kWave); }
  __device__ __forceinline__ int shfl(int v, int src) const { return __shfl(v, src, kWave); }
  __device__ __forceinline__ unsigned long long clock() const { return wall_clock64(); }     // 100 MHz
  __device__ __forceinline__ void count_max(unsigned long long* p, unsigned long long v) const {
    if (lane == 0) atomicMax(p, v);
  }
  // append every get(i) >= 0, i ascending, to a downward stack (base[--sp]); returns the new sp
  template <class F>
  __device__ __forceinline__ int push_desc(int n, F get, int32_t* base, int sp) const {
    for (int b = 0; b < n; b += kWave) {
      const int i = b + lane;
      const int c = i < n ? get(i) : -1;
      const unsigned long long mask = __ballot(c >= 0);
      if (c >= 0) base[sp - 1 - __popcll(mask & ((1ull << lane) - 1ull))] = c;
      sp -= __popcll(mask);
    }
    return sp;
  }
  __device__ __forceinline__ void amin(int32_t* p, int v) const { atomicMin(p, v); }
  __device__ __forceinline__ void amax(int32_t* p, int v) const { atomicMax(p, v); }
  __device__ __forceinline__ void aor(int32_t* p, int v) const { atomicOr(p, v); }
  __device__ __forceinline__ void count(unsigned long long* p, unsigned long long v) const {
    if (lane == 0 && v) atomicAdd(p, v);
  }
  __device__ __forceinline__ unsigned long long fetch_add(unsigned long long* p, unsigned long long v) const {
    unsigned long long r = 0;
    if (lane == 0) r = atomicAdd(p, v);
    const unsigned lo = __shfl((unsigned)(r & 0xffffffffull), 0, kWave);
    const unsigned hi = __shfl((unsigned)(r >> 32), 0, kWave);
    return ((unsigned long long)hi << 32) | lo;
  }
};

constexpr int kPPMax = 368, kAPMax = 368, kMaxdMax = 528;

#define AGZ_SCRATCH(S)                                                        \
  __shared__ int8_t s_sb[kPPMax];                                             \
  __shared__ int32_t s_label[kPPMax + kWave];                                 \
  __shared__ int32_t s_minlib[kPPMax];                                        \
  __shared__ int32_t s_maxlib[kPPMax];                                        \
  __shared__ int8_t s_flag[kAPMax];                                           \
  __shared__ double s_dbuf[kAPMax];                                           \
  __shared__ int32_t s_path[kMaxdMax];                                        \
  Scratch S{s_sb, s_label, s_minlib, s_maxlib, s_flag, s_dbuf, s_path};

__global__ __launch_bounds__(kWave) void k_pre(View V) {
  AGZ_SCRATCH(S)
  HipWave w;
  game_pre(w, V, S, (int)blockIdx.x);
}

// the board updates of the leaves k_pre created (node_create_child, defer): one wave per leaf
__global__ __launch_bounds__(kWave) void k_expand(View V) {
  AGZ_SCRATCH(S)
  HipWave w;
  const int half = kMaxPend / 2;                 // a game rarely creates more than parallel_readouts leaves per step
  const int g = (int)blockIdx.x / half, i = (int)blockIdx.x % half;
  game_expand(w, V, S, g, i);
  game_expand(w, V, S, g, i + half);
}

__global__ __launch_bounds__(kWave) void k_post(View V) {
  AGZ_SCRATCH(S)
  HipWave w;
  game_post(w, V, S, (int)blockIdx.x);
}

// exclusive prefix sum of GameState::nleaves over the games -> leaf_base, total -> batch_count
// arena: two batches, one per network.  Black players' leaves (even slots) get rows [0, n0), White
// players' leaves rows [half, half + n1) with half = games*par/2: each network sees a contiguous
// batch at a FIXED base, so no row offset has to come back to the host.
__global__ __launch_bounds__(256) void k_scan_arena(View V) {
  __shared__ int part[256];
  const int t = threadIdx.x, npairs = V.games / 2;
  const int chunk = (npairs + 255) / 256;
  const int lo = t * chunk, hi = min(npairs, lo + chunk);
  for (int c = 0; c < 2; ++c) {
    int s = 0;
    for (int p = lo; p < hi; ++p) s += V.gs[2 * p + c].nleaves;
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int v = t >= o ? part[t - o] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    int base = part[t] - s + c * (npairs * V.par);
    for (int p = lo; p < hi; ++p) {
      V.gs[2 * p + c].leaf_base = base;
      base += V.gs[2 * p + c].nleaves;
    }
    if (t == 255) V.batch_count[c] = part[255];
    __syncthreads();
  }
  if (t == 255) atomicAdd(&V.counters[CT_STEPS], 1ull);
}

__global__ __launch_bounds__(256) void k_scan(View V) {
  __shared__ int part[256];
  const int t = threadIdx.x, G = V.games;
  const int chunk = (G + 255) / 256;
  const int lo = t * chunk, hi = min(G, lo + chunk);
  int s = 0;
  for (int g = lo; g < hi; ++g) s += V.gs[g].nleaves;
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int base = part[t] - s;
  for (int g = lo; g < hi; ++g) {
    V.gs[g].leaf_base = base;
    base += V.gs[g].nleaves;
  }
  if (t == 255) {
    *V.batch_count = part[255];
    atomicAdd(&V.counters[CT_STEPS], 1ull);
  }
}

// One workgroup of four waves per leaf slot, each wave a quarter of the leaf's (point, quad) items.  (Neither this, nor
// one wave per leaf, nor four leaves per workgroup changes its 35-39 us per 8192 leaves of 9x9: see leaf_features.)
constexpr int kWavesPerLeaf = 4;
__global__ __launch_bounds__(kWavesPerLeaf * kWave) void k_leaf_features(View V, int g0, int slots, float* x32, float* whcn) {
  const int slot = (int)blockIdx.x;
  if (slot >= slots) return;
  const int g = g0 + slot / V.par, k = slot % V.par;
  if (k >= V.gs[g].nleaves) return;
  HipWave w((int)(threadIdx.x & (kWave - 1)));
  const long row = (long)V.gs[g].leaf_base + k;
  leaf_features(w, V, g, k, x32 ? x32 + row * V.P * 32 : nullptr, whcn ? whcn + row * 17 * V.P : nullptr, (int)(threadIdx.x >> 6),
                kWavesPerLeaf);
}

// The Vector{Position} a caller-supplied network receives (mcts_play.jl:89): per collected leaf of slot g, its own
// board and the up-to-7 older boards behind the history planes (the same sources k_leaf_features reads), its NodeMeta
// and the move before its last one -- gathered into contiguous rows for ONE copy to the host.
struct LeafPosRow {
  NodeMeta m;
  int32_t node, plen, prev_move, pad;
};
__global__ __launch_bounds__(kWave) void k_leaf_positions(View V, int g, int8_t* boards8, LeafPosRow* rows) {
  const int k = blockIdx.x;
  if (k >= V.gs[g].nleaves) return;
  const long li = (long)g * V.par + k;
  const int P = V.P;
  for (int s = 0; s < 8; ++s) {
    const int src = V.leaf_featsrc[li * 8 + s];
    const int8_t* b = src >= 0 ? V.board + node_index(V, g, src) * V.PP : V.hist + ((long)g * 7 + (-src - 1)) * V.PP;
    for (int p = threadIdx.x; p < P; p += kWave) boards8[((long)k * 8 + s) * P + p] = b[p];
  }
  if (threadIdx.x == 0) {
    LeafPosRow r;
    r.node = V.leaf_node[li];
    r.plen = V.leaf_plen[li];
    r.m = V.meta[node_index(V, g, r.node)];
    r.prev_move = r.m.parent >= 0 ? V.meta[node_index(V, g, r.m.parent)].last_move : -1;
    r.pad = 0;
    rows[k] = r;
  }
}

__global__ __launch_bounds__(kWave) void k_tree_op(View V, TreeArgs T) {
  AGZ_SCRATCH(S)
  HipWave w;
  tree_op(w, V, S, T);
}

__global__ __launch_bounds__(kWave) void k_go_play(View V, const int8_t* boards, const int8_t* tp, const int32_t* ko,
                                                    const int32_t* moves, int B, int8_t* bo, int32_t* ko_o,
                                                    int32_t* nc, int32_t* st) {
  AGZ_SCRATCH(S)
  HipWave w;
  const int b = blockIdx.x;
  if (b >= B) return;
  go_play_one(w, V, S, boards + (long)b * V.P, tp[b], ko[b], moves[b], bo + (long)b * V.P, ko_o + b, nc + b, st + b);
}

__global__ __launch_bounds__(kWave) void k_go_legal(View V, const int8_t* boards, const int8_t* tp,
                                                     const int32_t* ko, int B, int8_t* out) {
  AGZ_SCRATCH(S)
  HipWave w;
  const int b = blockIdx.x;
  if (b >= B) return;
  go_legal_one(w, V, S, boards + (long)b * V.P, tp[b], ko[b], out + (long)b * V.A);
}

__global__ __launch_bounds__(kWave) void k_go_score(View V, const int8_t* boards, const float* komi, int B, float* out) {
  AGZ_SCRATCH(S)
  HipWave w;
  const int b = blockIdx.x;
  if (b >= B) return;
  go_score_one(w, V, S, boards + (long)b * V.P, komi[b], out + b);
}

// replay_position (board.jl:557-578) on the device: rebuild the 17 planes of positions of a game
// from its move list.  One wave walks one game: block b replays moves[off[b] .. off[b]+nm[b]) from
// the empty board and writes the features of the position BEFORE move k for every k >= emit_from[b]
// to out + out_off[b] + (k - emit_from[b]) * 17*P.  (record_features: one block, emit_from 0;
// get_replay_batch, train.jl:4-12: one block per sampled (game, ply), nm = ply+1, emit_from = ply.)
template <class W>
__device__ __forceinline__ void replay_emit(W& w, const View& V, Scratch& S, const int16_t* moves, int nm, int from,
                                            int8_t* hist, float* out) {
  const int P = V.P;
  // hist[0] = current board, hist[k] = k moves ago; avail = number of real older boards
  w.for_each(8 * V.PP, [&](int i) { hist[i] = 0; });
  w.sync();
  int tp = 1, avail = 0;
  for (int k = 0; k < nm; ++k) {
    if (k >= from) {
      w.for_each(P, [&](int p) {
        float* dst = out + (long)(k - from) * 17 * P;
        for (int s = 0; s < 8; ++s) {
          const int t = s <= avail ? s : avail;
          const int c = hist[t * V.PP + p];
          dst[(2 * s) * P + p] = c == tp ? 1.f : 0.f;
          dst[(2 * s + 1) * P + p] = c == -tp ? 1.f : 0.f;
        }
        dst[16 * P + p] = (float)tp;
      });
      w.sync();
    }
    if (k + 1 == nm) break;      // the last listed move is never needed
    // play move k on hist[0]
    const int a = moves[k];
    w.for_each(P, [&](int p) { S.sb[p] = hist[p]; });
    w.sync();
    int ncap = 0, nko = -1;
    if (a >= 0 && a < P) {
      label_components(w, V, S, true);
      group_liberties(w, V, S);
      apply_move_in_scratch(w, V, S, a, tp, &ncap, &nko);
    }
    for (int s = 7; s >= 1; --s) {
      w.for_each(P, [&](int p) { hist[s * V.PP + p] = hist[(s - 1) * V.PP + p]; });
      w.sync();
    }
    w.for_each(P, [&](int p) { hist[p] = S.sb[p]; });
    w.sync();
    tp = -tp;
    if (avail < 7) avail++;
  }
}

__global__ __launch_bounds__(kWave) void k_replay_features(View V, const int16_t* moves, const int32_t* off,
                                                            const int32_t* nms, const int32_t* emit_from,
                                                            const int64_t* out_off,
                                                            int8_t* hist_all /*[blocks][8][PP] scratch in HBM*/,
                                                            float* out_all) {
  AGZ_SCRATCH(S)
  HipWave w;
  const int b = blockIdx.x;
  replay_emit(w, V, S, moves + off[b], nms[b], emit_from[b], hist_all + (long)b * 8 * V.PP, out_all + out_off[b]);
}

// ---- the device replay arena (agz_replay_*): packed records
//      [agz_game_header 32 B | moves i16[n] | pad 4 | pis f32[n][A] | qs f32[n] | pad 8] back to back in HBM.

__host__ __device__ inline size_t packed_bytes(int A, int nm) {
  size_t b = sizeof(agz_game_header) + sizeof(int16_t) * (size_t)nm;
  b = (b + 3) & ~(size_t)3;
  b += sizeof(float) * (size_t)nm * A + sizeof(float) * (size_t)nm;
  return (b + 7) & ~(size_t)7;
}

// one workgroup per finished record k: copy it from the engine's record ring into dst + off[k]
__global__ __launch_bounds__(256) void k_pack_records(View V, const int64_t* off, uint8_t* dst, long first) {
  const long k = first + blockIdx.x;
  const agz_game_header h = V.fin_hdr[k];
  const int nm = h.num_moves, mgl = V.max_game_length, A = V.A;
  uint8_t* r = dst + off[blockIdx.x];
  if (threadIdx.x == 0) *reinterpret_cast<agz_game_header*>(r) = h;
  int16_t* mv = reinterpret_cast<int16_t*>(r + sizeof(agz_game_header));
  const size_t o_pi = (sizeof(agz_game_header) + sizeof(int16_t) * (size_t)nm + 3) & ~(size_t)3;
  float* pi = reinterpret_cast<float*>(r + o_pi);
  float* q = pi + (size_t)nm * A;
  for (int i = threadIdx.x; i < nm; i += 256) {
    mv[i] = V.fin_moves[k * mgl + i];
    q[i] = V.fin_q[k * mgl + i];
  }
  if ((nm & 1) && threadIdx.x == 0) mv[nm] = 0;                       // the 2 pad bytes in front of pis
  for (long i = threadIdx.x; i < (long)nm * A; i += 256) pi[i] = V.fin_pi[k * (long)mgl * A + i];
  const size_t end = o_pi + sizeof(float) * (size_t)nm * (A + 1);
  if ((end & 7) && threadIdx.x == 0) *reinterpret_cast<uint32_t*>(r + end) = 0u;   // pad to 8
}

// one wave per source chunk (= one rank's part of the padded all-gather): walk its records and write where
// each one starts: out_off[first[c] + i] = byte offset of record i of chunk c inside `buf`.  nrec[c] >= 0 is
// the record count the sender announced (checked); nrec[c] = -(cap + 1) means "unknown, at most cap".
__global__ __launch_bounds__(kWave) void k_index_records(const uint8_t* buf, const int64_t* chunk_off,
                                                          const int64_t* chunk_bytes, const int64_t* nrec,
                                                          const int64_t* first, int A, int mgl, int64_t* out_off,
                                                          agz_game_header* out_hdr, int64_t* found, int32_t* bad) {
  if (threadIdx.x != 0) return;
  const int c = blockIdx.x;
  int64_t o = chunk_off[c];
  const int64_t end = o + chunk_bytes[c];
  const int64_t want = nrec[c], cap = want < 0 ? -(want + 1) : want;
  int64_t i = 0;
  bool ok = true;
  while (o < end) {
    if (i >= cap || o + (int64_t)sizeof(agz_game_header) > end) { ok = false; break; }
    const agz_game_header h = *reinterpret_cast<const agz_game_header*>(buf + o);
    if (h.num_moves < 0 || h.num_moves > mgl || o + (int64_t)packed_bytes(A, h.num_moves) > end) { ok = false; break; }
    out_off[first[c] + i] = o;
    out_hdr[first[c] + i] = h;
    o += (int64_t)packed_bytes(A, h.num_moves);
    ++i;
  }
  if (!ok || o != end || (want >= 0 && i != want)) atomicAdd(bad, 1);
  found[c] = i;
}

// get_replay_batch (train.jl:4-12) straight from the arena: sample b = (record at rec_off[b], ply[b]):
// features of the position before move ply (replay_position, board.jl:557-578), pi of that move, z = result
__global__ __launch_bounds__(kWave) void k_replay_arena_batch(View V, const uint8_t* arena, const int64_t* rec_off,
                                                               const int32_t* ply, int8_t* hist_all, float* feats,
                                                               float* pi_out, float* z_out) {
  AGZ_SCRATCH(S)
  HipWave w;
  const int b = blockIdx.x, A = V.A;
  const uint8_t* r = arena + rec_off[b];
  const agz_game_header h = *reinterpret_cast<const agz_game_header*>(r);
  const int16_t* mv = reinterpret_cast<const int16_t*>(r + sizeof(agz_game_header));
  const size_t o_pi = (sizeof(agz_game_header) + sizeof(int16_t) * (size_t)h.num_moves + 3) & ~(size_t)3;
  const float* pi = reinterpret_cast<const float*>(r + o_pi) + (size_t)ply[b] * A;
  replay_emit(w, V, S, mv, ply[b] + 1, ply[b], hist_all + (long)b * 8 * V.PP, feats + (long)b * 17 * V.P);
  if (pi_out) w.for_each(A, [&](int i) { pi_out[(long)b * A + i] = pi[i]; });
  if (z_out && w.leader()) z_out[b] = (float)h.result;
}

__global__ void k_debug_draws(uint64_t seed, uint64_t game, uint32_t move, int n, double alpha, double* out) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < n) out[a] = agz_dirichlet_gamma(seed, game, move, (uint32_t)a, alpha);
}

__global__ void k_debug_math(View V, int op, const double* x, const double* y, int n, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0.0;
  switch (op) {
    case 0: r = agz_log(x[i]); break;
    case 1: r = agz_exp(x[i]); break;
    case 2: r = agz_pow(x[i], 0.98); break;
    case 3: r = (double)sqrtf((float)x[i]); break;
    case 4: r = (double)((float)x[i] / (float)y[i]); break;
    case 5: {
      const float denom = 1.0f + (float)y[i];
      const float q = (float)x[i] / denom;
      const float qs = q * -1.0f;
      const double scale = puct_scale(V, (float)y[i] + 7.0f);
      r = (double)qs + (scale * (double)0.25f) / (double)denom;
    } break;
  }
  out[i] = r;
}

// ------------------------------------------------------------------ host engine

static void validate(const agz_config& c) {
  AGZ_REQUIRE(c.board_size >= 2 && c.board_size <= 19, AGZ_BAD_ARGUMENT, "board_size %d not in 2..19", c.board_size);
  AGZ_REQUIRE(c.tower_height >= 0 && c.tower_height <= 64, AGZ_BAD_ARGUMENT, "tower_height %d", c.tower_height);
  AGZ_REQUIRE(c.games >= 1, AGZ_BAD_ARGUMENT, "games must be >= 1");
  AGZ_REQUIRE(c.num_readouts >= 1, AGZ_BAD_ARGUMENT, "num_readouts must be >= 1");
  AGZ_REQUIRE(c.parallel_readouts >= 1 && c.parallel_readouts <= kMaxPar, AGZ_BAD_ARGUMENT,
              "parallel_readouts %d not in 1..%d", c.parallel_readouts, kMaxPar);
  AGZ_REQUIRE(!c.arena_mode || c.games % 2 == 0, AGZ_BAD_ARGUMENT, "arena_mode needs an even number of slots");
  // the word that is pool_policy now was `stagger_moves` until round 2 (agz_debug_set_stagger since) and a must-be-zero
  // reserved1 until round 4: an old-ABI caller that still puts a move count there must hear about it
  AGZ_REQUIRE(c.reserved0 == 0.f, AGZ_BAD_ARGUMENT, "agz_config.reserved0 must be 0");
  AGZ_REQUIRE(c.pool_policy == AGZ_POOL_MOVE_EARLY || c.pool_policy == AGZ_POOL_STALL, AGZ_BAD_ARGUMENT,
              "agz_config.pool_policy %d: AGZ_POOL_MOVE_EARLY (0) or AGZ_POOL_STALL (1); this word was stagger_moves / "
              "reserved1 in earlier headers (use agz_debug_set_stagger)", c.pool_policy);
  AGZ_REQUIRE(c.max_nodes_per_game >= 0, AGZ_BAD_ARGUMENT, "max_nodes_per_game %d", c.max_nodes_per_game);
}

Engine::Engine(const agz_config& cfg) : cfg_(cfg) {
  validate(cfg);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  AGZ_REQUIRE(e == hipSuccess && ndev > 0, AGZ_HIP_ERROR,
              "no HIP device available (%s): libagz has no CPU fallback", hipGetErrorString(e));
  AGZ_REQUIRE(cfg.device >= 0 && cfg.device < ndev, AGZ_BAD_ARGUMENT, "device %d of %d", cfg.device, ndev);
  AGZ_HIP(hipSetDevice(cfg.device));
  hipDeviceProp_t prop;
  AGZ_HIP(hipGetDeviceProperties(&prop, cfg.device));
  AGZ_REQUIRE(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0, AGZ_HIP_ERROR,
              "device %d is %s; libagz is built for gfx950 (MI355X) only", cfg.device, prop.gcnArchName);
  AGZ_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  fill_dims(V_, cfg);
  V_.defer_expand = 1;      // k_expand follows every k_pre (step, select_external)
  AGZ_REQUIRE(V_.PP <= kPPMax && V_.AP <= kAPMax && V_.maxd <= kMaxdMax, AGZ_BAD_ARGUMENT, "board too large");
  for_each_buffer(V_, [&](auto*& p, size_t n) {
    using T = std::remove_reference_t<decltype(*p)>;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    void* q = nullptr;
    AGZ_HIP(hipMalloc(&q, bytes));
    AGZ_HIP(hipMemsetAsync(q, 0, bytes, stream_));
    p = (T*)q;
    bufs_.push_back(q);
    state_bytes_ += bytes;
  });
  bcap_ = V_.games * V_.par;
  d_x32_.alloc((size_t)bcap_ * V_.P * 32);
  d_pi_.alloc((size_t)bcap_ * V_.A);
  d_v_.alloc((size_t)bcap_);
  V_.pi = d_pi_.p;
  V_.v = d_v_.p;
  s_iout_.alloc(4);
  net_.reset(new Net(V_.N, cfg.tower_height, stream_));
  if (cfg.arena_mode) net2_.reset(new Net(V_.N, cfg.tower_height, stream_));
  // every slot idle-retired until start()
  std::vector<GameState> gs(V_.games);
  std::memset(gs.data(), 0, sizeof(GameState) * gs.size());
  for (auto& g : gs) g.phase = G_RETIRED;
  AGZ_HIP(hipMemcpyAsync(V_.gs, gs.data(), sizeof(GameState) * gs.size(), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

Engine::~Engine() {
  (void)hipStreamSynchronize(stream_);
  for (auto e : sprof_ev_) (void)hipEventDestroy(e);
  // the trainers point into their networks (host-sync callback) and own device buffers: they go first, then the
  // networks, and the stream they all enqueue on last
  trainer_.reset();
  trainer2_.reset();
  for (void* p : bufs_) (void)hipFree(p);
  net_.reset();
  net2_.reset();
  if (stream_) (void)hipStreamDestroy(stream_);
}

void Engine::sync() {
  AGZ_HIP(hipStreamSynchronize(stream_));
  // an error word a persistent tower launch raised is reported by the first synchronising call behind it, not only by
  // the next persistent forward
  net_->check_async_error();
  if (net2_) net2_->check_async_error();
}

void Engine::net_select(int which) {
  AGZ_REQUIRE(which == 0 || (which == 1 && net2_), AGZ_BAD_ARGUMENT,
              "network %d does not exist (network 1 needs arena_mode)", which);
  net_sel_ = which;
}

void Engine::start(int64_t total_games) {
  V_.total_games = total_games;
  rec_sent_ = 0;
  abandoned_ = 0;
  stepped_ = false;
  AGZ_HIP(hipMemsetAsync(V_.counters, 0, sizeof(unsigned long long) * CT_COUNT, stream_));
  AGZ_HIP(hipMemsetAsync(V_.ar_hdr, 0, sizeof(int32_t) * 5 * (V_.games / 2 + 1), stream_));
  std::vector<GameState> gs(V_.games);
  std::memset(gs.data(), 0, sizeof(GameState) * gs.size());
  for (auto& g : gs) g.phase = G_IDLE;
  AGZ_HIP(hipMemcpyAsync(V_.gs, gs.data(), sizeof(GameState) * gs.size(), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::step(int nsteps) {
  AGZ_REQUIRE(!cfg_.external_network, AGZ_BAD_ARGUMENT,
              "engine was created with external_network=1: use select/incorporate");
  stepped_ = stepped_ || nsteps > 0;
  for (int s = 0; s < nsteps; ++s) {
    // (search-kernel timing, bench.py: seven events per step on the engine's stream; off: no event is recorded)
    hipEvent_t* ev = sprof_on_ && sprof_n_ < kSearchProfMax ? &sprof_ev_[(size_t)7 * sprof_n_] : nullptr;
    if (ev) (void)hipEventRecord(ev[0], stream_);
    hipLaunchKernelGGL(k_pre, dim3(V_.games), dim3(kWave), 0, stream_, V_);
    if (ev) (void)hipEventRecord(ev[1], stream_);
    hipLaunchKernelGGL(k_expand, dim3(V_.games * (kMaxPend / 2)), dim3(kWave), 0, stream_, V_);
    if (ev) (void)hipEventRecord(ev[2], stream_);
    hipLaunchKernelGGL(cfg_.arena_mode ? k_scan_arena : k_scan, dim3(1), dim3(256), 0, stream_, V_);
    if (ev) (void)hipEventRecord(ev[3], stream_);
    hipLaunchKernelGGL(k_leaf_features, dim3(bcap_), dim3(kWavesPerLeaf * kWave), 0, stream_, V_, 0, bcap_, d_x32_.p, (float*)nullptr);
    if (ev) (void)hipEventRecord(ev[4], stream_);
    if (cfg_.arena_mode) {   // evaluate(): Black's players ask network 0, White's network 1
      const int half = bcap_ / 2;
      net_->forward(d_x32_.p, V_.batch_count, half, d_pi_.p, d_v_.p);
      net2_->forward(d_x32_.p + (size_t)half * V_.P * 32, V_.batch_count + 1, half, d_pi_.p + (size_t)half * V_.A,
                     d_v_.p + half);
    } else {
      net_->forward(d_x32_.p, V_.batch_count, bcap_, d_pi_.p, d_v_.p);
    }
    if (ev) (void)hipEventRecord(ev[5], stream_);
    hipLaunchKernelGGL(k_post, dim3(V_.games), dim3(kWave), 0, stream_, V_);
    if (ev) {
      (void)hipEventRecord(ev[6], stream_);
      ++sprof_n_;
    }
  }
  AGZ_HIP(hipGetLastError());
}

void Engine::profile_search_enable(bool on) {
  if (on && sprof_ev_.empty()) {
    sprof_ev_.resize((size_t)7 * kSearchProfMax);
    for (auto& e : sprof_ev_) AGZ_HIP(hipEventCreate(&e));
  }
  sprof_on_ = on;
  if (on) sprof_n_ = 0;
}

// ms5: summed milliseconds of k_pre, k_expand, k_scan, k_leaf_features, k_post over the timed steps
void Engine::profile_search_read(double* ms5, int64_t* steps) {
  AGZ_HIP(hipStreamSynchronize(stream_));
  for (int k = 0; k < 5; ++k) ms5[k] = 0.0;
  static const int a[5] = {0, 1, 2, 3, 5}, b[5] = {1, 2, 3, 4, 6};
  for (int i = 0; i < sprof_n_; ++i)
    for (int k = 0; k < 5; ++k) {
      float ms = 0.f;
      AGZ_HIP(hipEventElapsedTime(&ms, sprof_ev_[(size_t)7 * i + a[k]], sprof_ev_[(size_t)7 * i + b[k]]));
      ms5[k] += ms;
    }
  if (steps) *steps = sprof_n_;
}

int Engine::select_external() {
  stepped_ = true;
  hipLaunchKernelGGL(k_pre, dim3(V_.games), dim3(kWave), 0, stream_, V_);
  hipLaunchKernelGGL(k_expand, dim3(V_.games * (kMaxPend / 2)), dim3(kWave), 0, stream_, V_);
  hipLaunchKernelGGL(cfg_.arena_mode ? k_scan_arena : k_scan, dim3(1), dim3(256), 0, stream_, V_);
  int32_t n[2] = {0, 0};
  AGZ_HIP(hipMemcpyAsync(n, V_.batch_count, sizeof(int32_t) * 2, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  external_batch_ = n[0];
  external_batch2_ = cfg_.arena_mode ? n[1] : 0;
  return external_batch_ + external_batch2_;
}

void Engine::arena_counts(int32_t* out) const {
  out[0] = external_batch_;
  out[1] = external_batch2_;
}

void Engine::leaf_features_external(float* feats_out) {
  const int B = external_batch_, B2 = external_batch2_;
  if (B + B2 <= 0) return;
  const size_t per = (size_t)17 * V_.P;
  d_whcn_.ensure((size_t)bcap_ * per);
  hipLaunchKernelGGL(k_leaf_features, dim3(bcap_), dim3(kWavesPerLeaf * kWave), 0, stream_, V_, 0, bcap_, (float*)nullptr, d_whcn_.p);
  if (B) AGZ_HIP(hipMemcpyAsync(feats_out, d_whcn_.p, sizeof(float) * (size_t)B * per, hipMemcpyDeviceToHost, stream_));
  // arena: the White players' rows follow the Black players' rows in the host buffer
  if (B2) AGZ_HIP(hipMemcpyAsync(feats_out + (size_t)B * per, d_whcn_.p + (size_t)(bcap_ / 2) * per,
                                 sizeof(float) * (size_t)B2 * per, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::incorporate_external(const float* pi, const float* v) {
  const int B = external_batch_, B2 = external_batch2_;
  if (B > 0) {
    AGZ_HIP(hipMemcpyAsync(d_pi_.p, pi, sizeof(float) * (size_t)B * V_.A, hipMemcpyHostToDevice, stream_));
    AGZ_HIP(hipMemcpyAsync(d_v_.p, v, sizeof(float) * (size_t)B, hipMemcpyHostToDevice, stream_));
  }
  if (B2 > 0) {
    const size_t half = (size_t)bcap_ / 2;
    AGZ_HIP(hipMemcpyAsync(d_pi_.p + half * V_.A, pi + (size_t)B * V_.A, sizeof(float) * (size_t)B2 * V_.A,
                           hipMemcpyHostToDevice, stream_));
    AGZ_HIP(hipMemcpyAsync(d_v_.p + half, v + B, sizeof(float) * (size_t)B2, hipMemcpyHostToDevice, stream_));
  }
  hipLaunchKernelGGL(k_post, dim3(V_.games), dim3(kWave), 0, stream_, V_);
  AGZ_HIP(hipStreamSynchronize(stream_));
  external_batch_ = 0;
  external_batch2_ = 0;
}

void Engine::debug_set_stagger(int moves) {
  AGZ_REQUIRE(moves >= 0 && !cfg_.arena_mode, AGZ_BAD_ARGUMENT, "stagger: >= 0 moves, not in arena_mode");
  AGZ_REQUIRE(!stepped_, AGZ_BAD_ARGUMENT, "stagger: set it before the first step of a run (before or right after agz_selfplay_start)");
  V_.stagger = moves;            // the View travels to the kernels by value: effective from the next launch
}

void Engine::debug_live_record(int g, int k, uint64_t* game_id, int32_t* num_moves, int32_t* move, float* pi, float* q) {
  AGZ_REQUIRE(g >= 0 && g < V_.games, AGZ_BAD_ARGUMENT, "slot %d of %d", g, V_.games);
  GameState G;
  AGZ_HIP(hipMemcpyAsync(&G, V_.gs + g, sizeof(G), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  if (game_id) *game_id = G.game_id;
  if (num_moves) *num_moves = G.nqs;
  if (k < 0) return;
  AGZ_REQUIRE(k < G.nqs, AGZ_BAD_ARGUMENT, "slot %d has recorded %d moves", g, G.nqs);
  const long at = (long)g * V_.max_game_length + k;
  int16_t m = 0;
  AGZ_HIP(hipMemcpyAsync(&m, V_.rec_moves + at, sizeof(m), hipMemcpyDeviceToHost, stream_));
  if (q) AGZ_HIP(hipMemcpyAsync(q, V_.rec_q + at, sizeof(float), hipMemcpyDeviceToHost, stream_));
  if (pi) AGZ_HIP(hipMemcpyAsync(pi, V_.rec_pi + at * V_.A, sizeof(float) * V_.A, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  if (move) *move = m;
}

int Engine::debug_counters(uint64_t* out, int cap) {
  unsigned long long c[CT_COUNT];
  AGZ_HIP(hipMemcpyAsync(c, V_.counters, sizeof(c), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  for (int i = 0; i < CT_COUNT && i < cap; ++i) out[i] = c[i];
  return CT_COUNT;
}

void Engine::stats(agz_stats* out) {
  unsigned long long c[CT_COUNT];
  AGZ_HIP(hipMemcpyAsync(c, V_.counters, sizeof(c), hipMemcpyDeviceToHost, stream_));
  std::vector<GameState> gs(V_.games);
  AGZ_HIP(hipMemcpyAsync(gs.data(), V_.gs, sizeof(GameState) * gs.size(), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  net_->check_async_error();
  if (net2_) net2_->check_async_error();
  std::memset(out, 0, sizeof(*out));
  out->steps = (int64_t)c[CT_STEPS];
  out->positions = (int64_t)c[CT_POSITIONS];
  out->games_started = (int64_t)c[CT_STARTED];
  out->games_finished = (int64_t)c[CT_FINISHED];
  out->evals = (int64_t)c[CT_EVALS];
  out->duplicate_evals = (int64_t)c[CT_DUP];
  out->terminal_visits = (int64_t)c[CT_TERMINAL];
  out->pool_exhausted = (int64_t)c[CT_POOL_EXHAUSTED];
  out->resigned_games = (int64_t)c[CT_RESIGNED];
  out->root_visits = (int64_t)c[CT_ROOTVISITS];
  out->records_dropped = c[CT_RECORDED] > (unsigned long long)V_.fin_cap ? (int64_t)(c[CT_RECORDED] - V_.fin_cap) : 0;
  out->pool_short_searches = (int64_t)c[CT_POOL_SHORT];
  out->peak_nodes_per_game = (int64_t)c[CT_PEAK_NODES];
  out->node_capacity = V_.cap;
  out->abandoned_games = abandoned_;
  for (const auto& g : gs) {
    out->nodes_in_use += g.nodes_used;
    out->peak_nodes_per_game = std::max<int64_t>(out->peak_nodes_per_game, g.nodes_used);
    out->live_games += (g.phase != G_RETIRED && g.phase != G_IDLE);
    out->stalled_games += (g.stalled != 0 && g.phase == G_SEARCH);
  }
}

// Per slot: AGZ_OK, or AGZ_POOL_EXHAUSTED for a game that is waiting on a full node pool (AGZ_POOL_STALL, or nothing
// to play yet); nodes its tree holds; moves it has played.  The host's side of "recoverable": look, then
// agz_slot_abandon the games it gives up on -- or let AGZ_POOL_MOVE_EARLY (the default) keep them moving.
void Engine::slot_status(int32_t* status, int32_t* nodes, int32_t* moves) {
  std::vector<GameState> gs(V_.games);
  AGZ_HIP(hipMemcpyAsync(gs.data(), V_.gs, sizeof(GameState) * gs.size(), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  for (int g = 0; g < V_.games; ++g) {
    if (status) status[g] = (gs[g].stalled && gs[g].phase == G_SEARCH) ? AGZ_POOL_EXHAUSTED : (gs[g].err == AGZ_POOL_EXHAUSTED ? AGZ_OK : gs[g].err);
    if (nodes) nodes[g] = gs[g].nodes_used;
    if (moves) moves[g] = gs[g].move_count;
  }
}

// Drop the game a slot is playing, without a record: the slot claims the next game id at the next step (or retires
// when the quota of agz_selfplay_start is used up).  Not for arena or single-tree (manual) slots.
void Engine::slot_abandon(int g) {
  AGZ_REQUIRE(g >= 0 && g < V_.games, AGZ_BAD_ARGUMENT, "slot %d of %d", g, V_.games);
  AGZ_REQUIRE(!V_.arena, AGZ_BAD_ARGUMENT, "arena slots come in pairs: not abandonable one by one");
  GameState G;
  AGZ_HIP(hipMemcpyAsync(&G, V_.gs + g, sizeof(G), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  AGZ_REQUIRE(G.phase == G_SEARCH || G.phase == G_INIT || G.phase == G_INIT_WAIT, AGZ_BAD_ARGUMENT,
              "slot %d is not playing a self-play game (phase %d)", g, G.phase);
  G.phase = G_IDLE;
  G.nleaves = 0;
  G.npend = 0;
  G.err = 0;
  G.stalled = 0;
  AGZ_HIP(hipMemcpyAsync(V_.gs + g, &G, sizeof(G), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  ++abandoned_;
}

// ---- records

int64_t Engine::records_count() {
  unsigned long long f = 0;
  AGZ_HIP(hipMemcpyAsync(&f, V_.counters + CT_RECORDED, sizeof(f), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  net_->check_async_error();
  if (net2_) net2_->check_async_error();
  return (int64_t)std::min<unsigned long long>(f, (unsigned long long)V_.fin_cap);
}

void Engine::record_header(int64_t k, agz_game_header* out) {
  AGZ_REQUIRE(k >= 0 && k < records_count(), AGZ_BAD_ARGUMENT, "record %lld out of range", (long long)k);
  AGZ_HIP(hipMemcpyAsync(out, V_.fin_hdr + k, sizeof(*out), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::record_game(int64_t k, int16_t* moves, float* pis, float* qs) {
  agz_game_header h;
  record_header(k, &h);
  const size_t mgl = V_.max_game_length, nm = (size_t)h.num_moves;
  if (nm == 0) return;
  if (moves) AGZ_HIP(hipMemcpyAsync(moves, V_.fin_moves + k * mgl, sizeof(int16_t) * nm, hipMemcpyDeviceToHost, stream_));
  if (qs) AGZ_HIP(hipMemcpyAsync(qs, V_.fin_q + k * mgl, sizeof(float) * nm, hipMemcpyDeviceToHost, stream_));
  if (pis) AGZ_HIP(hipMemcpyAsync(pis, V_.fin_pi + k * mgl * V_.A, sizeof(float) * nm * V_.A, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

static size_t packed_record_bytes(const View& V, int nm) { return packed_bytes(V.A, nm); }

// bytes of the packed form of records [first, count)
int64_t Engine::records_packed_size(int64_t first) {
  const int64_t n = records_count() - first;
  if (n <= 0) return 0;
  std::vector<agz_game_header> h((size_t)n);
  AGZ_HIP(hipMemcpy(h.data(), V_.fin_hdr + first, sizeof(agz_game_header) * (size_t)n, hipMemcpyDeviceToHost));
  size_t total = 0;
  for (auto& x : h) total += packed_record_bytes(V_, x.num_moves);
  return (int64_t)total;
}

// pack the finished records [first, count) into `dst` (device memory, >= records_packed_size(first) bytes): one D2H
// of the headers to lay the records out, one kernel to move them.  Returns the number of records packed.
int64_t Engine::pack_records_device(uint8_t* dst, int64_t capacity, int64_t* nbytes, int64_t first) {
  const int64_t n = records_count() - first;
  *nbytes = 0;
  if (n <= 0) return 0;
  std::vector<agz_game_header> h((size_t)n);
  AGZ_HIP(hipMemcpyAsync(h.data(), V_.fin_hdr + first, sizeof(agz_game_header) * (size_t)n, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  std::vector<int64_t> off((size_t)n);
  size_t total = 0;
  for (int64_t k = 0; k < n; ++k) {
    off[k] = (int64_t)total;
    total += packed_record_bytes(V_, h[k].num_moves);
  }
  AGZ_REQUIRE((int64_t)total <= capacity, AGZ_BAD_ARGUMENT, "export buffer too small");
  s_i64a_.ensure((size_t)n);
  AGZ_HIP(hipMemcpyAsync(s_i64a_.p, off.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_pack_records, dim3((unsigned)n), dim3(256), 0, stream_, V_, (const int64_t*)s_i64a_.p, dst, (long)first);
  AGZ_HIP(hipGetLastError());
  AGZ_HIP(hipStreamSynchronize(stream_));      // `off` is stack-owned
  *nbytes = (int64_t)total;
  return n;
}

void Engine::records_export_packed(void* dst, int64_t capacity, bool is_device) {
  int64_t nb = 0;
  if (is_device) {
    pack_records_device((uint8_t*)dst, capacity, &nb);
    return;
  }
  const int64_t need = records_packed_size();
  AGZ_REQUIRE(need <= capacity, AGZ_BAD_ARGUMENT, "export buffer too small");
  if (need == 0) return;
  s_pack_.ensure((size_t)need);
  pack_records_device(s_pack_.p, need, &nb);
  AGZ_HIP(hipMemcpyAsync(dst, s_pack_.p, (size_t)nb, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

// ---- device replay arena

void Engine::replay_reserve(size_t bytes) {
  if (bytes <= rp_buf_.n) return;
  size_t cap = std::max<size_t>(bytes, std::max<size_t>(2 * rp_buf_.n, (size_t)1 << 20));
  DevBuf<uint8_t> nb;
  nb.alloc(cap);
  if (rp_used_) AGZ_HIP(hipMemcpyAsync(nb.p, rp_buf_.p, rp_used_, hipMemcpyDeviceToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  std::swap(nb.p, rp_buf_.p);
  std::swap(nb.n, rp_buf_.n);
}

// Append packed records lying in device memory as `chunks` (offset, bytes, expected record count or -1) of
// `dbuf` -- a padded all-gather receive buffer has one chunk per rank.  The variable-length records are
// indexed ON THE DEVICE (one wave walks one chunk); only the index (8 B + 32 B per game) visits the host.
int64_t Engine::replay_ingest_chunks(const uint8_t* dbuf, const std::vector<int64_t>& coff,
                                     const std::vector<int64_t>& cbytes, const std::vector<int64_t>& cnrec) {
  const int nc = (int)coff.size();
  int64_t total_rec = 0, total_bytes = 0;
  std::vector<int64_t> first((size_t)nc), nrec(cnrec);
  for (int c = 0; c < nc; ++c) {
    // an unknown count is bounded by the smallest record (a bare header)
    if (nrec[c] < 0) nrec[c] = -(cbytes[c] / (int64_t)sizeof(agz_game_header)) - 1;
    first[c] = total_rec;
    total_rec += nrec[c] < 0 ? -(nrec[c] + 1) : nrec[c];
    total_bytes += cbytes[c];
  }
  if (total_rec == 0 || total_bytes == 0) return 0;
  // device scratch: [coff | cbytes | nrec | first] int64 x nc, then offsets, found counts, bad flag, headers
  DevBuf<int64_t> d_meta, d_off;
  DevBuf<agz_game_header> d_hdr;
  DevBuf<int32_t> d_flag;
  d_meta.alloc((size_t)5 * nc);
  d_off.alloc((size_t)total_rec);
  d_hdr.alloc((size_t)total_rec);
  d_flag.alloc(1);
  std::vector<int64_t> meta;
  meta.insert(meta.end(), coff.begin(), coff.end());
  meta.insert(meta.end(), cbytes.begin(), cbytes.end());
  meta.insert(meta.end(), nrec.begin(), nrec.end());
  meta.insert(meta.end(), first.begin(), first.end());
  meta.resize((size_t)5 * nc, 0);
  AGZ_HIP(hipMemcpyAsync(d_meta.p, meta.data(), sizeof(int64_t) * meta.size(), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemsetAsync(d_flag.p, 0, sizeof(int32_t), stream_));
  hipLaunchKernelGGL(k_index_records, dim3(nc), dim3(kWave), 0, stream_, dbuf, (const int64_t*)d_meta.p,
                     (const int64_t*)(d_meta.p + nc), (const int64_t*)(d_meta.p + 2 * nc),
                     (const int64_t*)(d_meta.p + 3 * nc), V_.A, V_.max_game_length, d_off.p, d_hdr.p, d_meta.p + 4 * nc,
                     d_flag.p);
  AGZ_HIP(hipGetLastError());
  std::vector<int64_t> off((size_t)total_rec), found((size_t)nc);
  std::vector<agz_game_header> hdr((size_t)total_rec);
  int32_t bad = 0;
  AGZ_HIP(hipMemcpyAsync(off.data(), d_off.p, sizeof(int64_t) * off.size(), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(hdr.data(), d_hdr.p, sizeof(agz_game_header) * hdr.size(), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(found.data(), d_meta.p + 4 * nc, sizeof(int64_t) * nc, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(&bad, d_flag.p, sizeof(bad), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  AGZ_REQUIRE(bad == 0, AGZ_BAD_ARGUMENT, "packed records are malformed (a record runs past its chunk or a count does not match)");
  replay_reserve(rp_used_ + (size_t)total_bytes);
  int64_t added = 0;
  for (int c = 0; c < nc; ++c) {
    if (cbytes[c] == 0) continue;
    AGZ_HIP(hipMemcpyAsync(rp_buf_.p + rp_used_, dbuf + coff[c], (size_t)cbytes[c], hipMemcpyDeviceToDevice, stream_));
    for (int64_t i = 0; i < found[c]; ++i) {
      rp_off_.push_back((int64_t)rp_used_ + off[first[c] + i] - coff[c]);
      rp_hdr_.push_back(hdr[first[c] + i]);
      rp_positions_ += hdr[first[c] + i].num_moves;
    }
    added += found[c];
    rp_used_ += (size_t)cbytes[c];
  }
  AGZ_HIP(hipStreamSynchronize(stream_));
  return added;
}

int64_t Engine::replay_ingest(const void* packed, int64_t nbytes, bool is_device) {
  AGZ_REQUIRE(nbytes >= 0 && (nbytes == 0 || packed), AGZ_BAD_ARGUMENT, "bad packed buffer");
  AGZ_REQUIRE(nbytes % 8 == 0, AGZ_BAD_ARGUMENT, "packed records are a multiple of 8 bytes");
  if (nbytes == 0) return 0;
  const uint8_t* d = (const uint8_t*)packed;
  if (!is_device) {
    s_pack_.ensure((size_t)nbytes);
    AGZ_HIP(hipMemcpyAsync(s_pack_.p, packed, (size_t)nbytes, hipMemcpyHostToDevice, stream_));
    d = s_pack_.p;
  }
  return replay_ingest_chunks(d, {0}, {nbytes}, {-1});
}

int64_t Engine::replay_ingest_gathered(const void* buf, bool is_device, size_t nbytes, const std::vector<int64_t>& coff,
                                       const std::vector<int64_t>& cbytes, const std::vector<int64_t>& cnrec) {
  const uint8_t* d = (const uint8_t*)buf;
  if (!is_device) {
    s_pack_.ensure(std::max<size_t>(nbytes, 8));
    AGZ_HIP(hipMemcpyAsync(s_pack_.p, buf, nbytes, hipMemcpyHostToDevice, stream_));
    d = s_pack_.p;
  }
  return replay_ingest_chunks(d, coff, cbytes, cnrec);
}

// files the engine's own finished records that no exchange has filed yet (the watermark rec_sent_ moves; the
// record ring itself is the caller's to clear)
int64_t Engine::replay_ingest_local() {
  const int64_t need = records_packed_size(rec_sent_);
  if (need == 0) return 0;
  s_pack_.ensure((size_t)need);
  int64_t nb = 0;
  const int64_t n = pack_records_device(s_pack_.p, need, &nb, rec_sent_);
  const int64_t added = replay_ingest_chunks(s_pack_.p, {0}, {nb}, {n});
  rec_sent_ += n;
  return added;
}

void Engine::replay_header(int64_t k, agz_game_header* out) const {
  AGZ_REQUIRE(k >= 0 && k < (int64_t)rp_hdr_.size(), AGZ_BAD_ARGUMENT, "replay game %lld out of range", (long long)k);
  *out = rp_hdr_[(size_t)k];
}

void Engine::replay_game(int64_t k, int16_t* moves, float* pis, float* qs) {
  agz_game_header h;
  replay_header(k, &h);
  const size_t nm = (size_t)h.num_moves;
  if (nm == 0) return;
  const uint8_t* r = rp_buf_.p + rp_off_[(size_t)k];
  const size_t o_pi = (sizeof(agz_game_header) + sizeof(int16_t) * nm + 3) & ~(size_t)3;
  if (moves) AGZ_HIP(hipMemcpyAsync(moves, r + sizeof(agz_game_header), sizeof(int16_t) * nm, hipMemcpyDeviceToHost, stream_));
  if (pis) AGZ_HIP(hipMemcpyAsync(pis, r + o_pi, sizeof(float) * nm * V_.A, hipMemcpyDeviceToHost, stream_));
  if (qs) AGZ_HIP(hipMemcpyAsync(qs, r + o_pi + sizeof(float) * nm * V_.A, sizeof(float) * nm, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

// the FIFO window of train() (`shrink`, train.jl:52): forget the oldest games until at most max_positions remain
void Engine::replay_trim(int64_t max_positions) {
  AGZ_REQUIRE(max_positions >= 0, AGZ_BAD_ARGUMENT, "negative window");
  size_t drop = 0;
  int64_t pos = rp_positions_;
  while (drop < rp_hdr_.size() && pos > max_positions) pos -= rp_hdr_[drop++].num_moves;
  if (drop == 0) return;
  if (drop == rp_hdr_.size()) { replay_clear(); return; }
  const size_t cut = (size_t)rp_off_[drop], keep = rp_used_ - cut;
  DevBuf<uint8_t> nb;
  nb.alloc(std::max(keep, (size_t)1 << 20));
  AGZ_HIP(hipMemcpyAsync(nb.p, rp_buf_.p + cut, keep, hipMemcpyDeviceToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  std::swap(nb.p, rp_buf_.p);
  std::swap(nb.n, rp_buf_.n);
  rp_off_.erase(rp_off_.begin(), rp_off_.begin() + drop);
  rp_hdr_.erase(rp_hdr_.begin(), rp_hdr_.begin() + drop);
  for (auto& o : rp_off_) o -= (int64_t)cut;
  rp_used_ = keep;
  rp_positions_ = pos;
}

void Engine::replay_clear() {
  rp_off_.clear();
  rp_hdr_.clear();
  rp_used_ = 0;
  rp_positions_ = 0;
}

void Engine::train_step(const float* feats, const float* pi, const float* z, int B, bool is_device, float eta, float rho,
                        float* losses_out) {
  std::unique_ptr<Trainer>& tr = (net_sel_ && net2_) ? trainer2_ : trainer_;
  if (!tr) tr.reset(new Trainer(net(), stream_));
  tr->step(feats, pi, z, B, is_device, eta, rho, losses_out);
}

void Engine::train_reset() {
  std::unique_ptr<Trainer>& tr = (net_sel_ && net2_) ? trainer2_ : trainer_;
  if (tr) tr->reset();
}

// every parameter of the selected network as one flat vector, in the order of the device master (Net::weight_keys)
std::vector<float> Engine::weights_flat() {
  std::vector<std::pair<int, int>> keys;
  Net::weight_keys(net().tower(), keys);
  std::vector<float> w;
  for (auto& lk : keys) {
    const int64_t n = net().param_count(lk.first, lk.second);
    const size_t at = w.size();
    w.resize(at + (size_t)n);
    net().get(lk.first, lk.second, w.data() + at, n);
  }
  return w;
}

void Engine::weights_set_flat(const std::vector<float>& w) {
  std::vector<std::pair<int, int>> keys;
  Net::weight_keys(net().tower(), keys);
  size_t at = 0;
  for (auto& lk : keys) {
    const int64_t n = net().param_count(lk.first, lk.second);
    AGZ_REQUIRE(at + (size_t)n <= w.size(), AGZ_BAD_SHAPE, "flat weight vector too short");
    net().set(lk.first, lk.second, w.data() + at, n);
    at += (size_t)n;
  }
  AGZ_REQUIRE(at == w.size(), AGZ_BAD_SHAPE, "flat weight vector too long");
}

void Engine::replay_batch(const int64_t* game, const int32_t* ply, int B, float* feats, float* pi, float* z,
                          bool out_is_device) {
  AGZ_REQUIRE(B >= 0, AGZ_BAD_ARGUMENT, "negative batch");
  if (B == 0) return;
  AGZ_REQUIRE(game && ply && feats, AGZ_BAD_ARGUMENT, "null pointer");
  std::vector<int64_t> off((size_t)B);
  for (int b = 0; b < B; ++b) {
    AGZ_REQUIRE(game[b] >= 0 && game[b] < (int64_t)rp_hdr_.size(), AGZ_BAD_ARGUMENT, "sample %d: game out of range", b);
    AGZ_REQUIRE(ply[b] >= 0 && ply[b] < rp_hdr_[(size_t)game[b]].num_moves, AGZ_BAD_ARGUMENT,
                "sample %d: ply outside the game (searches_pi has one entry per move played)", b);
    off[b] = rp_off_[(size_t)game[b]];
  }
  const size_t per = (size_t)17 * V_.P;
  s_i64a_.ensure((size_t)B);
  s_i32a_.ensure((size_t)B);
  s_boards_.ensure((size_t)B * 8 * V_.PP);
  float *df = feats, *dp = pi, *dz = z;
  if (!out_is_device) {
    s_f32a_.ensure(per * (size_t)B);
    s_f32b_.ensure((size_t)B * (V_.A + 1));
    df = s_f32a_.p;
    dp = pi ? s_f32b_.p : nullptr;
    dz = z ? s_f32b_.p + (size_t)B * V_.A : nullptr;
  }
  AGZ_HIP(hipMemcpyAsync(s_i64a_.p, off.data(), sizeof(int64_t) * (size_t)B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, ply, sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_replay_arena_batch, dim3(B), dim3(kWave), 0, stream_, V_, (const uint8_t*)rp_buf_.p,
                     (const int64_t*)s_i64a_.p, (const int32_t*)s_i32a_.p, s_boards_.p, df, dp, dz);
  AGZ_HIP(hipGetLastError());
  if (!out_is_device) {
    AGZ_HIP(hipMemcpyAsync(feats, df, sizeof(float) * per * (size_t)B, hipMemcpyDeviceToHost, stream_));
    if (pi) AGZ_HIP(hipMemcpyAsync(pi, dp, sizeof(float) * (size_t)B * V_.A, hipMemcpyDeviceToHost, stream_));
    if (z) AGZ_HIP(hipMemcpyAsync(z, dz, sizeof(float) * (size_t)B, hipMemcpyDeviceToHost, stream_));
  }
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::records_clear() {
  rec_sent_ = 0;
  AGZ_HIP(hipMemsetAsync(V_.counters + CT_RECORDED, 0, sizeof(unsigned long long), stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::record_features(int64_t k, float* out) {
  agz_game_header h;
  record_header(k, &h);
  if (h.num_moves <= 0) return;
  const size_t per = (size_t)17 * V_.P;
  s_f32a_.ensure(per * (size_t)h.num_moves);
  s_boards_.ensure((size_t)8 * V_.PP);
  s_i32a_.ensure(3);
  s_i64a_.ensure(1);
  const int32_t args[3] = {0, h.num_moves, 0};
  const int64_t zero = 0;
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, args, sizeof(args), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i64a_.p, &zero, sizeof(zero), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_replay_features, dim3(1), dim3(kWave), 0, stream_, V_,
                     (const int16_t*)(V_.fin_moves + k * V_.max_game_length), s_i32a_.p, s_i32a_.p + 1, s_i32a_.p + 2,
                     s_i64a_.p, s_boards_.p, s_f32a_.p);
  AGZ_HIP(hipMemcpyAsync(out, s_f32a_.p, sizeof(float) * per * (size_t)h.num_moves, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

// get_replay_batch (train.jl:4-12), feature side: B sampled (game, ply) pairs -> [B][17*P].
// moves: the games' action lists back to back; off[b] = where sample b's game starts; ply[b] = which
// position of that game (0 = empty board).
void Engine::replay_batch_features(const int16_t* moves, int64_t nmoves, const int32_t* off, const int32_t* ply,
                                   int B, float* out, bool out_is_device) {
  AGZ_REQUIRE(B >= 0 && nmoves >= 0, AGZ_BAD_ARGUMENT, "negative count");
  if (B == 0) return;
  std::vector<int32_t> h32((size_t)3 * B);
  std::vector<int64_t> h64(B);
  const size_t per = (size_t)17 * V_.P;
  for (int b = 0; b < B; ++b) {
    AGZ_REQUIRE(off[b] >= 0 && ply[b] >= 0 && (int64_t)off[b] + ply[b] <= nmoves, AGZ_BAD_ARGUMENT,
                "sample points outside the move list");
    AGZ_REQUIRE(ply[b] <= V_.max_game_length, AGZ_BAD_ARGUMENT, "ply beyond max_game_length");
    h32[b] = off[b];
    h32[B + b] = ply[b] + 1;
    h32[2 * B + b] = ply[b];
    h64[b] = (int64_t)b * (int64_t)per;
  }
  for (int64_t i = 0; i < nmoves; ++i)
    AGZ_REQUIRE(moves[i] >= 0 && moves[i] <= V_.P, AGZ_BAD_ARGUMENT, "move outside 0..N*N");
  s_i16a_.ensure((size_t)std::max<int64_t>(nmoves, 1));
  s_i32a_.ensure((size_t)3 * B);
  s_i64a_.ensure(B);
  s_boards_.ensure((size_t)B * 8 * V_.PP);
  float* dst = out;
  if (!out_is_device) {
    s_f32a_.ensure(per * (size_t)B);
    dst = s_f32a_.p;
  }
  if (nmoves) AGZ_HIP(hipMemcpyAsync(s_i16a_.p, moves, sizeof(int16_t) * nmoves, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, h32.data(), sizeof(int32_t) * h32.size(), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i64a_.p, h64.data(), sizeof(int64_t) * h64.size(), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_replay_features, dim3(B), dim3(kWave), 0, stream_, V_, (const int16_t*)s_i16a_.p, s_i32a_.p,
                     s_i32a_.p + B, s_i32a_.p + 2 * B, s_i64a_.p, s_boards_.p, dst);
  AGZ_HIP(hipGetLastError());
  if (!out_is_device)
    AGZ_HIP(hipMemcpyAsync(out, dst, sizeof(float) * per * (size_t)B, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));     // the staging vectors above are stack-owned
}

// ---- network entry points

void Engine::features(const int8_t* boards, const int8_t* deltas, const int32_t* ndeltas, const int8_t* to_play,
                      int B, float* out) {
  AGZ_REQUIRE(B >= 0, AGZ_BAD_ARGUMENT, "B < 0");
  if (B == 0) return;
  const size_t P = V_.P;
  s_boards_.ensure((size_t)B * P);
  s_deltas_.ensure((size_t)B * 7 * P);
  s_i32a_.ensure(B);
  s_tp_.ensure(B);
  s_f32a_.ensure((size_t)B * 17 * P);
  AGZ_HIP(hipMemcpyAsync(s_boards_.p, boards, (size_t)B * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_deltas_.p, deltas, (size_t)B * 7 * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, ndeltas, sizeof(int32_t) * B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_tp_.p, to_play, (size_t)B, hipMemcpyHostToDevice, stream_));
  launch_features_from_deltas(s_boards_.p, s_deltas_.p, s_i32a_.p, s_tp_.p, B, V_.N, nullptr, s_f32a_.p, stream_);
  AGZ_HIP(hipMemcpyAsync(out, s_f32a_.p, sizeof(float) * (size_t)B * 17 * P, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::net_forward_positions(const int8_t* boards, const int8_t* deltas, const int32_t* ndeltas,
                                   const int8_t* to_play, int B, float* pi_out, float* v_out) {
  AGZ_REQUIRE(B >= 0, AGZ_BAD_ARGUMENT, "B < 0");
  if (B == 0) return;
  const size_t P = V_.P;
  s_boards_.ensure((size_t)B * P);
  s_deltas_.ensure((size_t)B * 7 * P);
  s_i32a_.ensure(B);
  s_tp_.ensure(B);
  d_count_.ensure(1);
  s_f32a_.ensure((size_t)B * P * 32);
  s_f32b_.ensure((size_t)B * (V_.A + 1));
  AGZ_HIP(hipMemcpyAsync(s_boards_.p, boards, (size_t)B * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_deltas_.p, deltas, (size_t)B * 7 * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, ndeltas, sizeof(int32_t) * B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_tp_.p, to_play, (size_t)B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(d_count_.p, &B, sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  launch_features_from_deltas(s_boards_.p, s_deltas_.p, s_i32a_.p, s_tp_.p, B, V_.N, s_f32a_.p, nullptr, stream_);
  net().forward(s_f32a_.p, d_count_.p, B, s_f32b_.p, s_f32b_.p + (size_t)B * V_.A);
  AGZ_HIP(hipMemcpyAsync(pi_out, s_f32b_.p, sizeof(float) * (size_t)B * V_.A, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(v_out, s_f32b_.p + (size_t)B * V_.A, sizeof(float) * B, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  net().check_async_error();      // a persistent tower launch that gave up: these outputs are garbage, say so now
}

void Engine::net_forward_features(const float* feats, int B, float* pi_out, float* v_out) {
  AGZ_REQUIRE(B >= 0, AGZ_BAD_ARGUMENT, "B < 0");
  if (B == 0) return;
  const size_t P = V_.P;
  d_whcn_.ensure((size_t)B * 17 * P);
  d_count_.ensure(1);
  s_f32a_.ensure((size_t)B * P * 32);
  s_f32b_.ensure((size_t)B * (V_.A + 1));
  AGZ_HIP(hipMemcpyAsync(d_whcn_.p, feats, sizeof(float) * (size_t)B * 17 * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(d_count_.p, &B, sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  launch_whcn_to_x32(d_whcn_.p, B, V_.N, s_f32a_.p, stream_);
  net().forward(s_f32a_.p, d_count_.p, B, s_f32b_.p, s_f32b_.p + (size_t)B * V_.A);
  AGZ_HIP(hipMemcpyAsync(pi_out, s_f32b_.p, sizeof(float) * (size_t)B * V_.A, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(v_out, s_f32b_.p + (size_t)B * V_.A, sizeof(float) * B, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  net().check_async_error();      // a persistent tower launch that gave up: these outputs are garbage, say so now
}

// random stone features (not zeros: DVFS makes zero-filled operands look faster than real data)
void Engine::fill_synthetic_inputs(int B) {
  const size_t P = V_.P;
  std::vector<int8_t> boards((size_t)B * P), deltas((size_t)B * 7 * P, 0), tp(B);
  std::vector<int32_t> nd(B, 0);
  uint64_t s = 0x1234567ull;
  for (auto& x : boards) {
    s = agz_mix64(s + 0x9E3779B97F4A7C15ull);
    const int r = (int)(s % 3);
    x = (int8_t)(r == 2 ? -1 : r);
  }
  for (int b = 0; b < B; ++b) tp[b] = (b & 1) ? -1 : 1;
  s_boards_.ensure(boards.size());
  s_deltas_.ensure(deltas.size());
  s_i32a_.ensure(B);
  s_tp_.ensure(B);
  d_count_.ensure(1);
  s_f32a_.ensure((size_t)B * P * 32);
  s_f32b_.ensure((size_t)B * (V_.A + 1));
  AGZ_HIP(hipMemcpyAsync(s_boards_.p, boards.data(), boards.size(), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_deltas_.p, deltas.data(), deltas.size(), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, nd.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_tp_.p, tp.data(), (size_t)B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(d_count_.p, &B, sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  launch_features_from_deltas(s_boards_.p, s_deltas_.p, s_i32a_.p, s_tp_.p, B, V_.N, s_f32a_.p, nullptr, stream_);
  AGZ_HIP(hipStreamSynchronize(stream_));
}

float Engine::time_forward(int B, int iters) {
  AGZ_REQUIRE(B > 0 && iters > 0, AGZ_BAD_ARGUMENT, "B and iters must be positive");
  fill_synthetic_inputs(B);
  net().forward(s_f32a_.p, d_count_.p, B, s_f32b_.p, s_f32b_.p + (size_t)B * V_.A);   // warm-up
  hipEvent_t e0, e1;
  AGZ_HIP(hipEventCreate(&e0));
  AGZ_HIP(hipEventCreate(&e1));
  AGZ_HIP(hipEventRecord(e0, stream_));
  for (int i = 0; i < iters; ++i) net().forward(s_f32a_.p, d_count_.p, B, s_f32b_.p, s_f32b_.p + (size_t)B * V_.A);
  AGZ_HIP(hipEventRecord(e1, stream_));
  AGZ_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  AGZ_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ms / (float)iters;
}

float Engine::time_conv(int B, int iters) {
  AGZ_REQUIRE(B > 0 && iters > 0, AGZ_BAD_ARGUMENT, "B and iters must be positive");
  fill_synthetic_inputs(B);
  // one forward leaves real (post-ReLU, mixed-sign-weight) activations in the tower buffers
  net().forward(s_f32a_.p, d_count_.p, B, s_f32b_.p, s_f32b_.p + (size_t)B * V_.A);
  net_->launch_tower_conv_once(d_count_.p, B);
  hipEvent_t e0, e1;
  AGZ_HIP(hipEventCreate(&e0));
  AGZ_HIP(hipEventCreate(&e1));
  AGZ_HIP(hipEventRecord(e0, stream_));
  for (int i = 0; i < iters; ++i) net_->launch_tower_conv_once(d_count_.p, B);
  AGZ_HIP(hipEventRecord(e1, stream_));
  AGZ_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  AGZ_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ms / (float)iters;
}

void Engine::debug_draws(uint64_t seed, uint64_t game, uint32_t move, int n, double alpha, double* out) {
  if (n <= 0) return;
  s_f64_.ensure(n);
  hipLaunchKernelGGL(k_debug_draws, dim3(ceil_div(n, 64)), dim3(64), 0, stream_, seed, game, move, n, alpha, s_f64_.p);
  AGZ_HIP(hipMemcpyAsync(out, s_f64_.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::debug_math(int op, const double* x, const double* y, int n, double* out) {
  if (n <= 0) return;
  s_f64_.ensure((size_t)3 * n);
  AGZ_HIP(hipMemcpyAsync(s_f64_.p, x, sizeof(double) * n, hipMemcpyHostToDevice, stream_));
  if (y) AGZ_HIP(hipMemcpyAsync(s_f64_.p + n, y, sizeof(double) * n, hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_debug_math, dim3(ceil_div(n, 64)), dim3(64), 0, stream_, V_, op, (const double*)s_f64_.p,
                     (const double*)(s_f64_.p + n), n, s_f64_.p + 2 * (size_t)n);
  AGZ_HIP(hipMemcpyAsync(out, s_f64_.p + 2 * (size_t)n, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

// ---- Go rules

void Engine::go_play(const int8_t* boards, const int8_t* to_play, const int32_t* ko, const int32_t* moves, int B,
                     int8_t* boards_out, int32_t* ko_out, int32_t* ncap_out, int32_t* status_out) {
  if (B <= 0) return;
  const size_t P = V_.P;
  s_boards_.ensure((size_t)B * P);
  s_boards_out_.ensure((size_t)B * P);
  s_tp_.ensure(B);
  s_i32a_.ensure(B); s_i32b_.ensure(B); s_i32c_.ensure((size_t)3 * B);
  AGZ_HIP(hipMemcpyAsync(s_boards_.p, boards, (size_t)B * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_tp_.p, to_play, (size_t)B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, ko, sizeof(int32_t) * B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32b_.p, moves, sizeof(int32_t) * B, hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_go_play, dim3(B), dim3(kWave), 0, stream_, V_, (const int8_t*)s_boards_.p,
                     (const int8_t*)s_tp_.p, (const int32_t*)s_i32a_.p, (const int32_t*)s_i32b_.p, B,
                     s_boards_out_.p, s_i32c_.p, s_i32c_.p + B, s_i32c_.p + 2 * (size_t)B);
  AGZ_HIP(hipMemcpyAsync(boards_out, s_boards_out_.p, (size_t)B * P, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(ko_out, s_i32c_.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(ncap_out, s_i32c_.p + B, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(status_out, s_i32c_.p + 2 * (size_t)B, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::go_legal(const int8_t* boards, const int8_t* to_play, const int32_t* ko, int B, int8_t* out) {
  if (B <= 0) return;
  const size_t P = V_.P;
  s_boards_.ensure((size_t)B * P);
  s_tp_.ensure(B);
  s_i32a_.ensure(B);
  s_legal_.ensure((size_t)B * V_.A);
  AGZ_HIP(hipMemcpyAsync(s_boards_.p, boards, (size_t)B * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_tp_.p, to_play, (size_t)B, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_i32a_.p, ko, sizeof(int32_t) * B, hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_go_legal, dim3(B), dim3(kWave), 0, stream_, V_, (const int8_t*)s_boards_.p,
                     (const int8_t*)s_tp_.p, (const int32_t*)s_i32a_.p, B, s_legal_.p);
  AGZ_HIP(hipMemcpyAsync(out, s_legal_.p, (size_t)B * V_.A, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

void Engine::go_score(const int8_t* boards, const float* komi, int B, float* out) {
  if (B <= 0) return;
  const size_t P = V_.P;
  s_boards_.ensure((size_t)B * P);
  s_f32a_.ensure((size_t)2 * B);
  AGZ_HIP(hipMemcpyAsync(s_boards_.p, boards, (size_t)B * P, hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipMemcpyAsync(s_f32a_.p, komi, sizeof(float) * B, hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_go_score, dim3(B), dim3(kWave), 0, stream_, V_, (const int8_t*)s_boards_.p,
                     (const float*)s_f32a_.p, B, s_f32a_.p + B);
  AGZ_HIP(hipMemcpyAsync(out, s_f32a_.p + B, sizeof(float) * B, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

// ---- single-tree compat

void Engine::check_game(int g) const {
  AGZ_REQUIRE(g >= 0 && g < V_.games, AGZ_BAD_ARGUMENT, "game slot %d of %d", g, V_.games);
}
void Engine::check_node(int g, int node) const {
  check_game(g);
  AGZ_REQUIRE(node >= 0 && node < V_.cap, AGZ_BAD_ARGUMENT, "node %d of %d", node, V_.cap);
}

int Engine::tree_op(TreeArgs& T, int32_t* r0) {
  check_game(T.g);
  // stage host-side inputs
  if (T.probs) {
    s_f32a_.ensure(V_.A);
    AGZ_HIP(hipMemcpyAsync(s_f32a_.p, T.probs, sizeof(float) * V_.A, hipMemcpyHostToDevice, stream_));
    T.probs = s_f32a_.p;
  }
  if (T.board) {
    s_boards_.ensure(V_.P);
    AGZ_HIP(hipMemcpyAsync(s_boards_.p, T.board, (size_t)V_.P, hipMemcpyHostToDevice, stream_));
    T.board = s_boards_.p;
  }
  if (T.history && T.info.history_len > 0) {
    s_deltas_.ensure((size_t)7 * V_.P);
    AGZ_HIP(hipMemcpyAsync(s_deltas_.p, T.history, (size_t)std::min(T.info.history_len, 7) * V_.P,
                           hipMemcpyHostToDevice, stream_));
    T.history = s_deltas_.p;
  }
  double* host_dout = T.dout;
  if (T.dout) {
    s_f64_.ensure(V_.A);
    T.dout = s_f64_.p;
  }
  T.iout = s_iout_.p;
  hipLaunchKernelGGL(k_tree_op, dim3(1), dim3(kWave), 0, stream_, V_, T);
  int32_t iout[4] = {0, 0, 0, 0};
  AGZ_HIP(hipMemcpyAsync(iout, s_iout_.p, sizeof(iout), hipMemcpyDeviceToHost, stream_));
  if (host_dout) AGZ_HIP(hipMemcpyAsync(host_dout, s_f64_.p, sizeof(double) * V_.A, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  if (r0) *r0 = iout[1];
  return iout[0];
}

// tree_search!(player, parallel_readouts) on slot g, split at the network call so that any
// callable can stand in for the network (MCTSPlayer.network is duck-typed, mcts_play.jl:5,89)
int Engine::tree_search_select(int g, int par, int* nleaves) {
  check_game(g);
  AGZ_REQUIRE(par >= 1 && par <= V_.par, AGZ_BAD_ARGUMENT,
              "parallel_readouts %d exceeds the engine's configured %d", par, V_.par);
  TreeArgs T;
  std::memset(&T, 0, sizeof(T));
  T.op = TOP_SEARCH_SELECT; T.g = g; T.par = par;
  int32_t n = 0;
  const int st = tree_op(T, &n);
  *nleaves = n;
  tree_batch_ = n;
  return st;
}

void Engine::tree_leaf_features(int g, float* feats_out) {
  check_game(g);
  if (tree_batch_ <= 0) return;
  d_whcn_.ensure((size_t)bcap_ * 17 * V_.P);
  hipLaunchKernelGGL(k_leaf_features, dim3(V_.par), dim3(kWavesPerLeaf * kWave), 0, stream_, V_, g, V_.par, (float*)nullptr, d_whcn_.p);
  AGZ_HIP(hipMemcpyAsync(feats_out, d_whcn_.p, sizeof(float) * (size_t)tree_batch_ * 17 * V_.P,
                         hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

// agz_tree_leaf_positions: the leaves of the last tree_search_select as the reference's GoPosition fields
// (board.jl:271-306): board, board_deltas newest first (delta_k = B_k - B_{k+1}: +colour at the played point and where
// an opponent stone vanished, board.jl:479-481,505-506), to_play, n, ko, caps, the last two moves
void Engine::tree_leaf_positions(int g, int32_t* nodes, int8_t* boards, int8_t* deltas, int32_t* ndeltas,
                                 int8_t* to_play, agz_position_info* info) {
  check_game(g);
  const int n = tree_batch_;
  if (n <= 0) return;
  const int P = V_.P;
  s_leafb_.ensure((size_t)V_.par * 8 * P);
  s_leafrows_.ensure((size_t)V_.par * sizeof(LeafPosRow));
  hipLaunchKernelGGL(k_leaf_positions, dim3(n), dim3(kWave), 0, stream_, V_, g, s_leafb_.p, (LeafPosRow*)s_leafrows_.p);
  std::vector<int8_t> b8((size_t)n * 8 * P);
  std::vector<LeafPosRow> rows(n);
  GameState gs;
  AGZ_HIP(hipMemcpyAsync(b8.data(), s_leafb_.p, b8.size(), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(rows.data(), s_leafrows_.p, sizeof(LeafPosRow) * n, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipMemcpyAsync(&gs, V_.gs + g, sizeof(GameState), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  for (int k = 0; k < n; ++k) {
    const LeafPosRow& r = rows[k];
    const int avail = std::min(7, (r.plen - 1) + gs.hist_len);   // older boards that exist (record_leaf)
    const int8_t* b = b8.data() + (size_t)k * 8 * P;
    if (nodes) nodes[k] = r.node;
    if (boards) std::memcpy(boards + (size_t)k * P, b, P);
    if (deltas) {
      int8_t* d = deltas + (size_t)k * 7 * P;
      std::memset(d, 0, (size_t)7 * P);
      for (int s = 0; s < avail; ++s)
        for (int p = 0; p < P; ++p) d[(size_t)s * P + p] = (int8_t)(b[(size_t)s * P + p] - b[(size_t)(s + 1) * P + p]);
    }
    if (ndeltas) ndeltas[k] = avail;
    if (to_play) to_play[k] = r.m.to_play;
    if (info) {
      agz_position_info& o = info[k];
      o.n = r.m.n; o.to_play = r.m.to_play; o.ko = r.m.ko; o.caps_black = r.m.caps_b; o.caps_white = r.m.caps_w;
      o.last_move = r.m.last_move; o.prev_move = r.prev_move; o.history_len = avail; o.komi = gs.komi;
    }
  }
}

int Engine::tree_search_incorporate(int g, const float* pi, const float* v) {
  check_game(g);
  const int n = tree_batch_;
  if (n > 0) {
    if (pi && v) {
      AGZ_HIP(hipMemcpyAsync(d_pi_.p, pi, sizeof(float) * (size_t)n * V_.A, hipMemcpyHostToDevice, stream_));
      AGZ_HIP(hipMemcpyAsync(d_v_.p, v, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, stream_));
    } else {
      d_count_.ensure(1);
      AGZ_HIP(hipMemcpyAsync(d_count_.p, &n, sizeof(int32_t), hipMemcpyHostToDevice, stream_));
      hipLaunchKernelGGL(k_leaf_features, dim3(V_.par), dim3(kWavesPerLeaf * kWave), 0, stream_, V_, g, V_.par, d_x32_.p, (float*)nullptr);
      net_->forward(d_x32_.p, d_count_.p, std::min(bcap_, V_.par), d_pi_.p, d_v_.p);
    }
  }
  TreeArgs T;
  std::memset(&T, 0, sizeof(T));
  T.op = TOP_SEARCH_POST; T.g = g;
  tree_batch_ = 0;
  return tree_op(T, nullptr);
}

void Engine::game_state(int g, GameState* out) {
  check_game(g);
  AGZ_HIP(hipMemcpyAsync(out, V_.gs + g, sizeof(GameState), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
void Engine::game_patch(int g, const GameState& s) {
  check_game(g);
  AGZ_HIP(hipMemcpyAsync(V_.gs + g, &s, sizeof(GameState), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
void Engine::node_meta(int g, int node, NodeMeta* out) {
  check_node(g, node);
  AGZ_HIP(hipMemcpyAsync(out, V_.meta + node_index(V_, g, node), sizeof(NodeMeta), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
void Engine::node_meta_set(int g, int node, const NodeMeta& m) {
  check_node(g, node);
  AGZ_HIP(hipMemcpyAsync(V_.meta + node_index(V_, g, node), &m, sizeof(NodeMeta), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
static float* row_base(const View& V, int field) {
  return field == AGZ_F_CHILD_N ? V.childN : field == AGZ_F_CHILD_W ? V.childW : V.childP;
}
void Engine::node_row_get(int g, int node, int field, float* out) {
  check_node(g, node);
  AGZ_REQUIRE(field >= 0 && field <= 2, AGZ_BAD_ARGUMENT, "field %d", field);
  AGZ_HIP(hipMemcpyAsync(out, row_base(V_, field) + node_index(V_, g, node) * V_.AP, sizeof(float) * V_.A,
                         hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
void Engine::node_row_set(int g, int node, int field, const float* in) {
  check_node(g, node);
  AGZ_REQUIRE(field >= 0 && field <= 2, AGZ_BAD_ARGUMENT, "field %d", field);
  AGZ_HIP(hipMemcpyAsync(row_base(V_, field) + node_index(V_, g, node) * V_.AP, in, sizeof(float) * V_.A,
                         hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
void Engine::node_children(int g, int node, int32_t* out) {
  check_node(g, node);
  AGZ_HIP(hipMemcpyAsync(out, V_.child + node_index(V_, g, node) * V_.AP, sizeof(int32_t) * V_.A,
                         hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
void Engine::node_board(int g, int node, int8_t* out) {
  check_node(g, node);
  AGZ_HIP(hipMemcpyAsync(out, V_.board + node_index(V_, g, node) * V_.PP, (size_t)V_.P, hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}
float Engine::node_stat(int g, int node, int which) {
  NodeMeta m;
  node_meta(g, node, &m);
  float v = 0.f;
  const float* src;
  if (m.parent < 0) {
    src = which == 0 ? &V_.gs[g].rootN : &V_.gs[g].rootW;
  } else {
    src = (which == 0 ? V_.childN : V_.childW) + node_index(V_, g, m.parent) * V_.AP + m.fmove;
  }
  AGZ_HIP(hipMemcpyAsync(&v, src, sizeof(float), hipMemcpyDeviceToHost, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
  return v;
}
void Engine::node_set_N(int g, int node, float v) {
  NodeMeta m;
  node_meta(g, node, &m);
  float* dst = m.parent < 0 ? &V_.gs[g].rootN : V_.childN + node_index(V_, g, m.parent) * V_.AP + m.fmove;
  AGZ_HIP(hipMemcpyAsync(dst, &v, sizeof(float), hipMemcpyHostToDevice, stream_));
  AGZ_HIP(hipStreamSynchronize(stream_));
}

}  // namespace agz
