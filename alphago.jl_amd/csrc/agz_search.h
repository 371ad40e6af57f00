// agz_search.h -- Go rules, PUCT search, virtual loss, backup, noise and the per-move phase of
// the self-play hot path, written ONCE as wave-level templates.
//
// One 64-lane wavefront owns one game tree (DESIGN.md "Search kernels").  The template
// parameter W supplies the wave primitives:
//   * agz::HipWave  (agz_engine.hip)   -- the product: lanes = threadIdx, LDS scratch, DPP/shuffle
//                                         reductions, LDS atomics.  This is what runs on gfx950.
//   * a host wave simulator (tests/hostsim) -- TEST INFRASTRUCTURE that executes the same source
//     lane-serially on the CPU so that tree/rules logic can be diffed against the oracle in
//     this GPU-less container.  It is never linked into libagz.so.
//
// Discipline that makes the two agree: data-parallel work goes through w.for_each (pure per-index
// bodies), every cross-lane flow through memory is separated by w.sync(), per-lane partials are
// combined with w.reduce_*, and all other control flow is wave-uniform.
//
// Reference semantics restated here (file:line under /root/reference):
//   select_leaf mcts.jl:108-138 | maybe_add_child! :140-147 | virtual loss :149-171 |
//   revert_visits! :173-186 | incorporate_results! :188-213 | backup_value! :215-225 |
//   inject_noise! :233-239 | children_as_pi :241-252 | PUCT :84-92
//   tree_search! mcts_play.jl:73-98 | pick_move :52-71 | play_move! :26-50 | should_resign :124
//   selfplay loop selfplay.jl:1-45
//   play_move! board.jl:451-509 | pass_move! :426-440 | all_legal_moves :393-424 |
//   is_move_suicidal :354-374 | is_koish :47-56 | score :511-533
// Float types follow the reference exactly: Float32 statistics, Float64 c_puct
// and action score, Float32 values; this header must be compiled with FP contraction off.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/agz_draws.h"
#include "agz_state.h"

#if defined(__HIPCC__)
#define AGZ_FN __host__ __device__ __forceinline__
#else
#define AGZ_FN inline
#endif

// k_pre phase clocks (timing builds only; see CT_T_* in agz_state.h)
#ifdef AGZ_TIMING_EXPERIMENTS
#define AGZ_STAMP_BEGIN(w) unsigned long long agz_t_prev = (w).clock()
#define AGZ_STAMP_BEGIN_AGAIN(w) agz_t_prev = (w).clock()
#define AGZ_STAMP(w, V, slot)                                   \
  do {                                                          \
    const unsigned long long agz_t_now = (w).clock();           \
    (w).count(&(V).counters[slot], agz_t_now - agz_t_prev);      \
    agz_t_prev = agz_t_now;                                     \
  } while (0)
#else
#define AGZ_STAMP_BEGIN(w) unsigned long long agz_t_prev = 0
#define AGZ_STAMP_BEGIN_AGAIN(w) (void)agz_t_prev
#define AGZ_STAMP(w, V, slot) (void)agz_t_prev
#endif


#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace agz {

struct Scratch {
  int8_t* sb;        // [PP] working board
  int32_t* label;    // [PP]
  int32_t* minlib;   // [PP]
  int32_t* maxlib;   // [PP]
  int8_t* flag;      // [AP]
  double* dbuf;      // [AP]
  int32_t* path;     // [maxd]
};

constexpr int kIntMax = 0x7fffffff;

// ------------------------------------------------------------------ small helpers

template <class W>
AGZ_FN int nbrs(int N, int p, int out[4]) {
  const int i = p % N, j = p / N;
  int k = 0;
  if (i + 1 < N) out[k++] = p + 1;
  if (i - 1 >= 0) out[k++] = p - 1;
  if (j + 1 < N) out[k++] = p + N;
  if (j - 1 >= 0) out[k++] = p - N;
  return k;
}

AGZ_FN long node_index(const View& V, int g, int node) { return (long)g * V.cap + node; }

AGZ_FN bool legal_bit(const View& V, long ni, int a) {
  return (V.legal[ni * V.LW + (a >> 5)] >> (a & 31)) & 1u;
}

// N(x) / W(x): a node's own statistics live in its parent's child rows (mcts.jl:96-102); the
// root's live in the game record (the DummyNode of mcts.jl:27-39).
AGZ_FN float* slotN(const View& V, int g, int node) {
  const NodeMeta& m = V.meta[node_index(V, g, node)];
  return m.parent < 0 ? &V.gs[g].rootN : &V.childN[node_index(V, g, m.parent) * V.AP + m.fmove];
}
AGZ_FN float* slotW(const View& V, int g, int node) {
  const NodeMeta& m = V.meta[node_index(V, g, node)];
  return m.parent < 0 ? &V.gs[g].rootW : &V.childW[node_index(V, g, m.parent) * V.AP + m.fmove];
}

AGZ_FN bool node_is_done(const View& V, int g, int node) {
  const NodeMeta& m = V.meta[node_index(V, g, node)];
  return (m.flags & NF_DONE) || m.n >= V.max_game_length;   // mcts.jl:230-231
}

// ------------------------------------------------------------------ rules on a board in scratch

// Connected components by min-label propagation with pointer jumping.  stones=true labels
// stone groups, stones=false labels empty regions.  The fixed point (label = smallest point
// of the component) does not depend on the order in which lanes update.
template <class W>
AGZ_FN void label_components(W& w, const View& V, Scratch& S, bool stones) {
  const int P = V.P, N = V.N;
  w.for_each(P, [&](int p) {
    const bool in = stones ? (S.sb[p] != 0) : (S.sb[p] == 0);
    S.label[p] = in ? p : -1;
    S.minlib[p] = kIntMax;
    S.maxlib[p] = -1;
  });
  w.sync();
  for (int iter = 0; iter < 4 * P; ++iter) {
    bool changed = false;
    w.for_each(P, [&](int p) {
      const int l = S.label[p];
      if (l < 0) return;
      const int c = S.sb[p];
      int m = l, nb[4];
      const int k = nbrs<W>(N, p, nb);
      for (int t = 0; t < k; ++t)
        if (S.sb[nb[t]] == c) { const int lq = S.label[nb[t]]; m = lq < m ? lq : m; }
      const int lm = S.label[m];
      m = lm < m ? lm : m;
      if (m < l) { S.label[p] = m; changed = true; }
    });
    changed = w.any(changed);
    w.sync();
    if (!changed) break;
  }
}

// distinct-liberty summary per stone group: minlib/maxlib over the group's empty neighbours.
//   none  <=> maxlib < 0 ; exactly one <=> minlib == maxlib ; two or more <=> minlib < maxlib
template <class W>
AGZ_FN void group_liberties(W& w, const View& V, Scratch& S) {
  const int P = V.P, N = V.N;
  w.for_each(P, [&](int e) {
    if (S.sb[e] != 0) return;
    int nb[4];
    const int k = nbrs<W>(N, e, nb);
    for (int t = 0; t < k; ++t)
      if (S.sb[nb[t]] != 0) {
        const int gl = S.label[nb[t]];
        w.amin(&S.minlib[gl], e);
        w.amax(&S.maxlib[gl], e);
      }
  });
  w.sync();
}

// all_legal_moves (board.jl:393-424) for the side `tp` into S.flag[0..AP).
// An empty, non-ko point is legal iff it has an empty neighbour, or joins a friendly group that
// keeps another liberty, or captures an enemy group in atari (is_move_suicidal :354-374).
// Needs label_components(stones) + group_liberties on the board in S.sb.
template <class W>
AGZ_FN void compute_legal_flags(W& w, const View& V, Scratch& S, int tp, int ko) {
  const int P = V.P, N = V.N, A = V.A;
  w.for_each(V.AP, [&](int a) {
    bool ok = false;
    if (a == P) ok = true;
    else if (a < P && S.sb[a] == 0 && a != ko) {
      int nb[4];
      const int k = nbrs<W>(N, a, nb);
      for (int t = 0; t < k; ++t) {
        const int c = S.sb[nb[t]];
        if (c == 0) { ok = true; continue; }
        const int gl = S.label[nb[t]];
        if (c == tp) { if (S.minlib[gl] < S.maxlib[gl]) ok = true; }
        else if (S.minlib[gl] == S.maxlib[gl]) ok = true;
      }
    }
    S.flag[a] = ok && a < A;
  });
  w.sync();
}

template <class W>
AGZ_FN void write_legal_mask(W& w, const View& V, Scratch& S, long ni, int tp, int ko) {
  const int A = V.A;
  compute_legal_flags(w, V, S, tp, ko);
  w.for_each(V.LW, [&](int wi) {
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) {
      const int a = wi * 32 + b;
      if (a < A && S.flag[a]) bits |= 1u << b;
    }
    V.legal[ni * V.LW + wi] = bits;
  });
  w.sync();
}

// Place a stone of `color` at point a on the board in S.sb (labels + liberties of the PRE-move
// board must be current): removes the enemy groups whose only liberty was a (add_stone!
// board.jl:251-259) and reports the capture count and the ko point (board.jl:472,483-487).
template <class W>
AGZ_FN void apply_move_in_scratch(W& w, const View& V, Scratch& S, int a, int color, int* ncap_out, int* ko_out) {
  const int P = V.P, N = V.N;
  int nb[4];
  const int k = nbrs<W>(N, a, nb);
  // is_koish on the pre-move board (board.jl:47-56)
  int koish = S.sb[nb[0]];
  for (int t = 1; t < k; ++t)
    if (S.sb[nb[t]] != koish) koish = 0;
  int dead[4];
  for (int t = 0; t < 4; ++t) dead[t] = -1;
  for (int t = 0; t < k; ++t)
    if (S.sb[nb[t]] == -color) {
      const int gl = S.label[nb[t]];
      if (S.minlib[gl] == a && S.maxlib[gl] == a) dead[t] = gl;
    }
  w.sync();
  int ncap = 0, lastcap = -1;
  w.for_each(P, [&](int p) {
    if (S.sb[p] != -color) return;
    const int gl = S.label[p];
    if (gl == dead[0] || gl == dead[1] || gl == dead[2] || gl == dead[3]) {
      S.sb[p] = 0;
      ncap += 1;
      lastcap = p > lastcap ? p : lastcap;
    }
  });
  ncap = w.reduce_sum(ncap);
  lastcap = w.reduce_max(lastcap);
  if (w.leader()) S.sb[a] = (int8_t)color;
  w.sync();
  *ncap_out = ncap;
  *ko_out = (ncap == 1 && koish == -color) ? lastcap : -1;
}

// Tromp-Taylor area score minus komi (board.jl:511-533), Black-positive.
template <class W>
AGZ_FN float area_score(W& w, const View& V, Scratch& S, float komi) {
  const int P = V.P, N = V.N;
  label_components(w, V, S, false);
  // minlib doubles as the "touches" bitmask per empty region (1 = black border, 2 = white)
  w.for_each(P, [&](int p) { S.maxlib[p] = 0; });
  w.sync();
  w.for_each(P, [&](int e) {
    if (S.sb[e] != 0) return;
    int nb[4];
    const int k = nbrs<W>(N, e, nb);
    int bits = 0;
    for (int t = 0; t < k; ++t) {
      const int c = S.sb[nb[t]];
      if (c == 1) bits |= 1;
      if (c == -1) bits |= 2;
    }
    if (bits) w.aor(&S.maxlib[S.label[e]], bits);
  });
  w.sync();
  int diff = 0;
  w.for_each(P, [&](int p) {
    const int c = S.sb[p];
    if (c == 1) diff += 1;
    else if (c == -1) diff -= 1;
    else {
      const int t = S.maxlib[S.label[p]];
      if (t == 1) diff += 1;
      else if (t == 2) diff -= 1;
    }
  });
  diff = w.reduce_sum(diff);
  w.sync();
  return (float)diff - komi;
}

AGZ_FN int result_of(float score) { return score > 0.f ? 1 : score < 0.f ? -1 : 0; }

// ------------------------------------------------------------------ node pool

template <class W>
AGZ_FN int free_pending(W& w, const View& V, Scratch& S, int g, int budget);

template <class W>
AGZ_FN int pool_alloc(W& w, const View& V, Scratch& S, int g) {
  GameState& G = V.gs[g];
  int top = G.free_top;
  if (top <= 0 && G.garbage > 0) {      // nothing free but releases pending: do them now
    free_pending(w, V, S, g, V.cap);
    top = G.free_top;
  }
  if (top <= 0) {
    w.count(&V.counters[CT_POOL_EXHAUSTED], 1);
    if (w.leader()) G.err = AGZ_POOL_EXHAUSTED;
    w.sync();
    return -1;
  }
  const int id = V.freelist[(long)g * V.cap + top - 1];
  w.sync();
  if (w.leader()) { G.free_top = top - 1; G.nodes_used += 1; }
  w.sync();
  return id;
}

template <class W>
AGZ_FN void pool_free(W& w, const View& V, int g, int node) {
  GameState& G = V.gs[g];
  const int top = G.free_top;
  w.sync();
  if (w.leader()) {
    V.freelist[(long)g * V.cap + top] = node;
    G.free_top = top + 1;
    G.nodes_used -= 1;
  }
  w.sync();
}

// Deferred release of discarded subtrees.  play_move! drops the siblings of the chosen child
// (`root.parent.children = Dict()`, mcts_play.jl:48); walking them at once put a serial walk over
// thousands of nodes on the critical path of the ~2 % of games that move in a given step (k_pre took
// as long as its slowest wave: 2.9 ms).  Instead the detached old root is pushed on a per-game
// garbage stack -- it lives in the unused tail of the slot's free list and grows downward, while free
// entries grow upward; they cannot collide because every stacked node is an allocated node and
// #allocated + #free == cap -- and every k_pre releases at most `budget` nodes of it, children pushed
// in ascending action order (a deterministic order: node ids never depend on timing).
template <class W>
AGZ_FN int free_pending(W& w, const View& V, Scratch& S, int g, int budget) {
  GameState& G = V.gs[g];
  int32_t* fl = V.freelist + (long)g * V.cap;
  int sp = V.cap - G.garbage;
  int done = 0;
  w.sync();
  while (sp < V.cap && done < budget) {
    const int node = fl[sp];
    ++sp;
    const long ni = node_index(V, g, node);
    const bool expanded = V.meta[ni].flags & NF_EXPANDED;
    w.sync();
    if (expanded) sp = w.push_desc(V.A, [&](int a) { return V.child[ni * V.AP + a]; }, fl, sp);
    const int top = G.free_top;
    w.sync();
    if (w.leader()) { fl[top] = node; G.free_top = top + 1; G.nodes_used -= 1; V.meta[ni].flags = 0; }
    w.sync();
    ++done;
  }
  if (w.leader()) G.garbage = V.cap - sp;
  w.sync();
  return done;
}

// nodes released per step and game: a move creates <= R+7 nodes over ~R/8 steps, i.e. ~8 per step
// in steady state; 32 keeps up with 4x headroom and costs ~30 us of a wave per step
constexpr int kFreeBudget = 32;

// detach `keep` from `start` and hand `start` (with everything else below it) to the garbage stack
template <class W>
AGZ_FN void discard_except(W& w, const View& V, int g, int start, int keep_action) {
  GameState& G = V.gs[g];
  w.sync();
  if (w.leader()) {
    if (keep_action >= 0) V.child[node_index(V, g, start) * V.AP + keep_action] = -1;
    V.freelist[(long)g * V.cap + V.cap - G.garbage - 1] = start;
    G.garbage = G.garbage + 1;
  }
  w.sync();
}

// ------------------------------------------------------------------ node creation

template <class W>
AGZ_FN void load_board(W& w, const View& V, Scratch& S, long ni) {
  w.for_each(V.P, [&](int p) { S.sb[p] = V.board[ni * V.PP + p]; });
  w.sync();
}

// Initialise node `id` from the board currently in S.sb (rows zeroed, legal mask for `tp`).
template <class W>
AGZ_FN void node_init_from_scratch(W& w, const View& V, Scratch& S, int g, int id, const NodeMeta& m) {
  const long ni = node_index(V, g, id);
  w.for_each(V.AP, [&](int a) {
    V.childN[ni * V.AP + a] = 0.f;
    V.childW[ni * V.AP + a] = 0.f;
    V.childP[ni * V.AP + a] = 0.f;
    V.child[ni * V.AP + a] = -1;
  });
  w.for_each(V.P, [&](int p) { V.board[ni * V.PP + p] = S.sb[p]; });
  if (w.leader()) { V.meta[ni] = m; V.meta[ni].flags = (uint8_t)(m.flags | NF_ALLOC); }
  label_components(w, V, S, true);
  group_liberties(w, V, S);
  write_legal_mask(w, V, S, ni, m.to_play, m.ko);
}

// The board half of node_create_child: play the node's move (meta.fmove) on its parent's board in scratch and
// initialise the node's rows, board and legal mask from the result (board.jl:451-509 / pass_move! :426-440).
template <class W>
AGZ_FN void node_expand_child(W& w, const View& V, Scratch& S, int g, int id) {
  const long ni = node_index(V, g, id);
  NodeMeta m = V.meta[ni];
  const long pi = node_index(V, g, m.parent);
  const NodeMeta pm = V.meta[pi];
  const int a = m.fmove;
  load_board(w, V, S, pi);
  if (a != V.P) {
    const int color = pm.to_play;
    label_components(w, V, S, true);
    group_liberties(w, V, S);
    int ncap = 0, ko = -1;
    apply_move_in_scratch(w, V, S, a, color, &ncap, &ko);
    m.ko = ko;
    if (pm.to_play == 1) m.caps_b += ncap; else m.caps_w += ncap;
  }
  node_init_from_scratch(w, V, S, g, id, m);
}

// play_move!(position, a) into a fresh node.  Returns the new node id, -1 on pool exhaustion, -2 on an illegal move.
// defer (View::defer_expand, the engine's select phase): only allocate the node and link it -- nothing in the rest of
// the select phase looks at a new, unexpanded leaf's board, rows or legal mask -- and leave node_expand_child to
// k_expand, which runs one wave per new leaf instead of eight expansions in a row per game (21.7 us each: 62 % of the
// select phase).  Terminal children are finished at once: their board is scored right after the descent.
template <class W>
AGZ_FN int node_create_child(W& w, const View& V, Scratch& S, int g, int parent, int a, bool defer = false) {
  const long pi = node_index(V, g, parent);
  const NodeMeta pm = V.meta[pi];
  const int P = V.P;
  if (a < 0 || a >= V.A || !legal_bit(V, pi, a)) return -2;
  AGZ_STAMP_BEGIN(w);
  const int id = pool_alloc(w, V, S, g);
  if (id < 0) return -1;
  NodeMeta m;
  m.parent = parent;
  m.n = pm.n + 1;
  m.fmove = (int16_t)a;
  m.last_move = (int16_t)a;
  m.losses = 0;
  m.to_play = (int8_t)-pm.to_play;
  m.flags = NF_ALLOC;
  m.caps_b = pm.caps_b;
  m.caps_w = pm.caps_w;
  m.ko = -1;
  m.pad = 0;
  if (a == P && pm.last_move == P) m.flags |= NF_DONE;     // pass: done iff the previous move was a pass too
  GameState& G = V.gs[g];
  const bool terminal = (m.flags & NF_DONE) || m.n >= V.max_game_length;
  const bool later = defer && !terminal && G.npend < kMaxPend;
  w.sync();
  if (w.leader()) {
    V.meta[node_index(V, g, id)] = m;
    V.child[pi * V.AP + a] = id;
    if (later) { V.pend_node[(long)g * kMaxPend + G.npend] = id; G.npend = G.npend + 1; }
  }
  w.sync();
  if (!later) node_expand_child(w, V, S, g, id);
  AGZ_STAMP(w, V, CT_T_CREATE);
#ifdef AGZ_TIMING_EXPERIMENTS
  w.count(&V.counters[CT_N_CREATE], 1);
#endif
  return id;
}

// k_expand: the i-th deferred expansion of game g's select phase
template <class W>
AGZ_FN void game_expand(W& w, const View& V, Scratch& S, int g, int i) {
  if (i >= V.gs[g].npend) return;
  node_expand_child(w, V, S, g, V.pend_node[(long)g * kMaxPend + i]);
}

// ------------------------------------------------------------------ PUCT and select_leaf

// child_action_score[a] (mcts.jl:86-92): Float32 Q times to_play, plus Float64 U
AGZ_FN double action_score(const View& V, long ni, int a, float to_play, double scale) {
  const float denom = 1.0f + V.childN[ni * V.AP + a];
  const float q = V.childW[ni * V.AP + a] / denom;
  const float qs = q * to_play;
  const double u = (scale * (double)V.childP[ni * V.AP + a]) / (double)denom;
  return (double)qs + u;
}

AGZ_FN double puct_scale(const View& V, float n_node) {
  const float one_plus_n = 1.0f + n_node;
  return V.c_puct * (double)sqrtf(one_plus_n);
}

// k-th (0-based) set flag in ascending index order; every lane scans (LDS broadcast reads)
AGZ_FN int kth_flag(const int8_t* flag, int n, int k) {
  for (int a = 0; a < n; ++a)
    if (flag[a]) { if (k == 0) return a; --k; }
  return -1;
}

// PUCT score from register operands: the arithmetic of action_score, operation for operation
AGZ_FN double action_score_v(float n, float wv, float p, float to_play, double scale) {
  const float denom = 1.0f + n;
  const float q = wv / denom;
  const float qs = q * to_play;
  const double u = (scale * (double)p) / (double)denom;
  return (double)qs + u;
}

// select_leaf for waves that keep a node's child rows in registers (W::kRegisterRows, the GPU): R = ceil(AP / 64)
// row elements per lane.  The generic form below walks FOUR dependent global round trips per tree level (meta ->
// the node's own N in its parent's row -> its child rows for the scores -> the chosen child's id); here everything
// that depends only on the node id -- meta, N / W / P / child-id rows, the legal words -- is requested at once, the
// node's own N comes along from the parent's row of the level above, and the chosen child's id and N come out of the
// registers by shuffle: one round trip per level.  Same arithmetic, same tie-break draws: the tree is bit-identical.
template <int R, class W>
AGZ_FN int select_leaf_rows(W& w, const View& V, Scratch& S, int g, int from, int* plen_out, bool defer) {
  GameState& G = V.gs[g];
  const int A = V.A, AP = V.AP, pass = V.P;
  const uint32_t move_key = (uint32_t)V.meta[node_index(V, g, G.root)].n;
  const uint32_t sel = (uint32_t)G.sel;
  int cur = from, depth = 0, plen = 0;
  w.sync();
  float* np = slotN(V, g, cur);
  float n_cur = *np;
  for (;;) {
    const long ni = node_index(V, g, cur);
    const NodeMeta m = V.meta[ni];
    float cn[R], cw[R], cp[R];
    int cc[R];
    bool lg[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int a = w.lane + 64 * r;
      const bool in = a < AP;
      const long o = ni * AP + (in ? a : 0);
      cn[r] = V.childN[o];
      cw[r] = V.childW[o];
      cp[r] = V.childP[o];
      cc[r] = in ? V.child[o] : -1;
      lg[r] = a < A && legal_bit(V, ni, a);
    }
    const float n_new = n_cur + 1.0f;
    w.sync();
    if (w.leader()) { *np = n_new; if (plen < V.maxd) S.path[plen] = cur; }
    plen++;
    w.sync();
    if (!(m.flags & NF_EXPANDED)) break;
    auto row_f = [&](const float* v, int a) {              // element a of a row held across the lanes
      float x = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) { const float t = w.shfl(v[r], a & 63); if (r == (a >> 6)) x = t; }
      return x;
    };
    int pick;
    if (m.last_move == pass && row_f(cn, pass) == 0.0f) {
      pick = pass;    // HACK of mcts.jl:119-126: look at the double pass first
    } else {
      const double scale = puct_scale(V, n_new);
      const float tp = (float)m.to_play;
      double sc[R];
      double best = -1.0e300;
      bool have = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        sc[r] = action_score_v(cn[r], cw[r], cp[r], tp, scale);
        if (lg[r] && (!have || sc[r] > best)) { best = sc[r]; have = true; }
      }
      best = w.reduce_max(have ? best : -1.0e300);
      int cnt = 0, idx = kIntMax;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int a = w.lane + 64 * r;
        const bool f = lg[r] && sc[r] == best;
        if (a < AP) S.flag[a] = f;
        if (f) { cnt++; idx = a < idx ? a : idx; }
      }
      cnt = w.reduce_sum(cnt);
      idx = w.reduce_min(idx);
      w.sync();
      if (cnt > 1) {
        const uint64_t bits = agz_draw_u64(V.seed, G.game_id, move_key, AGZ_SITE_PUCT_TIE,
                                           (uint64_t)sel * 1024u + (uint64_t)depth);
        idx = kth_flag(S.flag, A, (int)agz_index(bits, (uint32_t)cnt));
      }
      if (cnt == 0) idx = pass;   // cannot happen: pass is always legal
      pick = idx;
      w.sync();
    }
    int nx = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) { const int t = w.shfl(cc[r], pick & 63); if (r == (pick >> 6)) nx = t; }
    const float n_next = row_f(cn, pick);
    if (nx < 0) nx = node_create_child(w, V, S, g, cur, pick, defer);
    if (nx < 0) break;   // pool exhausted: hand back the current node (flagged in the counters)
    np = &V.childN[ni * AP + pick];      // the child's own N lives in this row (slotN)
    n_cur = n_next;
    cur = nx;
    depth++;
  }
  if (w.leader()) G.sel = (int32_t)(sel + 1);
  w.sync();
  *plen_out = plen < V.maxd ? plen : V.maxd;
  return cur;
}

// select_leaf from `from`; the visited nodes are left in S.path[0..len).  Returns the leaf.
template <class W>
AGZ_FN int select_leaf(W& w, const View& V, Scratch& S, int g, int from, int* plen_out, bool defer = false) {
  if constexpr (W::kRegisterRows) {
    switch ((V.AP + 63) >> 6) {
      case 1: return select_leaf_rows<1>(w, V, S, g, from, plen_out, defer);
      case 2: return select_leaf_rows<2>(w, V, S, g, from, plen_out, defer);     // 9x9
      case 3: return select_leaf_rows<3>(w, V, S, g, from, plen_out, defer);     // 13x13
      case 6: return select_leaf_rows<6>(w, V, S, g, from, plen_out, defer);     // 19x19
      default: break;
    }
  }
  GameState& G = V.gs[g];
  const int A = V.A, pass = V.P;
  const uint32_t move_key = (uint32_t)V.meta[node_index(V, g, G.root)].n;
  const uint32_t sel = (uint32_t)G.sel;
  int cur = from, depth = 0, plen = 0;
  w.sync();
  for (;;) {
    const long ni = node_index(V, g, cur);
    const NodeMeta m = V.meta[ni];
    float* np = slotN(V, g, cur);
    const float n_new = *np + 1.0f;
    w.sync();
    if (w.leader()) { *np = n_new; if (plen < V.maxd) S.path[plen] = cur; }
    plen++;
    w.sync();
    if (!(m.flags & NF_EXPANDED)) break;
    int pick;
    if (m.last_move == pass && V.childN[ni * V.AP + pass] == 0.0f) {
      pick = pass;    // HACK of mcts.jl:119-126: look at the double pass first
    } else {
      const double scale = puct_scale(V, n_new);
      const float tp = (float)m.to_play;
      double best = -1.0e300;
      bool have = false;
      w.for_each(A, [&](int a) {
        if (!legal_bit(V, ni, a)) return;
        const double s = action_score(V, ni, a, tp, scale);
        if (!have || s > best) { best = s; have = true; }
      });
      best = w.reduce_max(have ? best : -1.0e300);
      int cnt = 0, idx = kIntMax;
      w.for_each(V.AP, [&](int a) {
        const bool f = a < A && legal_bit(V, ni, a) && action_score(V, ni, a, tp, scale) == best;
        S.flag[a] = f;
        if (f) { cnt++; idx = a < idx ? a : idx; }
      });
      cnt = w.reduce_sum(cnt);
      idx = w.reduce_min(idx);
      w.sync();
      if (cnt > 1) {
        const uint64_t bits = agz_draw_u64(V.seed, G.game_id, move_key, AGZ_SITE_PUCT_TIE,
                                           (uint64_t)sel * 1024u + (uint64_t)depth);
        idx = kth_flag(S.flag, A, (int)agz_index(bits, (uint32_t)cnt));
      }
      if (cnt == 0) idx = pass;   // cannot happen: pass is always legal
      pick = idx;
      w.sync();
    }
    int nx = V.child[ni * V.AP + pick];
    if (nx < 0) nx = node_create_child(w, V, S, g, cur, pick, defer);
    if (nx < 0) break;   // pool exhausted: hand back the current node (flagged in the counters)
    cur = nx;
    depth++;
  }
  if (w.leader()) G.sel = (int32_t)(sel + 1);
  w.sync();
  *plen_out = plen < V.maxd ? plen : V.maxd;
  return cur;
}

// walk parents from `node` up to and including `up_to` (or the root); path[0] = topmost
template <class W>
AGZ_FN int walk_path(W& w, const View& V, Scratch& S, int g, int node, int up_to) {
  int len = 0, x = node;
  for (;;) {
    ++len;
    if (x == up_to) break;
    const int p = V.meta[node_index(V, g, x)].parent;
    if (p < 0) break;
    x = p;
  }
  if (len > V.maxd) len = V.maxd;
  x = node;
  w.sync();
  for (int i = len - 1; i >= 0; --i) {
    if (w.leader()) S.path[i] = x;
    x = V.meta[node_index(V, g, x)].parent;
    if (x < 0) x = 0;
  }
  w.sync();
  return len;
}

// W(node) += sign * to_play(node) and losses_applied += sign along the path (mcts.jl:149-171)
template <class W>
AGZ_FN void path_virtual_loss(W& w, const View& V, int g, const int32_t* path, int len, int sign) {
  w.for_each(len, [&](int i) {
    const int node = path[i];
    NodeMeta& m = V.meta[node_index(V, g, node)];
    float* wp = slotW(V, g, node);
    *wp = *wp + (float)(sign * m.to_play);
    m.losses = (int16_t)(m.losses + sign);
  });
  w.sync();
}

// W(node) += value along the path (backup_value!, mcts.jl:215-225)
template <class W>
AGZ_FN void path_backup(W& w, const View& V, int g, const int32_t* path, int len, float value) {
  w.for_each(len, [&](int i) {
    float* wp = slotW(V, g, path[i]);
    *wp = *wp + value;
  });
  w.sync();
}

// N(node) -= 1 along the path (revert_visits!, mcts.jl:173-186)
template <class W>
AGZ_FN void path_revert_visits(W& w, const View& V, int g, const int32_t* path, int len) {
  w.for_each(len, [&](int i) {
    float* np = slotN(V, g, path[i]);
    *np = *np - 1.0f;
  });
  w.sync();
}

// incorporate_results!(node, probs, value, up_to) with the path root..node in `path`
template <class W>
AGZ_FN int incorporate(W& w, const View& V, int g, int node, const float* probs, float value,
                       const int32_t* path, int len) {
  const long ni = node_index(V, g, node);
  const uint8_t flags = V.meta[ni].flags;
  if (flags & NF_DONE) return AGZ_ASSERT_DONE_NODE;
  if (flags & NF_EXPANDED) {
    path_revert_visits(w, V, g, path, len);
    w.count(&V.counters[CT_DUP], 1);
    return AGZ_OK;
  }
  w.sync();
  if (w.leader()) V.meta[ni].flags = flags | NF_EXPANDED;
  w.for_each(V.A, [&](int a) {
    V.childP[ni * V.AP + a] = probs[a];
    V.childW[ni * V.AP + a] = value;   // initialise child Q as the parent's value, mcts.jl:203-211
  });
  w.sync();
  path_backup(w, V, g, path, len, value);
  return AGZ_OK;
}

// inject_noise!(node) (mcts.jl:233-239): Dirichlet(alpha) over ALL actions from the draw stream
template <class W>
AGZ_FN void inject_noise(W& w, const View& V, Scratch& S, int g, int node) {
  const GameState& G = V.gs[g];
  const long ni = node_index(V, g, node);
  const uint32_t move_key = (uint32_t)V.meta[ni].n;
  const int A = V.A;
  w.for_each(A, [&](int a) {
    S.dbuf[a] = agz_dirichlet_gamma(V.seed, G.game_id, move_key, (uint32_t)a, V.alpha);
  });
  w.sync();
  double sum = 0.0;
  for (int a = 0; a < A; ++a) sum += S.dbuf[a];   // fixed ascending order on every lane
  w.for_each(A, [&](int a) {
    const double d = sum > 0.0 ? S.dbuf[a] / sum : 1.0 / (double)A;
    const double mixed = (double)V.childP[ni * V.AP + a] * (1.0 - V.noise_w) + d * V.noise_w;
    V.childP[ni * V.AP + a] = (float)mixed;
  });
  w.sync();
}

// children_as_pi(root, squash) into out[A] (mcts.jl:241-252)
template <class W>
AGZ_FN void children_as_pi(W& w, const View& V, Scratch& S, long ni, bool squash, float* out) {
  const int A = V.A;
  if (!squash) {
    float part = 0.f;
    w.for_each(A, [&](int a) { part += V.childN[ni * V.AP + a]; });
    const float s = w.reduce_sum_f(part);   // visit counts are integers: order-free and exact
    w.for_each(A, [&](int a) { out[a] = V.childN[ni * V.AP + a] / s; });
  } else {
    w.for_each(A, [&](int a) { S.dbuf[a] = agz_pow((double)V.childN[ni * V.AP + a], 0.98); });
    w.sync();
    double s = 0.0;
    for (int a = 0; a < A; ++a) s += S.dbuf[a];
    w.for_each(A, [&](int a) { out[a] = (float)(S.dbuf[a] / s); });
  }
  w.sync();
}

// pick_move (mcts_play.jl:52-71).  Returns AGZ_OK / AGZ_ASSERT_SOFTPICK.
template <class W>
AGZ_FN int pick_move(W& w, const View& V, Scratch& S, int g, int* a_out) {
  const GameState& G = V.gs[g];
  const long ni = node_index(V, g, G.root);
  const int A = V.A, n = V.meta[ni].n;
  const int tau = V.two_player ? -1 : V.tau;
  if (n >= tau) {
    float mx = -1.0f;
    w.for_each(A, [&](int a) { const float c = V.childN[ni * V.AP + a]; mx = c > mx ? c : mx; });
    mx = w.reduce_max_f(mx);
    int cnt = 0, idx = kIntMax;
    w.for_each(V.AP, [&](int a) {
      const bool f = a < A && V.childN[ni * V.AP + a] == mx;
      S.flag[a] = f;
      if (f) { cnt++; idx = a < idx ? a : idx; }
    });
    cnt = w.reduce_sum(cnt);
    idx = w.reduce_min(idx);
    w.sync();
    if (cnt > 1) {
      const uint64_t bits = agz_draw_u64(V.seed, G.game_id, (uint32_t)n, AGZ_SITE_PICK_TIE, 0);
      idx = kth_flag(S.flag, A, (int)agz_index(bits, (uint32_t)cnt));
    }
    w.sync();
    *a_out = idx;
    return AGZ_OK;
  }
  // soft pick: cdf = cumsum(child_N) ./ cdf[end-1]; first index with !(cdf < u)
  float denom = 0.f;
  {
    float part = 0.f;
    w.for_each(A - 1, [&](int a) { part += V.childN[ni * V.AP + a]; });
    denom = w.reduce_sum_f(part);
  }
  const double u = agz_u01(agz_draw_u64(V.seed, G.game_id, (uint32_t)n, AGZ_SITE_SOFTPICK, 0));
  w.for_each(A, [&](int a) { S.dbuf[a] = (double)V.childN[ni * V.AP + a]; });
  w.sync();
  int f = A;
  float acc = 0.f;
  for (int a = 0; a < A; ++a) {
    acc += (float)S.dbuf[a];
    const float c = acc / denom;
    if (!((double)c < u)) { f = a; break; }
  }
  w.sync();
  if (f >= A || S.dbuf[f] == 0.0) return AGZ_ASSERT_SOFTPICK;
  *a_out = f;
  return AGZ_OK;
}

// ------------------------------------------------------------------ per-move phase / lifecycle

template <class W>
AGZ_FN void push_history(W& w, const View& V, int g, long ni_board) {
  GameState& G = V.gs[g];
  int8_t* h = V.hist + (long)g * 7 * V.PP;
  const int keep = G.hist_len < 6 ? G.hist_len : 6;
  for (int k = keep; k >= 1; --k) {
    w.for_each(V.P, [&](int p) { h[k * V.PP + p] = h[(k - 1) * V.PP + p]; });
    w.sync();
  }
  w.for_each(V.P, [&](int p) { h[p] = V.board[ni_board * V.PP + p]; });
  w.sync();
  if (w.leader()) G.hist_len = keep + 1;
  w.sync();
}

// Make `child` (a child of the current root via action a) the new root; everything else in
// the old tree is released (play_move!(player, c) mcts_play.jl:40-48: siblings dropped).
template <class W>
AGZ_FN void reroot(W& w, const View& V, Scratch& S, int g, int a, int child) {
  GameState& G = V.gs[g];
  const int old_root = G.root;
  const long oi = node_index(V, g, old_root);
  const float cn = V.childN[oi * V.AP + a], cw = V.childW[oi * V.AP + a];
  push_history(w, V, g, oi);
  discard_except(w, V, g, old_root, a);
  if (w.leader()) {
    G.rootN = cn;
    G.rootW = cw;
    G.root = child;
    G.sel = 0;
    NodeMeta& cm = V.meta[node_index(V, g, child)];
    cm.parent = -1;
  }
  w.sync();
}

// initialize_game! on an empty board (mcts_play.jl:110-118) + the selfplay.jl:9 resign coin
template <class W>
AGZ_FN void game_start(W& w, const View& V, Scratch& S, int g, uint64_t game_id) {
  GameState& G = V.gs[g];
  w.for_each(V.cap, [&](int i) { V.freelist[(long)g * V.cap + i] = V.cap - 1 - i; });
  w.sync();
  if (w.leader()) {
    G.game_id = game_id;
    const double u = agz_u01(agz_draw_u64(V.seed, game_id, 0, AGZ_SITE_RESIGN, 0));
    G.resign_disabled = u < V.resign_disable_frac;
    G.resign_threshold = G.resign_disabled ? -1.0 : V.resign_threshold;
    G.rootN = 0.f; G.rootW = 0.f; G.target = 0.f; G.komi = V.komi;
    G.sel = 0; G.move_count = 0; G.nqs = 0; G.hist_len = 0;
    G.free_top = V.cap; G.garbage = 0; G.nleaves = 0; G.err = 0; G.result = 0; G.was_resign = 0; G.nodes_used = 0;
    G.short_first = 0; G.short_searches = 0;
    G.phase = G_INIT;
  }
  w.sync();
  const int id = pool_alloc(w, V, S, g);
  w.for_each(V.P, [&](int p) { S.sb[p] = 0; });
  w.sync();
  NodeMeta m;
  m.parent = -1; m.n = 0; m.ko = -1; m.caps_b = 0; m.caps_w = 0; m.fmove = -1; m.last_move = -1;
  m.losses = 0; m.to_play = 1; m.flags = 0; m.pad = 0;
  node_init_from_scratch(w, V, S, g, id, m);
  if (w.leader()) G.root = id;
  w.sync();
  w.count(&V.counters[CT_STARTED], 1);
  // bench-only: a random opening prefix so that concurrent games are at mixed stages, and a
  // shortened first search so that they are also at mixed phases of their readout budget
  if (V.stagger > 0) {
    if (w.leader()) G.short_first = 1;
    w.sync();
    const int want = (int)agz_index(agz_draw_u64(V.seed, game_id, 0, AGZ_SITE_STAGGER, 0), (uint32_t)V.stagger + 1u);
    for (int t = 0; t < want; ++t) {
      const long ni = node_index(V, g, G.root);
      int cnt = 0;
      w.for_each(V.AP, [&](int a) {
        const bool f = a < V.P && legal_bit(V, ni, a);
        S.flag[a] = f;
        if (f) cnt++;
      });
      cnt = w.reduce_sum(cnt);
      w.sync();
      if (cnt == 0) break;
      const uint64_t bits = agz_draw_u64(V.seed, game_id, 0, AGZ_SITE_STAGGER, (uint64_t)t + 1u);
      const int a = kth_flag(S.flag, V.P, (int)agz_index(bits, (uint32_t)cnt));
      w.sync();
      const int ch = node_create_child(w, V, S, g, G.root, a);
      if (ch < 0) break;
      reroot(w, V, S, g, a, ch);
      if (w.leader()) { G.rootN = 0.f; G.rootW = 0.f; }
      w.sync();
    }
  }
}

// set_result! + extract_data (mcts_play.jl:100-108,126-139): copy the finished game into the
// record arena and release the slot.
template <class W>
AGZ_FN void game_finish(W& w, const View& V, Scratch& S, int g, int winner, int was_resign, float score) {
  GameState& G = V.gs[g];
  const int nm = G.move_count;
  const long slot = (long)(w.fetch_add(&V.counters[CT_RECORDED], 1ull) % (unsigned long long)V.fin_cap);
  w.count(&V.counters[CT_FINISHED], 1);
  const int mgl = V.max_game_length;
  if (w.leader()) {
    agz_game_header h;
    h.game_id = G.game_id; h.num_moves = nm; h.result = winner; h.was_resign = was_resign;
    h.resign_disabled = G.resign_disabled; h.final_score = score; h.short_searches = G.short_searches;
    V.fin_hdr[slot] = h;
    G.result = winner; G.was_resign = was_resign; G.phase = G_IDLE;
  }
  w.for_each(nm, [&](int k) {
    V.fin_moves[slot * mgl + k] = V.rec_moves[(long)g * mgl + k];
    V.fin_q[slot * mgl + k] = V.rec_q[(long)g * mgl + k];
  });
  w.for_each(nm * V.A, [&](int i) { V.fin_pi[slot * mgl * V.A + i] = V.rec_pi[(long)g * mgl * V.A + i]; });
  w.sync();
  if (was_resign) w.count(&V.counters[CT_RESIGNED], 1);
}

// A game whose node pool refused an allocation in the last select phase (G.err).  The reference's tree is unbounded
// (mcts.jl:140-147: a child is a fresh heap object); a fixed pool has to answer "and when it is full?".  With
// AGZ_POOL_MOVE_EARLY the search of this move ends here: the move phase runs on the visits the root has -- re-rooting
// hands the siblings' subtrees back -- and the shortened search is counted (agz_stats, game header).  That needs a
// child to play: an expanded root with a visited board move (the soft pick normalises by the non-pass visits,
// mcts_play.jl:64-67), or any visited child once the pick is the arg-max.  Otherwise, and always with AGZ_POOL_STALL,
// the slot waits for the host (agz_slot_status / agz_slot_abandon); the other slots are not affected.
template <class W>
AGZ_FN bool pool_full_can_move(W& w, const View& V, int g) {
  const GameState& G = V.gs[g];
  if (G.err != AGZ_POOL_EXHAUSTED || V.pool_policy != AGZ_POOL_MOVE_EARLY) return false;
  const long ri = node_index(V, g, G.root);
  const NodeMeta rm = V.meta[ri];
  if (!(rm.flags & NF_EXPANDED)) return false;
  float s = 0.f;
  w.for_each(V.P, [&](int a) { s += V.childN[ri * V.AP + a]; });
  s = w.reduce_sum_f(s);
  const bool argmax = V.two_player || rm.n >= V.tau;
  return s > 0.f || (argmax && V.childN[ri * V.AP + V.P] > 0.f);
}

// The selfplay.jl:22-43 loop body between two readout phases, for a game whose budget is spent:
// resign check -> pick -> play (record pi and Q, re-root) -> done check -> noise for the next move.
template <class W>
AGZ_FN void game_move_phase(W& w, const View& V, Scratch& S, int g) {
  GameState& G = V.gs[g];
  const int root = G.root;
  const long ri = node_index(V, g, root);
  const NodeMeta rm = V.meta[ri];
  // should_resign: Q_perspective(root) < resign_threshold (mcts_play.jl:124)
  const float q = G.rootW / (1.0f + G.rootN);
  const float qp = q * (float)rm.to_play;
  w.count_max(&V.counters[CT_PEAK_NODES], (unsigned long long)G.nodes_used);
  const bool early = G.rootN < G.target;          // only game_pre's full-pool rule sends a game here before its budget is spent
  if (early) {
    w.count(&V.counters[CT_POOL_SHORT], 1);
    if (w.leader()) { G.short_searches = G.short_searches + 1; G.err = 0; }
    w.sync();
  }
  if ((double)qp < G.resign_threshold) {
    game_finish(w, V, S, g, -rm.to_play, 1, 0.f);
    return;
  }
  AGZ_STAMP_BEGIN(w);
  int a = V.P;
  if (pick_move(w, V, S, g, &a) != AGZ_OK) a = V.P;  // the reference dies on its assertion; we pass
  // play_move!(player, c): record pi and Q, then re-root (mcts_play.jl:26-50)
  const int k = G.move_count;
  if (k < V.max_game_length) {
    children_as_pi(w, V, S, ri, rm.n <= V.tau, V.rec_pi + ((long)g * V.max_game_length + k) * V.A);
    if (w.leader()) {
      V.rec_moves[(long)g * V.max_game_length + k] = (int16_t)a;
      V.rec_q[(long)g * V.max_game_length + k] = q;
    }
  }
  w.sync();
  AGZ_STAMP(w, V, CT_T_PICK);
  int child = V.child[ri * V.AP + a];
  if (child < 0) child = node_create_child(w, V, S, g, root, a);
  if (child < 0) { game_finish(w, V, S, g, 0, 0, 0.f); return; }
  AGZ_STAMP(w, V, CT_T_CHILD);
  reroot(w, V, S, g, a, child);
  if (w.leader()) { G.move_count = k + 1; G.nqs = k + 1; }
  w.sync();
  if (!G.short_first) w.count(&V.counters[CT_POSITIONS], 1);
  w.sync();
  if (w.leader()) G.short_first = 0;
  w.sync();
  AGZ_STAMP(w, V, CT_T_REROOT);
  if (node_is_done(V, g, child)) {
    load_board(w, V, S, node_index(V, g, child));
    const float sc = area_score(w, V, S, G.komi);
    game_finish(w, V, S, g, result_of(sc), 0, sc);
    return;
  }
  inject_noise(w, V, S, g, child);
  if (w.leader()) G.target = G.rootN + (float)V.R;
  w.sync();
  AGZ_STAMP(w, V, CT_T_NOISE);
}

// Record which boards feed the eight history planes of leaf `k` (features.jl:8-14): path
// nodes newest first, then the game's history ring, then "repeat the oldest available".
template <class W>
AGZ_FN void record_leaf(W& w, const View& V, Scratch& S, int g, int k, int leaf, int plen) {
  const GameState& G = V.gs[g];
  const long li = (long)g * V.par + k;
  const int d = plen - 1;   // path[0] is the root when the descent started there
  int avail = d + G.hist_len;
  if (avail > 7) avail = 7;
  w.for_each(8, [&](int s) {
    const int t = s <= avail ? s : avail;
    V.leaf_featsrc[li * 8 + s] = t <= d ? S.path[d - t] : -(t - d - 1) - 1;
  });
  w.for_each(plen, [&](int i) { V.leaf_path[li * V.maxd + i] = S.path[i]; });
  if (w.leader()) {
    V.leaf_node[li] = leaf;
    V.leaf_plen[li] = plen;
    V.leaf_tp[li] = V.meta[node_index(V, g, leaf)].to_play;
  }
  w.sync();
}

// One tree_search! select phase (mcts_play.jl:73-87): up to `par` leaves, `2*par` attempts.
template <class W>
AGZ_FN void game_select_phase(W& w, const View& V, Scratch& S, int g, int par, bool defer = false) {
  GameState& G = V.gs[g];
  int nleaves = 0, failsafe = 0, terminal = 0;
  const float n_before = G.rootN;
  // (G.err says what THIS select phase ran into: a refused allocation of an earlier phase must not shorten a later search --
  // the tree may have been re-rooted, or its garbage released, since)
  if (w.leader()) { G.npend = 0; G.err = 0; }
  w.sync();
  while (nleaves < par && failsafe < 2 * par) {
    failsafe++;
    int plen = 0;
    const int leaf = select_leaf(w, V, S, g, G.root, &plen, defer);
    if (node_is_done(V, g, leaf)) {
      load_board(w, V, S, node_index(V, g, leaf));
      const float value = (float)result_of(area_score(w, V, S, G.komi));
      path_backup(w, V, g, S.path, plen, value);
      terminal++;
      continue;
    }
    path_virtual_loss(w, V, g, S.path, plen, +1);
    record_leaf(w, V, S, g, nleaves, leaf, plen);
    nleaves++;
  }
  if (w.leader()) G.nleaves = nleaves;
  w.sync();
  w.count(&V.counters[CT_TERMINAL], (unsigned long long)terminal);
  w.count(&V.counters[CT_EVALS], (unsigned long long)nleaves);
  w.count(&V.counters[CT_ROOTVISITS], (unsigned long long)(G.rootN - n_before));
}

// ---------------------------------------------------------------- evaluate() arena ----
// neural_net.jl:103-158 for many games at once.  Slots come in pairs: slot 2i is the Black
// player of game i (network 0), slot 2i+1 the White player (network 1), each with its own tree
// (`black` / `white` MCTSPlayers, two_player_mode: arg-max picks, no noise, no pi).  The side to
// move searches; when its budget is spent it moves at the END of k_post and publishes the move in
// the pair's mailbox; the partner reads the mailbox at the START of the next k_pre, plays the same
// move on its tree and takes over.  Writer and reader therefore always run in different kernel
// launches: no intra-kernel communication, deterministic.

// arena mailbox words of a slot pair: ar_hdr[4 * pair .. +3] = {local game index, plies played, done, result};
// after the (games/2 + 1) headers, one abort word per pair = 1 + local game index of a game one side had to drop
constexpr int kArenaWordsPerPair = 5;
AGZ_FN int32_t* arena_abort_word(const View& V, int pair) { return V.ar_hdr + 4 * (V.games / 2 + 1) + pair; }

template <class W>
AGZ_FN void arena_finish(W& w, const View& V, Scratch& S, int g, bool emit, int winner, int was_resign) {
  GameState& G = V.gs[g];
  if (emit) {
    // `result(black.root.position)` (:147) is the Tromp-Taylor result of the final position
    load_board(w, V, S, node_index(V, g, G.root));
    const float sc = area_score(w, V, S, G.komi);
    game_finish(w, V, S, g, winner, was_resign, sc);
  }
  if (w.leader()) { G.phase = G_IDLE; G.arena_k = G.arena_k + 1; G.nleaves = 0; }
  w.sync();
}

template <class W>
AGZ_FN void arena_start(W& w, const View& V, Scratch& S, int g, uint64_t index) {
  GameState& G = V.gs[g];
  const int k = G.arena_k;
  game_start(w, V, S, g, 2 * index + (uint64_t)(g & 1));
  if (w.leader()) {
    G.arena_k = k;
    G.resign_disabled = 0;
    G.resign_threshold = V.resign_threshold;     // MCTSPlayer default, evaluate passes none (:110-111)
    if (g & 1) G.phase = G_ARENA_WAIT;
    else { G.target = G.rootN + (float)V.R; G.phase = G_SEARCH; }     // :121-126, root not pre-expanded
  }
  w.sync();
}

// play the partner's move on this slot's tree: play_move!(inactive, move) (:137)
template <class W>
AGZ_FN bool arena_apply(W& w, const View& V, Scratch& S, int g, int a, float q) {
  GameState& G = V.gs[g];
  const long ri = node_index(V, g, G.root);
  const int k = G.move_count;
  if (k < V.max_game_length) {
    w.for_each(V.A, [&](int i) { V.rec_pi[((long)g * V.max_game_length + k) * V.A + i] = 0.f; });
    if (w.leader()) {
      V.rec_moves[(long)g * V.max_game_length + k] = (int16_t)a;
      V.rec_q[(long)g * V.max_game_length + k] = q;
    }
  }
  w.sync();
  int child = V.child[ri * V.AP + a];
  if (child < 0) child = node_create_child(w, V, S, g, G.root, a);
  if (child < 0) return false;
  reroot(w, V, S, g, a, child);
  if (w.leader()) { G.move_count = k + 1; G.nqs = k + 1; }
  w.sync();
  return true;
}

template <class W>
AGZ_FN void arena_pre(W& w, const View& V, Scratch& S, int g) {
  GameState& G = V.gs[g];
  const int pair = g >> 1, npairs = V.games >> 1;
  if (G.phase == G_MANUAL || G.phase == G_RETIRED) {
    if (w.leader()) G.nleaves = 0;
    w.sync();
    return;
  }
  if (G.phase == G_IDLE) {
    if (w.leader()) G.nleaves = 0;
    w.sync();
    const long long idx = (long long)pair + (long long)G.arena_k * npairs;    // both slots derive the same sequence
    if (V.total_games > 0 && idx >= V.total_games) {
      if (w.leader()) G.phase = G_RETIRED;
      w.sync();
      return;
    }
    arena_start(w, V, S, g, V.id_base + (uint64_t)idx * V.id_stride);
  }
  if (G.phase == G_ARENA_WAIT) {
    if (w.leader()) G.nleaves = 0;
    w.sync();
    const int32_t* hdr = V.ar_hdr + 4 * pair;
    const int hk = hdr[0], hply = hdr[1], hdone = hdr[2], hres = hdr[3];
    if (hk != G.arena_k) return;                            // nothing published for this game yet
    if (hply > G.move_count) {
      const long pg = (long)(g ^ 1) * V.max_game_length + G.move_count;
      const int a = V.rec_moves[pg];
      const float q = V.rec_q[pg];
      if (!arena_apply(w, V, S, g, a, q)) {
        // node pool exhausted while following the partner's move: the partner is parked in G_ARENA_WAIT for
        // this slot's reply and would wait forever.  Raise the pair's abort word; the partner reads it in
        // k_post (a different launch, so still no intra-kernel communication) and files the game as void.
        if (w.leader()) arena_abort_word(V, pair)[0] = G.arena_k + 1;
        w.sync();
        arena_finish(w, V, S, g, false, 0, 0);
        return;
      }
    } else if (!hdone) {
      return;
    }
    if (hdone) {                                            // set_result!(inactive, ...) (:131,144)
      if (w.leader()) { G.result = (hres & 3) - 1; G.was_resign = hres >> 2; }
      w.sync();
      arena_finish(w, V, S, g, false, 0, 0);
      return;
    }
    if (w.leader()) { G.target = G.rootN + (float)V.R; G.phase = G_SEARCH; }     // :121-126
    w.sync();
  }
  if (G.phase == G_SEARCH) game_select_phase(w, V, S, g, V.par, V.defer_expand != 0);
}

// the active player's budget is spent: :128-146
template <class W>
AGZ_FN void arena_move_phase(W& w, const View& V, Scratch& S, int g) {
  GameState& G = V.gs[g];
  const int pair = g >> 1;
  int32_t* hdr = V.ar_hdr + 4 * pair;
  const long ri = node_index(V, g, G.root);
  const NodeMeta rm = V.meta[ri];
  const float q = G.rootW / (1.0f + G.rootN);
  const float qp = q * (float)rm.to_play;
  if ((double)qp < G.resign_threshold) {                    // should_resign(active) (:129-133)
    const int winner = -rm.to_play;
    if (w.leader()) { hdr[0] = G.arena_k; hdr[1] = G.move_count; hdr[2] = 1; hdr[3] = (winner + 1) | 4; }
    w.sync();
    arena_finish(w, V, S, g, true, winner, 1);
    return;
  }
  int a = V.P;
  if (pick_move(w, V, S, g, &a) != AGZ_OK) a = V.P;
  if (!arena_apply(w, V, S, g, a, q)) {
    if (w.leader()) { hdr[0] = G.arena_k; hdr[1] = G.move_count; hdr[2] = 1; hdr[3] = 1; }
    w.sync();
    arena_finish(w, V, S, g, true, 0, 0);
    return;
  }
  w.count(&V.counters[CT_POSITIONS], 1);
  if (node_is_done(V, g, G.root)) {                         // :140-145
    load_board(w, V, S, node_index(V, g, G.root));
    const int winner = result_of(area_score(w, V, S, G.komi));
    if (w.leader()) { hdr[0] = G.arena_k; hdr[1] = G.move_count; hdr[2] = 1; hdr[3] = winner + 1; }
    w.sync();
    arena_finish(w, V, S, g, true, winner, 0);
    return;
  }
  if (w.leader()) { hdr[0] = G.arena_k; hdr[1] = G.move_count; hdr[2] = 0; hdr[3] = 0; G.phase = G_ARENA_WAIT; }
  w.sync();
}

// Phase A+B of a self-play step for game slot g: lifecycle, per-move phase, select.
template <class W>
AGZ_FN void game_pre(W& w, const View& V, Scratch& S, int g) {
  GameState& G = V.gs[g];
  if (w.leader()) G.npend = 0;       // k_expand looks at every game, whether it selects this step or not
  AGZ_STAMP_BEGIN(w);
  const unsigned long long agz_t_start = agz_t_prev;
  (void)agz_t_start;
  if (G.garbage > 0 && G.phase != G_MANUAL) free_pending(w, V, S, g, kFreeBudget);
  AGZ_STAMP(w, V, CT_T_FREE);
  if (V.arena && G.phase != G_MANUAL) { arena_pre(w, V, S, g); return; }
  if (G.phase == G_MANUAL || G.phase == G_RETIRED) {
    if (w.leader() && G.phase == G_RETIRED) G.nleaves = 0;
    w.sync();
    return;
  }
  bool agz_moved = false;
  if (G.phase == G_SEARCH) {
    const bool full = G.err == AGZ_POOL_EXHAUSTED;
    const bool can_move = !(G.rootN < G.target) || pool_full_can_move(w, V, g);
    if (w.leader()) G.stalled = full && !can_move;      // what agz_stats.stalled_games / agz_slot_status report
    w.sync();
    if (can_move) { game_move_phase(w, V, S, g); agz_moved = true; }
  } else if (G.stalled) {
    if (w.leader()) G.stalled = 0;
    w.sync();
  }
  (void)agz_moved;
  AGZ_STAMP_BEGIN_AGAIN(w);
  if (G.phase == G_IDLE) {
    if (w.leader()) G.nleaves = 0;
    w.sync();
    // claim the next global game index; retire the slot when the quota is used up
    const long long idx = (long long)w.fetch_add(&V.counters[CT_CLAIMED], 1ull);
    if (V.total_games > 0 && idx >= V.total_games) {
      if (w.leader()) G.phase = G_RETIRED;
      w.sync();
      return;
    }
    game_start(w, V, S, g, V.id_base + (uint64_t)idx * V.id_stride);
  }
  if (G.phase == G_INIT) {
    // selfplay.jl:16-20: the very first select_leaf returns the unexpanded root; it is sent to
    // the network alone, without virtual loss, and incorporated with up_to = itself.
    int plen = 0;
    const int leaf = select_leaf(w, V, S, g, G.root, &plen);
    record_leaf(w, V, S, g, 0, leaf, plen);
    if (w.leader()) { G.nleaves = 1; G.phase = G_INIT_WAIT; }
    w.sync();
    w.count(&V.counters[CT_EVALS], 1);
    return;
  }
  if (G.phase == G_SEARCH) {
    game_select_phase(w, V, S, g, V.par, V.defer_expand != 0);
#ifdef AGZ_TIMING_EXPERIMENTS
    const unsigned long long t = w.clock();
    w.count(&V.counters[agz_moved ? CT_T_MOVE_SELECT : CT_T_SELECT], t - agz_t_prev);
    w.count(&V.counters[agz_moved ? CT_N_MOVE : CT_N_SELECT], 1);
    if (agz_moved) w.count_max(&V.counters[CT_T_MOVE_MAX], t - agz_t_start);
#endif
  }
}

// Phase C: revert virtual losses and incorporate the network outputs in collection order
// (mcts_play.jl:89-96), or finish the selfplay.jl:16-20 pre-expansion.
template <class W>
AGZ_FN void game_post(W& w, const View& V, Scratch& S, int g) {
  GameState& G = V.gs[g];
  const int nl = G.nleaves;
  if (V.arena && G.phase == G_ARENA_WAIT) {                 // the partner dropped this game in k_pre (pool exhausted)
    if (arena_abort_word(V, g >> 1)[0] == G.arena_k + 1) arena_finish(w, V, S, g, true, 0, 0);
    return;
  }
  if (V.arena && G.phase == G_SEARCH && nl <= 0) {          // a select phase of terminal leaves only
    if (!(G.rootN < G.target)) arena_move_phase(w, V, S, g);
    return;
  }
  if (nl <= 0 || (G.phase != G_SEARCH && G.phase != G_INIT_WAIT && G.phase != G_MANUAL)) return;
  const float n_before = G.rootN;
  for (int k = 0; k < nl; ++k) {
    const long li = (long)g * V.par + k;
    const int leaf = V.leaf_node[li], plen = V.leaf_plen[li];
    const long b = (long)G.leaf_base + k;
    const int32_t* path = V.leaf_path + li * V.maxd;
    if (G.phase != G_INIT_WAIT) path_virtual_loss(w, V, g, path, plen, -1);
    incorporate(w, V, g, leaf, V.pi + b * V.A, V.v[b], path, plen);
  }
  (void)n_before;
  if (G.phase == G_INIT_WAIT) {
    inject_noise(w, V, S, g, G.root);
    if (w.leader()) {
      float budget = (float)V.R;
      if (G.short_first) {
        const double u = agz_u01(agz_draw_u64(V.seed, G.game_id, 0, AGZ_SITE_STAGGER, 1000000u));
        budget = (float)(1 + (int)(u * (double)(V.R - 1)));
      }
      G.target = G.rootN + budget;
      G.phase = G_SEARCH;
    }
  }
  if (w.leader()) G.nleaves = 0;
  w.sync();
  if (V.arena && G.phase == G_SEARCH && !(G.rootN < G.target)) arena_move_phase(w, V, S, g);
}

// Feature planes of one leaf slot into the stem-input layout [P][32] (features.jl:3-26)
template <class W>
// part / parts: this wave does the part-th of `parts` equal shares of the leaf's items (k_leaf_features gives a leaf four waves)
AGZ_FN void leaf_features(W& w, const View& V, int g, int k, float* x32, float* whcn, int part = 0, int parts = 1) {
  const long li = (long)g * V.par + k;
  const int tp = V.leaf_tp[li];
  const int P = V.P;
  int src[8];
  for (int s = 0; s < 8; ++s) src[s] = V.leaf_featsrc[li * 8 + s];
  auto stone = [&](int s, int p) -> int {
    return src[s] >= 0 ? V.board[node_index(V, g, src[s]) * V.PP + p] : V.hist[((long)g * 7 + (-src[s] - 1)) * V.PP + p];
  };
  if (x32) {
    // The stem input row of point p is 32 floats = eight 16-byte quads: quad c < 4 holds the planes of history boards
    // 2c and 2c + 1 (own / opponent stones each), quad 4 the colour plane, quads 5..7 the padding.  Item = (point, quad),
    // quad fastest: the 64 lanes of a wave write 1 KB of consecutive bytes per store instruction, and a lane reads the two
    // board bytes its quad needs instead of all eight.  (Round 6 tried, and timed, four things on this kernel -- this store
    // pattern against lane = point, four leaves per workgroup, four waves per leaf, streaming stores: 35-39 us per 8192
    // leaves of 9x9 every time, 0.3 of the HBM rate in bytes.  What it waits for is the chain leaf record -> eight node
    // boards scattered over a 14 GB pool -> store, three dependent round trips with a TLB miss each, not bandwidth.)
    struct alignas(16) Quad { float a, b, c, d; };
    Quad* dst = reinterpret_cast<Quad*>(x32);
    const int per = (P * 8 + parts - 1) / parts, lo = part * per, hi = lo + per < P * 8 ? lo + per : P * 8;
    w.for_each(hi - lo, [&](int j) {
      const int i = lo + j;
      const int p = i >> 3, c = i & 7;
      Quad q{0.f, 0.f, 0.f, 0.f};
      if (c < 4) {
        const int s0 = stone(2 * c, p), s1 = stone(2 * c + 1, p);
        q.a = s0 == tp ? 1.f : 0.f;
        q.b = s0 == -tp ? 1.f : 0.f;
        q.c = s1 == tp ? 1.f : 0.f;
        q.d = s1 == -tp ? 1.f : 0.f;
      } else if (c == 4) {
        q.a = (float)tp;
      }
      dst[i] = q;
    });
  }
  if (whcn) {
    const int per = (P + parts - 1) / parts, lo = part * per, hi = lo + per < P ? lo + per : P;
    w.for_each(hi - lo, [&](int j) {
      const int p = lo + j;
      for (int s = 0; s < 8; ++s) {
        const int c = stone(s, p);
        whcn[(long)P * (2 * s) + p] = c == tp ? 1.f : 0.f;
        whcn[(long)P * (2 * s + 1) + p] = c == -tp ? 1.f : 0.f;
      }
      whcn[(long)P * 16 + p] = (float)tp;
    });
  }
}


// ------------------------------------------------------------------ batched Go-rule entry points
// (agz_go_play / agz_go_legal / agz_go_score): one wave per position, no tree involved.

template <class W>
AGZ_FN void go_legal_one(W& w, const View& V, Scratch& S, const int8_t* board, int tp, int ko, int8_t* out) {
  w.for_each(V.P, [&](int p) { S.sb[p] = board[p]; });
  w.sync();
  label_components(w, V, S, true);
  group_liberties(w, V, S);
  compute_legal_flags(w, V, S, tp, ko);
  w.for_each(V.A, [&](int a) { out[a] = S.flag[a]; });
  w.sync();
}

template <class W>
AGZ_FN void go_play_one(W& w, const View& V, Scratch& S, const int8_t* board, int tp, int ko, int move,
                        int8_t* board_out, int32_t* ko_out, int32_t* ncap_out, int32_t* status_out) {
  const int P = V.P;
  w.for_each(P, [&](int p) { S.sb[p] = board[p]; });
  w.sync();
  int ncap = 0, nko = -1, status = AGZ_OK;
  if (move == P) {
    // pass_move!: board unchanged, ko cleared
  } else if (move < 0 || move > P) {
    status = AGZ_ILLEGAL_MOVE;
  } else {
    label_components(w, V, S, true);
    group_liberties(w, V, S);
    compute_legal_flags(w, V, S, tp, ko);
    const bool ok = S.flag[move];
    w.sync();
    if (!ok) status = AGZ_ILLEGAL_MOVE;
    else apply_move_in_scratch(w, V, S, move, tp, &ncap, &nko);
  }
  if (status != AGZ_OK) {
    w.for_each(P, [&](int p) { board_out[p] = board[p]; });
    nko = ko;
    ncap = 0;
  } else {
    w.for_each(P, [&](int p) { board_out[p] = S.sb[p]; });
  }
  if (w.leader()) { *ko_out = nko; *ncap_out = ncap; *status_out = status; }
  w.sync();
}

template <class W>
AGZ_FN void go_score_one(W& w, const View& V, Scratch& S, const int8_t* board, float komi, float* out) {
  w.for_each(V.P, [&](int p) { S.sb[p] = board[p]; });
  w.sync();
  const float sc = area_score(w, V, S, komi);
  if (w.leader()) *out = sc;
  w.sync();
}

// ------------------------------------------------------------------ single-tree compat ops
// The reference's MCTSNode / MCTSPlayer calls, one at a time, on game slot g (agz_tree_*).

enum TreeOpCode : int32_t {
  TOP_INIT = 0, TOP_SELECT, TOP_ADD_CHILD, TOP_VLOSS_ADD, TOP_VLOSS_REVERT, TOP_INCORPORATE, TOP_NOISE,
  TOP_SEARCH_SELECT, TOP_SEARCH_POST, TOP_PICK, TOP_PLAY, TOP_RESIGN, TOP_SCORES, TOP_PENDING
};

struct TreeArgs {
  int32_t op, g, node, a, up_to, par;
  float value;
  agz_position_info info;
  const float* probs;      // [A]
  const int8_t* board;     // [P]
  const int8_t* history;   // [history_len][P]
  int32_t* iout;           // [4]
  double* dout;            // [A]
};

template <class W>
AGZ_FN void tree_op(W& w, const View& V, Scratch& S, const TreeArgs& T) {
  const int g = T.g;
  GameState& G = V.gs[g];
  int status = AGZ_OK, r0 = 0;
  switch (T.op) {
    case TOP_INIT: {
      // initialize_game!(player, pos), mcts_play.jl:110-118
      w.for_each(V.cap, [&](int i) { V.freelist[(long)g * V.cap + i] = V.cap - 1 - i; });
      w.sync();
      if (w.leader()) {
        G.rootN = 0.f; G.rootW = 0.f; G.target = 0.f; G.komi = T.info.komi;
        G.sel = 0; G.move_count = 0; G.nqs = 0; G.hist_len = T.info.history_len;
        G.free_top = V.cap; G.garbage = 0; G.nleaves = 0; G.err = 0; G.result = 0; G.was_resign = 0; G.nodes_used = 0;
        G.short_searches = 0;
        G.phase = G_MANUAL;
        G.resign_threshold = V.resign_threshold; G.resign_disabled = 0;
      }
      w.sync();
      for (int h = 0; h < T.info.history_len && h < 7; ++h)
        w.for_each(V.P, [&](int p) { V.hist[((long)g * 7 + h) * V.PP + p] = T.history[(long)h * V.P + p]; });
      const int id = pool_alloc(w, V, S, g);
      w.for_each(V.P, [&](int p) { S.sb[p] = T.board[p]; });
      w.sync();
      NodeMeta m;
      m.parent = -1; m.n = T.info.n; m.ko = T.info.ko; m.caps_b = T.info.caps_black; m.caps_w = T.info.caps_white;
      m.fmove = -1; m.last_move = (int16_t)T.info.last_move; m.losses = 0; m.to_play = (int8_t)T.info.to_play;
      m.flags = 0; m.pad = 0;
      node_init_from_scratch(w, V, S, g, id, m);
      if (w.leader()) G.root = id;
      w.sync();
      r0 = id;
    } break;
    case TOP_SELECT: {
      int plen = 0;
      r0 = select_leaf(w, V, S, g, T.node, &plen);
    } break;
    case TOP_ADD_CHILD: {
      const long ni = node_index(V, g, T.node);
      int c = (T.a >= 0 && T.a < V.A) ? V.child[ni * V.AP + T.a] : -2;
      if (c == -1) c = node_create_child(w, V, S, g, T.node, T.a);
      if (c == -2) status = AGZ_ILLEGAL_MOVE;
      else if (c < 0) status = AGZ_POOL_EXHAUSTED;
      r0 = c;
    } break;
    case TOP_VLOSS_ADD:
    case TOP_VLOSS_REVERT: {
      const int len = walk_path(w, V, S, g, T.node, T.up_to);
      path_virtual_loss(w, V, g, S.path, len, T.op == TOP_VLOSS_ADD ? +1 : -1);
    } break;
    case TOP_INCORPORATE: {
      const int len = walk_path(w, V, S, g, T.node, T.up_to);
      status = incorporate(w, V, g, T.node, T.probs, T.value, S.path, len);
    } break;
    case TOP_NOISE: inject_noise(w, V, S, g, T.node); break;
    case TOP_SEARCH_SELECT: {
      game_select_phase(w, V, S, g, T.par);
      r0 = G.nleaves;
      if (w.leader()) G.leaf_base = 0;
      w.sync();
    } break;
    case TOP_SEARCH_POST: game_post(w, V, S, g); break;
    case TOP_PICK: status = pick_move(w, V, S, g, &r0); break;
    case TOP_PLAY: {
      // play_move!(player, c), mcts_play.jl:26-50
      const int root = G.root;
      const long ri = node_index(V, g, root);
      int c = (T.a >= 0 && T.a < V.A) ? V.child[ri * V.AP + T.a] : -2;
      if (c == -1) c = node_create_child(w, V, S, g, root, T.a);
      if (c < 0) { r0 = 0; status = c == -2 ? AGZ_OK : AGZ_POOL_EXHAUSTED; break; }
      const int k = G.move_count;
      const float q = G.rootW / (1.0f + G.rootN);
      if (!V.two_player && k < V.max_game_length)
        children_as_pi(w, V, S, ri, V.meta[ri].n <= V.tau, V.rec_pi + ((long)g * V.max_game_length + k) * V.A);
      if (w.leader() && G.nqs < V.max_game_length) {
        V.rec_moves[(long)g * V.max_game_length + G.nqs] = (int16_t)T.a;
        V.rec_q[(long)g * V.max_game_length + G.nqs] = q;
      }
      w.sync();
      reroot(w, V, S, g, T.a, c);
      free_pending(w, V, S, g, V.cap);      // single-tree API: release the dropped siblings right away
      if (w.leader()) { if (!V.two_player) G.move_count = k + 1; G.nqs = G.nqs + 1; }
      w.sync();
      r0 = 1;
    } break;
    case TOP_RESIGN: {
      const float q = G.rootW / (1.0f + G.rootN);
      const float qp = q * (float)V.meta[node_index(V, g, G.root)].to_play;
      r0 = (double)qp < G.resign_threshold;
    } break;
    case TOP_SCORES: {
      const long ni = node_index(V, g, T.node);
      const NodeMeta m = V.meta[ni];
      const double scale = puct_scale(V, *slotN(V, g, T.node));
      w.for_each(V.A, [&](int a) { T.dout[a] = action_score(V, ni, a, (float)m.to_play, scale); });
      w.sync();
    } break;
    case TOP_PENDING: {
      int cnt = 0;
      w.for_each(V.cap, [&](int i) {
        const NodeMeta& m = V.meta[node_index(V, g, i)];
        if ((m.flags & NF_ALLOC) && m.losses != 0) cnt++;
      });
      r0 = w.reduce_sum(cnt);
    } break;
    default: status = AGZ_BAD_ARGUMENT;
  }
  w.sync();
  if (w.leader()) { T.iout[0] = status; T.iout[1] = r0; }
  w.sync();
}

}  // namespace agz
