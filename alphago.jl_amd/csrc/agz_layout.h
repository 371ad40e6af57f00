// agz_layout.h -- dimensions and buffer list of the engine state (shared by the HIP engine and
// by the host wave simulator used in tests, so both allocate the same View).
#pragma once
#include <cstddef>
#include <cstdint>

#include "agz_state.h"

namespace agz {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

inline void fill_dims(View& V, const agz_config& c) {
  V.N = c.board_size;
  V.P = V.N * V.N;
  V.PP = round_up(V.P, 16);
  V.A = V.P + 1;
  V.AP = round_up(V.A, 16);
  V.LW = (V.A + 31) / 32;
  V.games = c.games;
  V.par = c.parallel_readouts;
  V.R = c.num_readouts;
  V.max_game_length = (V.P * 7) / 5;                      // mcts.jl:21
  V.maxd = V.max_game_length + 8;
  V.tau = ((V.P / 12) / 2) * 2;                           // mcts_play.jl:19
  V.arena = c.arena_mode ? 1 : 0;
  V.two_player = c.two_player_mode || c.arena_mode;       // evaluate(): both players are two_player_mode
  V.stagger = 0;                 // agz_debug_set_stagger (bench / soak tests only)
  // default pool: with subtree reuse a game's tree settles near R / (1 - f), f = the played child's share of the root's
  // visits (16 R covers f < 0.94); a sharp policy over a long game adds to that whatever R is, hence the term in the
  // game length (19x19, 16-32 readouts, 500 moves needed ~8000 nodes: tools/soak_selfplay.py).  A full pool is handled
  // by agz_config.pool_policy, not by stopping.
  V.cap = c.max_nodes_per_game > 0 ? c.max_nodes_per_game : 16 * c.num_readouts + 256 + 16 * V.max_game_length;
  V.pool_policy = c.pool_policy;
  V.fin_cap = c.record_capacity_games > 0 ? c.record_capacity_games : 2 * c.games + 64;
  V.total_games = 0;
  V.seed = c.seed;
  V.id_base = c.game_id_base;
  V.id_stride = c.game_id_stride ? c.game_id_stride : 1;
  V.c_puct = c.c_puct;
  V.noise_w = c.dirichlet_noise_weight;
  V.alpha = (double)(float)(0.03 * 361.0 / (double)V.A);  // mcts.jl:22 with go.jl:24
  V.resign_threshold = c.resign_threshold;
  V.resign_disable_frac = c.resign_disable_fraction;
  V.komi = c.komi;
  V.defer_expand = 0;
}

// visits every buffer of the View: f(pointer-reference, element count)
template <class F>
inline void for_each_buffer(View& V, F&& f) {
  const size_t nodes = (size_t)V.games * V.cap;
  const size_t leaves = (size_t)V.games * V.par;
  const size_t mgl = (size_t)V.max_game_length;
  f(V.childN, nodes * V.AP);
  f(V.childW, nodes * V.AP);
  f(V.childP, nodes * V.AP);
  f(V.child, nodes * V.AP);
  f(V.board, nodes * V.PP);
  f(V.meta, nodes);
  f(V.legal, nodes * V.LW);
  f(V.gs, (size_t)V.games);
  f(V.hist, (size_t)V.games * 7 * V.PP);
  f(V.freelist, nodes);
  f(V.leaf_node, leaves);
  f(V.leaf_featsrc, leaves * 8);
  f(V.leaf_tp, leaves);
  f(V.leaf_plen, leaves);
  f(V.leaf_path, leaves * V.maxd);
  f(V.pend_node, (size_t)V.games * kMaxPend);
  f(V.rec_moves, (size_t)V.games * mgl);
  f(V.rec_pi, (size_t)V.games * mgl * V.A);
  f(V.rec_q, (size_t)V.games * mgl);
  f(V.fin_hdr, (size_t)V.fin_cap);
  f(V.fin_moves, (size_t)V.fin_cap * mgl);
  f(V.fin_pi, (size_t)V.fin_cap * mgl * V.A);
  f(V.fin_q, (size_t)V.fin_cap * mgl);
  f(V.counters, (size_t)CT_COUNT);
  f(V.batch_count, (size_t)2);
  f(V.ar_hdr, (size_t)5 * (V.games / 2 + 1));        // 4 header words + 1 abort word per pair (agz_search.h)
}

}  // namespace agz
