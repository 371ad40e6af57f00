// agz_capi.hip -- the extern "C" surface of libagz.so (include/agz.h): argument checks,
// exception -> status translation, nothing else.
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "agz_engine.h"
#include "agz_search.h"

struct agz_engine {
  agz::Engine* impl;
  std::string err;
};

namespace {
std::string g_create_error;

template <class F>
agz_status guard(agz_engine* e, F&& f) {
  if (!e || !e->impl) return AGZ_BAD_ARGUMENT;
  try {
    f(*e->impl);
    return AGZ_OK;
  } catch (const agz::Error& x) {
    e->err = x.what();
    return x.status;
  } catch (const std::exception& x) {
    e->err = x.what();
    return AGZ_BAD_ARGUMENT;
  }
}

// like guard, but the callable returns a reference-level status (IllegalMove, assertion ...)
template <class F>
agz_status guard_status(agz_engine* e, F&& f) {
  if (!e || !e->impl) return AGZ_BAD_ARGUMENT;
  try {
    const int st = f(*e->impl);
    if (st != AGZ_OK) e->err = "reference-level condition " + std::to_string(st);
    return st;
  } catch (const agz::Error& x) {
    e->err = x.what();
    return x.status;
  } catch (const std::exception& x) {
    e->err = x.what();
    return AGZ_BAD_ARGUMENT;
  }
}

agz::TreeArgs targs(int op, int g, int node = 0, int a = 0, int up_to = -1) {
  agz::TreeArgs T;
  std::memset(&T, 0, sizeof(T));
  T.op = op; T.g = g; T.node = node; T.a = a; T.up_to = up_to;
  return T;
}
}  // namespace

extern "C" {

int32_t agz_version(void) { return AGZ_VERSION; }

void agz_config_default(agz_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->board_size = 19;          // GoEnv(board_size = 19), go.jl:10
  c->tower_height = 19;        // NeuralNet(env; tower_height = 19), neural_net.jl:13
  c->games = 1;
  c->num_readouts = 800;       // mcts_play.jl:17
  c->parallel_readouts = 8;    // mcts_play.jl:73
  c->two_player_mode = 0;
  c->arena_mode = 0;
  c->komi = 7.5f;              // board.jl:297
  c->c_puct = 0.96;            // mcts.jl:11
  c->dirichlet_noise_weight = 0.25;   // mcts.jl:13
  c->resign_threshold = -0.9;  // mcts_play.jl:18
  c->resign_disable_fraction = 0.05;  // selfplay.jl:9
  c->seed = 0;
  c->game_id_base = 0;
  c->game_id_stride = 1;
}

agz_status agz_engine_create(const agz_config* cfg, agz_engine** out) {
  if (!cfg || !out) return AGZ_BAD_ARGUMENT;
  *out = nullptr;
  try {
    agz_engine* e = new agz_engine{nullptr, {}};
    e->impl = new agz::Engine(*cfg);
    *out = e;
    return AGZ_OK;
  } catch (const agz::Error& x) {
    g_create_error = x.what();
    return x.status;
  } catch (const std::exception& x) {
    g_create_error = x.what();
    return AGZ_BAD_ARGUMENT;
  }
}

void agz_engine_destroy(agz_engine* e) {
  if (!e) return;
  delete e->impl;
  delete e;
}

const char* agz_last_error(const agz_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

agz_status agz_engine_sync(agz_engine* e) { return guard(e, [&](agz::Engine& E) { E.sync(); }); }

// ---- network
agz_status agz_net_set_weights(agz_engine* e, int32_t layer, int32_t kind, const float* data, int64_t count) {
  return guard(e, [&](agz::Engine& E) { E.net().set(layer, kind, data, count); });
}
int64_t agz_net_param_count(const agz_engine* e, int32_t layer, int32_t kind) {
  if (!e || !e->impl) return -1;
  return e->impl->net().param_count(layer, kind);
}
agz_status agz_net_get_weights(agz_engine* e, int32_t layer, int32_t kind, float* out, int64_t count) {
  return guard(e, [&](agz::Engine& E) { E.net().get(layer, kind, out, count); });
}
agz_status agz_net_select(agz_engine* e, int32_t which) {
  return guard(e, [&](agz::Engine& E) { E.net_select(which); });
}
agz_status agz_arena_counts(agz_engine* e, int32_t* counts_out) {
  return guard(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(counts_out, AGZ_BAD_ARGUMENT, "null pointer");
    E.arena_counts(counts_out);
  });
}
agz_status agz_net_init_synthetic(agz_engine* e, uint64_t seed) {
  return guard(e, [&](agz::Engine& E) { E.net().init_synthetic(seed); });
}
agz_status agz_net_forward(agz_engine* e, const int8_t* boards, const int8_t* deltas, const int32_t* ndeltas,
                           const int8_t* to_play, int32_t B, float* pi_out, float* v_out) {
  return guard(e, [&](agz::Engine& E) { E.net_forward_positions(boards, deltas, ndeltas, to_play, B, pi_out, v_out); });
}
agz_status agz_net_forward_features(agz_engine* e, const float* feats, int32_t B, float* pi_out, float* v_out) {
  return guard(e, [&](agz::Engine& E) { E.net_forward_features(feats, B, pi_out, v_out); });
}
agz_status agz_features(agz_engine* e, const int8_t* boards, const int8_t* deltas, const int32_t* ndeltas,
                        const int8_t* to_play, int32_t B, float* out) {
  return guard(e, [&](agz::Engine& E) { E.features(boards, deltas, ndeltas, to_play, B, out); });
}
agz_status agz_net_time_forward(agz_engine* e, int32_t B, int32_t iters, float* ms_out) {
  return guard(e, [&](agz::Engine& E) { *ms_out = E.time_forward(B, iters); });
}
agz_status agz_net_time_conv(agz_engine* e, int32_t B, int32_t iters, float* ms_out) {
  return guard(e, [&](agz::Engine& E) { *ms_out = E.time_conv(B, iters); });
}

agz_status agz_net_set_winograd(agz_engine* e, int32_t on) {
  return guard(e, [&](agz::Engine& E) { AGZ_REQUIRE(on >= 0 && on <= 3, AGZ_BAD_ARGUMENT, "agz_net_set_winograd: 0, 1, 2 or 3"); E.net().set_winograd(on); });
}
agz_status agz_net_set_tower_persistent(agz_engine* e, int32_t on) {
  return guard(e, [&](agz::Engine& E) { E.net().set_tower_persistent(on != 0); });
}
agz_status agz_net_set_tower_streams(agz_engine* e, int32_t n) {
  return guard(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(n >= 1 && n <= 4, AGZ_BAD_ARGUMENT, "agz_net_set_tower_streams: 1..4");
    E.net().set_tower_streams(n);
  });
}
agz_status agz_net_set_precision(agz_engine* e, int32_t precision) {
  return guard(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(precision == AGZ_PRECISION_F32 || precision == AGZ_PRECISION_F16 || precision == AGZ_PRECISION_F32S,
                AGZ_BAD_ARGUMENT, "precision %d (0 = f32, 1 = f16 tower, 2 = f32 as split f16 operands)", precision);
    E.net().set_precision(precision);
  });
}
agz_status agz_profile_conv_enable(agz_engine* e, int32_t on) {
  return guard(e, [&](agz::Engine& E) { E.net().profile_enable(on != 0); });
}
agz_status agz_profile_conv_read(agz_engine* e, double* total_ms, double* total_flop, int64_t* launches) {
  return guard(e, [&](agz::Engine& E) { E.net().profile_read(total_ms, total_flop, launches); });
}
agz_status agz_profile_search_enable(agz_engine* e, int32_t on) {
  return guard(e, [&](agz::Engine& E) { E.profile_search_enable(on != 0); });
}
agz_status agz_profile_search_read(agz_engine* e, double* ms5, int64_t* steps) {
  return guard(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(ms5, AGZ_BAD_ARGUMENT, "ms5 is NULL");
    E.profile_search_read(ms5, steps);
  });
}

// ---- Go rules
agz_status agz_go_play(agz_engine* e, const int8_t* boards, const int8_t* to_play, const int32_t* ko,
                       const int32_t* moves, int32_t B, int8_t* boards_out, int32_t* ko_out,
                       int32_t* ncaptured_out, int32_t* status_out) {
  return guard(e, [&](agz::Engine& E) { E.go_play(boards, to_play, ko, moves, B, boards_out, ko_out, ncaptured_out, status_out); });
}
agz_status agz_go_legal(agz_engine* e, const int8_t* boards, const int8_t* to_play, const int32_t* ko, int32_t B,
                        int8_t* legal_out) {
  return guard(e, [&](agz::Engine& E) { E.go_legal(boards, to_play, ko, B, legal_out); });
}
agz_status agz_go_score(agz_engine* e, const int8_t* boards, const float* komi, int32_t B, float* score_out) {
  return guard(e, [&](agz::Engine& E) { E.go_score(boards, komi, B, score_out); });
}

// ---- batched self-play
agz_status agz_selfplay_start(agz_engine* e, int64_t total_games) {
  return guard(e, [&](agz::Engine& E) { E.start(total_games); });
}
agz_status agz_selfplay_step(agz_engine* e, int32_t nsteps) {
  return guard(e, [&](agz::Engine& E) { E.step(nsteps); });
}
agz_status agz_engine_stats(agz_engine* e, agz_stats* out) {
  return guard(e, [&](agz::Engine& E) { E.stats(out); });
}
agz_status agz_selfplay_select(agz_engine* e, int32_t* nleaves_out) {
  return guard(e, [&](agz::Engine& E) { *nleaves_out = E.select_external(); });
}
agz_status agz_selfplay_leaf_features(agz_engine* e, float* feats_out) {
  return guard(e, [&](agz::Engine& E) { E.leaf_features_external(feats_out); });
}
agz_status agz_selfplay_incorporate(agz_engine* e, const float* pi, const float* v) {
  return guard(e, [&](agz::Engine& E) { E.incorporate_external(pi, v); });
}

// ---- records
int64_t agz_records_count(agz_engine* e) {
  int64_t n = -1;
  guard(e, [&](agz::Engine& E) { n = E.records_count(); });
  return n;
}
agz_status agz_records_header(agz_engine* e, int64_t k, agz_game_header* out) {
  return guard(e, [&](agz::Engine& E) { E.record_header(k, out); });
}
agz_status agz_records_game(agz_engine* e, int64_t k, int16_t* moves, float* pis, float* qs) {
  return guard(e, [&](agz::Engine& E) { E.record_game(k, moves, pis, qs); });
}
agz_status agz_records_packed_size(agz_engine* e, int64_t* nbytes_out) {
  return guard(e, [&](agz::Engine& E) { *nbytes_out = E.records_packed_size(); });
}
agz_status agz_records_export_packed(agz_engine* e, void* dst, int64_t capacity, int32_t is_device) {
  return guard(e, [&](agz::Engine& E) { E.records_export_packed(dst, capacity, is_device != 0); });
}
agz_status agz_records_clear(agz_engine* e) { return guard(e, [&](agz::Engine& E) { E.records_clear(); }); }
agz_status agz_slot_status(agz_engine* e, int32_t* status_out, int32_t* nodes_out, int32_t* moves_out) {
  return guard(e, [&](agz::Engine& E) { E.slot_status(status_out, nodes_out, moves_out); });
}
agz_status agz_slot_abandon(agz_engine* e, int32_t slot) {
  return guard(e, [&](agz::Engine& E) { E.slot_abandon(slot); });
}
agz_status agz_records_features(agz_engine* e, int64_t k, float* out) {
  return guard(e, [&](agz::Engine& E) { E.record_features(k, out); });
}
agz_status agz_replay_ingest_packed(agz_engine* e, const void* packed, int64_t nbytes, int32_t is_device,
                                    int64_t* added_out) {
  return guard(e, [&](agz::Engine& E) {
    const int64_t n = E.replay_ingest(packed, nbytes, is_device != 0);
    if (added_out) *added_out = n;
  });
}
agz_status agz_replay_ingest_gathered(agz_engine* e, const void* buf, int32_t is_device, int32_t world,
                                      int64_t chunk_stride, const int64_t* counts, int64_t* added_out) {
  return guard(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(buf && counts && world >= 1 && chunk_stride >= 0 && chunk_stride % 8 == 0, AGZ_BAD_ARGUMENT, "bad gather buffer");
    std::vector<int64_t> coff((size_t)world), cbytes((size_t)world), cnrec((size_t)world);
    for (int r = 0; r < world; ++r) {
      AGZ_REQUIRE(counts[2 * r] >= 0 && counts[2 * r + 1] >= 0 && counts[2 * r + 1] <= chunk_stride && counts[2 * r + 1] % 8 == 0,
                  AGZ_BAD_ARGUMENT, "chunk %d: %lld records in %lld bytes", r, (long long)counts[2 * r], (long long)counts[2 * r + 1]);
      coff[r] = (int64_t)r * chunk_stride;
      cnrec[r] = counts[2 * r];
      cbytes[r] = counts[2 * r + 1];
    }
    const int64_t n = E.replay_ingest_gathered(buf, is_device != 0, (size_t)world * (size_t)chunk_stride, coff, cbytes, cnrec);
    if (added_out) *added_out = n;
  });
}
int64_t agz_replay_count(agz_engine* e) { return (e && e->impl) ? e->impl->replay_count() : -1; }
int64_t agz_replay_positions(agz_engine* e) { return (e && e->impl) ? e->impl->replay_positions() : -1; }
agz_status agz_replay_header(agz_engine* e, int64_t k, agz_game_header* out) {
  return guard(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(out != nullptr, AGZ_BAD_ARGUMENT, "null pointer");
    E.replay_header(k, out);
  });
}
agz_status agz_replay_game(agz_engine* e, int64_t k, int16_t* moves, float* pis, float* qs) {
  return guard(e, [&](agz::Engine& E) { E.replay_game(k, moves, pis, qs); });
}
agz_status agz_replay_trim(agz_engine* e, int64_t max_positions) {
  return guard(e, [&](agz::Engine& E) { E.replay_trim(max_positions); });
}
agz_status agz_replay_clear(agz_engine* e) { return guard(e, [&](agz::Engine& E) { E.replay_clear(); }); }
agz_status agz_replay_batch(agz_engine* e, const int64_t* game, const int32_t* ply, int32_t B, float* feats,
                            float* pi, float* z, int32_t out_is_device) {
  return guard(e, [&](agz::Engine& E) { E.replay_batch(game, ply, B, feats, pi, z, out_is_device != 0); });
}

agz_status agz_train_step(agz_engine* e, const float* feats, const float* pi, const float* z, int32_t B,
                          int32_t inputs_are_device, float eta, float rho, float* losses_out) {
  return guard(e, [&](agz::Engine& E) { E.train_step(feats, pi, z, B, inputs_are_device != 0, eta, rho, losses_out); });
}
agz_status agz_train_reset(agz_engine* e) { return guard(e, [&](agz::Engine& E) { E.train_reset(); }); }

// ---- RCCL exchange (agz_comm.hip)
agz_status agz_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return AGZ_BAD_ARGUMENT;
  try {
    agz::comm_unique_id(id_out);
    return AGZ_OK;
  } catch (const agz::Error& x) {
    g_create_error = x.what();
    return x.status;
  } catch (const std::exception& x) {
    g_create_error = x.what();
    return AGZ_RCCL_ERROR;
  }
}
agz_status agz_comm_create(agz_engine* e, int32_t rank, int32_t world, const uint8_t* id, agz_comm** out) {
  if (!out) return AGZ_BAD_ARGUMENT;
  *out = nullptr;
  return guard(e, [&](agz::Engine& E) { *out = reinterpret_cast<agz_comm*>(agz::comm_create(E, rank, world, id)); });
}
void agz_comm_destroy(agz_comm* c) { agz::comm_destroy(reinterpret_cast<agz::Comm*>(c)); }
agz_status agz_allgather_records(agz_engine* e, agz_comm* comm, int64_t* added_out) {
  return guard(e, [&](agz::Engine& E) {
    const int64_t n = agz::comm_allgather_records(E, reinterpret_cast<agz::Comm*>(comm));
    if (added_out) *added_out = n;
  });
}
agz_status agz_gather_plan(const int64_t* counts, int32_t world, int64_t* chunk_stride_out, int64_t* total_records_out) {
  if (!chunk_stride_out) return AGZ_BAD_ARGUMENT;
  try {
    *chunk_stride_out = agz::gather_plan(counts, world, total_records_out);
    return AGZ_OK;
  } catch (const agz::Error& x) {
    g_create_error = x.what();
    return x.status;
  }
}
agz_status agz_broadcast_weights(agz_engine* e, agz_comm* comm, int32_t root, int64_t* nfloats_out) {
  return guard(e, [&](agz::Engine& E) {
    const int64_t n = agz::comm_broadcast_weights(E, reinterpret_cast<agz::Comm*>(comm), root);
    if (nfloats_out) *nfloats_out = n;
  });
}

// ---- ABI self-description
int32_t agz_abi_layout(const char* name, int32_t* out, int32_t cap) {
  if (!name || !out) return -1;
  std::vector<int32_t> v;
#define AGZ_SZ(T) v.push_back((int32_t)sizeof(T))
#define AGZ_OFF(T, f) v.push_back((int32_t)offsetof(T, f))
  const std::string n(name);
  if (n == "agz_config") {
    AGZ_SZ(agz_config);
    AGZ_OFF(agz_config, board_size); AGZ_OFF(agz_config, tower_height); AGZ_OFF(agz_config, games);
    AGZ_OFF(agz_config, num_readouts); AGZ_OFF(agz_config, parallel_readouts); AGZ_OFF(agz_config, two_player_mode);
    AGZ_OFF(agz_config, komi); AGZ_OFF(agz_config, reserved0); AGZ_OFF(agz_config, c_puct);
    AGZ_OFF(agz_config, dirichlet_noise_weight); AGZ_OFF(agz_config, resign_threshold);
    AGZ_OFF(agz_config, resign_disable_fraction); AGZ_OFF(agz_config, seed); AGZ_OFF(agz_config, game_id_base);
    AGZ_OFF(agz_config, game_id_stride); AGZ_OFF(agz_config, max_nodes_per_game); AGZ_OFF(agz_config, device);
    AGZ_OFF(agz_config, external_network); AGZ_OFF(agz_config, pool_policy);
    AGZ_OFF(agz_config, record_capacity_games); AGZ_OFF(agz_config, arena_mode);
  } else if (n == "agz_stats") {
    AGZ_SZ(agz_stats);
    AGZ_OFF(agz_stats, steps); AGZ_OFF(agz_stats, positions); AGZ_OFF(agz_stats, games_started);
    AGZ_OFF(agz_stats, games_finished); AGZ_OFF(agz_stats, evals); AGZ_OFF(agz_stats, duplicate_evals);
    AGZ_OFF(agz_stats, terminal_visits); AGZ_OFF(agz_stats, root_visits); AGZ_OFF(agz_stats, nodes_in_use);
    AGZ_OFF(agz_stats, pool_exhausted); AGZ_OFF(agz_stats, resigned_games); AGZ_OFF(agz_stats, live_games);
    AGZ_OFF(agz_stats, records_dropped); AGZ_OFF(agz_stats, pool_short_searches);
    AGZ_OFF(agz_stats, peak_nodes_per_game); AGZ_OFF(agz_stats, stalled_games); AGZ_OFF(agz_stats, node_capacity);
    AGZ_OFF(agz_stats, abandoned_games);
  } else if (n == "agz_game_header") {
    AGZ_SZ(agz_game_header);
    AGZ_OFF(agz_game_header, game_id); AGZ_OFF(agz_game_header, num_moves); AGZ_OFF(agz_game_header, result);
    AGZ_OFF(agz_game_header, was_resign); AGZ_OFF(agz_game_header, resign_disabled);
    AGZ_OFF(agz_game_header, final_score); AGZ_OFF(agz_game_header, short_searches);
  } else if (n == "agz_position_info") {
    AGZ_SZ(agz_position_info);
    AGZ_OFF(agz_position_info, n); AGZ_OFF(agz_position_info, to_play); AGZ_OFF(agz_position_info, ko);
    AGZ_OFF(agz_position_info, caps_black); AGZ_OFF(agz_position_info, caps_white);
    AGZ_OFF(agz_position_info, last_move); AGZ_OFF(agz_position_info, prev_move);
    AGZ_OFF(agz_position_info, history_len); AGZ_OFF(agz_position_info, komi);
  } else if (n == "agz_node_info") {
    AGZ_SZ(agz_node_info);
    AGZ_OFF(agz_node_info, N); AGZ_OFF(agz_node_info, W); AGZ_OFF(agz_node_info, Q); AGZ_OFF(agz_node_info, parent);
    AGZ_OFF(agz_node_info, fmove); AGZ_OFF(agz_node_info, is_expanded); AGZ_OFF(agz_node_info, losses_applied);
    AGZ_OFF(agz_node_info, done); AGZ_OFF(agz_node_info, pos);
  } else {
    return -1;
  }
#undef AGZ_SZ
#undef AGZ_OFF
  if ((int32_t)v.size() > cap) return -1;
  for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
  return (int32_t)v.size() - 1;
}

agz_status agz_replay_features(agz_engine* e, const int16_t* moves, int64_t nmoves, const int32_t* game_offset,
                               const int32_t* ply, int32_t B, float* out, int32_t out_is_device) {
  return guard(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(B == 0 || (game_offset && ply && out && (moves || nmoves == 0)), AGZ_BAD_ARGUMENT, "null pointer");
    E.replay_batch_features(moves, nmoves, game_offset, ply, B, out, out_is_device != 0);
  });
}

// ---- single-tree compat
agz_status agz_tree_init(agz_engine* e, int32_t g, const int8_t* board, const agz_position_info* info,
                         const int8_t* history) {
  return guard_status(e, [&](agz::Engine& E) {
    AGZ_REQUIRE(board && info, AGZ_BAD_ARGUMENT, "agz_tree_init: null board/info");
    agz::TreeArgs T = targs(agz::TOP_INIT, g);
    T.info = *info;
    T.board = board;
    T.history = history;
    return E.tree_op(T, nullptr);
  });
}
agz_status agz_tree_root(agz_engine* e, int32_t g, int32_t* node_out) {
  return guard(e, [&](agz::Engine& E) {
    agz::GameState s;
    E.game_state(g, &s);
    *node_out = s.root;
  });
}
agz_status agz_tree_select_leaf(agz_engine* e, int32_t g, int32_t from_node, int32_t* leaf_out) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_SELECT, g, from_node);
    return E.tree_op(T, leaf_out);
  });
}
agz_status agz_tree_maybe_add_child(agz_engine* e, int32_t g, int32_t node, int32_t a, int32_t* child_out) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_ADD_CHILD, g, node, a);
    return E.tree_op(T, child_out);
  });
}
agz_status agz_tree_add_virtual_loss(agz_engine* e, int32_t g, int32_t node, int32_t up_to) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_VLOSS_ADD, g, node, 0, up_to);
    return E.tree_op(T, nullptr);
  });
}
agz_status agz_tree_revert_virtual_loss(agz_engine* e, int32_t g, int32_t node, int32_t up_to) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_VLOSS_REVERT, g, node, 0, up_to);
    return E.tree_op(T, nullptr);
  });
}
agz_status agz_tree_incorporate(agz_engine* e, int32_t g, int32_t node, const float* probs, int32_t nprobs,
                                float value, int32_t up_to) {
  return guard_status(e, [&](agz::Engine& E) {
    if (nprobs != E.view().A) return (int)AGZ_BAD_SHAPE;   // @assert size(move_probs) == (A,), mcts.jl:190
    agz::TreeArgs T = targs(agz::TOP_INCORPORATE, g, node, 0, up_to);
    T.probs = probs;
    T.value = value;
    return E.tree_op(T, nullptr);
  });
}
agz_status agz_tree_inject_noise(agz_engine* e, int32_t g, int32_t node) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_NOISE, g, node);
    return E.tree_op(T, nullptr);
  });
}
agz_status agz_tree_search_select(agz_engine* e, int32_t g, int32_t par, int32_t* nleaves_out) {
  return guard_status(e, [&](agz::Engine& E) { return E.tree_search_select(g, par, nleaves_out); });
}
agz_status agz_tree_leaf_features(agz_engine* e, int32_t g, float* feats_out) {
  return guard(e, [&](agz::Engine& E) { E.tree_leaf_features(g, feats_out); });
}
agz_status agz_tree_leaf_positions(agz_engine* e, int32_t g, int32_t* nodes_out, int8_t* boards_out, int8_t* deltas_out,
                                   int32_t* ndeltas_out, int8_t* to_play_out, agz_position_info* info_out) {
  return guard(e, [&](agz::Engine& E) {
    E.tree_leaf_positions(g, nodes_out, boards_out, deltas_out, ndeltas_out, to_play_out, info_out);
  });
}
agz_status agz_tree_search_incorporate(agz_engine* e, int32_t g, const float* pi, const float* v) {
  return guard_status(e, [&](agz::Engine& E) { return E.tree_search_incorporate(g, pi, v); });
}
agz_status agz_tree_search(agz_engine* e, int32_t g, int32_t par, int32_t* nleaves_out) {
  return guard_status(e, [&](agz::Engine& E) {
    const int st = E.tree_search_select(g, par, nleaves_out);
    if (st != AGZ_OK) return st;
    return E.tree_search_incorporate(g, nullptr, nullptr);
  });
}
agz_status agz_tree_pick_move(agz_engine* e, int32_t g, int32_t* a_out) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_PICK, g);
    return E.tree_op(T, a_out);
  });
}
agz_status agz_tree_play_move(agz_engine* e, int32_t g, int32_t a, int32_t* ok_out) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_PLAY, g, 0, a);
    return E.tree_op(T, ok_out);
  });
}
agz_status agz_tree_should_resign(agz_engine* e, int32_t g, int32_t* out) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_RESIGN, g);
    return E.tree_op(T, out);
  });
}
agz_status agz_tree_is_done(agz_engine* e, int32_t g, int32_t node, int32_t* out) {
  return guard(e, [&](agz::Engine& E) {
    agz::NodeMeta m;
    E.node_meta(g, node, &m);
    *out = (m.flags & agz::NF_DONE) || m.n >= E.view().max_game_length;   // mcts.jl:230-231
  });
}
agz_status agz_tree_node_info(agz_engine* e, int32_t g, int32_t node, agz_node_info* out) {
  return guard(e, [&](agz::Engine& E) {
    agz::NodeMeta m;
    agz::GameState s;
    E.node_meta(g, node, &m);
    E.game_state(g, &s);
    std::memset(out, 0, sizeof(*out));
    out->N = E.node_stat(g, node, 0);
    out->W = E.node_stat(g, node, 1);
    out->Q = out->W / (1.0f + out->N);
    out->parent = m.parent;
    out->fmove = m.fmove;
    out->is_expanded = (m.flags & agz::NF_EXPANDED) != 0;
    out->losses_applied = m.losses;
    out->done = (m.flags & agz::NF_DONE) != 0;
    out->pos.n = m.n;
    out->pos.to_play = m.to_play;
    out->pos.ko = m.ko;
    out->pos.caps_black = m.caps_b;
    out->pos.caps_white = m.caps_w;
    out->pos.last_move = m.last_move;
    out->pos.prev_move = -1;
    out->pos.history_len = s.hist_len;
    out->pos.komi = s.komi;
  });
}
agz_status agz_tree_node_floats(agz_engine* e, int32_t g, int32_t node, int32_t field, float* out) {
  return guard(e, [&](agz::Engine& E) { E.node_row_get(g, node, field, out); });
}
agz_status agz_tree_node_scores(agz_engine* e, int32_t g, int32_t node, double* out) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_SCORES, g, node);
    T.dout = out;
    return E.tree_op(T, nullptr);
  });
}
agz_status agz_tree_node_set_floats(agz_engine* e, int32_t g, int32_t node, int32_t field, const float* in) {
  return guard(e, [&](agz::Engine& E) { E.node_row_set(g, node, field, in); });
}
agz_status agz_tree_node_set_N(agz_engine* e, int32_t g, int32_t node, float value) {
  return guard(e, [&](agz::Engine& E) { E.node_set_N(g, node, value); });
}
agz_status agz_tree_node_set_n(agz_engine* e, int32_t g, int32_t node, int32_t n) {
  return guard(e, [&](agz::Engine& E) {
    agz::NodeMeta m;
    E.node_meta(g, node, &m);
    m.n = n;
    E.node_meta_set(g, node, m);
  });
}
agz_status agz_tree_node_children(agz_engine* e, int32_t g, int32_t node, int32_t* out) {
  return guard(e, [&](agz::Engine& E) { E.node_children(g, node, out); });
}
agz_status agz_tree_node_board(agz_engine* e, int32_t g, int32_t node, int8_t* out) {
  return guard(e, [&](agz::Engine& E) { E.node_board(g, node, out); });
}
agz_status agz_tree_pending_vlosses(agz_engine* e, int32_t g, int32_t* out) {
  return guard_status(e, [&](agz::Engine& E) {
    agz::TreeArgs T = targs(agz::TOP_PENDING, g);
    return E.tree_op(T, out);
  });
}
agz_status agz_tree_set_draw(agz_engine* e, int32_t g, uint64_t game_id, uint32_t sel) {
  return guard(e, [&](agz::Engine& E) {
    agz::GameState s;
    E.game_state(g, &s);
    s.game_id = game_id;
    s.sel = (int32_t)sel;
    E.game_patch(g, s);
  });
}

// ---- diagnostics
agz_status agz_debug_draws(agz_engine* e, uint64_t seed, uint64_t game, uint32_t move, int32_t n, double alpha,
                           double* gamma_out) {
  return guard(e, [&](agz::Engine& E) { E.debug_draws(seed, game, move, n, alpha, gamma_out); });
}
agz_status agz_debug_math(agz_engine* e, int32_t op, const double* x, const double* y, int32_t n, double* out) {
  return guard(e, [&](agz::Engine& E) { E.debug_math(op, x, y, n, out); });
}
int32_t agz_debug_counters(agz_engine* e, uint64_t* out, int32_t cap) {
  int32_t n = -1;
  (void)guard(e, [&](agz::Engine& E) { n = E.debug_counters(out, cap); });
  return n;
}
agz_status agz_debug_live_record(agz_engine* e, int32_t g, int32_t k, uint64_t* game_id_out, int32_t* num_moves_out,
                                 int32_t* move_out, float* pi_out, float* q_out) {
  return guard(e, [&](agz::Engine& E) { E.debug_live_record(g, k, game_id_out, num_moves_out, move_out, pi_out, q_out); });
}
agz_status agz_debug_set_stagger(agz_engine* e, int32_t moves) {
  return guard(e, [&](agz::Engine& E) { E.debug_set_stagger(moves); });
}
agz_status agz_debug_mfma_sustained(agz_engine* e, int32_t millis, float* tflops_out) {
  return guard(e, [&](agz::Engine& E) { *tflops_out = E.net().mfma_sustained_tflops(millis); });
}
agz_status agz_debug_mfma_sustained_data(agz_engine* e, int32_t millis, int32_t mode, float* tflops_out) {
  return guard(e, [&](agz::Engine& E) { *tflops_out = E.net().mfma_sustained_tflops(millis, mode); });
}
agz_status agz_debug_pack_diff(agz_engine* e, int32_t which, int64_t* mismatches_out) {
  return guard(e, [&](agz::Engine& E) { *mismatches_out = (int64_t)E.net().debug_pack_diff(which); });
}

}  // extern "C"
