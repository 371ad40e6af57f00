// agz_engine.h -- host-side engine object behind the C ABI (include/agz.h).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "agz_common.h"
#include "agz_layout.h"
#include "agz_nn.h"
#include "agz_state.h"

namespace agz {

struct TreeArgs;

class Engine {
 public:
  explicit Engine(const agz_config& cfg);
  ~Engine();

  const agz_config& config() const { return cfg_; }
  const View& view() const { return V_; }
  Net& net() { return (net_sel_ && net2_) ? *net2_ : *net_; }   // the network agz_net_* calls address
  void net_select(int which);
  void arena_counts(int32_t* out) const;
  hipStream_t stream() const { return stream_; }
  void sync();

  // batched self-play
  void start(int64_t total_games);
  void step(int nsteps);
  void stats(agz_stats* out);
  int debug_counters(uint64_t* out, int cap);
  void debug_live_record(int g, int k, uint64_t* game_id, int32_t* num_moves, int32_t* move, float* pi, float* q);
  void debug_set_stagger(int moves);
  int select_external();
  void leaf_features_external(float* feats_out);
  void incorporate_external(const float* pi, const float* v);

  // records
  int64_t records_count();
  void record_header(int64_t k, agz_game_header* out);
  void record_game(int64_t k, int16_t* moves, float* pis, float* qs);
  int64_t records_packed_size(int64_t first = 0);
  void records_export_packed(void* dst, int64_t capacity, bool is_device);
  void records_clear();
  int64_t pack_records_device(uint8_t* dst, int64_t capacity, int64_t* nbytes, int64_t first = 0);
  // records [0, records_exchanged()) have been filed by agz_allgather_records since the last agz_records_clear
  int64_t records_exchanged() const { return rec_sent_; }
  void records_mark_exchanged(int64_t upto) { rec_sent_ = upto; }
  void record_features(int64_t k, float* out);

  // device replay arena: finished games of every rank, packed, resident in HBM
  int64_t replay_ingest(const void* packed, int64_t nbytes, bool is_device);
  int64_t replay_ingest_local();
  int64_t replay_ingest_gathered(const void* buf, bool is_device, size_t nbytes, const std::vector<int64_t>& coff,
                                 const std::vector<int64_t>& cbytes, const std::vector<int64_t>& cnrec);
  int64_t replay_ingest_chunks(const uint8_t* dbuf, const std::vector<int64_t>& coff,
                               const std::vector<int64_t>& cbytes, const std::vector<int64_t>& cnrec);
  int64_t replay_count() const { return (int64_t)rp_hdr_.size(); }
  int64_t replay_positions() const { return rp_positions_; }
  int64_t replay_bytes() const { return (int64_t)rp_used_; }
  void replay_header(int64_t k, agz_game_header* out) const;
  void replay_game(int64_t k, int16_t* moves, float* pis, float* qs);
  void replay_trim(int64_t max_positions);
  void replay_clear();
  void replay_batch(const int64_t* game, const int32_t* ply, int B, float* feats, float* pi, float* z,
                    bool out_is_device);
  DevBuf<uint8_t>& pack_scratch() { return s_pack_; }
  // one optimisation step on the selected network (agz_train.hip)
  void train_step(const float* feats, const float* pi, const float* z, int B, bool is_device, float eta, float rho,
                  float* losses_out);
  void train_reset();
  // flat parameter vector of the selected network in layers() order (broadcast_weights)
  std::vector<float> weights_flat();
  void weights_set_flat(const std::vector<float>& w);
  void replay_batch_features(const int16_t* moves, int64_t nmoves, const int32_t* off, const int32_t* ply, int B,
                             float* out, bool out_is_device);

  // network
  void net_forward_positions(const int8_t* boards, const int8_t* deltas, const int32_t* ndeltas,
                             const int8_t* to_play, int B, float* pi_out, float* v_out);
  void net_forward_features(const float* feats, int B, float* pi_out, float* v_out);
  void features(const int8_t* boards, const int8_t* deltas, const int32_t* ndeltas, const int8_t* to_play,
                int B, float* out);
  float time_forward(int B, int iters);
  // HIP events around the five search kernels of the next steps (bench.py's `search_kernels`)
  void profile_search_enable(bool on);
  void profile_search_read(double* ms5, int64_t* steps);
  float time_conv(int B, int iters);
  void slot_status(int32_t* status, int32_t* nodes, int32_t* moves);
  void slot_abandon(int g);

  void debug_draws(uint64_t seed, uint64_t game, uint32_t move, int n, double alpha, double* out);
  void debug_math(int op, const double* x, const double* y, int n, double* out);

  // Go rules
  void go_play(const int8_t* boards, const int8_t* to_play, const int32_t* ko, const int32_t* moves, int B,
               int8_t* boards_out, int32_t* ko_out, int32_t* ncap_out, int32_t* status_out);
  void go_legal(const int8_t* boards, const int8_t* to_play, const int32_t* ko, int B, int8_t* out);
  void go_score(const int8_t* boards, const float* komi, int B, float* out);

  // single-tree compat
  int tree_op(TreeArgs& T, int32_t* r0);           // returns the op's agz_status
  int tree_search_select(int g, int par, int* nleaves);
  void tree_leaf_features(int g, float* feats_out);
  void tree_leaf_positions(int g, int32_t* nodes, int8_t* boards, int8_t* deltas, int32_t* ndeltas, int8_t* to_play,
                           agz_position_info* info);
  int tree_search_incorporate(int g, const float* pi, const float* v);   // pi == NULL: engine's own network
  void game_state(int g, GameState* out);
  void game_patch(int g, const GameState& s);
  void node_meta(int g, int node, NodeMeta* out);
  void node_meta_set(int g, int node, const NodeMeta& m);
  void node_row_get(int g, int node, int field, float* out);
  void node_row_set(int g, int node, int field, const float* in);
  void node_children(int g, int node, int32_t* out);
  void node_board(int g, int node, int8_t* out);
  float node_stat(int g, int node, int which);     // 0 = N, 1 = W
  void node_set_N(int g, int node, float v);

  std::string last_error;

 private:
  void replay_reserve(size_t bytes);
  void upload_view_outputs();
  void fill_synthetic_inputs(int B);
  void check_game(int g) const;
  void check_node(int g, int node) const;

  agz_config cfg_;
  View V_{};
  hipStream_t stream_ = nullptr;
  std::unique_ptr<Net> net_, net2_;   // net2_: White's network in arena mode
  std::unique_ptr<Trainer> trainer_, trainer2_;
  int net_sel_ = 0;
  static constexpr int kSearchProfMax = 512;      // steps whose search kernels are timed after profile_search_enable(true)
  bool sprof_on_ = false;
  int sprof_n_ = 0;
  std::vector<hipEvent_t> sprof_ev_;              // [step][7]: before k_pre, behind k_pre / k_expand / k_scan / k_leaf_features, in front of / behind k_post
  int external_batch2_ = 0;
  std::vector<void*> bufs_;
  size_t state_bytes_ = 0;
  int bcap_ = 0;
  DevBuf<float> d_x32_, d_pi_, d_v_, d_whcn_;
  DevBuf<int32_t> d_count_;
  // staging for ABI calls
  DevBuf<int8_t> s_boards_, s_deltas_, s_tp_, s_boards_out_, s_legal_, s_leafb_;
  DevBuf<uint8_t> s_leafrows_;
  DevBuf<int32_t> s_i32a_, s_i32b_, s_i32c_, s_i32d_;
  DevBuf<float> s_f32a_, s_f32b_;
  DevBuf<int16_t> s_i16a_;
  DevBuf<int64_t> s_i64a_;
  DevBuf<double> s_f64_;
  DevBuf<int32_t> s_iout_;
  int external_batch_ = 0;
  int tree_batch_ = 0;
  int64_t abandoned_ = 0;       // games dropped by slot_abandon since the last selfplay_start
  // replay arena
  DevBuf<uint8_t> rp_buf_, s_pack_;
  size_t rp_used_ = 0;
  int64_t rp_positions_ = 0;
  std::vector<int64_t> rp_off_;
  std::vector<agz_game_header> rp_hdr_;
  int64_t rec_sent_ = 0;
  bool stepped_ = false;           // a step has run since the last start(): agz_debug_set_stagger is refused
};

// RCCL exchange (agz_comm.hip)
struct Comm;
void comm_unique_id(uint8_t* out);
Comm* comm_create(Engine& E, int rank, int world, const uint8_t* id);
void comm_destroy(Comm* c);
int64_t comm_allgather_records(Engine& E, Comm* c);
// host logic of the exchange between its two collectives (also the C ABI's agz_gather_plan): checks the gathered
// {records, bytes} pairs, returns the padded per-rank chunk size; throws AGZ_RCCL_ERROR naming the offending rank
int64_t gather_plan(const int64_t* counts, int world, int64_t* total_records);
int64_t comm_broadcast_weights(Engine& E, Comm* c, int root);

}  // namespace agz
