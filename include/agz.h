/*
 * agz.h -- C ABI of libagz.so, the MI355X-native self-play engine for AlphaGo.jl's hot path.
 *
 * The reference has no FFI of its own: its boundary is the Julia call surface that
 * train()/evaluate()/play() and the test-suite use.  Each entry point below
 * names the reference interface it stands in for (file:line under /root/reference); the thin
 * Julia `ccall` wrapper a maintainer would add is alphago.jl_amd/julia/AlphaGoMI.jl and the
 * binding recipe is INTEGRATION.md.  Everything is extern "C", plain pointers and sizes.
 *
 * Conventions
 *   - all indices are 0-based: board point p = row + N*col (Julia's column-major linear index
 *     minus one, src/game/go/coords.jl:6-7); action a in [0, A), A = N*N + 1, a == N*N = pass.
 *   - colours: BLACK = +1, WHITE = -1, EMPTY = 0 (src/game/go/board.jl:12).
 *   - tensors cross in the layouts Flux stores them: conv [kw,kh,cin,cout] column-major,
 *     dense [out,in] column-major, features N x N x 17 x B (WHCN), pi A x B, v B.
 *   - every function returns an agz_status (0 = OK); agz_last_error() describes the last
 *     failure of that engine.  Pointers are HOST pointers unless the name says _device.
 *   - an engine handle is bound to one HIP device and is not thread-safe (the reference is
 *     single-threaded with mutable module globals, src/mcts.jl:11-13).
 *   - the engine fails loudly (AGZ_HIP_ERROR) when no gfx950 device is present; there is no
 *     CPU fallback anywhere behind this ABI.
 */
#ifndef AGZ_H
#define AGZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGZ_VERSION 103

typedef int32_t agz_status;
#define AGZ_OK 0
#define AGZ_ILLEGAL_MOVE 1        /* IllegalMove   src/AlphaGo.jl:8, board.jl:265,470        */
#define AGZ_ASSERT_DONE_NODE 2    /* AssertionError src/mcts.jl:196                          */
#define AGZ_HISTORY_INCOMPLETE 3  /* AssertionError board.jl:568, mcts_play.jl:127           */
#define AGZ_BAD_SHAPE 4           /* AssertionError src/mcts.jl:190                          */
#define AGZ_ASSERT_SOFTPICK 5     /* AssertionError src/mcts_play.jl:67                      */
#define AGZ_BAD_ARGUMENT 6
#define AGZ_HIP_ERROR 7
#define AGZ_POOL_EXHAUSTED 8      /* a game's node pool overflowed (ours; no reference analogue) */
#define AGZ_RCCL_ERROR 9
#define AGZ_NOT_READY 10

#define AGZ_POOL_MOVE_EARLY 0
#define AGZ_POOL_STALL 1

typedef struct agz_engine agz_engine;

/* One POD for every knob of the hot path (SURVEY.md section 5 "Config / flags"):
 * GoEnv(board_size) go.jl:10; NeuralNet(env; tower_height) neural_net.jl:13;
 * MCTSPlayer(env, net; num_readouts, two_player_mode, resign_threshold) mcts_play.jl:17-18;
 * tree_search!(player, parallel_readouts) mcts_play.jl:73; komi board.jl:297;
 * c_puct / dirichlet_noise_weight mcts.jl:11-13. */
typedef struct {
  int32_t board_size;              /* N; 19 */
  int32_t tower_height;            /* residual blocks; 19 */
  int32_t games;                   /* concurrent game slots on this GPU */
  int32_t num_readouts;            /* 800 */
  int32_t parallel_readouts;       /* 8 */
  int32_t two_player_mode;         /* 0 */
  float komi;                      /* 7.5 */
  float reserved0;
  double c_puct;                   /* 0.96 */
  double dirichlet_noise_weight;   /* 0.25 */
  double resign_threshold;         /* -0.9 */
  double resign_disable_fraction;  /* 0.05, selfplay.jl:9 */
  uint64_t seed;                   /* draw-stream seed (include/agz_draws.h) */
  uint64_t game_id_base;           /* first global game id played by this engine */
  uint64_t game_id_stride;         /* id increment when a slot is recycled (= total slots) */
  int32_t max_nodes_per_game;      /* 0 = auto: 16*num_readouts + 256 + 16*max_game_length (see pool_policy) */
  int32_t device;                  /* HIP device ordinal */
  int32_t external_network;        /* 1: pi/v are supplied by the caller (duck-typed network) */
  int32_t pool_policy;             /* what a game does when its node pool is full (the reference's tree is garbage-
                                    * collected and unbounded, mcts.jl:140-147): AGZ_POOL_MOVE_EARLY (0, default) ends
                                    * the search of the current move there and plays it from the visits it has -- counted
                                    * in agz_stats.pool_short_searches and in the game's header; AGZ_POOL_STALL (1) never
                                    * shortens a search: the slot waits (agz_slot_status) until the host abandons it
                                    * (agz_slot_abandon).  Other slots keep stepping either way.  (This word was
                                    * reserved1 = 0 until round 4, and a bench-only knob before that.) */
  int32_t record_capacity_games;   /* finished-game record slots kept on the device; 0 = auto */
  int32_t arena_mode;              /* 1: evaluate() arena -- slots 2i / 2i+1 are the Black / White player of one
                                    * game with networks 0 / 1 (agz_net_select); `games` must be even */
} agz_config;

int32_t agz_version(void);
void agz_config_default(agz_config* cfg);
agz_status agz_engine_create(const agz_config* cfg, agz_engine** out);
void agz_engine_destroy(agz_engine* e);
const char* agz_last_error(const agz_engine* e);   /* e may be NULL: last create() failure */
agz_status agz_engine_sync(agz_engine* e);

/* ---------------------------------------------------------------- network ------------- */
/* NeuralNet(env; tower_height), neural_net.jl:13-33.  layer ids: 0 = stem conv+BN;
 * 1..2*tower = tower convs (block b, conv c -> 1 + 2b + c); negative = heads. */
#define AGZ_L_VALUE_CONV (-1)
#define AGZ_L_POLICY_CONV (-2)
#define AGZ_L_VALUE_FC1 (-3)
#define AGZ_L_VALUE_FC2 (-4)
#define AGZ_L_POLICY_FC (-5)
#define AGZ_K_WEIGHT 0
#define AGZ_K_BIAS 1
#define AGZ_K_BN_BETA 2
#define AGZ_K_BN_GAMMA 3
#define AGZ_K_BN_MEAN 4
#define AGZ_K_BN_VAR 5
#define AGZ_K_BN_EPS 6
/* copies `count` floats (caller keeps ownership); Flux layouts, kernel flip done inside */
agz_status agz_net_set_weights(agz_engine* e, int32_t layer, int32_t kind, const float* data,
                               int64_t count);
int64_t agz_net_param_count(const agz_engine* e, int32_t layer, int32_t kind);
/* read a parameter back in the layout it was set in (save_model, train.jl:14-35) */
agz_status agz_net_get_weights(agz_engine* e, int32_t layer, int32_t kind, float* out, int64_t count);
/* evaluate(env, black_net, white_net) neural_net.jl:103-158 needs two networks in one engine
 * (arena_mode): `which` = 0 (Black's, the default) or 1 (White's) selects the network that the
 * agz_net_set_weights / get_weights / init_synthetic / forward* calls after it address. */
agz_status agz_net_select(agz_engine* e, int32_t which);
/* Flux-default-equivalent init from the draw stream (glorot-uniform, zero bias, BN identity,
 * eps 1e-5) -- the synthetic weights of SURVEY.md 8d */
agz_status agz_net_init_synthetic(agz_engine* e, uint64_t seed);
/* (nn)(positions::Vector{Position}) -> (pi A x B, v B), neural_net.jl:57-68.
 * Position SoA: boards int8[B][N*N]; deltas int8[B][7][N*N] newest first; ndeltas int32[B];
 * to_play int8[B]. */
agz_status agz_net_forward(agz_engine* e, const int8_t* boards, const int8_t* deltas,
                           const int32_t* ndeltas, const int8_t* to_play, int32_t B,
                           float* pi_out, float* v_out);
/* same on a feature tensor N x N x 17 x B already in host memory */
agz_status agz_net_forward_features(agz_engine* e, const float* feats, int32_t B, float* pi_out,
                                    float* v_out);
/* get_feats(pos) -> N x N x 17 (x B), features.jl:3-26 */
agz_status agz_features(agz_engine* e, const int8_t* boards, const int8_t* deltas,
                        const int32_t* ndeltas, const int8_t* to_play, int32_t B, float* out);
/* micro-benchmark hook: run the network `iters` times on B resident synthetic positions and
 * return the average milliseconds per forward (HIP events on the engine's stream) */
agz_status agz_net_time_forward(agz_engine* e, int32_t B, int32_t iters, float* ms_out);
/* average duration (ms) of the dominant 3x3 256->256 conv launch over the same kind of run */
agz_status agz_net_time_conv(agz_engine* e, int32_t B, int32_t iters, float* ms_out);

/* tower-convolution algorithm: 1 (default) = Winograd on the f32 MFMA -- F(3x3,3x3), and for boards of 13x13 and
 * larger in the exact-f32 arithmetic F(4x4,3x3) (a quarter fewer multiplies at 19x19); 2 = Winograd F(3x3,3x3) on every
 * board size (comparison runs); 0 = direct implicit GEMM on the f32 MFMA; 3 = 1 with the tower layers of boards whose tile
 * blocks hold whole boards (N <= 12) on the five-pass 64-tile x 128-cout form of F(3x3,3x3) (agz_wino5.hip: 25 % fewer
 * operand bytes per flop, the same layer time within 0.5 % on the 9x9 headline -- opt-in, tests/test_gpu_wino5.py).  All
 * are f32 end to end; they differ by rounding only (each within 1e-4 of the float64 network: tests/test_gpu_nn.py,
 * tests/test_gpu_configs.py). */
agz_status agz_net_set_winograd(agz_engine* e, int32_t on);
/* the f32 Winograd tower as one launch per layer (0, default) or as ONE persistent launch over all its layers (1: used
 * wherever it applies -- board sizes whose tile blocks hold whole boards (N <= 12), a 256-CU device).  The same device
 * function does the work either way: outputs are bit-identical (tests/test_gpu_tower.py).  The persistent form needs
 * 3.7 % fewer cycles (no partly filled last workgroup round per layer) and, on a power-limited MI355X, runs at a
 * clock 4 % lower: the same wall time (HISTORY.md 4f). */
agz_status agz_net_set_tower_persistent(agz_engine* e, int32_t on);
/* the Winograd tower of a large batch as n = 1..4 independent layer chains (default 2): ranges of the batch's tile blocks, cut
 * at board boundaries, run their layers on n HIP streams and the hardware interleaves their workgroups (-5 % per forward
 * at 19x19 / 2048 positions, -1.6 % per step at 9x9 / 8192: the CUs stop moving through K loops and store bursts in
 * lockstep).  Same kernels, same rows: outputs are bit-identical for every n (tests/test_gpu_tower.py).  Small batches
 * (fewer than 256 tile blocks per chain) run as one chain. */
agz_status agz_net_set_tower_streams(agz_engine* e, int32_t n);
/* tower arithmetic of the network selected by agz_net_select.  AGZ_PRECISION_F32 (default): exact
 * f32 end to end -- the parity target of BASELINE.json's metric.  AGZ_PRECISION_F16: the "fp16 MFMA
 * path" of BASELINE.json configs[4]: tower activations and weights are rounded to IEEE half, products
 * accumulate in f32 (v_mfma_f32_32x32x16_f16); stem, heads, BatchNorm affine and residual adds stay
 * f32.  Mixed-precision inference: outputs agree with the f32 network to ~1e-3, not 1e-4. */
#define AGZ_PRECISION_F32 0
#define AGZ_PRECISION_F16 1
/* AGZ_PRECISION_F32S: the f32 network of AGZ_PRECISION_F32 -- f32 activations, weights, accumulation, BatchNorm,
 * residuals -- with the OPERANDS of the Winograd GEMMs carried as two IEEE halves each (x ~ hi + lo, 22 mantissa
 * bits instead of 24) so that the products run on the fp16 MFMA (v_mfma_f32_32x32x16_f16, all four cross products,
 * exact in f32) at 4x the f32 MFMA rate.  Opt-in; NOT what bench.py measures by default.  Outputs agree with the
 * float64 oracle to ~1e-6 (bar 1e-4, tests/test_gpu_nn32s.py). */
#define AGZ_PRECISION_F32S 2
agz_status agz_net_set_precision(agz_engine* e, int32_t precision);

/* HIP-event timing of every 3x3 256->256 tower-conv launch issued by subsequent steps /
 * forwards (up to 4096 launches), on the engine's own stream.  read() synchronises and returns
 * the summed launch time, the summed ALGORITHMIC flops (2 * rows * 9 * 256 * 256 with the rows
 * each launch actually processed) and the launch count. */
agz_status agz_profile_conv_enable(agz_engine* e, int32_t on);
agz_status agz_profile_conv_read(agz_engine* e, double* total_ms, double* total_flop, int64_t* launches);
/* The same for the search kernels of agz_selfplay_step (SURVEY.md 8d asks for them as HBM GB/s beside the tower's
 * TFLOP/s): HIP events on the engine's stream around k_pre (select_leaf / pick_move / play_move!, mcts.jl:108-138,
 * mcts_play.jl:52-71,126-139), k_expand (maybe_add_child!'s play_move!, mcts.jl:140-147, board.jl:451-509), k_scan,
 * k_leaf_features (features.jl:3-26) and k_post (incorporate_results! / backup_value!, mcts.jl:186-225) of the next
 * <= 512 steps.  read() synchronises; ms5 = summed milliseconds in that order. */
agz_status agz_profile_search_enable(agz_engine* e, int32_t on);
agz_status agz_profile_search_read(agz_engine* e, double* ms5 /* [5] */, int64_t* steps);

/* ---------------------------------------------------------------- Go rules (batched) --- */
/* play_move!(pos, c), board.jl:451-509 / pass_move! :426-440.  In/out SoA per position:
 * boards int8[B][N*N], to_play int8[B], ko int32[B] (-1 = none), moves int32[B].
 * status_out[b] = AGZ_OK or AGZ_ILLEGAL_MOVE (then the outputs for b repeat the input). */
agz_status agz_go_play(agz_engine* e, const int8_t* boards, const int8_t* to_play, const int32_t* ko,
                       const int32_t* moves, int32_t B, int8_t* boards_out, int32_t* ko_out,
                       int32_t* ncaptured_out, int32_t* status_out);
/* all_legal_moves(pos) -> Int8[A], board.jl:393-424 */
agz_status agz_go_legal(agz_engine* e, const int8_t* boards, const int8_t* to_play, const int32_t* ko,
                        int32_t B, int8_t* legal_out /* [B][A] */);
/* score(pos) board.jl:511-533 (area - komi, Black-positive) */
agz_status agz_go_score(agz_engine* e, const int8_t* boards, const float* komi, int32_t B,
                        float* score_out);

/* ---------------------------------------------------------------- batched self-play ----- */
/* selfplay(env, nn, num_ro) selfplay.jl:1-45, many games at once.  Start (re)initialises
 * every slot; each step is one tree_search! (mcts_play.jl:73-98) for every live game plus the
 * per-move phase for games whose readout budget is spent; finished games are recorded and
 * their slot recycled until `total_games` have been started (0 = recycle forever). */
agz_status agz_selfplay_start(agz_engine* e, int64_t total_games);
agz_status agz_selfplay_step(agz_engine* e, int32_t nsteps);          /* asynchronous */
typedef struct {
  int64_t steps;               /* tree_search! rounds executed                          */
  int64_t positions;           /* self-play moves played (= searches_pi entries)        */
  int64_t games_started;
  int64_t games_finished;
  int64_t evals;               /* network evaluations (leaves sent to the NN)            */
  int64_t duplicate_evals;     /* evaluations discarded by revert_visits! (mcts.jl:173)  */
  int64_t terminal_visits;     /* select_leaf hits on finished positions                */
  int64_t root_visits;         /* sum of N(root) increments                             */
  int64_t nodes_in_use;
  int64_t pool_exhausted;      /* allocations refused by a full pool (see pool_policy)  */
  int64_t resigned_games;
  int64_t live_games;
  int64_t records_dropped;     /* finished games overwritten in the record ring since the last
                                * agz_records_clear (ring = record_capacity_games): drain more often */
  int64_t pool_short_searches; /* moves played before their readout budget was spent because the game's pool was
                                * full (AGZ_POOL_MOVE_EARLY); 0 = every move had the reference's R readouts */
  int64_t peak_nodes_per_game; /* largest tree any slot has held at the moment it moved (of max_nodes_per_game) */
  int64_t stalled_games;       /* slots waiting on a full pool right now (AGZ_POOL_STALL, or no visited child to play) */
  int64_t node_capacity;       /* max_nodes_per_game in effect */
  int64_t abandoned_games;     /* games given up by agz_slot_abandon: they produce no record, so a run started with
                                * agz_selfplay_start(total) is over when games_finished + abandoned_games == total */
} agz_stats;
agz_status agz_engine_stats(agz_engine* e, agz_stats* out);            /* synchronises */
/* Per slot (arrays of `games` int32, any of them may be NULL): status = AGZ_OK or AGZ_POOL_EXHAUSTED (the game is
 * waiting on a full node pool: agz_config.pool_policy), nodes its tree holds, moves it has played.  The reference has
 * no analogue (its tree is unbounded, mcts.jl:140-147, mcts_play.jl:48); synchronises. */
agz_status agz_slot_status(agz_engine* e, int32_t* status_out, int32_t* nodes_out, int32_t* moves_out);
/* give up the game in `slot` without a record (counted in agz_stats.abandoned_games; its game id is not played again); the
 * slot starts the next game id at the next step */
agz_status agz_slot_abandon(agz_engine* e, int32_t slot);
/* external-network mode (MCTSPlayer.network duck typing, mcts_play.jl:5,89): after a step's
 * select phase the caller reads the leaf feature tensor and supplies pi/v. */
agz_status agz_selfplay_select(agz_engine* e, int32_t* nleaves_out);
agz_status agz_selfplay_leaf_features(agz_engine* e, float* feats_out /* N x N x 17 x B */);
agz_status agz_selfplay_incorporate(agz_engine* e, const float* pi /* A x B */, const float* v);

/* finished-game records: extract_data(player), mcts_play.jl:126-139 */
typedef struct {
  uint64_t game_id;
  int32_t num_moves;           /* position.n == length(searches_pi)                      */
  int32_t result;              /* +1 Black, -1 White, 0 draw (Black-absolute)             */
  int32_t was_resign;
  int32_t resign_disabled;
  float final_score;           /* score(position) when not resigned                       */
  int32_t short_searches;      /* moves of this game played on fewer than num_readouts readouts (full node pool,
                                * AGZ_POOL_MOVE_EARLY); 0 for a game that is the reference's game */
} agz_game_header;
int64_t agz_records_count(agz_engine* e);                               /* synchronises */
agz_status agz_records_header(agz_engine* e, int64_t k, agz_game_header* out);
/* moves int16[num_moves] (action index), pis float[num_moves][A], qs float[num_moves] */
agz_status agz_records_game(agz_engine* e, int64_t k, int16_t* moves, float* pis, float* qs);
/* packed export for the replay all-gather: writes every finished record as
 * [header | moves | pis | qs] back to back; returns bytes via nbytes_out. dst may be a host
 * or a device pointer (is_device). */
agz_status agz_records_packed_size(agz_engine* e, int64_t* nbytes_out);
agz_status agz_records_export_packed(agz_engine* e, void* dst, int64_t capacity, int32_t is_device);
agz_status agz_records_clear(agz_engine* e);
/* arena_mode with external_network: after agz_selfplay_select, counts_out[0] leaves belong to Black
 * players (network 0) and counts_out[1] to White players (network 1); agz_selfplay_leaf_features
 * and agz_selfplay_incorporate order the rows [Black players' | White players'].  A finished
 * arena game is one record: game_id = 2*game + colour of the player that ended it, moves, qs =
 * Q(root) of the mover, pis zero (two_player_mode records none, mcts_play.jl:33-36), result = what
 * set_result! stored, final_score = score(final position) -- evaluate's tally (:147) is
 * final_score > 0, also for resigned games. */
agz_status agz_arena_counts(agz_engine* e, int32_t* counts_out);
/* replay_position(pos, result) board.jl:557-578 on the device: rebuild the feature tensors of
 * every position of record k: out float[num_moves][N*N*17] (WHC per position) */
agz_status agz_records_features(agz_engine* e, int64_t k, float* out);
/* get_replay_batch(pos_buffer, ...) train.jl:4-12, feature side, for records from ANY rank: the
 * caller keeps games as action lists (moves int16[nmoves], games back to back); sample b is the
 * position before move ply[b] of the game starting at moves[game_offset[b]] (ply 0 = empty board).
 * One wave per sample replays the game on the device (board.jl:557-578) and writes
 * out float[B][N*N*17] (same WHC order as agz_features); out may be a device pointer. */
agz_status agz_replay_features(agz_engine* e, const int16_t* moves, int64_t nmoves,
                               const int32_t* game_offset, const int32_t* ply, int32_t B, float* out,
                               int32_t out_is_device);

/* ---------------------------------------------------------------- replay arena + exchange -- */
/* The replay buffer of train() (pos_buffer / pi_buffer / res_buffer, train.jl:47-66) as a device-resident
 * arena of packed game records: games of EVERY rank, in arrival order, addressed by index 0..count-1.
 * A (game, ply) pair is a training sample: position before move `ply` (rebuilt on the device by
 * replay_position, board.jl:557-578), pi = searches_pi[ply], z = result (extract_data, mcts_play.jl:126-139). */
/* append packed records ([header | moves | pis | qs] as written by agz_records_export_packed) from a host or
 * device buffer, e.g. games loaded from disk or received by other means; added_out may be NULL */
agz_status agz_replay_ingest_packed(agz_engine* e, const void* packed, int64_t nbytes, int32_t is_device,
                                    int64_t* added_out);
/* the same for the receive buffer of a padded all-gather the HOST performed with its own communication library
 * (MPI.jl, Distributed): `world` chunks of `chunk_stride` bytes, chunk r holding counts[2r] records in its first
 * counts[2r+1] bytes (what agz_records_count / agz_records_packed_size said on rank r).  This is also the second
 * half of agz_allgather_records (which does the gather itself over RCCL). */
agz_status agz_replay_ingest_gathered(agz_engine* e, const void* buf, int32_t is_device, int32_t world,
                                      int64_t chunk_stride, const int64_t* counts, int64_t* added_out);
int64_t agz_replay_count(agz_engine* e);                 /* games in the arena                       */
int64_t agz_replay_positions(agz_engine* e);             /* sum of num_moves = length(pos_buffer)    */
agz_status agz_replay_header(agz_engine* e, int64_t k, agz_game_header* out);
agz_status agz_replay_game(agz_engine* e, int64_t k, int16_t* moves, float* pis, float* qs);
/* `shrink` (train.jl:52): forget the oldest games until at most max_positions positions remain */
agz_status agz_replay_trim(agz_engine* e, int64_t max_positions);
agz_status agz_replay_clear(agz_engine* e);
/* get_replay_batch (train.jl:4-12) for B sampled (game, ply) pairs, ply < num_moves(game):
 * feats float[B][N*N*17] (order of agz_features), pi float[B][A], z float[B]; pi / z may be NULL;
 * the three outputs are host pointers, or device pointers when out_is_device != 0 */
agz_status agz_replay_batch(agz_engine* e, const int64_t* game, const int32_t* ply, int32_t B, float* feats,
                            float* pi, float* z, int32_t out_is_device);

/* ---------------------------------------------------------------- training step --------- */
/* One optimisation step of `_train` (neural_net.jl:75-101; optimiser Momentum(2f-2), train.jl:54; call
 * train.jl:67-74) on the network selected by agz_net_select, entirely on the device:
 *   training-mode forward (BatchNorm normalises with the batch's statistics; its running statistics move by
 *   momentum 0.1), loss = 0.01 * crossentropy(p, pi) + 0.01 * mse(v, z) + 1e-4 * sum(theta^2), backward,
 *   vel = rho * vel - eta * grad; theta += vel for every parameter (Flux Momentum: eta 0.02, rho 0.9).
 * `_train` does not run at the reference's HEAD; this is its intended step, pinned against a
 * float64 autograd twin.  feats float[B][N*N*17] (agz_features / agz_replay_batch order), pi float[B][A],
 * z float[B]: three host pointers, or three device pointers when inputs_are_device != 0 (the outputs of
 * agz_replay_batch with out_is_device).  losses_out float[4] = {total, policy, value, regulariser} BEFORE the
 * update; may be NULL.  B >= 2.  The optimiser state lives in the engine until agz_train_reset. */
agz_status agz_train_step(agz_engine* e, const float* feats, const float* pi, const float* z, int32_t B,
                          int32_t inputs_are_device, float eta, float rho, float* losses_out);
agz_status agz_train_reset(agz_engine* e);

/* The one exchange step of the path: RCCL over xGMI, one rank per GPU.  Rank 0 calls
 * agz_comm_unique_id and hands the 128 bytes to the other ranks by whatever means the host has (Julia:
 * Distributed / a shared file; Python: torch.distributed / gloo); every rank then calls agz_comm_create with
 * the engine that lives on its GPU.  RCCL is bound at run time (dlopen librccl.so.1); a missing library or a
 * failing collective returns AGZ_RCCL_ERROR with the RCCL error string in agz_last_error. */
#define AGZ_COMM_ID_BYTES 128
typedef struct agz_comm agz_comm;
agz_status agz_comm_unique_id(uint8_t* id_out /* [AGZ_COMM_ID_BYTES] */);
agz_status agz_comm_create(agz_engine* e, int32_t rank, int32_t world, const uint8_t* id, agz_comm** out);
void agz_comm_destroy(agz_comm* c);
/* all-gather the finished records of every rank (what agz_records_* shows on each) into THIS rank's replay
 * arena, rank 0's games first: this rank's unsent records are packed on the device, then a count exchange and one
 * padded ncclAllGather, device to device.  Records a completed payload collective has carried are not sent again,
 * whether or not this rank's own ingest then succeeded (a second call without new finished games adds nothing);
 * agz_records_clear empties the record ring and resets that mark.  comm == NULL: single-GPU run, files the
 * engine's own records.  added_out (may be NULL) = games appended.  Collective: every rank of the
 * communicator must call it.  Everything that can fail on one rank alone (counting, packing) happens BEFORE the
 * first collective; such a rank announces {-1, its status} in the count collective and EVERY rank, the failing
 * one included, returns AGZ_RCCL_ERROR (its message names the rank; the failing rank's also carries its own
 * reason) -- nobody is left waiting in ncclAllGather. */
agz_status agz_allgather_records(agz_engine* e, agz_comm* comm, int64_t* added_out);
/* The host logic between the two collectives of that exchange, for a host that carries the bytes with its own
 * library (MPI.jl, Distributed, torch.distributed/gloo) and finishes with agz_replay_ingest_gathered:
 * counts[2r], counts[2r+1] = {agz_records_count, agz_records_packed_size} of rank r as gathered (a rank that could
 * not pack sends {-1, its agz_status}).  Checks every pair and returns the chunk stride (largest rank, padded to
 * 256 B) each rank pads its packed export to; total_records_out may be NULL.  AGZ_RCCL_ERROR names the
 * offending rank (text via agz_last_error(NULL)).  Pure host code: needs no engine and no GPU. */
agz_status agz_gather_plan(const int64_t* counts, int32_t world, int64_t* chunk_stride_out, int64_t* total_records_out);
/* overwrite every rank's weight replica with rank `root`'s parameters of the selected network (after a
 * training step on one rank); nfloats_out may be NULL.  Collective. */
agz_status agz_broadcast_weights(agz_engine* e, agz_comm* comm, int32_t root, int64_t* nfloats_out);

/* ---------------------------------------------------------------- ABI self-description --- */
/* sizeof and field offsets of the PODs above as this library was compiled, so that a host mirror (ctypes
 * Structure, Julia struct) can be checked against them: name in {"agz_config", "agz_stats",
 * "agz_game_header", "agz_position_info", "agz_node_info"}; out[0] = sizeof, out[1..n] = offsetof of the n
 * fields in declaration order; returns n, or -1 for an unknown name / too small a buffer. */
int32_t agz_abi_layout(const char* name, int32_t* out, int32_t cap);

/* ---------------------------------------------------------------- single-tree compat ---- */
/* The reference's MCTSPlayer / MCTSNode API on game slot g (tests drive these one call at a
 * time exactly like test/test_mcts.jl and test/test_mcts_player.jl).  Node handles are
 * slot-local int32 ids; the root is whatever agz_tree_root returns. */
typedef struct {
  int32_t n;                   /* moves played so far                                   */
  int32_t to_play;
  int32_t ko;                  /* -1 none                                               */
  int32_t caps_black, caps_white;
  int32_t last_move;           /* -1 none, N*N pass  (recent[end].move)                 */
  int32_t prev_move;           /* -1 none            (recent[end-1].move)               */
  int32_t history_len;         /* boards of real history available before this one (<=7) */
  float komi;
} agz_position_info;
/* initialize_game!(player, pos) mcts_play.jl:110-118; history = int8[history_len][N*N] older
 * boards newest first (NULL when history_len == 0) */
agz_status agz_tree_init(agz_engine* e, int32_t g, const int8_t* board, const agz_position_info* info,
                         const int8_t* history);
agz_status agz_tree_root(agz_engine* e, int32_t g, int32_t* node_out);
agz_status agz_tree_select_leaf(agz_engine* e, int32_t g, int32_t from_node, int32_t* leaf_out);
agz_status agz_tree_maybe_add_child(agz_engine* e, int32_t g, int32_t node, int32_t a, int32_t* child_out);
agz_status agz_tree_add_virtual_loss(agz_engine* e, int32_t g, int32_t node, int32_t up_to);
agz_status agz_tree_revert_virtual_loss(agz_engine* e, int32_t g, int32_t node, int32_t up_to);
agz_status agz_tree_incorporate(agz_engine* e, int32_t g, int32_t node, const float* probs,
                                int32_t nprobs, float value, int32_t up_to);
agz_status agz_tree_inject_noise(agz_engine* e, int32_t g, int32_t node);
/* tree_search!(player, parallel_readouts): select phase then (with an internal network) the
 * evaluation and incorporate phases; returns the number of leaves */
agz_status agz_tree_search(agz_engine* e, int32_t g, int32_t parallel_readouts, int32_t* nleaves_out);
/* the same split at the network call, for a caller-supplied network (DummyNet, a Flux model):
 * select -> read the leaves' feature tensor (N x N x 17 x nleaves) -> hand back pi (A x nleaves)
 * and v (nleaves); pi == NULL makes the engine evaluate the leaves with its own network */
agz_status agz_tree_search_select(agz_engine* e, int32_t g, int32_t parallel_readouts, int32_t* nleaves_out);
agz_status agz_tree_leaf_features(agz_engine* e, int32_t g, float* feats_out);
/* ... or read the leaves as the `Vector{Position}` the reference hands its network (`mcts_player.network([leaf.position
 * for leaf in leaves])`, mcts_play.jl:89; DummyNet sizes its answer by length(positions), test/test_mcts_player.jl:25-32):
 * per collected leaf, in collection order, the GoPosition fields of board.jl:271-306 -- nodes_out int32[B] (the leaf's
 * node handle), boards_out int8[B][N*N], deltas_out int8[B][7][N*N] (board_deltas newest first, zero beyond ndeltas),
 * ndeltas_out int32[B], to_play_out int8[B] (together the SoA agz_net_forward / agz_features take), info_out[B] (n, ko,
 * caps, last two moves, komi; history_len = ndeltas).  Any output may be NULL.  Valid between agz_tree_search_select and
 * agz_tree_search_incorporate. */
agz_status agz_tree_leaf_positions(agz_engine* e, int32_t g, int32_t* nodes_out, int8_t* boards_out, int8_t* deltas_out,
                                   int32_t* ndeltas_out, int8_t* to_play_out, agz_position_info* info_out);
agz_status agz_tree_search_incorporate(agz_engine* e, int32_t g, const float* pi, const float* v);
agz_status agz_tree_pick_move(agz_engine* e, int32_t g, int32_t* a_out);
agz_status agz_tree_play_move(agz_engine* e, int32_t g, int32_t a, int32_t* ok_out);
agz_status agz_tree_should_resign(agz_engine* e, int32_t g, int32_t* out);
agz_status agz_tree_is_done(agz_engine* e, int32_t g, int32_t node, int32_t* out);
typedef struct {
  float N, W, Q;
  int32_t parent, fmove, is_expanded, losses_applied, done;
  agz_position_info pos;
} agz_node_info;
agz_status agz_tree_node_info(agz_engine* e, int32_t g, int32_t node, agz_node_info* out);
#define AGZ_F_CHILD_N 0
#define AGZ_F_CHILD_W 1
#define AGZ_F_CHILD_PRIOR 2
#define AGZ_F_ACTION_SCORE 3     /* Float64 scores narrowed to double[A] */
agz_status agz_tree_node_floats(agz_engine* e, int32_t g, int32_t node, int32_t field, float* out);
agz_status agz_tree_node_scores(agz_engine* e, int32_t g, int32_t node, double* out);
agz_status agz_tree_node_set_floats(agz_engine* e, int32_t g, int32_t node, int32_t field, const float* in);
agz_status agz_tree_node_set_N(agz_engine* e, int32_t g, int32_t node, float value);
agz_status agz_tree_node_set_n(agz_engine* e, int32_t g, int32_t node, int32_t n);
agz_status agz_tree_node_children(agz_engine* e, int32_t g, int32_t node, int32_t* out /* [A] */);
agz_status agz_tree_node_board(agz_engine* e, int32_t g, int32_t node, int8_t* out /* [N*N] */);
agz_status agz_tree_pending_vlosses(agz_engine* e, int32_t g, int32_t* out);
agz_status agz_tree_set_draw(agz_engine* e, int32_t g, uint64_t game_id, uint32_t sel);

/* Test hooks (device-side evaluation of the draw stream, single-tree introspection setters) are declared in
 * include/agz_debug.h: exported by the library for the parity tests, not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* AGZ_H */
