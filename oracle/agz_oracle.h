/*
 * agz_oracle.h -- CPU restatement of the AlphaGo.jl self-play hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so; the HIP engine never does.
 *
 * The reference is Julia and cannot be built or run in this image (no julia binary, and
 * at HEAD it does not load under its own Manifest -- SURVEY.md section 0), so this is a
 * restatement written from the reference sources, in the reference's own execution
 * shape (one game at a time, pointer-linked tree, positions that carry their delta
 * history).  It is pinned against every known-answer in the reference's test-suite
 * (tests/test_oracle_go.py, test_oracle_mcts.py, test_oracle_player.py,
 * test_oracle_features.py).  The network forward has NO pin in the reference's tests
 * (SURVEY.md 8c): NN parity is "unpinned" and rests on the documented Flux/NNlib layer
 * semantics checked against torch fp64 (tests/test_oracle_nn.py).
 *
 * Conventions: everything 0-based.  Board point p = row + N*col (the reference's
 * column-major linear index minus one, src/game/go/coords.jl:6-7); action a in [0,A),
 * a == N*N is the pass move.  Colours: BLACK = +1, WHITE = -1, EMPTY = 0
 * (src/game/go/board.jl:12).
 */
#ifndef AGZ_ORACLE_H
#define AGZ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_MAXN 19
#define OR_MAXP (OR_MAXN * OR_MAXN)
#define OR_MAXA (OR_MAXP + 1)
#define OR_MAXRECENT 1024

#define OR_OK 0
#define OR_ILLEGAL_MOVE 1      /* IllegalMove          src/AlphaGo.jl:8, board.jl:265,470 */
#define OR_ASSERT_DONE_NODE 2  /* AssertionError       src/mcts.jl:196                    */
#define OR_HISTORY_INCOMPLETE 3/* AssertionError       board.jl:568, mcts_play.jl:127     */
#define OR_BAD_SHAPE 4         /* AssertionError       src/mcts.jl:190                    */
#define OR_ASSERT_SOFTPICK 5   /* AssertionError       src/mcts_play.jl:67                */

/* GoPosition, src/game/go/board.jl:271-306 (the liberty tracker is an implementation
 * detail of the reference and is not mirrored: captures, suicide and legality are
 * computed by flood fill, which is what find_reached board.jl:28-45 does anyway). */
typedef struct {
  int N, A;
  int8_t board[OR_MAXP];
  int n;
  float komi;
  int caps[2];
  int ko;                    /* point or -1 */
  int to_play;
  int done;
  int ndeltas;               /* <= 7, newest first (board.jl:505-506) */
  int8_t deltas[7][OR_MAXP];
  int recent_len;
  int16_t recent_move[OR_MAXRECENT]; /* point, or N*N for a pass */
  int8_t recent_color[OR_MAXRECENT];
} OPos;

/* MCTSRules + module globals, src/mcts.jl:11-25 */
typedef struct {
  int N, A;
  int max_game_length;       /* (N^2*7) div 5                                        */
  float dirichlet_alpha;     /* Float32(0.03*361/A): max_action_space=361 go.jl:24   */
  double c_puct;             /* 0.96  (Float64 global, mcts.jl:11)                    */
  double noise_weight;       /* 0.25  (Float64 global, mcts.jl:13)                    */
} OEnv;

typedef struct { uint64_t seed, game; uint32_t move, sel; } ODraw;

typedef struct ONode ONode;
typedef struct OPlayer OPlayer;

/* network(positions) -> (pi A x B column-major, v B); duck-typed field mcts_play.jl:5 */
typedef void (*or_net_fn)(void* ctx, const OPos* const* positions, int B, float* pi, float* v);

/* ---- env / positions ---- */
void or_env_init(OEnv* env, int N);
void or_pos_init(OPos* pos, int N, float komi);                       /* Position(env) */
void or_pos_from_board(OPos* pos, int N, const int8_t* board, int n, float komi, int cap_b,
                       int cap_w, int ko, int to_play, int nrecent, const int16_t* recent_move,
                       const int8_t* recent_color);
int or_play_move(const OPos* pos, int a, OPos* out);                  /* board.jl:451-509 */
int or_play_move_color(const OPos* pos, int a, int color, OPos* out);
void or_pass_move(const OPos* pos, OPos* out);                        /* board.jl:426-440 */
void or_flip_playerturn(const OPos* pos, OPos* out);                  /* board.jl:442-447 */
int or_is_koish(int N, const int8_t* board, int p);                   /* board.jl:47-56  */
int or_is_eyeish(int N, const int8_t* board, int p);                  /* board.jl:58-81  */
int or_is_move_suicidal(const OPos* pos, int p);                      /* board.jl:354-374 */
int or_is_move_legal(const OPos* pos, int a);                         /* board.jl:376-391 */
void or_all_legal_moves(const OPos* pos, int8_t* out /*A*/);          /* board.jl:393-424 */
float or_score(const OPos* pos);                                      /* board.jl:511-533 */
int or_result(const OPos* pos);                                       /* board.jl:535-544 */
void or_result_string(const OPos* pos, char* out /*>=16*/);           /* board.jl:546-555 */
/* group containing stone p: marks stones[] / libs[] (0/1 per point), returns #libs, or -1
 * if p is empty.  Used to pin the liberty-tracker known answers of test_go.jl:74-262. */
int or_group_info(int N, const int8_t* board, int p, int8_t* stones, int8_t* libs);
int or_count_groups(int N, const int8_t* board);

/* ---- features, src/features.jl:3-26; out is N x N x 17 column-major (WHC) ---- */
void or_get_feats(const OPos* pos, float* out);
void or_get_feats_f64(const OPos* pos, double* out);

/* ---- search tree, src/mcts.jl ---- */
ONode* or_node_new(const OEnv* env, const OPos* pos);                 /* MCTSNode(pos)   */
void or_node_free_tree(ONode* root);
ONode* or_select_leaf(const OEnv* env, ONode* root, ODraw* draw);     /* mcts.jl:108-138 */
int or_maybe_add_child(const OEnv* env, ONode* node, int a, ONode** out); /* :140-147    */
void or_add_virtual_loss(ONode* node, ONode* up_to);                  /* :149-163 */
void or_revert_virtual_loss(ONode* node, ONode* up_to);               /* :165-171 */
void or_revert_visits(ONode* node, ONode* up_to);                     /* :173-186 */
int or_incorporate_results(const OEnv* env, ONode* node, const float* probs, int nprobs,
                           float value, ONode* up_to);                /* :188-213 */
void or_backup_value(ONode* node, float value, ONode* up_to);         /* :215-225 */
int or_node_is_done(const OEnv* env, const ONode* node);              /* :230-231 */
void or_inject_noise(const OEnv* env, ONode* node, const ODraw* draw);/* :233-239 */
void or_children_as_pi(const ONode* node, int squash, float* out);    /* :241-252 */
void or_child_action_score(const OEnv* env, const ONode* node, double* out); /* :86-92 */
/* accessors */
float or_node_N(const ONode* node);
float or_node_W(const ONode* node);
float or_node_Q(const ONode* node);
void or_node_set_N(ONode* node, float v);
int or_node_fmove(const ONode* node);
int or_node_is_expanded(const ONode* node);
int or_node_losses_applied(const ONode* node);
ONode* or_node_child(const ONode* node, int a);
ONode* or_node_parent(const ONode* node);
const OPos* or_node_pos(const ONode* node);
OPos* or_node_pos_mut(ONode* node);
float* or_node_child_N(ONode* node);
float* or_node_child_W(ONode* node);
float* or_node_child_prior(ONode* node);
float* or_node_original_prior(ONode* node);
int or_tree_pending_vlosses(const ONode* root);   /* test_utils.jl:76-85 */
int or_tree_count_nodes(const ONode* root);

/* ---- player, src/mcts_play.jl ---- */
OPlayer* or_player_new(int N, or_net_fn net, void* net_ctx, int num_readouts, int two_player_mode,
                       double resign_threshold, uint64_t seed, uint64_t game);
void or_player_free(OPlayer* p);
void or_player_initialize_game(OPlayer* p, const OPos* pos /* or NULL */);
int or_player_tree_search(OPlayer* p, int parallel_readouts);         /* returns #leaves */
int or_player_pick_move(OPlayer* p, int* a_out);                      /* mcts_play.jl:52-71 */
int or_player_play_move(OPlayer* p, int a);                           /* 1 ok / 0 illegal */
int or_player_should_resign(const OPlayer* p);
int or_player_is_done(const OPlayer* p);
void or_player_set_result(OPlayer* p, int winner, int was_resign);
ONode* or_player_root(OPlayer* p);
const OEnv* or_player_env(const OPlayer* p);
int or_player_result(const OPlayer* p);
const char* or_player_result_string(const OPlayer* p);
int or_player_tau_threshold(const OPlayer* p);
int or_player_num_moves(const OPlayer* p);        /* length(searches_pi) */
const float* or_player_search_pi(const OPlayer* p, int k);
float or_player_q(const OPlayer* p, int k);
int or_player_nqs(const OPlayer* p);
uint64_t or_player_evals(const OPlayer* p);
/* extract_data mcts_play.jl:126-139: replays the game; positions[k] is the position before
 * move k.  Returns n or a negative error.  Any of the outputs may be NULL. */
int or_player_extract_data(const OPlayer* p, OPos* positions, float* pis, int* results);

/* ---- selfplay, src/selfplay.jl:1-45 (with the D2 ternary typo read as intended) ---- */
OPlayer* or_selfplay(int N, or_net_fn net, void* net_ctx, int num_readouts, uint64_t seed,
                     uint64_t game, int max_moves /* <=0: play to the end */);
/* same with the resign threshold (-0.9) and the resign-disable fraction (0.05) exposed */
OPlayer* or_selfplay_ex(int N, or_net_fn net, void* net_ctx, int num_readouts, uint64_t seed,
                        uint64_t game, int max_moves, double threshold, double disable_fraction);

/* ---- evaluate, src/neural_net.jl:103-158: ONE game of the two-network arena ----
 * black_net always plays Black, white_net White; both players are two_player_mode MCTSPlayers
 * (tau_threshold = -1: pick_move is always the arg-max; no pi recording; no Dirichlet noise) with
 * their own trees, advanced with the same moves.  Draw streams: Black's player is keyed
 * (seed, 2*game), White's (seed, 2*game+1).  `black_won` is the reference's tally
 * `result(black.root.position) == BLACK` (:147) -- the Tromp-Taylor result of the final position,
 * also for resigned games. */
typedef struct {
  int num_moves;       /* moves played */
  int result;          /* what set_result! stored: winner (+1 Black / -1 White / 0) */
  int was_resign;
  int black_won;       /* :147 */
  float final_score;   /* score(black.root.position) */
  uint64_t evals_black, evals_white;
} OEvalGame;
void or_evaluate_game(int N, or_net_fn black_net, void* black_ctx, or_net_fn white_net, void* white_ctx,
                      int num_readouts, double resign_threshold, uint64_t seed, uint64_t game,
                      int16_t* moves_out /* >= max_game_length, may be NULL */,
                      float* qs_out /* Q(root) of the mover before each move, may be NULL */, OEvalGame* out);

/* ---- network, src/neural_net.jl:13-33,57-73 + src/resnet.jl:11-32 ---- */
typedef struct ONet ONet;
ONet* or_net_new(int N, int tower_height);
void or_net_free(ONet* net);
/* layer ids: 0 = stem, 1..2t = tower convs (block b conv c -> 1+2b+c), then heads */
#define OR_L_VALUE_CONV (-1)
#define OR_L_POLICY_CONV (-2)
#define OR_L_VALUE_FC1 (-3)
#define OR_L_VALUE_FC2 (-4)
#define OR_L_POLICY_FC (-5)
/* kinds */
#define OR_K_WEIGHT 0   /* conv [kw,kh,cin,cout] column-major / dense [out,in] column-major */
#define OR_K_BIAS 1
#define OR_K_BN_BETA 2
#define OR_K_BN_GAMMA 3
#define OR_K_BN_MEAN 4
#define OR_K_BN_VAR 5
#define OR_K_BN_EPS 6   /* one float */
int or_net_set(ONet* net, int layer, int kind, const float* data, int64_t count);
int64_t or_net_param_count(const ONet* net, int layer, int kind);
/* Flux-default-equivalent init (glorot-uniform, zero bias, BN identity, eps 1e-5) from the
 * draw stream -- identical tensors for the oracle and the engine. */
void or_net_init_synthetic(ONet* net, uint64_t seed);
int or_net_get(const ONet* net, int layer, int kind, float* out, int64_t count);
/* forward on feature tensors x (N*N*17 per position, WHCN); precision 32 or 64 */
/* precision: 32 = float, 64 = double, 16 = the fp16-operand tower of agz_net_set_precision(F16)
 * restated (tower weights / activations rounded to IEEE half where the GPU stores them, float64 sums) */
void or_net_forward_feats(const ONet* net, const float* x, int B, float* pi, float* v, int precision);
float or_quant_half(float f);                      /* float -> nearest half (ties to even) -> float */
void or_net_forward_feats_f64(const ONet* net, const double* x, int B, double* pi, double* v);
/* a or_net_fn over an ONet (ctx = ONet*), fp32 compute */
void or_net_callable(void* ctx, const OPos* const* positions, int B, float* pi, float* v);
void or_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
