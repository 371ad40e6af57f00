// agz_state.h -- HBM-resident state of the batched self-play engine (SoA, one pool per game).
//
// The reference keeps one heap-allocated MCTSNode per tree node, each owning a deep-copied
// GoPosition (src/mcts.jl:41-82, src/game/go/board.jl:271-315).  Here every game slot owns a
// fixed pool of `cap` node records in HBM; a node's statistics are rows of [cap*games][AP]
// arrays so that one wavefront reads a level's child_N / child_W / child_prior / child-index
// rows with fully coalesced loads.  A node stores only its own board (N*N bytes): the eight
// history planes the network needs come from its ancestors' boards (features.jl:8-14 rebuilds
// exactly those boards from deltas), plus a 7-deep per-game ring for boards older than the root.
#pragma once
#include <cstdint>

#include "../../include/agz.h"

namespace agz {

constexpr int kMaxPar = 64;        // upper bound on parallel_readouts
constexpr int kWave = 64;

enum GamePhase : int32_t {
  G_IDLE = 0, G_INIT = 1, G_INIT_WAIT = 2, G_SEARCH = 3, G_MANUAL = 4, G_RETIRED = 5,
  G_ARENA_WAIT = 6    // arena: the partner slot (other colour, other network) is to move
};

enum NodeFlags : uint8_t { NF_EXPANDED = 1, NF_DONE = 2, NF_ALLOC = 4 };

struct NodeMeta {
  int32_t parent;      // slot-local node id, -1 for a root
  int32_t n;           // position.n
  int32_t ko;          // point or -1
  int32_t caps_b, caps_w;
  int16_t fmove;       // action that led here (-1 for a root)
  int16_t last_move;   // recent[end].move: point, N*N for pass, -1 for none
  int16_t losses;      // losses_applied, mcts.jl:45
  int8_t to_play;
  uint8_t flags;
  int32_t pad;
};
static_assert(sizeof(NodeMeta) == 32, "NodeMeta layout");

struct GameState {
  uint64_t game_id;
  double resign_threshold;
  float rootN, rootW;        // DummyNode entries of the root, mcts.jl:27-39,96-102
  float target;              // current_readouts + readouts, selfplay.jl:24-27
  float komi;
  int32_t root;
  int32_t phase;
  int32_t sel;               // select_leaf calls since the last move (draw-stream index)
  int32_t move_count;        // length(searches_pi)
  int32_t nqs;               // length(qs)
  int32_t hist_len;          // boards of history older than the root (<= 7)
  int32_t free_top;
  int32_t nleaves;
  int32_t leaf_base;
  int32_t resign_disabled;
  int32_t err;
  int32_t result;
  int32_t was_resign;
  int32_t nodes_used;
  int32_t short_first;       // bench stagger: the first search of this game has a shortened budget
  int32_t arena_k;           // arena: games this slot has finished (= local index of the current one)
  int32_t garbage;           // nodes on the deferred-free stack (tail of the slot's free list)
  int32_t npend;             // leaves created by this step's select phase whose board update waits for k_expand
  int32_t short_searches;    // moves of this game played before their budget was spent (full pool, AGZ_POOL_MOVE_EARLY)
  int32_t stalled;           // set by game_pre: the pool is full AND this game could not move early -- it is waiting
                             // (G.err alone also reads EXHAUSTED between a refused allocation and the early move of
                             // the next step: ADVICE r5)
};
static_assert(sizeof(GameState) == 112, "GameState layout (tests/hs.py mirrors it)");

constexpr int kMaxPend = 16;    // deferred leaf expansions per game and step (2 x parallel_readouts at most)

enum Counter : int {
  CT_STEPS = 0, CT_POSITIONS, CT_STARTED, CT_FINISHED, CT_EVALS, CT_DUP, CT_TERMINAL, CT_ROOTVISITS,
  CT_POOL_EXHAUSTED, CT_RESIGNED, CT_CLAIMED,
  CT_RECORDED,     // records written to the finished-game ring since the last records_clear (CT_FINISHED never resets)
  // k_pre phase clocks (100 MHz ticks summed over games), written only in -DAGZ_TIMING_EXPERIMENTS builds and read
  // through agz_debug_counters (tools/pre_phases.py): games in their per-move phase / all searching games
  CT_T_FREE, CT_T_PICK, CT_T_CHILD, CT_T_REROOT, CT_T_NOISE, CT_T_MOVE_SELECT, CT_N_MOVE, CT_T_SELECT, CT_N_SELECT,
  CT_T_MOVE_MAX,   // max over games of one move phase + its select phase
  CT_T_CREATE, CT_N_CREATE,   // node_create_child (leaf expansion: board update in scratch), all callers
  CT_POOL_SHORT,   // moves played early because the pool was full
  CT_PEAK_NODES,   // max over games of nodes_used at the moment of a move
  CT_COUNT
};

struct View {
  // dimensions
  int N, P, PP, A, AP, LW, cap, games, par, maxd;
  int R, max_game_length, tau, two_player, stagger;
  int arena;           // evaluate() arena: slots 2i / 2i+1 are Black's / White's player of one game
  int pool_policy;     // AGZ_POOL_MOVE_EARLY / AGZ_POOL_STALL
  int fin_cap;
  int64_t total_games;
  uint64_t seed, id_base, id_stride;
  double c_puct, noise_w, alpha, resign_threshold, resign_disable_frac;
  float komi;
  int32_t defer_expand;   // 1: the select phase only allocates new leaves, k_expand (one wave per leaf) plays the move
  // node pools  [games*cap]
  float* childN;
  float* childW;
  float* childP;
  int32_t* child;
  int8_t* board;
  NodeMeta* meta;
  uint32_t* legal;
  // per game
  GameState* gs;
  int8_t* hist;        // [games][7][PP]
  int32_t* freelist;   // [games][cap]
  // per step leaf bookkeeping  [games][par]
  int32_t* leaf_node;
  int32_t* leaf_featsrc;  // [games][par][8]: node id >= 0, or -(h+1) for history slot h
  int8_t* leaf_tp;
  int32_t* leaf_plen;
  int32_t* leaf_path;     // [games][par][maxd]
  int32_t* pend_node;     // [games][kMaxPend]: nodes of GameState::npend
  // live game records [games][mgl]
  int16_t* rec_moves;
  float* rec_pi;          // [games][mgl][A]
  float* rec_q;
  // finished-game arena [fin_cap]
  agz_game_header* fin_hdr;
  int16_t* fin_moves;
  float* fin_pi;
  float* fin_q;
  // counters / batch
  unsigned long long* counters;
  int32_t* batch_count;   // device scalars: leaves in this step's batch ([0]; arena: [0] Black's, [1] White's players)
  int32_t* ar_hdr;        // arena mailbox [games/2][4]: {local game index, plies played, done, -}
  const float* pi;        // [batch][A]
  const float* v;         // [batch]
};

}  // namespace agz
