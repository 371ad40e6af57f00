/*
 * agz_oracle_mcts.c -- search tree, player and self-play loop of the CPU oracle.
 * TEST INFRASTRUCTURE (see agz_oracle.h).  Restates
 *   /root/reference/src/mcts.jl:11-252
 *   /root/reference/src/mcts_play.jl:3-151
 *   /root/reference/src/selfplay.jl:1-45
 * in the reference's execution shape: one pointer-linked tree, positions stored per node,
 * legality recomputed at every level of every descent, leaves evaluated 8 at a time.
 *
 * Floating-point types follow the reference exactly: tree statistics are
 * Float32; c_puct and the noise weight are Float64 globals, so U and the action score are
 * Float64; values entering the tree are Float32.  Build with -ffp-contract=off.
 */
#include "agz_oracle.h"
#include "../include/agz_draws.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ONode {
  ONode* parent;
  int fmove;                 /* action that led here; -1 for a root */
  OPos pos;
  int is_expanded;
  int losses_applied;
  float* child_N;
  float* child_W;
  float* original_prior;
  float* child_prior;
  ONode** children;
  float dummy_N, dummy_W;    /* DummyNode defaults, mcts.jl:27-39 */
};

ONode* or_node_new(const OEnv* env, const OPos* pos) {
  ONode* nd = (ONode*)calloc(1, sizeof(ONode));
  int A = env->A;
  nd->fmove = -1;
  memcpy(&nd->pos, pos, sizeof(OPos));
  nd->child_N = (float*)calloc((size_t)A, sizeof(float));
  nd->child_W = (float*)calloc((size_t)A, sizeof(float));
  nd->original_prior = (float*)calloc((size_t)A, sizeof(float));
  nd->child_prior = (float*)calloc((size_t)A, sizeof(float));
  nd->children = (ONode**)calloc((size_t)A, sizeof(ONode*));
  return nd;
}

static void node_free_subtree(ONode* nd) {
  if (!nd) return;
  int A = nd->pos.A;
  for (int a = 0; a < A; ++a) node_free_subtree(nd->children[a]);
  free(nd->child_N); free(nd->child_W); free(nd->original_prior); free(nd->child_prior);
  free(nd->children);
  free(nd);
}

void or_node_free_tree(ONode* root) {
  if (!root) return;
  while (root->parent) root = root->parent;
  node_free_subtree(root);
}

float or_node_N(const ONode* x) { return x->parent ? x->parent->child_N[x->fmove] : x->dummy_N; }
float or_node_W(const ONode* x) { return x->parent ? x->parent->child_W[x->fmove] : x->dummy_W; }
void or_node_set_N(ONode* x, float v) {
  if (x->parent) x->parent->child_N[x->fmove] = v; else x->dummy_N = v;
}
static void set_W(ONode* x, float v) {
  if (x->parent) x->parent->child_W[x->fmove] = v; else x->dummy_W = v;
}
/* Q(x) = W(x) / (1 + N(x)), Float32, mcts.jl:94 */
float or_node_Q(const ONode* x) { return or_node_W(x) / (1.0f + or_node_N(x)); }
int or_node_fmove(const ONode* x) { return x->fmove; }
int or_node_is_expanded(const ONode* x) { return x->is_expanded; }
int or_node_losses_applied(const ONode* x) { return x->losses_applied; }
ONode* or_node_child(const ONode* x, int a) { return x->children[a]; }
ONode* or_node_parent(const ONode* x) { return x->parent; }
const OPos* or_node_pos(const ONode* x) { return &x->pos; }
OPos* or_node_pos_mut(ONode* x) { return &x->pos; }
float* or_node_child_N(ONode* x) { return x->child_N; }
float* or_node_child_W(ONode* x) { return x->child_W; }
float* or_node_child_prior(ONode* x) { return x->child_prior; }
float* or_node_original_prior(ONode* x) { return x->original_prior; }

int or_tree_pending_vlosses(const ONode* root) {
  int A = root->pos.A, s = root->losses_applied != 0;
  for (int a = 0; a < A; ++a)
    if (root->children[a]) s += or_tree_pending_vlosses(root->children[a]);
  return s;
}

int or_tree_count_nodes(const ONode* root) {
  int A = root->pos.A, s = 1;
  for (int a = 0; a < A; ++a)
    if (root->children[a]) s += or_tree_count_nodes(root->children[a]);
  return s;
}

/* child_action_score = child_Q .* to_play .+ child_U, mcts.jl:86-92.
 *   child_Q = child_W ./ (1 .+ child_N)                                  Float32
 *   child_U = (c_puct * sqrt(1 + N(x))) * child_prior ./ (1 .+ child_N)  Float64
 * sqrt is taken on the Float32 value 1+N(x) and is itself Float32. */
void or_child_action_score(const OEnv* env, const ONode* x, double* out) {
  int A = env->A;
  float one_plus_n = 1.0f + or_node_N(x);
  double scale = env->c_puct * (double)sqrtf(one_plus_n);
  for (int a = 0; a < A; ++a) {
    float denom = 1.0f + x->child_N[a];
    float q = x->child_W[a] / denom;
    float qs = q * (float)x->pos.to_play;
    double u = (scale * (double)x->child_prior[a]) / (double)denom;
    out[a] = (double)qs + u;
  }
}

int or_maybe_add_child(const OEnv* env, ONode* node, int a, ONode** out) {
  if (!node->children[a]) {
    OPos np;
    int rc = or_play_move(&node->pos, a, &np);
    if (rc != OR_OK) return rc;
    ONode* ch = or_node_new(env, &np);
    ch->parent = node;
    ch->fmove = a;
    node->children[a] = ch;
  }
  if (out) *out = node->children[a];
  return OR_OK;
}

ONode* or_select_leaf(const OEnv* env, ONode* root, ODraw* draw) {
  ONode* cur = root;
  int A = env->A, pass = A - 1, depth = 0;
  double cas[OR_MAXA];
  int8_t legal[OR_MAXA];
  int possible[OR_MAXA];
  for (;;) {
    or_node_set_N(cur, or_node_N(cur) + 1.0f);
    if (!cur->is_expanded) break;
    /* HACK: if the last move was a pass, investigate double-pass first, mcts.jl:119-126 */
    int rl = cur->pos.recent_len;
    if (rl != 0 && cur->pos.recent_move[rl - 1] == pass && cur->child_N[pass] == 0.0f) {
      ONode* nx = NULL;
      or_maybe_add_child(env, cur, pass, &nx);
      cur = nx;
      depth++;
      continue;
    }
    or_child_action_score(env, cur, cas);
    or_all_legal_moves(&cur->pos, legal);
    double best = 0.0;
    int have = 0, np = 0;
    for (int a = 0; a < A; ++a)
      if (legal[a] && (!have || cas[a] > best)) { best = cas[a]; have = 1; }
    for (int a = 0; a < A; ++a)
      if (legal[a] && cas[a] == best) possible[np++] = a;
    int pick = possible[0];
    if (np > 1) {
      uint64_t bits = agz_draw_u64(draw->seed, draw->game, draw->move, AGZ_SITE_PUCT_TIE,
                                   (uint64_t)draw->sel * 1024u + (uint64_t)depth);
      pick = possible[agz_index(bits, (uint32_t)np)];
    }
    ONode* nx = NULL;
    or_maybe_add_child(env, cur, pick, &nx);
    cur = nx;
    depth++;
  }
  draw->sel++;
  return cur;
}

void or_add_virtual_loss(ONode* node, ONode* up_to) {
  for (;;) {
    node->losses_applied += 1;
    set_W(node, or_node_W(node) + (float)node->pos.to_play);
    if (node->parent == NULL || node == up_to) return;
    node = node->parent;
  }
}

void or_revert_virtual_loss(ONode* node, ONode* up_to) {
  for (;;) {
    node->losses_applied -= 1;
    set_W(node, or_node_W(node) + (float)(-node->pos.to_play));
    if (node->parent == NULL || node == up_to) return;
    node = node->parent;
  }
}

void or_revert_visits(ONode* node, ONode* up_to) {
  for (;;) {
    or_node_set_N(node, or_node_N(node) - 1.0f);
    if (node->parent == NULL || node == up_to) return;
    node = node->parent;
  }
}

void or_backup_value(ONode* node, float value, ONode* up_to) {
  for (;;) {
    set_W(node, or_node_W(node) + value);
    if (node->parent == NULL || node == up_to) return;
    node = node->parent;
  }
}

int or_incorporate_results(const OEnv* env, ONode* node, const float* probs, int nprobs,
                           float value, ONode* up_to) {
  int A = env->A;
  if (nprobs != A) return OR_BAD_SHAPE;
  if (node->pos.done) return OR_ASSERT_DONE_NODE;
  if (node->is_expanded) { or_revert_visits(node, up_to); return OR_OK; }
  node->is_expanded = 1;
  for (int a = 0; a < A; ++a) {
    node->original_prior[a] = node->child_prior[a] = probs[a];
    node->child_W[a] = value;   /* initialise child Q as the parent's value, mcts.jl:203-211 */
  }
  or_backup_value(node, value, up_to);
  return OR_OK;
}

int or_node_is_done(const OEnv* env, const ONode* node) {
  return node->pos.done || node->pos.n >= env->max_game_length;
}

/* inject_noise!, mcts.jl:233-239.  d ~ Dirichlet(alpha * 1_A) over ALL actions, built from
 * the draw stream's gammas; prior = Float32(Float64(prior)*(1-w) + d*w). */
void or_inject_noise(const OEnv* env, ONode* node, const ODraw* draw) {
  int A = env->A;
  double g[OR_MAXA], sum = 0.0;
  double alpha = (double)env->dirichlet_alpha;
  for (int a = 0; a < A; ++a) {
    g[a] = agz_dirichlet_gamma(draw->seed, draw->game, draw->move, (uint32_t)a, alpha);
    sum += g[a];
  }
  for (int a = 0; a < A; ++a) {
    double d = sum > 0.0 ? g[a] / sum : 1.0 / (double)A;
    node->child_prior[a] =
        (float)((double)node->child_prior[a] * (1.0 - env->noise_weight) + d * env->noise_weight);
  }
}

/* children_as_pi, mcts.jl:241-252.  Without squash: Float32 child_N ./ Float32 sum.  With
 * squash: child_N .^ 0.98 is Float64 (0.98 is a Float64 literal), normalised in Float64 and
 * narrowed when pushed into searches_pi::Vector{Vector{Float32}}.  pow comes from the draw
 * header's deterministic exp/log so the oracle and the GPU agree bit for bit; it differs from
 * Julia's pow by a few ulp of Float64, far below Float32 resolution. */
void or_children_as_pi(const ONode* node, int squash, float* out) {
  int A = node->pos.A;
  if (!squash) {
    float s = 0.0f;
    for (int a = 0; a < A; ++a) s += node->child_N[a];
    for (int a = 0; a < A; ++a) out[a] = node->child_N[a] / s;
  } else {
    double p[OR_MAXA], s = 0.0;
    for (int a = 0; a < A; ++a) { p[a] = agz_pow((double)node->child_N[a], 0.98); s += p[a]; }
    for (int a = 0; a < A; ++a) out[a] = (float)(p[a] / s);
  }
}

/* ------------------------------------------------------------------ player ---- */

struct OPlayer {
  OEnv env;
  or_net_fn net;
  void* net_ctx;
  int num_readouts;
  int two_player_mode;
  int tau_threshold;
  double resign_threshold;
  ONode* root;
  int result;
  char result_string[16];
  /* searches_pi / qs */
  int npi, nqs, cap;
  float* pis;
  float* qs;
  ODraw draw;
  uint64_t evals;
};

OPlayer* or_player_new(int N, or_net_fn net, void* net_ctx, int num_readouts, int two_player_mode,
                       double resign_threshold, uint64_t seed, uint64_t game) {
  OPlayer* p = (OPlayer*)calloc(1, sizeof(OPlayer));
  or_env_init(&p->env, N);
  p->net = net;
  p->net_ctx = net_ctx;
  p->num_readouts = num_readouts;
  p->two_player_mode = two_player_mode;
  p->tau_threshold = two_player_mode ? -1 : ((N * N / 12) / 2) * 2;   /* mcts_play.jl:19 */
  p->resign_threshold = resign_threshold;
  p->draw.seed = seed;
  p->draw.game = game;
  return p;
}

void or_player_free(OPlayer* p) {
  if (!p) return;
  or_node_free_tree(p->root);
  free(p->pis);
  free(p->qs);
  free(p);
}

void or_player_initialize_game(OPlayer* p, const OPos* pos) {
  OPos fresh;
  if (!pos) { or_pos_init(&fresh, p->env.N, 7.5f); pos = &fresh; }
  or_node_free_tree(p->root);
  p->root = or_node_new(&p->env, pos);
  p->result = 0;
  p->result_string[0] = 0;
  p->npi = p->nqs = 0;
  p->draw.move = (uint32_t)pos->n;
  p->draw.sel = 0;
}

ONode* or_player_root(OPlayer* p) { return p->root; }
const OEnv* or_player_env(const OPlayer* p) { return &p->env; }
int or_player_result(const OPlayer* p) { return p->result; }
const char* or_player_result_string(const OPlayer* p) { return p->result_string; }
int or_player_tau_threshold(const OPlayer* p) { return p->tau_threshold; }
int or_player_num_moves(const OPlayer* p) { return p->npi; }
const float* or_player_search_pi(const OPlayer* p, int k) { return p->pis + (size_t)k * p->env.A; }
float or_player_q(const OPlayer* p, int k) { return p->qs[k]; }
int or_player_nqs(const OPlayer* p) { return p->nqs; }
uint64_t or_player_evals(const OPlayer* p) { return p->evals; }

/* tree_search!, mcts_play.jl:73-98 */
int or_player_tree_search(OPlayer* p, int parallel_readouts) {
  int A = p->env.A, nleaves = 0, failsafe = 0;
  ONode** leaves = (ONode**)malloc(sizeof(ONode*) * (size_t)parallel_readouts);
  while (nleaves < parallel_readouts && failsafe < 2 * parallel_readouts) {
    failsafe++;
    ONode* leaf = or_select_leaf(&p->env, p->root, &p->draw);
    if (or_node_is_done(&p->env, leaf)) {
      float value = (float)or_result(&leaf->pos);
      or_backup_value(leaf, value, p->root);
      continue;
    }
    or_add_virtual_loss(leaf, p->root);
    leaves[nleaves++] = leaf;
  }
  if (nleaves) {
    const OPos** positions = (const OPos**)malloc(sizeof(OPos*) * (size_t)nleaves);
    float* pi = (float*)malloc(sizeof(float) * (size_t)A * (size_t)nleaves);
    float* v = (float*)malloc(sizeof(float) * (size_t)nleaves);
    for (int k = 0; k < nleaves; ++k) positions[k] = &leaves[k]->pos;
    p->net(p->net_ctx, positions, nleaves, pi, v);
    p->evals += (uint64_t)nleaves;
    for (int k = 0; k < nleaves; ++k) {
      or_revert_virtual_loss(leaves[k], p->root);
      or_incorporate_results(&p->env, leaves[k], pi + (size_t)k * A, A, v[k], p->root);
    }
    free(positions); free(pi); free(v);
  }
  free(leaves);
  return nleaves;
}

/* pick_move, mcts_play.jl:52-71 */
int or_player_pick_move(OPlayer* p, int* a_out) {
  int A = p->env.A;
  ONode* root = p->root;
  if (root->pos.n >= p->tau_threshold) {
    float mx = root->child_N[0];
    int possible[OR_MAXA] = {0}, np = 0;
    for (int a = 1; a < A; ++a) if (root->child_N[a] > mx) mx = root->child_N[a];
    for (int a = 0; a < A; ++a) if (root->child_N[a] == mx) possible[np++] = a;
    int pick = possible[0];
    if (np > 1) {
      uint64_t bits = agz_draw_u64(p->draw.seed, p->draw.game, (uint32_t)root->pos.n,
                                   AGZ_SITE_PICK_TIE, 0);
      pick = possible[agz_index(bits, (uint32_t)np)];
    }
    *a_out = pick;
    return OR_OK;
  }
  /* soft pick: cdf = cumsum(child_N); cdf /= cdf[end-1]; searchsortedfirst(cdf, rand()) */
  float cdf[OR_MAXA], acc = 0.0f;
  for (int a = 0; a < A; ++a) { acc += root->child_N[a]; cdf[a] = acc; }
  float denom = cdf[A - 2];
  for (int a = 0; a < A; ++a) cdf[a] = cdf[a] / denom;
  double u = agz_u01(agz_draw_u64(p->draw.seed, p->draw.game, (uint32_t)root->pos.n,
                                  AGZ_SITE_SOFTPICK, 0));
  int f = A;   /* searchsortedfirst returns length+1 when nothing qualifies */
  for (int a = 0; a < A; ++a)
    if (!((double)cdf[a] < u)) { f = a; break; }   /* isless-style: NaN is not less */
  if (f >= A || root->child_N[f] == 0.0f) return OR_ASSERT_SOFTPICK;
  *a_out = f;
  return OR_OK;
}

/* play_move!(player, c), mcts_play.jl:26-50 */
int or_player_play_move(OPlayer* p, int a) {
  int A = p->env.A;
  ONode* root = p->root;
  if (p->npi + 1 > p->cap || p->nqs + 1 > p->cap) {
    p->cap = p->cap ? p->cap * 2 : 128;
    p->pis = (float*)realloc(p->pis, sizeof(float) * (size_t)A * (size_t)p->cap);
    p->qs = (float*)realloc(p->qs, sizeof(float) * (size_t)p->cap);
  }
  if (!p->two_player_mode) {
    or_children_as_pi(root, root->pos.n <= p->tau_threshold, p->pis + (size_t)p->npi * A);
    p->npi++;
  }
  p->qs[p->nqs++] = or_node_Q(root);
  ONode* child = NULL;
  if (or_maybe_add_child(&p->env, root, a, &child) != OR_OK) {
    if (!p->two_player_mode) p->npi--;
    p->nqs--;
    return 0;
  }
  /* root.parent.children = Dict(): siblings are dropped, the path to the old root stays */
  for (int b = 0; b < A; ++b)
    if (b != a && root->children[b]) { node_free_subtree(root->children[b]); root->children[b] = NULL; }
  p->root = child;
  p->draw.move = (uint32_t)child->pos.n;
  p->draw.sel = 0;
  return 1;
}

/* should_resign, mcts_play.jl:124: Q_perspective(root) < resign_threshold */
int or_player_should_resign(const OPlayer* p) {
  float qp = or_node_Q(p->root) * (float)p->root->pos.to_play;
  return (double)qp < p->resign_threshold;
}

int or_player_is_done(const OPlayer* p) {
  return p->result != 0 || or_node_is_done(&p->env, p->root);
}

void or_player_set_result(OPlayer* p, int winner, int was_resign) {
  p->result = winner;
  if (was_resign) snprintf(p->result_string, sizeof(p->result_string), winner == 1 ? "B+R" : "W+R");
  else or_result_string(&p->root->pos, p->result_string);
}

/* extract_data, mcts_play.jl:126-139 + replay_position board.jl:557-578 */
int or_player_extract_data(const OPlayer* p, OPos* positions, float* pis, int* results) {
  const OPos* fin = &p->root->pos;
  int A = p->env.A;
  if (p->npi != fin->n) return -OR_HISTORY_INCOMPLETE;
  if (fin->n != fin->recent_len) return -OR_HISTORY_INCOMPLETE;
  OPos cur, nxt;
  or_pos_init(&cur, p->env.N, fin->komi);
  for (int k = 0; k < fin->recent_len; ++k) {
    if (positions) memcpy(&positions[k], &cur, sizeof(OPos));
    if (results) results[k] = p->result;
    int rc = or_play_move_color(&cur, fin->recent_move[k], fin->recent_color[k], &nxt);
    if (rc != OR_OK) return -rc;
    memcpy(&cur, &nxt, sizeof(OPos));
  }
  if (pis) memcpy(pis, p->pis, sizeof(float) * (size_t)A * (size_t)p->npi);
  return fin->n;
}

/* selfplay, selfplay.jl:1-45.  resign_threshold: "rand() < 0.05 : -1.0 : -0.9" is read as
 * the intended ternary. */
OPlayer* or_selfplay_ex(int N, or_net_fn net, void* net_ctx, int num_readouts, uint64_t seed,
                        uint64_t game, int max_moves, double threshold, double disable_fraction) {
  double u = agz_u01(agz_draw_u64(seed, game, 0, AGZ_SITE_RESIGN, 0));
  double resign_threshold = u < disable_fraction ? -1.0 : threshold;
  OPlayer* p = or_player_new(N, net, net_ctx, num_readouts, 0, resign_threshold, seed, game);
  int A = p->env.A;
  or_player_initialize_game(p, NULL);
  /* pre-expand the root so that noise affects the first move, selfplay.jl:16-20 */
  {
    ONode* first = or_select_leaf(&p->env, p->root, &p->draw);
    const OPos* pp = &first->pos;
    float* pi = (float*)malloc(sizeof(float) * (size_t)A);
    float v;
    net(net_ctx, &pp, 1, pi, &v);
    p->evals += 1;
    or_incorporate_results(&p->env, first, pi, A, v, first);
    free(pi);
  }
  int moves = 0;
  for (;;) {
    or_inject_noise(&p->env, p->root, &p->draw);
    float current = or_node_N(p->root);
    while (or_node_N(p->root) < current + (float)num_readouts) or_player_tree_search(p, 8);
    if (or_player_should_resign(p)) {
      or_player_set_result(p, -p->root->pos.to_play, 1);
      break;
    }
    int a = -1;
    if (or_player_pick_move(p, &a) != OR_OK) {
      /* the reference would die on its assertion here; the engine passes instead */
      a = A - 1;
    }
    or_player_play_move(p, a);
    moves++;
    if (or_node_is_done(&p->env, p->root)) {
      or_player_set_result(p, or_result(&p->root->pos), 0);
      break;
    }
    if (max_moves > 0 && moves >= max_moves) break;
  }
  return p;
}

OPlayer* or_selfplay(int N, or_net_fn net, void* net_ctx, int num_readouts, uint64_t seed,
                     uint64_t game, int max_moves) {
  return or_selfplay_ex(N, net, net_ctx, num_readouts, seed, game, max_moves, -0.9, 0.05);
}

/* evaluate(), neural_net.jl:103-158, the body of `for i = 1:num_games` */
void or_evaluate_game(int N, or_net_fn black_net, void* black_ctx, or_net_fn white_net, void* white_ctx,
                      int num_readouts, double resign_threshold, uint64_t seed, uint64_t game,
                      int16_t* moves_out, float* qs_out, OEvalGame* out) {
  OPlayer* black = or_player_new(N, black_net, black_ctx, num_readouts, 1, resign_threshold, seed, 2 * game);
  OPlayer* white = or_player_new(N, white_net, white_ctx, num_readouts, 1, resign_threshold, seed, 2 * game + 1);
  or_player_initialize_game(black, NULL);
  or_player_initialize_game(white, NULL);
  int A = black->env.A;
  int num_move = 0;
  memset(out, 0, sizeof(*out));
  for (;;) {
    OPlayer* active = (num_move % 2) ? white : black;      /* :118-119 */
    OPlayer* inactive = (num_move % 2) ? black : white;
    float current = or_node_N(active->root);                /* :121-126 */
    while (or_node_N(active->root) < current + (float)active->num_readouts) or_player_tree_search(active, 8);
    if (or_player_should_resign(active)) {                  /* :129-133 */
      int winner = -active->root->pos.to_play;
      or_player_set_result(active, winner, 1);
      or_player_set_result(inactive, winner, 1);
      out->was_resign = 1;
      break;
    }
    int a = -1;
    if (or_player_pick_move(active, &a) != OR_OK) a = A - 1;
    if (qs_out) qs_out[num_move] = or_node_Q(active->root);
    or_player_play_move(active, a);                         /* :135-137 */
    or_player_play_move(inactive, a);
    if (moves_out) moves_out[num_move] = (int16_t)a;
    num_move++;
    if (or_node_is_done(&active->env, active->root)) {      /* :140-145 */
      int winner = or_result(&active->root->pos);
      or_player_set_result(active, winner, 0);
      or_player_set_result(inactive, winner, 0);
      break;
    }
  }
  out->num_moves = num_move;
  out->result = black->result;
  out->final_score = or_score(&black->root->pos);
  out->black_won = or_result(&black->root->pos) == 1;      /* :147 */
  out->evals_black = black->evals;
  out->evals_white = white->evals;
  or_player_free(black);
  or_player_free(white);
}

/* ---- thin exports of the draw-stream header for tests/test_draws.py ---- */
uint64_t or_draw_u64(uint64_t seed, uint64_t game, uint32_t move, uint32_t site, uint64_t idx) {
  return agz_draw_u64(seed, game, move, site, idx);
}
double or_draw_u01(uint64_t bits) { return agz_u01(bits); }
uint32_t or_draw_index(uint64_t bits, uint32_t n) { return agz_index(bits, n); }
double or_det_log(double x) { return agz_log(x); }
double or_det_exp(double x) { return agz_exp(x); }
double or_det_pow(double x, double y) { return agz_pow(x, y); }
double or_det_sqrt(double x) { return agz_sqrt(x); }
double or_dirichlet_gamma(uint64_t seed, uint64_t game, uint32_t move, uint32_t a, double alpha) {
  return agz_dirichlet_gamma(seed, game, move, a, alpha);
}
