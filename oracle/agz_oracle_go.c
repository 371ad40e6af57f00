/*
 * agz_oracle_go.c -- Go rules and board-plane features of the CPU oracle.
 * TEST INFRASTRUCTURE (see agz_oracle.h).  Restates
 *   /root/reference/src/game/go/board.jl:28-81,354-578, go.jl:1-26, coords.jl:5-12
 *   /root/reference/src/features.jl:3-26
 */
#include "agz_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BLACK 1
#define WHITE (-1)
#define EMPTY 0
#define UNKNOWN 4

/* NEIGHBORS table, go.jl:18-19 */
static int neighbors(int N, int p, int out[4]) {
  int i = p % N, j = p / N, k = 0;
  if (i + 1 < N) out[k++] = p + 1;
  if (i - 1 >= 0) out[k++] = p - 1;
  if (j + 1 < N) out[k++] = p + N;
  if (j - 1 >= 0) out[k++] = p - N;
  return k;
}

static int diagonals(int N, int p, int out[4]) {
  int i = p % N, j = p / N, k = 0;
  if (i + 1 < N && j + 1 < N) out[k++] = p + 1 + N;
  if (i + 1 < N && j - 1 >= 0) out[k++] = p + 1 - N;
  if (i - 1 >= 0 && j + 1 < N) out[k++] = p - 1 + N;
  if (i - 1 >= 0 && j - 1 >= 0) out[k++] = p - 1 - N;
  return k;
}

/* find_reached, board.jl:28-45: chain = connected same-coloured region of c,
 * reached = every differently-coloured neighbour of the chain. */
static void find_reached(int N, const int8_t* board, int c, int8_t* chain, int8_t* reached) {
  int P = N * N, color = board[c], top = 0;
  int frontier[OR_MAXP];
  memset(chain, 0, (size_t)P);
  memset(reached, 0, (size_t)P);
  frontier[top++] = c;
  chain[c] = 1;
  while (top) {
    int cur = frontier[--top], nb[4];
    int k = neighbors(N, cur, nb);
    for (int t = 0; t < k; ++t) {
      int q = nb[t];
      if (board[q] == color) {
        if (!chain[q]) { chain[q] = 1; frontier[top++] = q; }
      } else {
        reached[q] = 1;
      }
    }
  }
}

void or_env_init(OEnv* env, int N) {
  env->N = N;
  env->A = N * N + 1;
  env->max_game_length = (N * N * 7) / 5;                         /* mcts.jl:21 */
  env->dirichlet_alpha = (float)(0.03 * 361.0 / (double)env->A);  /* mcts.jl:22, go.jl:24 */
  env->c_puct = 0.96;
  env->noise_weight = 0.25;
}

void or_pos_init(OPos* pos, int N, float komi) {
  memset(pos, 0, sizeof(*pos));
  pos->N = N;
  pos->A = N * N + 1;
  pos->komi = komi;
  pos->ko = -1;
  pos->to_play = BLACK;
}

void or_pos_from_board(OPos* pos, int N, const int8_t* board, int n, float komi, int cap_b,
                       int cap_w, int ko, int to_play, int nrecent, const int16_t* recent_move,
                       const int8_t* recent_color) {
  or_pos_init(pos, N, komi);
  if (board) memcpy(pos->board, board, (size_t)(N * N));
  pos->n = n;
  pos->caps[0] = cap_b;
  pos->caps[1] = cap_w;
  pos->ko = ko;
  pos->to_play = to_play;
  pos->recent_len = nrecent;
  for (int k = 0; k < nrecent; ++k) {
    pos->recent_move[k] = recent_move[k];
    pos->recent_color[k] = recent_color[k];
  }
}

/* deepcopy(GoPosition), board.jl:308-315: note the constructor resets done to false */
static void pos_copy(const OPos* src, OPos* dst) {
  if (dst != src) memcpy(dst, src, sizeof(OPos));
  dst->done = 0;
}

static void push_delta(OPos* pos, const int8_t* delta) {
  /* cat(new, get_first_n(deltas, planes-2 = 6)), board.jl:435,505-506 */
  int P = pos->N * pos->N;
  int keep = pos->ndeltas < 6 ? pos->ndeltas : 6;
  for (int k = keep; k >= 1; --k) memcpy(pos->deltas[k], pos->deltas[k - 1], (size_t)P);
  if (delta) memcpy(pos->deltas[0], delta, (size_t)P);
  else memset(pos->deltas[0], 0, (size_t)P);
  pos->ndeltas = keep + 1;
}

static void push_recent(OPos* pos, int color, int move) {
  if (pos->recent_len < OR_MAXRECENT) {
    pos->recent_move[pos->recent_len] = (int16_t)move;
    pos->recent_color[pos->recent_len] = (int8_t)color;
  }
  pos->recent_len++;
}

void or_pass_move(const OPos* pos, OPos* out) {
  int P = pos->N * pos->N;
  pos_copy(pos, out);
  out->n += 1;
  push_recent(out, out->to_play, P);
  push_delta(out, NULL);
  out->to_play = -out->to_play;
  out->ko = -1;
  if (out->recent_len > 1 && out->recent_len <= OR_MAXRECENT &&
      out->recent_move[out->recent_len - 2] == P)
    out->done = 1;
}

void or_flip_playerturn(const OPos* pos, OPos* out) {
  pos_copy(pos, out);
  out->ko = -1;
  out->to_play = -out->to_play;
}

int or_is_koish(int N, const int8_t* board, int p) {
  if (board[p] != EMPTY) return 0;
  int nb[4], k = neighbors(N, p, nb);
  int c0 = board[nb[0]];
  if (c0 == EMPTY) return 0;
  for (int t = 1; t < k; ++t)
    if (board[nb[t]] != c0) return 0;
  return c0;
}

int or_is_eyeish(int N, const int8_t* board, int p) {
  int color = or_is_koish(N, board, p);
  if (color == 0) return 0;
  int dg[4], k = diagonals(N, p, dg), faults = 0;
  if (k < 4) faults += 1;
  for (int t = 0; t < k; ++t)
    if (board[dg[t]] != color && board[dg[t]] != EMPTY) faults += 1;
  return faults > 1 ? 0 : color;
}

int or_group_info(int N, const int8_t* board, int p, int8_t* stones, int8_t* libs) {
  int P = N * N, nl = 0;
  int8_t chain[OR_MAXP], reached[OR_MAXP];
  if (board[p] == EMPTY) return -1;
  find_reached(N, board, p, chain, reached);
  for (int q = 0; q < P; ++q) {
    int l = reached[q] && board[q] == EMPTY;
    if (stones) stones[q] = chain[q];
    if (libs) libs[q] = (int8_t)l;
    nl += l;
  }
  return nl;
}

int or_count_groups(int N, const int8_t* board) {
  int P = N * N, count = 0;
  int8_t seen[OR_MAXP], chain[OR_MAXP], reached[OR_MAXP];
  memset(seen, 0, sizeof(seen));
  for (int p = 0; p < P; ++p) {
    if (board[p] == EMPTY || seen[p]) continue;
    find_reached(N, board, p, chain, reached);
    for (int q = 0; q < P; ++q) seen[q] |= chain[q];
    count++;
  }
  return count;
}

/* is_move_suicidal, board.jl:354-374 (group liberties by flood fill) */
int or_is_move_suicidal(const OPos* pos, int p) {
  int N = pos->N, P = N * N;
  int nb[4], k = neighbors(N, p, nb);
  int8_t potential[OR_MAXP], stones[OR_MAXP], libs[OR_MAXP];
  memset(potential, 0, (size_t)P);
  for (int t = 0; t < k; ++t) {
    int q = nb[t];
    if (pos->board[q] == EMPTY) return 0;
    int nl = or_group_info(N, pos->board, q, stones, libs);
    if (pos->board[q] == pos->to_play) {
      for (int r = 0; r < P; ++r) potential[r] |= libs[r];
    } else if (nl == 1) {
      return 0;
    }
  }
  potential[p] = 0;
  for (int r = 0; r < P; ++r)
    if (potential[r]) return 0;
  return 1;
}

int or_is_move_legal(const OPos* pos, int a) {
  int P = pos->N * pos->N;
  if (a == P) return 1;
  if (pos->board[a] != EMPTY) return 0;
  if (a == pos->ko) return 0;
  if (or_is_move_suicidal(pos, a)) return 0;
  return 1;
}

/* all_legal_moves, board.jl:393-424 */
void or_all_legal_moves(const OPos* pos, int8_t* out) {
  int N = pos->N, P = N * N;
  for (int p = 0; p < P; ++p) {
    out[p] = pos->board[p] == EMPTY;
    if (!out[p]) continue;
    /* padded "adjacent" count: the edge always counts as a lost liberty */
    int nb[4], k = neighbors(N, p, nb), adj = 4 - k;
    for (int t = 0; t < k; ++t) adj += pos->board[nb[t]] != EMPTY;
    if (adj == 4 && or_is_move_suicidal(pos, p)) out[p] = 0;
  }
  if (pos->ko >= 0) out[pos->ko] = 0;
  out[P] = 1;
}

int or_play_move_color(const OPos* pos, int a, int color, OPos* out) {
  int N = pos->N, P = N * N;
  if (a == P) { or_pass_move(pos, out); return OR_OK; }
  if (a < 0 || a > P) return OR_ILLEGAL_MOVE;
  if (!or_is_move_legal(pos, a)) return OR_ILLEGAL_MOVE;
  OPos tmp;
  pos_copy(pos, &tmp);
  int potential_ko = or_is_koish(N, tmp.board, a);
  tmp.board[a] = (int8_t)color;
  /* add_stone!, board.jl:227-269: opponent neighbour groups left without liberties die */
  int8_t captured[OR_MAXP], stones[OR_MAXP];
  int ncap = 0, nb[4], k = neighbors(N, a, nb);
  memset(captured, 0, (size_t)P);
  for (int t = 0; t < k; ++t) {
    int q = nb[t];
    if (tmp.board[q] != -color || captured[q]) continue;
    if (or_group_info(N, tmp.board, q, stones, NULL) == 0)
      for (int r = 0; r < P; ++r)
        if (stones[r] && !captured[r]) { captured[r] = 1; ncap++; }
  }
  for (int r = 0; r < P; ++r)
    if (captured[r]) tmp.board[r] = EMPTY;
  /* suicide is illegal, board.jl:263-266 */
  if (or_group_info(N, tmp.board, a, NULL, NULL) == 0) return OR_ILLEGAL_MOVE;

  int8_t delta[OR_MAXP];
  memset(delta, 0, (size_t)P);
  delta[a] = (int8_t)color;
  int lastcap = -1;
  for (int r = 0; r < P; ++r)
    if (captured[r]) { delta[r] = (int8_t)color; lastcap = r; }
  int new_ko = (ncap == 1 && potential_ko == -color) ? lastcap : -1;
  if (tmp.to_play == BLACK) tmp.caps[0] += ncap; else tmp.caps[1] += ncap;
  tmp.n += 1;
  tmp.ko = new_ko;
  push_recent(&tmp, color, a);
  push_delta(&tmp, delta);
  tmp.to_play = -tmp.to_play;
  memcpy(out, &tmp, sizeof(OPos));
  return OR_OK;
}

int or_play_move(const OPos* pos, int a, OPos* out) {
  return or_play_move_color(pos, a, pos->to_play, out);
}

/* score, board.jl:511-533: Tromp-Taylor area minus komi, Black-positive */
float or_score(const OPos* pos) {
  int N = pos->N, P = N * N;
  int8_t work[OR_MAXP], terr[OR_MAXP], borders[OR_MAXP];
  memcpy(work, pos->board, (size_t)P);
  for (int p = 0; p < P; ++p) {
    if (work[p] != EMPTY) continue;
    find_reached(N, work, p, terr, borders);
    int xb = 0, ob = 0;
    for (int q = 0; q < P; ++q)
      if (borders[q]) { xb |= work[q] == BLACK; ob |= work[q] == WHITE; }
    int color = (xb && !ob) ? BLACK : (ob && !xb) ? WHITE : UNKNOWN;
    for (int q = 0; q < P; ++q)
      if (terr[q]) work[q] = (int8_t)color;
  }
  int nb = 0, nw = 0;
  for (int p = 0; p < P; ++p) { nb += work[p] == BLACK; nw += work[p] == WHITE; }
  return (float)(nb - nw) - pos->komi;
}

int or_result(const OPos* pos) {
  float s = or_score(pos);
  return s > 0 ? 1 : s < 0 ? -1 : 0;
}

void or_result_string(const OPos* pos, char* out) {
  float s = or_score(pos);
  if (s > 0) snprintf(out, 16, "B+%.1f", (double)s);
  else if (s < 0) snprintf(out, 16, "W+%.1f", (double)-s);
  else snprintf(out, 16, "DRAW");
}

/* ---- features.jl:3-26 ---- */
void or_get_feats_f64(const OPos* pos, double* out) {
  int N = pos->N, P = N * N;
  int last_eight[8][OR_MAXP];
  int avail = pos->ndeltas;
  for (int p = 0; p < P; ++p) last_eight[0][p] = pos->board[p];
  /* last_eight[k] = board - cumsum(deltas)[k], features.jl:8-12 */
  for (int k = 1; k <= avail; ++k)
    for (int p = 0; p < P; ++p) last_eight[k][p] = last_eight[k - 1][p] - pos->deltas[k - 1][p];
  /* no more deltas: repeat the oldest board, features.jl:14 */
  for (int k = avail + 1; k < 8; ++k)
    for (int p = 0; p < P; ++p) last_eight[k][p] = last_eight[avail][p];
  for (int k = 0; k < 8; ++k)
    for (int p = 0; p < P; ++p) {
      out[(size_t)(2 * k) * P + p] = last_eight[k][p] == pos->to_play;
      out[(size_t)(2 * k + 1) * P + p] = last_eight[k][p] == -pos->to_play;
    }
  /* colour-to-play plane is +1 / -1, features.jl:22 */
  for (int p = 0; p < P; ++p) out[(size_t)16 * P + p] = pos->to_play;
}

void or_get_feats(const OPos* pos, float* out) {
  int P = pos->N * pos->N;
  double tmp[17 * OR_MAXP];
  or_get_feats_f64(pos, tmp);
  for (int k = 0; k < 17 * P; ++k) out[k] = (float)tmp[k];
}
