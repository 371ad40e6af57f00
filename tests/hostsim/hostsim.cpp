// hostsim.cpp -- TEST INFRASTRUCTURE: a host "wave simulator" for alphago.jl_amd/csrc/agz_search.h.
//
// The search / rules logic of the engine is written once as wave-level templates.  The product
// instantiates them with agz::HipWave inside HIP kernels (agz_engine.hip).  This file
// instantiates the very same source with a lane-serial SimWave so that, in a container without
// a GPU, the tree / rules / lifecycle logic can be diffed bit-for-bit against the CPU oracle
// (tests/test_hostsim_*.py).  It is built into tests/hostsim/libhostsim.so, is loaded only by
// tests, and is never linked into or loaded by libagz.so.  It contains no network: pi/v always
// come from the caller.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../alphago.jl_amd/csrc/agz_layout.h"
#include "../../alphago.jl_amd/csrc/agz_search.h"

namespace {

struct SimWave {
  static constexpr bool kRegisterRows = false;    // the generic select_leaf (agz_search.h)
  template <class F>
  void for_each(int n, F f) const { for (int i = 0; i < n; ++i) f(i); }
  void sync() const {}
  bool leader() const { return true; }
  int reduce_sum(int v) const { return v; }
  int reduce_min(int v) const { return v; }
  int reduce_max(int v) const { return v; }
  double reduce_max(double v) const { return v; }
  float reduce_sum_f(float v) const { return v; }
  float reduce_max_f(float v) const { return v; }
  bool any(bool v) const { return v; }
  template <class F>
  int push_desc(int n, F get, int32_t* base, int sp) const {
    for (int i = 0; i < n; ++i) {
      const int c = get(i);
      if (c >= 0) base[--sp] = c;
    }
    return sp;
  }
  void amin(int32_t* p, int v) const { if (v < *p) *p = v; }
  void amax(int32_t* p, int v) const { if (v > *p) *p = v; }
  void aor(int32_t* p, int v) const { *p |= v; }
  void count(unsigned long long* p, unsigned long long v) const { *p += v; }
  unsigned long long clock() const { return 0; }
  void count_max(unsigned long long* p, unsigned long long v) const { if (v > *p) *p = v; }
  unsigned long long fetch_add(unsigned long long* p, unsigned long long v) const {
    const unsigned long long old = *p;
    *p += v;
    return old;
  }
};

struct Sim {
  agz::View V{};
  std::vector<void*> bufs;
  std::vector<int8_t> sb;
  std::vector<int32_t> label, minlib, maxlib, path;
  std::vector<int8_t> flag;
  std::vector<double> dbuf;
  std::vector<float> pi, v;
  agz::Scratch S{};
  int batch = 0;
};

template <class T>
void alloc_one(Sim* s, T*& p, size_t n) {
  p = (T*)calloc(n ? n : 1, sizeof(T));
  s->bufs.push_back(p);
}

}  // namespace

extern "C" {

void* hs_create(const agz_config* cfg) {
  Sim* s = new Sim();
  agz::fill_dims(s->V, *cfg);
  agz::for_each_buffer(s->V, [&](auto*& p, size_t n) { alloc_one(s, p, n); });
  s->sb.resize(s->V.PP);
  s->label.resize(s->V.PP + 64);
  s->minlib.resize(s->V.PP);
  s->maxlib.resize(s->V.PP);
  s->flag.resize(s->V.AP);
  s->dbuf.resize(s->V.AP);
  s->path.resize(s->V.maxd);
  s->S = agz::Scratch{s->sb.data(), s->label.data(), s->minlib.data(), s->maxlib.data(),
                      s->flag.data(), s->dbuf.data(), s->path.data()};
  for (int g = 0; g < s->V.games; ++g) s->V.gs[g].phase = agz::G_RETIRED;
  return s;
}

void hs_destroy(void* h) {
  Sim* s = (Sim*)h;
  for (void* p : s->bufs) free(p);
  delete s;
}

void hs_dims(void* h, int32_t* out /*N,P,A,AP,cap,games,par,mgl,tau,maxd*/) {
  const agz::View& V = ((Sim*)h)->V;
  int32_t v[10] = {V.N, V.P, V.A, V.AP, V.cap, V.games, V.par, V.max_game_length, V.tau, V.maxd};
  memcpy(out, v, sizeof(v));
}

void hs_start(void* h, int64_t total_games) {
  Sim* s = (Sim*)h;
  s->V.total_games = total_games;
  memset(s->V.counters, 0, sizeof(unsigned long long) * agz::CT_COUNT);
  memset(s->V.ar_hdr, 0, sizeof(int32_t) * 5 * (s->V.games / 2 + 1));
  for (int g = 0; g < s->V.games; ++g) {
    memset(&s->V.gs[g], 0, sizeof(agz::GameState));
    s->V.gs[g].phase = agz::G_IDLE;
  }
}

// phase A+B for every slot, then the prefix scan that assigns batch rows
int hs_pre(void* h) {
  Sim* s = (Sim*)h;
  SimWave w;
  for (int g = 0; g < s->V.games; ++g) agz::game_pre(w, s->V, s->S, g);
  int base = 0;
  if (s->V.arena) {   // rows: [Black players' leaves | White players' leaves]
    for (int c = 0; c < 2; ++c) {
      const int before = base;
      for (int g = c; g < s->V.games; g += 2) {
        s->V.gs[g].leaf_base = base;
        base += s->V.gs[g].nleaves;
      }
      s->V.batch_count[c] = base - before;
    }
    s->batch = base;
    s->V.counters[agz::CT_STEPS] += 1;
    return base;
  }
  for (int g = 0; g < s->V.games; ++g) {
    s->V.gs[g].leaf_base = base;
    base += s->V.gs[g].nleaves;
  }
  s->batch = base;
  *s->V.batch_count = base;
  s->V.counters[agz::CT_STEPS] += 1;
  return base;
}

void hs_leaf_features(void* h, float* whcn) {
  Sim* s = (Sim*)h;
  SimWave w;
  const agz::View& V = s->V;
  for (int g = 0; g < V.games; ++g)
    for (int k = 0; k < V.gs[g].nleaves; ++k)
      agz::leaf_features(w, V, g, k, (float*)nullptr, whcn + (size_t)(V.gs[g].leaf_base + k) * 17 * V.P);
}

void hs_post(void* h, const float* pi, const float* v) {
  Sim* s = (Sim*)h;
  SimWave w;
  s->V.pi = pi;
  s->V.v = v;
  for (int g = 0; g < s->V.games; ++g) agz::game_post(w, s->V, s->S, g);
}

void hs_arena_counts(void* h, int32_t* out) {
  out[0] = ((Sim*)h)->V.batch_count[0];
  out[1] = ((Sim*)h)->V.batch_count[1];
}

void hs_counters(void* h, unsigned long long* out) {
  memcpy(out, ((Sim*)h)->V.counters, sizeof(unsigned long long) * agz::CT_COUNT);
}

int hs_live_games(void* h) {
  Sim* s = (Sim*)h;
  int n = 0;
  for (int g = 0; g < s->V.games; ++g) n += s->V.gs[g].phase != agz::G_RETIRED && s->V.gs[g].phase != agz::G_IDLE;
  return n;
}

long hs_records_count(void* h) {
  Sim* s = (Sim*)h;
  const unsigned long long f = s->V.counters[agz::CT_RECORDED];
  return (long)(f < (unsigned long long)s->V.fin_cap ? f : s->V.fin_cap);
}
void hs_record_header(void* h, long k, agz_game_header* out) { *out = ((Sim*)h)->V.fin_hdr[k]; }
void hs_record_game(void* h, long k, int16_t* moves, float* pis, float* qs) {
  const agz::View& V = ((Sim*)h)->V;
  const int nm = V.fin_hdr[k].num_moves, mgl = V.max_game_length;
  memcpy(moves, V.fin_moves + (size_t)k * mgl, sizeof(int16_t) * nm);
  memcpy(qs, V.fin_q + (size_t)k * mgl, sizeof(float) * nm);
  memcpy(pis, V.fin_pi + (size_t)k * mgl * V.A, sizeof(float) * (size_t)nm * V.A);
}

// ---- Go rules, batched
void hs_go_play(void* h, const int8_t* boards, const int8_t* to_play, const int32_t* ko, const int32_t* moves,
                int B, int8_t* boards_out, int32_t* ko_out, int32_t* ncap_out, int32_t* status_out) {
  Sim* s = (Sim*)h;
  SimWave w;
  const int P = s->V.P;
  for (int b = 0; b < B; ++b)
    agz::go_play_one(w, s->V, s->S, boards + (size_t)b * P, to_play[b], ko[b], moves[b], boards_out + (size_t)b * P,
                     ko_out + b, ncap_out + b, status_out + b);
}
void hs_go_legal(void* h, const int8_t* boards, const int8_t* to_play, const int32_t* ko, int B, int8_t* out) {
  Sim* s = (Sim*)h;
  SimWave w;
  for (int b = 0; b < B; ++b)
    agz::go_legal_one(w, s->V, s->S, boards + (size_t)b * s->V.P, to_play[b], ko[b], out + (size_t)b * s->V.A);
}
void hs_go_score(void* h, const int8_t* boards, const float* komi, int B, float* out) {
  Sim* s = (Sim*)h;
  SimWave w;
  for (int b = 0; b < B; ++b) agz::go_score_one(w, s->V, s->S, boards + (size_t)b * s->V.P, komi[b], out + b);
}

// ---- single-tree compat ops
int hs_tree_op(void* h, const agz::TreeArgs* args, int32_t* r0_out) {
  Sim* s = (Sim*)h;
  SimWave w;
  int32_t iout[4] = {0, 0, 0, 0};
  agz::TreeArgs T = *args;
  T.iout = iout;
  s->V.pi = s->pi.data();
  s->V.v = s->v.data();
  agz::tree_op(w, s->V, s->S, T);
  if (r0_out) *r0_out = iout[1];
  return iout[0];
}
void hs_set_batch_outputs(void* h, const float* pi, const float* v, int B) {
  Sim* s = (Sim*)h;
  s->pi.assign(pi, pi + (size_t)B * s->V.A);
  s->v.assign(v, v + B);
}
void hs_tree_leaf_features(void* h, int g, float* whcn) {
  Sim* s = (Sim*)h;
  SimWave w;
  for (int k = 0; k < s->V.gs[g].nleaves; ++k)
    agz::leaf_features(w, s->V, g, k, (float*)nullptr, whcn + (size_t)k * 17 * s->V.P);
}
void hs_game_state(void* h, int g, agz::GameState* out) { *out = ((Sim*)h)->V.gs[g]; }
void hs_game_set(void* h, int g, int field, double value) {
  agz::GameState& G = ((Sim*)h)->V.gs[g];
  if (field == 0) G.game_id = (uint64_t)value;
  if (field == 1) G.sel = (int32_t)value;
  if (field == 2) G.rootN = (float)value;
  if (field == 3) G.resign_threshold = value;
}
void hs_node_meta(void* h, int g, int node, agz::NodeMeta* out) {
  Sim* s = (Sim*)h;
  *out = s->V.meta[agz::node_index(s->V, g, node)];
}
void hs_node_set_n(void* h, int g, int node, int n) {
  Sim* s = (Sim*)h;
  s->V.meta[agz::node_index(s->V, g, node)].n = n;
}
float hs_node_N(void* h, int g, int node) { Sim* s = (Sim*)h; return *agz::slotN(s->V, g, node); }
float hs_node_W(void* h, int g, int node) { Sim* s = (Sim*)h; return *agz::slotW(s->V, g, node); }
void hs_node_set_N(void* h, int g, int node, float v) { Sim* s = (Sim*)h; *agz::slotN(s->V, g, node) = v; }
float* hs_node_row(void* h, int g, int node, int field) {
  Sim* s = (Sim*)h;
  const long ni = agz::node_index(s->V, g, node);
  float* base = field == 0 ? s->V.childN : field == 1 ? s->V.childW : s->V.childP;
  return base + ni * s->V.AP;
}
int32_t* hs_node_children(void* h, int g, int node) {
  Sim* s = (Sim*)h;
  return s->V.child + agz::node_index(s->V, g, node) * s->V.AP;
}
int8_t* hs_node_board(void* h, int g, int node) {
  Sim* s = (Sim*)h;
  return s->V.board + agz::node_index(s->V, g, node) * s->V.PP;
}
void hs_node_legal(void* h, int g, int node, int8_t* out) {
  Sim* s = (Sim*)h;
  const long ni = agz::node_index(s->V, g, node);
  for (int a = 0; a < s->V.A; ++a) out[a] = agz::legal_bit(s->V, ni, a);
}

}  // extern "C"
