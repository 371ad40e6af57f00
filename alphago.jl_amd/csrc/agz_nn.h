// agz_nn.h -- the dual-head ResNet of AlphaGo.jl as resident HIP state + hand-written
// gfx950 kernels (inference only).  Topology: /root/reference/src/neural_net.jl:13-33,57-73,
// /root/reference/src/resnet.jl:11-32.  See DESIGN.md "Network kernels".
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "agz_common.h"

namespace agz {

constexpr int kC = 256;        // tower width, neural_net.jl:16
constexpr int kCinStem = 17;   // 2*planes + 1, neural_net.jl:19
constexpr int kCinStemPad = 32;

struct ConvHost {
  int k = 3, cin = 0, cout = 0;
  std::vector<float> w, b, beta, gamma, mean, var;
  float eps = 1e-5f;
};
struct DenseHost {
  int in = 0, out = 0;
  std::vector<float> w, b;
};

bool wino4_applies(int N);      // agz_wino4.hip (declared with the rest of its interface below)
bool wino5_applies(int N);      // agz_wino5.hip

class Net {
 public:
  Net(int N, int tower, hipStream_t stream);
  ~Net();
  int N() const { return N_; }
  int P() const { return P_; }
  int A() const { return A_; }
  int tower() const { return tower_; }

  int64_t param_count(int layer, int kind) const;
  void set(int layer, int kind, const float* data, int64_t count);
  void get(int layer, int kind, float* out, int64_t count) const;
  void init_synthetic(uint64_t seed);

  // The MASTER copy of every parameter lives on the device: one flat f32 array in the Flux layouts of the C ABI, in
  // weight_keys order (per conv layer: w, b, beta, gamma, mean, var, eps; then the head convs; then the dense layers'
  // w, b).  agz_net_set_weights writes through to it; the trainer and agz_broadcast_weights write it directly
  // (device_master_written) and every inference image -- direct, F(3x3,3x3), F(4x4,3x3), fp16, split, folded affines --
  // is derived from it by kernels on the engine's stream: new weights reach the next forward without visiting the host
  // (/root/reference/src/train.jl:67-74 changes the weights every iteration).  The host vectors below are a cache.
  static void weight_keys(int tower, std::vector<std::pair<int, int>>& keys);
  float* flux_device() { return d_flux_.p; }
  size_t flux_count() const { return flux_n_; }
  size_t flux_offset(int layer, int kind) const;
  void device_master_written() { host_stale_ = true; derived_dirty_ = true; ++param_version_; }
  // (test hook) words of the device images that differ from the host restatement of the same packs; -1 if `which` is unknown
  long debug_pack_diff(int which);

  // Reserve activation workspace for up to `bcap` positions.
  void reserve(int bcap);
  // d_x32: [bcap*P][32] stem input (17 planes + zero pad) ; d_count: device int, number of
  // positions actually present (<= bcap); outputs d_pi [bcap][A] (row per position), d_v.
  void forward(const float* d_x32, const int* d_count, int bcap, float* d_pi, float* d_v);
  // one tower conv launch on resident synthetic activations (for roofline timing)
  void launch_tower_conv_once(const int* d_count, int bcap);

  // tower convolution algorithm: 1 = Winograd (default: F(3x3,3x3), and F(4x4,3x3) for boards of 13x13 and larger in the
  // exact-f32 arithmetic), 2 = Winograd F(3x3,3x3) on every board size (A/B runs), 0 = the direct implicit GEMM
  // 3 = 1 with the tower layers of small boards (whole-board tile blocks, N <= 12) on the five-pass 64 x 128 form of
  // F(3x3,3x3) (agz_wino5.hip) instead of k_wino_gemm4
  void set_winograd(int mode) { winograd_ = mode != 0; wino_f33_only_ = mode == 2; wino5_ = mode == 3; }
  bool use_wino5() const { return winograd_ && wino5_ && precision_ == 0 && tower_ > 0 && wino5_applies(N_); }
  bool winograd() const { return winograd_; }
  // The Winograd tower of a large batch as n independent layer chains (ranges of its tile blocks, cut at board boundaries)
  // on n streams: the hardware scheduler interleaves their workgroups, the CUs stop marching through K loops and store
  // bursts in lockstep (-5 % per forward at 19x19 / 2048 positions, -1.6 % per step at 9x9 / 8192; outputs are
  // bit-identical: same kernels, same rows).  Applies to whole-board F(3x3,3x3) blocks and to the F(4x4,3x3) tower.
  void set_tower_streams(int n) { tower_streams_ = n < 1 ? 1 : n > kMaxTowerStreams ? kMaxTowerStreams : n; }
  int tower_streams() const { return tower_streams_; }
  bool use_wino4() const { return winograd_ && !wino_f33_only_ && precision_ == 0 && tower_ > 0 && wino4_applies(N_); }
  // the f32 Winograd tower as ONE persistent launch (k_wino_tower) instead of one launch per layer, where it applies
  // (whole-board tile blocks, 256 CUs with 32 resident workgroups per XCD); same arithmetic, same bits.  Off by
  // default: it needs 3.7 % fewer cycles and the clock comes down by as much (HISTORY.md 4f) -- same wall time.
  void set_tower_persistent(bool on) { tower_persistent_ = on; }
  // Throws if a persistent tower launch that has completed raised its scheduler's error word (its outputs were
  // garbage).  Called by every forward and by the engine's synchronising calls; the caller has synchronised the stream,
  // or accepts hearing about the error one call later.
  void check_async_error();
  bool tower_persistent() const { return tower_persistent_; }
  // tower arithmetic: 0 = exact f32 (default), 1 = fp16 operands / f32 accumulate (agz_conv16.hip)
  // 2 = exact-f32 network with the Winograd operands carried as two f16 halves (agz_wino.hip, split form)
  void set_precision(int p) { precision_ = p; }
  int precision() const { return precision_; }

  // what this board sustains on nothing but independent v_mfma_f32_32x32x2_f32 (one launch of ~10 ms after another for
  // `millis`, the median of the second half): the power-limited f32 MFMA rate bench.py quotes next to the nominal peak
  float mfma_sustained_tflops(int millis, int mode = 0);
  // HIP-event timing of every tower-conv launch inside forward() (bench.py roofline leg)
  void profile_enable(bool on);
  void profile_read(double* total_ms, double* total_flop, int64_t* launches);

  // training (agz_train.hip): host parameters by layer id, and "the device packs are stale"
  ConvHost* conv(int layer);
  const ConvHost* conv(int layer) const;
  DenseHost* dense(int layer);
  const DenseHost* dense(int layer) const;
  // the host vectors are refreshed lazily from the device master: whoever is about to read a host parameter calls
  // sync_host() first.  param_version() moves whenever the master is written (set, init_synthetic, trainer, broadcast).
  void sync_host() const;
  uint64_t param_version() const { return param_version_; }

  double flops_per_eval() const;         // BASELINE.md F_eval
  double conv_flops_per_launch(int B) const { return 2.0 * B * P_ * 9.0 * kC * kC; }

 private:
  void pack();

  int N_, P_, A_, tower_;
  hipStream_t stream_;
  ConvHost stem_, vconv_, pconv_;
  std::vector<ConvHost> tconv_;
  DenseHost vfc1_, vfc2_, pfc_;
  struct FluxSlot { int layer, kind; size_t off, n; };
  std::vector<FluxSlot> slots_;
  size_t flux_n_ = 0;
  DevBuf<float> d_flux_;
  bool derived_dirty_ = true;          // the inference images are older than the device master
  mutable bool host_stale_ = false;    // the host vectors are older than the device master
  uint64_t param_version_ = 0;
  void upload_slot(const FluxSlot& s);
  void upload_host_all();
  const float* host_slot(const FluxSlot& s) const;

  // device-resident packed parameters
  DevBuf<float> d_wstem_, d_wtower_;      // [cout][9*cin_pad] per layer
  DevBuf<float> d_scale_, d_shift_;       // (1 + 2*tower) x 256
  DevBuf<float> d_head_;                  // head conv weights + affine
  const float *d_vfc1w_ = nullptr, *d_vfc1b_ = nullptr, *d_vfc2w_ = nullptr, *d_vfc2b_ = nullptr, *d_pfcw_ = nullptr,
              *d_pfcb_ = nullptr;      // the dense layers, read in place in the master
  // workspace
  int bcap_ = 0;
  DevBuf<float> d_a_, d_b_, d_t_, d_vh_, d_ph_;
  bool winograd_ = true, wino_f33_only_ = false, wino5_ = false;
  DevBuf<float> d_uwino5_;                     // five-pass F(3x3,3x3) weights (agz_wino5.hip), packed when first used
  bool packed5_ = false;
  DevBuf<float> d_uwino4_;                     // F(4x4,3x3) transformed weights (agz_wino4.hip), packed when first used
  bool packed4_ = false;
  static constexpr int kMaxTowerStreams = 4;
  int tower_streams_ = 2;
  hipStream_t streamx_[kMaxTowerStreams - 1] = {nullptr, nullptr, nullptr};      // chains 1.. (chain 0 runs on stream_)
  hipEvent_t ev_fork_ = nullptr, ev_join_[kMaxTowerStreams - 1] = {nullptr, nullptr, nullptr};
  bool fork_chains(int parts);                 // side streams wait for stream_; true if this tower is being timed
  void join_chains(int parts, bool timed_tower);      // stream_ waits for the side streams
  DevBuf<float> d_uwino_s_, d_scale_s_;        // split form: weights as halves, scale x 1 / (operand scales)
  bool packed_split_ = false;
  DevBuf<float> d_uwino_, d_vimg_, d_vimg2_;   // transformed weights (stage images) / transformed activations (ping-pong)
  DevBuf<float> d_ustem_, d_ustem_s_;          // the stem's transformed weights (8 stages), f32 / split form
  int precision_ = 0;
  bool packed16_ = false;
  DevBuf<uint16_t> d_wi16_;               // fp16 tower weights as padded LDS tile images [layer][stage 72][256][40]
  DevBuf<uint16_t> d_ha_, d_hb_, d_ht_;   // fp16 tower activations [rows][256]
  // persistent tower launch
  bool tower_persistent_ = false;
  DevBuf<int> d_tower_sched_;
  DevBuf<char> d_tower_layers_;
  std::vector<char> tower_layers_host_;      // what d_tower_layers_ holds
  int32_t* tower_err_ = nullptr;             // pinned host: the scheduler's error word of the previous forward
  // profiling
  std::vector<int> prof_mult_;               // layers behind event pair i (1, or the whole tower for the persistent launch)
  bool prof_on_ = false;
  std::vector<hipEvent_t> prof_ev_;
  std::vector<int> prof_fwd_of_;
  int32_t* prof_counts_ = nullptr;   // pinned host
  int prof_n_ = 0, prof_fwd_ = 0;
  static constexpr int kProfMax = 4096;
};

// One optimisation step of `_train` on the device (agz_train.hip; /root/reference/src/neural_net.jl:75-101)
class Trainer {
 public:
  Trainer(Net& net, hipStream_t s);
  ~Trainer();
  // feats [B][17 P] (agz_features order), pi [B][A], z [B]: all host or all device pointers; losses_out[4] =
  // {total, policy, value, regulariser} before the update
  void step(const float* feats, const float* pi, const float* z, int B, bool is_device, float eta, float rho,
            float* losses_out);
  void reset();     // forget the optimiser state (Momentum velocities)

 private:
  struct Param;
  void upload();              // the network's device master -> training layouts (device to device)
  void publish(long M);       // training layouts + running statistics -> the network's device master
  uint64_t uploaded_version_ = ~0ull;
  Net& net_;
  hipStream_t stream_;
  std::vector<std::unique_ptr<Param>> params_;
  bool have_vel_ = false;
  DevBuf<float> d_x32_, d_u_, d_o_, d_ga_, d_gb_, d_gc_, d_stats_, d_ones_, d_zero_, d_wd_, d_small_, d_in_, d_part_, d_wpart_;
  DevBuf<double> d_sums_;
  DevBuf<int> d_cnt_;
  DevBuf<unsigned char> d_opta_, d_optw_;     // k_momentum_all's array table and work list (built with the parameter set)
  int n_optw_ = 0;
};

// `layers` Flux tensors [3][3][cin][256] on the device, wstride floats apart -> Wt[cout][tap][cin_pad] images (the direct
// kernel's and the trainer's layout)
void launch_pack_direct(const float* d_w, long wstride, int cin, int cin_pad, int layers, float* d_out, hipStream_t s);
// the direct implicit-GEMM 3x3 convolution of agz_nn.hip (y = act(scale * conv + shift (+ res))), cin_pad = 32 or 256
void launch_conv3x3_direct_taps(const float* x, const float* wt, const float* ones, const float* zeros, const float* shift,
                                float* y, float* part, const int* d_count, int bcap, int N, int cin_pad, hipStream_t s);
int conv3x3_direct_blocks(int bcap, int N);
void launch_conv3x3_direct(const float* x, const float* wt, const float* scale, const float* shift, const float* res,
                           float* y, const int* d_count, int bcap, int N, int relu, int cin_pad, hipStream_t s);

// Winograd F(3x3,3x3) tower convolution (agz_wino.hip)
constexpr int kWinoStages = 64, kWinoStemStages = 8;   // K-loop stages (input channels / 4) of a tower layer / the stem
void wino_pack_weights(const ConvHost& c, float* out, int ns = kWinoStages);          // host restatement (test reference)
// the product: `layers` Flux tensors [3][3][cin][256] on the device, wstride floats apart -> `layers` U images
void launch_wino_pack(const float* d_w, long wstride, int cin, int layers, float* d_out, int ns, bool split, hipStream_t s);
size_t wino_weight_floats(int ns = kWinoStages);
size_t wino_v_floats(int bcap, int T);
// x -> V (the 25 transformed planes as GEMM stage images); needed in front of the first Winograd layer, and
// in front of every layer when the board's tiles do not pack into whole-board tile blocks (!wino_fusable)
// fixup: dense tile blocks (!wino_fusable(N), e.g. 19x19) whose previous GEMM already emitted the V of every tile whose
// 5x5 patch lies inside its block: only the remaining rows (the ends of every block, the rows past the batch) are done
void launch_wino_in(const float* x, float* vimg, const int* d_count, int bcap, int N, bool split, hipStream_t s,
                    int ns = kWinoStages, bool fixup = false);
// V, U -> y (if y != NULL: affine, residual, ReLU applied) and / or the NEXT layer's V (if vnext != NULL)
void launch_wino_gemm(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                      float* y, float* vnext, const int* d_count, int bcap, int N, int relu, bool split, hipStream_t s,
                      int ns = kWinoStages, int part = 0, int parts = 1);
bool wino_fusable(int N);
// the tower in one persistent launch (agz_wino.hip: k_wino_tower).  Layer table entry, device side:
struct WinoTowerLayer {
  const float *v, *u, *scale, *shift, *res;
  float *y, *vnext;
  int mode, relu;      // mode bit 0: write y; bit 1: emit the next layer's V
};
size_t wino_tower_sched_ints(int layers, int bcap, int N);
bool wino_tower_supported(hipStream_t s);
void launch_wino_tower(const void* d_layers, int layers, int* d_sched, const int* d_count, int bcap, int N, bool split, hipStream_t s);
constexpr int kWinoTowerErrWord = 8;         // int offset of the scheduler's error word in d_sched
// split-operand form (AGZ_PRECISION_F32S): weights as (hi, lo) halves of 2^10 u; 1 / (operand scales) for the epilogue
void wino_pack_weights_split(const ConvHost& c, float* out, int ns = kWinoStages);
float wino_split_descale();

// Winograd F(3x3,3x3) tower layer in five one-row passes over a 64-tile x 128-cout workgroup tile (agz_wino5.hip): reads the
// V images of agz_wino.hip's kernels, its own U image; whole-board tile blocks (N <= 12), exact f32
bool wino5_applies(int N);
void wino5_pack_weights(const ConvHost& c, float* out);                              // host restatement (test reference)
void launch_wino5_pack(const float* d_w, long wstride, int layers, float* d_out, hipStream_t s);
size_t wino5_weight_floats();
void launch_wino5_gemm(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                       float* y, float* vnext, const int* d_count, int bcap, int N, int relu, hipStream_t s, int part = 0,
                       int parts = 1);

// Winograd F(4x4,3x3) tower convolution for boards of 13x13 and larger (agz_wino4.hip): 36 planes in six one-row passes over
// the input channels, each folded into the inverse transform when its K loop ends; tiles of 4x4 outputs, T = ceil(N / 4)
constexpr int kWino4Stages = 96;             // K-loop stages of a layer: six passes (transform rows) x 16 stages of 6 planes x 16 cin
bool wino4_applies(int N);                   // N >= 13: fewer multiplies per output point than F(3x3,3x3)
bool wino4_whole_boards(int N);              // tile blocks hold whole boards (N = 13..16); else dense blocks + fix-up transform
void wino4_pack_weights(const ConvHost& c, float* out);                              // host restatement (test reference)
void launch_wino4_pack(const float* d_w, long wstride, int layers, float* d_out, hipStream_t s);
size_t wino4_weight_floats();
size_t wino4_v_floats(int bcap, int N);
// x -> V (all tiles), or with fixup only the tiles the previous GEMM's epilogue could not emit (dense blocks)
void launch_wino4_in(const float* x, float* vimg, const int* d_count, int bcap, int N, hipStream_t s, bool fixup, int part = 0,
                     int parts = 1);
// V, U -> y (if y != NULL) and / or the next layer's V (if vnext != NULL).  y_for_fixup_only (five boards per block pair,
// no residual): y is stored only where launch_wino4_in(fixup) will read it -- 28 of a pair's 125 rows
void launch_wino4_gemm(const float* vimg, const float* uimg, const float* scale, const float* shift, const float* res,
                       float* y, float* vnext, const int* d_count, int bcap, int N, int relu, hipStream_t s, int part = 0,
                       int parts = 1, bool y_for_fixup_only = false);
bool wino4_paired(int N);                    // five boards per two tile blocks (N = 17..19)
void wino4_validate(int bcap, int N);        // throws if the batch's tile index / byte offsets leave 32 bits (before any launch)
// (part / parts: the part-th of `parts` ranges of tile blocks, cut at board boundaries: ranges are independent layer chains)

// fp16-operand tower convolution (agz_conv16.hip); x is half, res / y are float* or half* as flagged
void conv16_pack_images(const ConvHost& c, uint16_t* out);                           // host restatement (test reference)
void launch_conv16_pack(const float* d_w, long wstride, int layers, uint16_t* d_out, hipStream_t s);
size_t conv16_image_halves();
void launch_conv16_dma(const uint16_t* x, const uint16_t* wi, const float* scale, const float* shift, const void* res,
                       int res_f32, void* y, int out_f32, const int* d_count, int bcap, int N, int relu, hipStream_t s);
void launch_f32_to_f16(const float* x, uint16_t* y, const int* d_count, int bcap, int N, hipStream_t s);

// feature extraction entry points (features.jl:3-26) from the reference's own position format
void launch_features_from_deltas(const int8_t* d_boards, const int8_t* d_deltas, const int32_t* d_ndeltas,
                                 const int8_t* d_to_play, int B, int N, float* d_x32, float* d_whcn,
                                 hipStream_t stream);
// WHCN feature tensor -> stem input layout
void launch_whcn_to_x32(const float* d_whcn, int B, int N, float* d_x32, hipStream_t stream);

}  // namespace agz
