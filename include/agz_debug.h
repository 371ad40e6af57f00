/* agz_debug.h -- test hooks of libagz.so.  NOT part of the drop-in boundary (include/agz.h): nothing a host of
 * the reference would call.  The parity tests use them to check that gfx950 and the CPU oracle evaluate the shared
 * draw stream (include/agz_draws.h) and the mixed Float32/Float64 PUCT arithmetic bit for bit. */
#ifndef AGZ_DEBUG_H
#define AGZ_DEBUG_H

#include "agz.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- diagnostics ----------- */
/* Evaluate the draw stream (include/agz_draws.h) ON THE DEVICE so tests can check that gfx950
 * and the host produce bit-identical draws: gamma_out[a] = a-th un-normalised Dirichlet
 * component for (seed, game, move). */
agz_status agz_debug_draws(agz_engine* e, uint64_t seed, uint64_t game, uint32_t move, int32_t n,
                           double alpha, double* gamma_out);
/* op 0: agz_log(x) 1: agz_exp(x) 2: agz_pow(x, 0.98) 3: (double)sqrtf((float)x)
 * 4: (double)((float)x / (float)y) 5: PUCT score of (W=x, N=y, P=0.25, to_play=-1, N_node=y+7) */
agz_status agz_debug_math(agz_engine* e, int32_t op, const double* x, const double* y, int32_t n,
                          double* out);
/* The engine's raw device counters (enum Counter of agz_state.h), including the k_pre phase clocks that only a
 * -DAGZ_TIMING_EXPERIMENTS build of libagz.so writes (tools/pre_phases.py); returns how many there are, copies
 * min(cap, that) of them.  Synchronises. */
int32_t agz_debug_counters(agz_engine* e, uint64_t* out, int32_t cap);
/* Bench / soak-test population shaping, never part of the reference's behaviour: every game started from now on (also
 * the recycled ones) begins with a random legal opening prefix of up to `moves` plies and its first search gets a
 * random fraction of the readout budget, so that concurrent games sit at mixed stages and phases from the first timed
 * step (that first, shortened move is not counted as a position).  0 = off (the default).  Call it before the first
 * step of a run (AGZ_BAD_ARGUMENT once agz_selfplay_step has run since the last agz_selfplay_start); not available
 * in arena_mode. */
agz_status agz_debug_set_stagger(agz_engine* e, int32_t moves);

/* The record of a game that is still being played: slot g's game id and the moves it has recorded so far (return value
 * through *num_moves_out), and -- for 0 <= k < that -- move k, its pi[A] and q (any of the three may be NULL; k < 0 reads
 * only the header).  Lets a parity test hold the first moves of full-size runs against the oracle without waiting for
 * the games to end (tests/test_gpu_selfplay.py).  Synchronises. */
agz_status agz_debug_live_record(agz_engine* e, int32_t g, int32_t k, uint64_t* game_id_out, int32_t* num_moves_out,
                                 int32_t* move_out, float* pi_out, float* q_out);

/* Measurement context for bench.py's roofline object: the f32 MFMA rate (TFLOP/s) this board SUSTAINS on nothing but
 * independent v_mfma_f32_32x32x2_f32 from registers -- back-to-back ~10 ms launches for `millis` (50..5000), median of the
 * second half.  On an MI355X at its power limit: ~124, i.e. 0.79 of the nominal 157.3 (HISTORY.md 4f).  Synchronises. */
agz_status agz_debug_mfma_sustained(agz_engine* e, int32_t millis, float* tflops_out);
/* The same with operands that change in front of every MFMA as a layer's do (a matrix pipe's power follows its operands'
 * toggling, and the rate above is measured on constants): mode 1 = f32 MFMA, pseudo-random A, half-zero B (post-ReLU-like);
 * mode 2 = v_mfma_f32_32x32x16_f16 likewise; modes 3 / 4 = f32 / fp16 with a dense pseudo-random B (a Winograd-transformed
 * operand); mode 0 = the constant-operand measurement above.  TFLOP/s. */
agz_status agz_debug_mfma_sustained_data(agz_engine* e, int32_t millis, int32_t mode, float* tflops_out);
/* The inference weight images are built on the device from the device master copy of the parameters (DESIGN.md "weights").
 * This hook rebuilds image family `which` on the HOST from the host copies (the round-1..4 pack code, kept as the
 * reference) and counts the 32-bit words in which the device image differs: 0 = direct Wt, 1 = F(3x3,3x3) U (+ stem),
 * 2 = F(4x4,3x3) U, 3 = fp16 images (precision f16 selected), 4 = split-operand U + scales (precision f32s selected),
 * 5 = folded BatchNorm affines + head block.  -1 for an unknown family. */
agz_status agz_debug_pack_diff(agz_engine* e, int32_t which, int64_t* mismatches_out);

#ifdef __cplusplus
}
#endif

#endif
