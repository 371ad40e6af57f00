// agz_wino_common.h -- what the two Winograd GEMM kernels (agz_wino.hip: k_wino_gemm4, agz_wino6.hip: k_wino_gemm6) and
// the input transform share: the V stage-image layout, the tile-block geometry and the LDS-DMA helpers.
#pragma once
#include "agz_nn.h"

namespace agz {

constexpr int WT = 64;            // tile rows per tile block
constexpr int WK = 4;             // input channels per stage
constexpr int WNS = kC / WK;      // 64 stages
constexpr int WXI = 25;
constexpr int WPL = 13;           // a V image holds 13 plane pairs (26 plane slots, the last one padding)
constexpr int A_STAGE = WPL * WT * 8;    // floats of V per (tile block, stage): 26,624 B

// V stage image (activations): [plane pair q 13][plane parity 2][tile row 64][4 dwords] = [plane 26][row 64][4]; the 4
// dwords of a row are the plane's 4 channels as two pairs, pair h at slot (h + (row >> 4)) & 1.  Row stride 4 dwords:
// rows r and r + 16 start on the same bank and take different slots, so a 32-row ds_read_b64 is conflict-free -- and a
// producer whose lane = tile row writes each 16-byte row whole, 64 lanes = 1 KB contiguous per store instruction.
__host__ __device__ __forceinline__ int wino_v_off(int xi, int row, int h) {      // dword offset inside a stage image
  return (xi >> 1) * (WT * 8) + (xi & 1) * (WT * 4) + row * 4 + 2 * ((h + (row >> 4)) & 1);
}

// rows of a 64-row tile block that carry tiles: whole boards when a board's tiles pack into 64 rows with
// <= 10 % waste (N <= 12: T*T = 1, 4, 9, 16 -> 64, 64, 63, 64 rows), else dense packing
__host__ __device__ inline int wino_rows_per_block(int T) {
  const int tt = T * T;
  const int whole = (WT / tt) * tt;
  return (tt <= WT && whole * 10 >= WT * 9) ? whole : WT;
}
__host__ __device__ inline bool wino_whole_boards(int T) { return wino_rows_per_block(T) % (T * T) == 0 && T * T <= WT; }

#ifdef __HIPCC__
// 16 bytes per lane straight from global memory into LDS (wave-uniform LDS base in M0 + lane*16).
// Issued through inline asm on purpose: hipcc cannot prove that the DMA target (the OTHER stage
// buffer) does not alias the ds_reads of the current stage and would put an s_waitcnt vmcnt(0) in
// front of them, serialising load and compute (measured: 38 % MFMA utilisation).  An asm statement
// is outside its vmcnt book-keeping, so the wait is placed by hand, once per stage, right before
// the barrier that hands the buffer over.
__device__ __forceinline__ void glds16(const float* g, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);   // make the SGPR operand provable
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_byte_addr)
      : "memory");
}

// SGPR base + per-lane 32-bit offset: no 64-bit VALU address arithmetic per piece
__device__ __forceinline__ void glds16s(const float* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_off), "s"(gbase_uniform), "s"(lds_byte_addr) : "memory");
}
#endif

}  // namespace agz
