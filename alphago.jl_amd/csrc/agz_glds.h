// agz_glds.h -- direct global -> LDS loads (global_load_lds_dwordx4) for the Winograd GEMMs (agz_wino.hip: F(3x3,3x3);
// agz_wino4.hip: F(4x4,3x3)).  Device code only.
#pragma once

namespace agz {

// 16 bytes per lane straight from global memory into LDS (wave-uniform LDS base in M0 + lane*16).
// Issued through inline asm on purpose: hipcc cannot prove that the DMA target (the OTHER stage
// buffer) does not alias the ds_reads of the current stage and would put an s_waitcnt vmcnt(0) in
// front of them, serialising load and compute (measured: 38 % MFMA utilisation).  An asm statement
// is outside its vmcnt book-keeping, so the wait is placed by hand, once per stage, right before
// the barrier that hands the buffer over.
// (M0 is saved and restored around every piece: the compiler keeps M0 reserved and does not accept it as a clobber.
// Dropping the two s_mov measured -1.5 % per layer in round 2 and +-0 in a round-3 same-box A/B; not worth the risk.)
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

// the same with the sc1 bit: served by L2, never by this CU's vector L1 (data another CU of the XCD has just written)
__device__ __forceinline__ void glds16_l2(const float* g, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_byte_addr)
      : "memory");
}
__device__ __forceinline__ void glds16s_l2(const float* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_off), "s"(gbase_uniform), "s"(lds_byte_addr) : "memory");
}


}  // namespace agz
