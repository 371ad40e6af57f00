// mfma_valu.hip -- does a wave's f32 VALU work run beside ANOTHER wave's MFMAs on the same SIMD?  gfx950's f32-input MFMA
// runs at the f32 vector rate (64 FLOP/clk/SIMD); if it executes on the vector ALU's own FMA lanes, a VALU-bound epilogue
// cannot hide under a co-resident workgroup's f32 K loop (k_wino_gemm6, HISTORY.md 4d), whereas under an f16/bf16 MFMA it can.
// A workgroup = 8 waves, two per SIMD: waves 0-3 issue MFMAs, waves 4-7 independent v_fma_f32 chains.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// mode bit 0: MFMA waves work; bit 1: VALU waves work.  KIND 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_f16
template <int KIND, int GAP, int PACE = 0>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = 1.f + threadIdx.x * 1e-6f, b = 0.5f;
    const h8 ah = {1, 2, 3, 4, 5, 6, 7, 8}, bh = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
        // PACE: the wave stays away from the issue port while its MFMA runs (64 cycles), instead of parking the next one there
        if (PACE == 1) asm volatile("s_nop 9" ::: "memory");
        if (PACE == 2) asm volatile("s_nop 11" ::: "memory");
        if (PACE == 3) asm volatile("s_nop 13" ::: "memory");
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    if (s == 123.456f) out[0] = s;
  } else {
    if (!(mode & 2)) return;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.f + i + threadIdx.x * 1e-6f;
    const float m = 1.0000001f, c = 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)         // 32 independent-enough FMAs per iteration (8 chains x 4)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
          if (GAP == 1) asm volatile("s_nop 3");
          if (GAP == 2) asm volatile("s_nop 15");
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) out[1] = s;
  }
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int kind = 0; kind < 7; ++kind) {
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (kind == 0) hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        else if (kind == 1) hipLaunchKernelGGL((k<1, 0>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        else if (kind == 2) hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        else if (kind == 3) hipLaunchKernelGGL((k<0, 2>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        else if (kind == 4) hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        else if (kind == 5) hipLaunchKernelGGL((k<0, 0, 2>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        else hipLaunchKernelGGL((k<0, 0, 3>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[mode], e0, e1));
      }
    }
    // per SIMD and iteration: 4 MFMAs (KIND 0: 64 cycles each; KIND 1: 32 cycles each... at ~2.4 GHz) and 32 v_fma_f32
    printf("%s: MFMA waves alone %.3f ms | VALU waves alone %.3f ms | both %.3f ms  -> both / max = %.2f, both / sum = %.2f\n",
           kind == 0 ? "v_mfma_f32_32x32x2_f32 , dense v_fma" : kind == 1 ? "v_mfma_f32_32x32x16_f16, dense v_fma" : kind == 2 ? "v_mfma_f32_32x32x2_f32 , v_fma + s_nop 3" : kind == 3 ? "v_mfma_f32_32x32x2_f32 , v_fma + s_nop 15" : kind == 4 ? "f32 MFMA + s_nop 9, dense v_fma" : kind == 5 ? "f32 MFMA + s_nop 11, dense v_fma" : "f32 MFMA + s_nop 13, dense v_fma", t[1], t[2], t[3],
           t[3] / (t[1] > t[2] ? t[1] : t[2]), t[3] / (t[1] + t[2]));
  }
  return 0;
}
