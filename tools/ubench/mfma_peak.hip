// mfma_peak.hip -- what f32 MFMA rate does an MI355X SUSTAIN?  157.3 TFLOP/s is 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz;
// a kernel that keeps the matrix pipes busy runs into the board's power limit first and the clock comes down.
// One wave per SIMD issuing nothing but independent v_mfma_f32_32x32x2_f32 from registers (no memory, no LDS), in launches
// of ~10 ms repeated for ~0.5 s so that the clock settles; mode 1 adds the two ds_read_b64 per MFMA pair the Winograd
// GEMM's K loop issues.  Prints the sustained TFLOP/s per launch.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KIND, int LDS>
__global__ __launch_bounds__(256, 1) void k(int iters, float* out) {
  __shared__ float2 buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = make_float2(1.f + i * 1e-6f, 0.5f);
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = 1.f + threadIdx.x * 1e-6f, b = 0.5f;
  const h8 ah = {1, 2, 3, 4, 5, 6, 7, 8}, bh = {1, 1, 1, 1, 1, 1, 1, 1};
  const float2* p = buf + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (LDS) {
        const float2 x = p[(i * 256 + it * 64) & 2047], y = p[2048 + ((i * 256 + it * 64) & 1023)];
        a = x.x; b = y.y;
      }
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  if (s == 123.456f) out[0] = s;
}

template <int KIND, int LDS>
static void run(const char* name, double flop_per_mfma) {
  float* out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = KIND == 0 ? 40000 : 160000;      // ~10 ms per launch
  double last = 0;
  for (int rep = 0; rep < 50; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, LDS>), dim3(256 * 4), dim3(256), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = 256.0 * 4 * 4 * iters * 8.0 * flop_per_mfma / (ms * 1e-3) / 1e12;
    if (rep % 10 == 9 || rep == 0) printf("%-46s launch %2d: %7.3f ms  %8.1f TFLOP/s\n", name, rep, ms, tf);
    last = tf;
  }
  (void)last;
  CK(hipFree(out));
}

int main() {
  run<0, 0>("v_mfma_f32_32x32x2_f32, registers only", 4096.0);
  run<0, 1>("v_mfma_f32_32x32x2_f32 + 2 ds_read_b64 each", 4096.0);
  run<1, 0>("v_mfma_f32_32x32x16_f16, registers only", 32768.0);
  return 0;
}
