// ingest.hip -- how many bytes per clock a gfx950 CU can pull out of L2: LDS-DMA (global_load_lds_dwordx4) against plain
// global_load_dwordx4 into registers, by waves per CU and working-set size.  Measurement tool for HISTORY.md 4d (round 3): the
// Winograd GEMM's K loop is fed by LDS-DMA, and its tile shape (flop per DMA byte) has to respect this ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o ingest ingest.hip && ./ingest
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16s(const float* gbase_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
  unsigned keep;
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_byte_off), "s"(gbase_uniform), "s"(lds_byte_addr) : "memory");
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every wave streams `pieces` 1-KB pieces; piece p of wave w of block b comes from offset ((b * waves + w) * 37 + p) mod region
template <int MODE>      // 0 = LDS-DMA, 1 = global_load_dwordx4 -> VGPR
__global__ __launch_bounds__(256) void k_ingest(const float* __restrict__ src, long region_pieces, int pieces, float* out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[0];
  const unsigned p = ((unsigned)blockIdx.x * nw + wave) * 37u, mask = (unsigned)region_pieces - 1u;     // region_pieces is a power of two
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < pieces; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned idx = __builtin_amdgcn_readfirstlane((p + i + j) & mask);
      const float* g = src + (size_t)idx * 256;
      if (MODE == 0) {
        glds16s(g, (unsigned)lane * 16u, lds0 + (unsigned)((wave * 16 + ((i + j) & 15)) * 1024));
      } else {
        acc += *reinterpret_cast<const f32x4*>(g + lane * 4);
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // 8 .. 16 pieces in flight per wave
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 0 && lds[threadIdx.x] == 123.456f) out[0] = 1.f;
  if (acc[0] == 1.f) out[1] = acc[1];
}

int main() {
  const size_t bytes = 512ull << 20;
  float* d;
  CK(hipMalloc(&d, bytes));
  CK(hipMemset(d, 0, bytes));
  float* out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int pieces = 4096;       // 4 MB per wave
  printf("mode waves/CU region_MB  GB/s   B/clk/CU(at 2.4GHz)\n");
  for (int mode = 0; mode < 2; ++mode)
    for (int wpc : {4, 8, 16})
      for (double region_mb : {1.0, 2.0, 16.0, 256.0}) {
        const long region_pieces = (long)(region_mb * 1024);
        const int threads = wpc >= 4 ? 256 : 64 * wpc, blocks_per_cu = wpc / 4;
        const int grid = 256 * blocks_per_cu;
        const size_t shm = (size_t)(threads / 64) * 16 * 1024;
        auto run = [&] {
          if (mode == 0) hipLaunchKernelGGL(k_ingest<0>, dim3(grid), dim3(threads), shm, 0, d, region_pieces, pieces, out);
          else hipLaunchKernelGGL(k_ingest<1>, dim3(grid), dim3(threads), shm, 0, d, region_pieces, pieces, out);
        };
        run();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) run();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double total = 3.0 * grid * (threads / 64) * (double)pieces * 1024.0;
        const double gbs = total / (ms * 1e-3) / 1e9;
        printf("%s %8d %9.1f %8.0f %8.1f\n", mode == 0 ? "lds-dma " : "vgpr    ", wpc, region_mb, gbs, gbs * 1e9 / 256 / 2.4e9);
      }
  return 0;
}
