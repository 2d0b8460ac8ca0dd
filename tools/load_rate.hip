// tools/load_rate.hip — per-CU issue rate of 16-B/lane vector loads: global_load_dwordx4 -> VGPR versus
// global_load_lds_dwordx4 (LDS-DMA), on a per-workgroup working set that stays in L2.
//   hipcc -O3 --offload-arch=gfx950 tools/load_rate.hip -o build/load_rate && ./build/load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int WAVES>  // 0: loads to VGPR, 1: LDS-DMA
__global__ __launch_bounds__(WAVES * 64) void k(const char* __restrict src, int iters, size_t ws_bytes, int* sink, long long* clk) {
  __shared__ __attribute__((aligned(16))) char smem[WAVES * 8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (size_t)blockIdx.x * ws_bytes;
  const size_t nchunk = ws_bytes / 1024;  // 1-KiB chunks in this workgroup's working set
  i32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = clock64();
  size_t c = wave;
  for (int it = 0; it < iters; ++it) {
    i32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const char* p = base + (c % nchunk) * 1024 + lane * 16;
      c += WAVES;
      if (MODE == 0) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(p) : "memory");
      } else {
        __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)(smem + (wave * 8 + u) * 1024), 16, 0, 0);
      }
    }
    // the destination registers stay live across the wait: the compiler must not recycle them while the loads are in flight
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  const long long t1 = clock64();
  if (MODE == 1) acc[0] = *reinterpret_cast<int*>(smem + threadIdx.x * 4);
  if (acc[0] == 0x12345678 && acc[1] == 77) sink[0] = acc[2];
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

static int g_scale = 1;
template <int MODE, int WAVES>
void run(const char* name, const char* src, size_t ws, int* sink, long long* clk, int ncu) {
  const int iters = 2000 * g_scale;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(ncu), dim3(WAVES * 64), 0, 0, src, 50, ws, sink, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(ncu), dim3(WAVES * 64), 0, 0, src, iters, ws, sink, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[1024];
  hipMemcpy(h, clk, sizeof(long long) * ncu, hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < ncu; ++i) avg += h[i];
  avg /= ncu;
  const double instr_per_cu = (double)iters * 8 * WAVES;
  printf("%-28s waves %d  ws/WG %6zu KiB  %7.3f ms  %6.1f clk/instr/CU (clock64 100MHz ticks x?)  %6.1f ns/instr/CU  %7.1f GB/s/CU  %6.2f TB/s total\n", name, WAVES, ws / 1024, ms,
         avg / instr_per_cu, ms * 1e6 / instr_per_cu, 1024.0 / (ms * 1e6 / instr_per_cu), 1024.0 * instr_per_cu * ncu / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  // load_rate [scale] [ws_kib]: scale multiplies the iteration count (power probes), ws_kib restricts to one working-set size
  g_scale = argc > 1 ? atoi(argv[1]) : 1;
  const size_t only_ws = argc > 2 ? (size_t)atoi(argv[2]) << 10 : 0;
  const int ncu = 256;
  const size_t maxws = 4 << 20;
  char* src;
  hipMalloc((void**)&src, maxws * ncu);
  hipMemset(src, 1, maxws * ncu);
  int* sink;
  long long* clk;
  hipMalloc((void**)&sink, 64);
  hipMalloc((void**)&clk, sizeof(long long) * 1024);
  for (size_t ws : {(size_t)64 << 10, (size_t)512 << 10, (size_t)4 << 20}) {
    if (only_ws && ws != only_ws) continue;
    run<0, 8>("global_load_dwordx4 -> VGPR", src, ws, sink, clk, ncu);
    run<1, 8>("global_load_lds_dwordx4", src, ws, sink, clk, ncu);
    run<0, 4>("global_load_dwordx4 -> VGPR", src, ws, sink, clk, ncu);
    run<1, 4>("global_load_lds_dwordx4", src, ws, sink, clk, ncu);
  }
  return 0;
}
