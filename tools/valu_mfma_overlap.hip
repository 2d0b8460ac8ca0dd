// tools/valu_mfma_overlap.hip — do VALU work and MFMA work issued by DIFFERENT waves of one SIMD overlap?
// Block = 8 waves (two per SIMD: waves w and w+4 share a SIMD).  Waves 0-3 run `mfma_iters` batches
// of 8 independent 32x32x16 MFMAs, waves 4-7 run `valu_iters` batches of 16 independent v_fma_f32
// (+ optionally 4 v_exp_f32).  Each wave times itself with s_memtime (shader clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int YIELD>
__global__ __launch_bounds__(512, 2) void k(const uint4* in, float* out, long long* cyc, int mfma_iters, int valu_iters, int with_exp, int valu_prio) {
  const int tid = threadIdx.x, wave = tid >> 6;
  float res = 0.f;
  if (wave >= 4 && valu_prio == 1) __builtin_amdgcn_s_setprio(1);
  if (wave >= 4 && valu_prio == 3) __builtin_amdgcn_s_setprio(3);
  long long t0 = clock64();
  if (wave < 4) {
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, in[tid & 255]), b = __builtin_bit_cast(bf16x8_t, in[256 + (tid & 255)]);
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        // does the MFMA wave have to yield explicitly for the other wave's VALU to get issue slots?
        if (YIELD == 1) asm volatile("s_nop 7");
        if (YIELD == 2) asm volatile("s_nop 15");
        if (YIELD == 3) asm volatile("s_nop 15\n\ts_nop 7");
        if (YIELD == 4) __builtin_amdgcn_s_sleep(1);
      }
    }
    for (int i = 0; i < 8; ++i) res += acc[i][0];
  } else {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = tid * 0.001f + i;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
      if (with_exp) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __builtin_amdgcn_exp2f(x[i] * 1e-9f);
      }
    }
    for (int i = 0; i < 16; ++i) res += x[i];
  }
  long long t1 = clock64();
  out[blockIdx.x * 512 + tid] = res;
  if (blockIdx.x == 0 && (tid & 63) == 0) cyc[wave] = t1 - t0;
}

// Same wave: each MFMA followed by NV independent v_fma_f32 (all 8 waves do this)
template <int NV>
__global__ __launch_bounds__(512, 2) void k_same(const uint4* in, float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x;
  bf16x8_t a = __builtin_bit_cast(bf16x8_t, in[tid & 255]), b = __builtin_bit_cast(bf16x8_t, in[256 + (tid & 255)]);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = tid * 0.001f + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[v] = __builtin_fmaf(x[v], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float res = 0.f;
  for (int i = 0; i < 8; ++i) res += acc[i][0] + x[i];
  out[blockIdx.x * 512 + tid] = res;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

int main() {
  uint4* in;
  float* out;
  long long* cyc;
  hipMalloc(&in, 512 * 16);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 64);
  unsigned h[2048];
  for (int i = 0; i < 2048; ++i) h[i] = 0x3f803f80u ^ (i * 2654435761u & 0x007f007fu);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int MI = 4000, VI = 4000;  // 8 MFMAs (256 clk of pipe) vs 16 FMAs (68 clk) per iteration
  struct Case { int mi, vi, ex, prio; const char* what; } cases[] = {
      {MI, 0, 0, 0, "MFMA waves alone"},           {0, VI * 4, 0, 0, "VALU (fma) waves alone"},          {0, VI * 2, 1, 0, "VALU (fma+exp) waves alone"},
      {MI, VI * 4, 0, 0, "MFMA + VALU(fma) together"}, {MI, VI * 2, 1, 0, "MFMA + VALU(fma+exp) together"},
      {MI, VI * 4, 0, 3, "MFMA + VALU(fma), VALU prio 3"}, {MI, VI * 2, 1, 3, "MFMA + VALU(fma+exp), VALU prio 3"},
      {MI, VI, 0, 3, "MFMA + short VALU(fma), prio 3"},
      {MI, VI * 4, 0, 10, "MFMA(+s_nop 7) + VALU(fma)"}, {MI, VI * 4, 0, 11, "MFMA(+s_nop 15) + VALU(fma)"}, {MI, VI * 4, 0, 12, "MFMA(+s_nop 15+7) + VALU(fma)"},
      {MI, VI * 4, 0, 13, "MFMA(+s_sleep 1) + VALU(fma)"},
  };
  for (auto& c : cases) {
    if (c.prio == 10) k<1><<<256, 512>>>(in, out, cyc, c.mi, c.vi, c.ex, 0);
    else if (c.prio == 11) k<2><<<256, 512>>>(in, out, cyc, c.mi, c.vi, c.ex, 0);
    else if (c.prio == 12) k<3><<<256, 512>>>(in, out, cyc, c.mi, c.vi, c.ex, 0);
    else if (c.prio == 13) k<4><<<256, 512>>>(in, out, cyc, c.mi, c.vi, c.ex, 0);
    else k<0><<<256, 512>>>(in, out, cyc, c.mi, c.vi, c.ex, c.prio);
    hipDeviceSynchronize();
    long long t[8];
    hipMemcpy(t, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-32s MFMA wave: %9lld clk", c.what, t[0]);
    if (c.mi) printf(" (%.1f per MFMA)", (double)t[0] / (c.mi * 8.0));
    printf("   VALU wave: %9lld clk", t[4]);
    if (c.vi) printf(" (%.2f per VALU op)", (double)t[4] / (c.vi * (16.0 + (c.ex ? 8 : 0))));
    printf("\n");
  }
  auto same = [&](auto kern, int nv, int threads) {
    kern<<<256, threads>>>(in, out, cyc, 2000);
    hipDeviceSynchronize();
    long long t;
    hipMemcpy(&t, cyc, 8, hipMemcpyDeviceToHost);
    printf("same wave, %d wave(s)/SIMD: MFMA + %d v_fma each: %.1f clk per (MFMA + VALU group)\n", threads / 256, nv, (double)t / (2000 * 8.0));
  };
  for (int threads : {256, 512}) {
    same(k_same<0>, 0, threads);
    same(k_same<2>, 2, threads);
    same(k_same<4>, 4, threads);
    same(k_same<6>, 6, threads);
    same(k_same<8>, 8, threads);
  }
  return 0;
}
