// tools/mfma_peak.hip — what the matrix pipes sustain with nothing else going on: back-to-back
// v_mfma_f32_32x32x16_bf16 on register operands (random bf16, not zeros: DVFS), W waves per SIMD.
// Usage: mfma_peak [ms_target]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x;
  uint4 ua = in[tid], ub = in[512 + tid];
  bf16x8_t a = __builtin_bit_cast(bf16x8_t, ua), b = __builtin_bit_cast(bf16x8_t, ub);
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

// GEMM-like register use: NA x NB accumulators, acc[i][j] += mfma(a[i], b[j]) with distinct operand tuples
template <int NA, int NB>
__global__ __launch_bounds__(512, 2) void mfma_tile(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8_t a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = __builtin_bit_cast(bf16x8_t, in[(tid + 64 * i) & 1023]);
  for (int j = 0; j < NB; ++j) b[j] = __builtin_bit_cast(bf16x8_t, in[(tid + 64 * j + 512) & 1023]);
  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    asm volatile("" : "+v"(a[0]));  // keep the loop from being folded; no instruction emitted
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][7];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

// The vendor library's choice: v_mfma_f32_16x16x32_bf16 on an NA x NB tile of 16 x 16 accumulators (4 registers each), the A
// fragment held across NB consecutive MFMAs.  Same FLOPs per instruction-pair, half the pipe time per instruction.
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NA, int NB>
__global__ __launch_bounds__(256, 1) void mfma_tile16(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8_t a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = __builtin_bit_cast(bf16x8_t, in[(tid + 64 * i) & 1023]);
  for (int j = 0; j < NB; ++j) b[j] = __builtin_bit_cast(bf16x8_t, in[(tid + 64 * j + 512) & 1023]);
  f32x4 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    asm volatile("" : "+v"(a[0]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}
template <int NA, int NB>
__global__ __launch_bounds__(256, 1) void mfma_tile32(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8_t a[NA], b[NB];
  for (int i = 0; i < NA; ++i) a[i] = __builtin_bit_cast(bf16x8_t, in[(tid + 64 * i) & 1023]);
  for (int j = 0; j < NB; ++j) b[j] = __builtin_bit_cast(bf16x8_t, in[(tid + 64 * j + 512) & 1023]);
  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    asm volatile("" : "+v"(a[0]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][7];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

// fp8 (e4m3) forms: 32x32x64 (the fp8 GEMM's) vs 16x16x128, 128 x 128 wave tile, one wave per SIMD
typedef __attribute__((ext_vector_type(8))) int i32x8;
template <int SHAPE>  // 32 or 16
__global__ __launch_bounds__(256, 1) void mfma_tile_fp8(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x;
  constexpr int NT = SHAPE == 32 ? 4 : 8;
  i32x8 a[NT], b[NT];
  for (int i = 0; i < NT; ++i) {
    const uint4 x = in[(tid + 64 * i) & 1023], y = in[(tid + 64 * i + 512) & 1023];
    a[i] = i32x8{(int)x.x & 0x3f3f3f3f, (int)x.y & 0x3f3f3f3f, (int)x.z & 0x3f3f3f3f, (int)x.w & 0x3f3f3f3f, (int)y.x & 0x3f3f3f3f, (int)y.y & 0x3f3f3f3f, (int)y.z & 0x3f3f3f3f, (int)y.w & 0x3f3f3f3f};
    b[i] = i32x8{(int)y.x & 0x3f3f3f3f, (int)y.y & 0x3f3f3f3f, (int)x.z & 0x3f3f3f3f, (int)x.w & 0x3f3f3f3f, (int)x.x & 0x3f3f3f3f, (int)y.z & 0x3f3f3f3f, (int)y.w & 0x3f3f3f3f, (int)x.y & 0x3f3f3f3f};
  }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[NT][NT];
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j)
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b[j], a[i], acc[i][j], 0, 0, 0, 0, 0, 0);
      asm volatile("" : "+v"(a[0]));
    }
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][7];
  } else {
    f32x4 acc[NT][NT];
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j)
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a[i], acc[i][j], 0, 0, 0, 0, 0, 0);
      asm volatile("" : "+v"(a[0]));
    }
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
  }
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

// int8 forms (the int8 mode's GEMM, round 4): 32x32x32 (two per 32-byte fragment pair, as gemm_pp_kernel<2> issues them) vs 16x16x64, 128 x 128 wave tile
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
template <int SHAPE>  // 32 or 16
__global__ __launch_bounds__(256, 1) void mfma_tile_i8(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x;
  constexpr int NT = SHAPE == 32 ? 4 : 8;
  i32x4 a[NT][2], b[NT][2];
  for (int i = 0; i < NT; ++i)
    for (int h = 0; h < 2; ++h) {
      const uint4 x = in[(tid + 64 * i + 128 * h) & 1023], y = in[(tid + 64 * i + 512 + 32 * h) & 1023];
      a[i][h] = i32x4{(int)x.x, (int)x.y, (int)x.z, (int)x.w};  // all 256 codes occur (the bf16 bit patterns of main() as bytes)
      b[i][h] = i32x4{(int)y.w, (int)y.x, (int)y.y, (int)y.z};
    }
  int s = 0;
  if constexpr (SHAPE == 32) {
    i32x16 acc[NT][NT];
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j)
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[j][h], a[i][h], acc[i][j], 0, 0, 0);
      asm volatile("" : "+v"(a[0][0]));
    }
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][7];
  } else {
    i32x4 acc[NT][NT];
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j)
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[j][h], a[i][h], acc[i][j], 0, 0, 0);
      asm volatile("" : "+v"(a[0][0]));
    }
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
  }
  out[(size_t)blockIdx.x * blockDim.x + tid] = (float)s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  uint4* in;
  float* out;
  hipMalloc((void**)&in, 1024 * 16);
  hipMalloc((void**)&out, 512 * 512 * 4);
  uint32_t h[4096];
  uint32_t x = 12345;
  for (int i = 0; i < 4096; ++i) {
    x = x * 1664525u + 1013904223u;
    // two bf16 in [-1,1): sign + exponent 0x3f00..0x3f7f
    uint32_t lo = 0x3f00 | ((x >> 8) & 0x7f) | ((x & 1) << 15), hi = 0x3f00 | ((x >> 16) & 0x7f) | ((x & 2) << 14);
    h[i] = lo | (hi << 16);
  }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](auto kern, const char* what, int nacc, int blocks, int threads, int n) {
    kern<<<blocks, threads>>>(in, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<blocks, threads>>>(in, out, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * (double)n * nacc * 2.0 * 32 * 32 * 16;
    printf("%-34s blocks %3d x %d waves, %8d iters: %8.2f ms  %7.1f TFLOP/s\n", what, blocks, threads / 64, n, ms, flop / (ms * 1e-3) / 1e12);
  };
  for (int threads : {256, 512})
    for (int blocks : {256, 512})
      for (int rep = 0; rep < 2; ++rep) run(mfma_loop<8>, "8 accumulators round-robin", 8, blocks, threads, rep ? iters * 20 : iters);
  // accumulator reuse distance (dependent-issue latency of the 8-pass MFMA), one wave per SIMD
  run(mfma_loop<4>, "4 accumulators round-robin", 4, 256, 256, iters * 10);
  run(mfma_loop<2>, "2 accumulators round-robin", 2, 256, 256, iters * 10);
  run(mfma_loop<1>, "1 accumulator (back to back)", 1, 256, 256, iters * 10);
  run(mfma_loop<2>, "2 accumulators, 2 waves per SIMD", 2, 256, 512, iters * 10);
  run(mfma_tile<4, 2>, "4x2 tile, distinct A/B tuples, 1 w/SIMD", 8, 256, 256, iters * 10);
  run(mfma_tile<4, 2>, "4x2 tile, distinct A/B tuples, 2 w/SIMD", 8, 256, 512, iters * 10);
  // 128 x 128 wave tile, one wave per SIMD: 4 x 4 of 32x32x16 (32 FLOP-units per round) vs 8 x 8 of 16x16x32 (same FLOPs per round)
  auto run2 = [&](auto kern, const char* what, double flop_per_iter, int n) {
    kern<<<256, 256>>>(in, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<256, 256>>>(in, out, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.2f ms  %7.1f TFLOP/s\n", what, ms, 256.0 * 4 * n * flop_per_iter / (ms * 1e-3) / 1e12);
  };
  for (int rep = 0; rep < 2; ++rep) {
    run2(mfma_tile32<4, 4>, "4x4 tile of 32x32x16 (256 acc regs), 1 w/SIMD", 16 * 2.0 * 32 * 32 * 16, iters * 5);
    run2(mfma_tile16<8, 8>, "8x8 tile of 16x16x32 (256 acc regs), 1 w/SIMD", 64 * 2.0 * 16 * 16 * 32, iters * 5);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run2(mfma_tile_fp8<32>, "fp8: 4x4 tile of 32x32x64 f8f6f4, 1 w/SIMD", 16 * 2.0 * 32 * 32 * 64, iters * 5);
    run2(mfma_tile_fp8<16>, "fp8: 8x8 tile of 16x16x128 f8f6f4, 1 w/SIMD", 64 * 2.0 * 16 * 16 * 128, iters * 5);
    run2(mfma_tile_i8<32>, "int8: 4x4 tile of 2 x 32x32x32 i8, 1 w/SIMD", 32 * 2.0 * 32 * 32 * 32, iters * 5);
    run2(mfma_tile_i8<16>, "int8: 8x8 tile of 2 x 16x16x64 i8, 1 w/SIMD", 128 * 2.0 * 16 * 16 * 64, iters * 5);
  }
  return 0;
}
