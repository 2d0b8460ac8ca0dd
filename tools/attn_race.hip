// tools/attn_race.hip — do the attention kernels reproduce their results while ANOTHER process keeps the GPU busy?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o build/attn_race tools/attn_race.hip
//   attn_race victim [launches] [kind]     repeated launches on fixed inputs (kind 2 one-wave, 1 ping-pong, 0 single barrier);
//                                          counts the launches whose output differs from the first one
//       env: LEN, HEADS (shape; default 4608 x 24)   WAIT=s (compute the reference alone, then wait s seconds for the co-runner)
//            DUMP=1 (where the wrong elements of the first bad launch are)   VONE=1 / KCONST=1 (V = 1 / identical keys: tells
//            numerator, denominator and score errors apart)   POLLUTE=1 (fill LDS and registers with NaN between launches)
//   attn_race copy|lds|exp|mfma [seconds]  co-runner for a second shell: HBM copies / LDS + VALU / transcendental / matrix pipe
//
// Found with it (round 2, profiles/r02_attn_race_shared_gpu.txt): the one-wave kernel refilled an MFMA's own A-operand buffer
// right behind the MFMA; next to copy- or LDS-heavy work of another process the LDS data landed before the queued MFMA had
// read the operand, and ~1 % of the workgroups produced different second-query-block rows.  The read now refills the buffer
// of the PREVIOUS MFMA (attention_w4.h); tests/test_gpu_shared_device.py keeps watch.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../diffusion-rs_amd/csrc/attention.hip"

namespace fmi {
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(fmi_status st, const std::string& m) {
  fprintf(stderr, "error: %s\n", m.c_str());
  return (int)st;
}
}  // namespace fmi
using namespace fmi;

__global__ void fill_kernel(bf16_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = f32_to_bf16(((float)(x & 0xffff) / 32768.0f - 1.0f) * 1.7f);
  }
}
__global__ void diff_kernel(const uint16_t* a, const uint16_t* b, size_t n, unsigned* count) {
  unsigned c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(count, c);
}
__global__ void burn_mfma(float* out, int iters) {
  typedef __attribute__((ext_vector_type(8))) short bf8;
  typedef __attribute__((ext_vector_type(16))) float f16v;
  bf8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (short)(threadIdx.x + i), b[i] = (short)(threadIdx.x * 3 + i);
  f16v acc = {};
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
}
__global__ void burn_exp(float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = x + 0.5f;
  for (int i = 0; i < iters; ++i) {
    x = __builtin_amdgcn_exp2f(x) - 1.0f;
    y = __builtin_amdgcn_exp2f(y) - 1.0f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
__global__ void burn_copy(const uint4* a, uint4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void burn_lds(float* out, int iters) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = i;
  __syncthreads();
  float acc = 0;
  int idx = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    acc += s[idx];
    idx = (idx * 5 + 1) & 4095;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// fills the CU's LDS and a large part of the register file (arch + accumulator VGPRs) with NaN patterns
__global__ __launch_bounds__(256, 1) void pollute(float* out) {
  __shared__ float s[40000];
  for (int i = threadIdx.x; i < 40000; i += 256) s[i] = __int_as_float(0x7fc12345);
  typedef __attribute__((ext_vector_type(8))) short bf8;
  typedef __attribute__((ext_vector_type(16))) float f16v;
  bf8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (short)0x7fc1, b[i] = (short)0x7fc1;
  f16v acc[12];
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = __int_as_float(0x7fc54321);
  float r[160];
#pragma unroll
  for (int i = 0; i < 160; ++i) r[i] = __int_as_float(0x7fc00000 + i + threadIdx.x);
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 160; ++i) asm volatile("" : "+v"(r[i]));
  __syncthreads();
  float t = s[(threadIdx.x * 7) % 40000];
#pragma unroll
  for (int j = 0; j < 12; ++j) t += acc[j][threadIdx.x & 15];
#pragma unroll
  for (int i = 0; i < 160; ++i) t += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "victim";
  if (mode == "victim") {
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const int kind = argc > 3 ? atoi(argv[3]) : 2;
    const int B = 1, H = getenv("HEADS") ? atoi(getenv("HEADS")) : 24, L = getenv("LEN") ? atoi(getenv("LEN")) : 4608, Lpad = (L + 63) / 64 * 64;
    const size_t n = (size_t)B * H * L * 128, nv = (size_t)B * H * 128 * Lpad;
    bf16_t *q, *k, *vt, *o0, *o;
    unsigned* cnt;
    hipMalloc((void**)&q, n * 2); hipMalloc((void**)&k, n * 2); hipMalloc((void**)&vt, nv * 2); hipMalloc((void**)&o0, n * 2); hipMalloc((void**)&o, n * 2);
    hipMalloc((void**)&cnt, 4);
    float* pol_out;
    hipMalloc((void**)&pol_out, 2048 * 256 * 4);
    fill_kernel<<<2048, 256>>>(q, n, 1u); fill_kernel<<<2048, 256>>>(k, n, 2u); fill_kernel<<<2048, 256>>>(vt, nv, 3u);
    if (getenv("VONE")) {  // V = 1: the output is 1 whatever the scores are, unless P (or O, l) is corrupted between softmax and P V
      std::vector<uint16_t> ones(nv, 0x3f80);
      hipMemcpy(vt, ones.data(), nv * 2, hipMemcpyHostToDevice);
    }
    if (getenv("KCONST")) {  // every key identical: the scores cannot be wrong through a stale K fragment
      std::vector<uint16_t> kk(n);
      hipMemcpy(kk.data(), k, n * 2, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < n; ++i) kk[i] = kk[i % 128];
      hipMemcpy(k, kk.data(), n * 2, hipMemcpyHostToDevice);
    }
    set_attention_pingpong(kind != 0);
    set_attention_w4(kind == 2);
    launch_attention(q, k, vt, o0, B, H, L, L, Lpad, 0.08838834764f, 1, nullptr);
    hipDeviceSynchronize();
    if (getenv("WAIT")) {  // the reference launch ran alone; now let the co-runner start
      const auto t0 = std::chrono::steady_clock::now();
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < atof(getenv("WAIT"))) {}
    }
    int bad = 0;
    size_t bad_elems = 0;
    for (int i = 0; i < iters; ++i) {
      hipMemsetAsync(cnt, 0, 4, nullptr);
      hipMemsetAsync(o, 0xff, n * 2, nullptr);
      if (getenv("POLLUTE")) pollute<<<2048, 256>>>(pol_out);
      launch_attention(q, k, vt, o, B, H, L, L, Lpad, 0.08838834764f, 1, nullptr);
      diff_kernel<<<1024, 256>>>((const uint16_t*)o0, (const uint16_t*)o, n, cnt);
      unsigned c = 0;
      hipMemcpy(&c, cnt, 4, hipMemcpyDeviceToHost);
      if (c && !bad && getenv("DUMP")) {  // first bad launch: where are the wrong elements?  out is token-major (L, H*128)
        std::vector<uint16_t> ha(n), hb(n);
        hipMemcpy(ha.data(), o0, n * 2, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), o, n * 2, hipMemcpyDeviceToHost);
        auto tof = [](uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return (double)f; };
        long by_wave[4] = {}, by_blk[2] = {}, by_l31[32] = {}, by_dt[4] = {}, by_head[4096] = {}, rows_bad = 0, wg_bad = 0, nan = 0;
        double maxd = 0, sumd = 0;
        std::vector<char> wgflag((size_t)H * ((L + 255) / 256), 0);
        std::vector<char> rowflag((size_t)H * L, 0);
        for (size_t r = 0; r < (size_t)L; ++r)
          for (int hh = 0; hh < H; ++hh)
            for (int d = 0; d < 128; ++d) {
              const size_t i = (r * H + hh) * 128 + d;
              if (ha[i] == hb[i]) continue;
              const double dd = std::fabs(tof(ha[i]) - tof(hb[i]));
              if (!(dd == dd)) ++nan; else maxd = std::max(maxd, dd), sumd += dd;
              ++by_wave[(r % 256) / 64], ++by_blk[(r % 64) / 32], ++by_l31[r % 32], ++by_dt[d / 32], ++by_head[hh];
              rowflag[hh * (size_t)L + r] = 1, wgflag[hh * ((L + 255) / 256) + r / 256] = 1;
            }
        for (char f : rowflag) rows_bad += f;
        for (char f : wgflag) wg_bad += f;
        printf("  first bad launch: %u elements, %ld (head,row) pairs of %d, %ld workgroups of %zu; max |diff| %.4g, mean %.4g, NaN %ld\n", c, rows_bad, H * L, wg_bad,
               wgflag.size(), maxd, sumd / c, nan);
        printf("  by wave: %ld %ld %ld %ld   by query block b: %ld %ld   by d block: %ld %ld %ld %ld\n", by_wave[0], by_wave[1], by_wave[2], by_wave[3], by_blk[0], by_blk[1],
               by_dt[0], by_dt[1], by_dt[2], by_dt[3]);
        printf("  by row & 31:");
        for (int i = 0; i < 32; ++i) printf(" %ld", by_l31[i]);
        printf("\n");
        // per bad (head,row): how many of its 128 d differ
        long full = 0, partial = 0;
        for (int hh = 0; hh < H; ++hh)
          for (size_t r = 0; r < (size_t)L; ++r)
            if (rowflag[hh * (size_t)L + r]) {
              int cnt_d = 0;
              for (int d = 0; d < 128; ++d) cnt_d += ha[(r * H + hh) * 128 + d] != hb[(r * H + hh) * 128 + d];
              (cnt_d > 100 ? full : partial)++;
            }
        int shown = 0;
        for (int hh = 0; hh < H && shown < 6; ++hh)
          for (size_t r = 0; r < (size_t)L && shown < 6; ++r)
            if (rowflag[hh * (size_t)L + r]) {
              printf("  head %d row %zu: ", hh, r);
              for (int d = 0; d < 128; d += 19) printf(" %.4f/%.4f(%.3f)", tof(hb[(r * H + hh) * 128 + d]), tof(ha[(r * H + hh) * 128 + d]), tof(hb[(r * H + hh) * 128 + d]) / tof(ha[(r * H + hh) * 128 + d]));
              printf("\n");
              ++shown;
              r += 40;
            }
        printf("  bad rows with > 100 of 128 d wrong: %ld, with fewer: %ld\n", full, partial);
      }
      bad += c != 0;
      bad_elems += c;
    }
    printf("victim kind %d: %d of %d launches differ from the first (%zu elements in total)%s\n", kind, bad, iters, bad_elems, hipGetLastError() == hipSuccess ? "" : "  (HIP ERROR)");
    return 0;
  }
  const double secs = argc > 2 ? atof(argv[2]) : 10.0;
  float* out;
  hipMalloc((void**)&out, 2048 * 256 * 4);
  uint4 *ca, *cb;
  const size_t cn = (size_t)1 << 26;  // 1 GiB
  hipMalloc((void**)&ca, cn * 16); hipMalloc((void**)&cb, cn * 16);
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 8; ++i) {
      if (mode == "mfma") burn_mfma<<<1024, 256>>>(out, 20000);
      else if (mode == "exp") burn_exp<<<2048, 256>>>(out, 20000);
      else if (mode == "copy") burn_copy<<<2048, 256>>>(ca, cb, cn);
      else if (mode == "lds") burn_lds<<<2048, 256>>>(out, 40000);
      ++launches;
    }
    hipDeviceSynchronize();
  }
  printf("co-runner %s: %ld launches\n", mode.c_str(), launches);
  return 0;
}
