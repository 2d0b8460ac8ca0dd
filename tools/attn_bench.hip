// tools/attn_bench.hip — micro-benchmark of the fused attention kernel at the FLUX C2 shape
// (B=1, H=24, L=4608, d=128) on random bf16 data.  Build with extra -D flags to try variants.
#include <cstdio>
#include <cstdlib>
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

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  struct Shape { int B, H, L; };
  std::vector<Shape> shapes = {{1, 24, 4608}, {1, 24, 4112}, {2, 24, 4608}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (auto& s : shapes) {
    const int Lpad = (s.L + 63) / 64 * 64;
    const size_t n = (size_t)s.B * s.H * s.L * 128, nv = (size_t)s.B * s.H * 128 * Lpad;
    bf16_t *q, *k, *vt, *o;
    hipMalloc((void**)&q, n * 2); hipMalloc((void**)&k, n * 2); hipMalloc((void**)&vt, nv * 2); hipMalloc((void**)&o, n * 2);
    fill_kernel<<<2048, 256>>>(q, n, 1u); fill_kernel<<<2048, 256>>>(k, n, 2u); fill_kernel<<<2048, 256>>>(vt, nv, 3u);
    hipDeviceSynchronize();
    const float scale = 0.08838834764f;
    for (int i = 0; i < 3; ++i) launch_attention(q, k, vt, o, s.B, s.H, s.L, s.L, Lpad, scale, 1, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) launch_attention(q, k, vt, o, s.B, s.H, s.L, s.L, Lpad, scale, 1, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    printf("B=%d H=%d L=%d  %8.3f ms  %7.1f TF\n", s.B, s.H, s.L, ms, 4.0 * s.B * s.H * (double)s.L * s.L * 128 / (ms * 1e-3) / 1e12);
    hipFree(q); hipFree(k); hipFree(vt); hipFree(o);
  }
  return 0;
}
