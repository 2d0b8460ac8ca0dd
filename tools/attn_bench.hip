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
    bf16_t* o2;
    hipMalloc((void**)&o2, n * 2);
    double tf[2];
    for (int pp = 0; pp < 2; ++pp) {  // single-barrier kernel, then the ping-pong kernel
      set_attention_pingpong(pp != 0);
      bf16_t* dst = pp ? o2 : o;
      for (int i = 0; i < 3; ++i) launch_attention(q, k, vt, dst, s.B, s.H, s.L, s.L, Lpad, scale, 1, nullptr);
      hipDeviceSynchronize();
      hipEventRecord(e0, nullptr);
      for (int i = 0; i < iters; ++i) launch_attention(q, k, vt, dst, s.B, s.H, s.L, s.L, Lpad, scale, 1, nullptr);
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= iters;
      tf[pp] = 4.0 * s.B * s.H * (double)s.L * s.L * 128 / (ms * 1e-3) / 1e12;
    }
    std::vector<uint16_t> ha(n), hb(n);
    hipMemcpy(ha.data(), o, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), o2, n * 2, hipMemcpyDeviceToHost);
    size_t mis = 0;
    for (size_t i = 0; i < n; ++i) mis += ha[i] != hb[i];
    printf("B=%d H=%d L=%d  single-barrier %7.1f TF   ping-pong %7.1f TF   mismatching elements %zu%s\n", s.B, s.H, s.L, tf[0], tf[1], mis,
           hipGetLastError() == hipSuccess ? "" : "  (HIP ERROR)");
    hipFree(o2);
#ifdef ATT_PP_TRACE
    {
      long long tr[8 * 64];
      hipMemcpyFromSymbol(tr, HIP_SYMBOL(fmi::g_att_trace), sizeof(tr));
      for (int w : {0, 4}) {
        printf("wave %d (clock64 deltas): per tile [V softmax | wait barrier | M issue | vmcnt wait | barrier wait]  and tile period\n", w);
        for (int t = 0; t < 7; ++t) {
          const long long* a = tr + w * 64 + t * 8;
          printf("  tile %2d: V %5lld  b %5lld  M %5lld  vm %5lld  b %5lld   period %5lld\n", 16 + t, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[8] - a[0]);
        }
      }
    }
#endif
    hipFree(q); hipFree(k); hipFree(vt); hipFree(o);
  }
  return 0;
}
