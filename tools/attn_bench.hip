// tools/attn_bench.hip — micro-benchmark of the fused attention kernel at the FLUX C2 shape
// (B=1, H=24, L=4608, d=128) on random bf16 data.  Build with extra -D flags to try variants.
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
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

__global__ void fill_e4m3_kernel(uint8_t* p, size_t n, uint32_t seed) {  // random e4m3 codes of moderate magnitude (|x| <= 4), no NaN codes
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = (uint8_t)((x & 0x80u) | (x >> 8) % 0x48u);
  }
}

// fp8-QK^T attention (the model's fp8 mode): the one-wave stream (attention_w16 QK8) against the 8-wave fp8 kernel
static void fp8_section(int iters) {
  struct Shape { int B, H, L; };
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (Shape s : {Shape{1, 24, 4608}, Shape{2, 24, 4112}, Shape{1, 4, 1000}, Shape{1, 2, 129}, Shape{1, 3, 200}, Shape{1, 2, 96}, Shape{1, 2, 128}, Shape{2, 2, 176}}) {
    const int Lpad = (s.L + 63) / 64 * 64;
    const size_t n = (size_t)s.B * s.H * s.L * 128, nv = (size_t)s.B * s.H * 128 * Lpad;
    uint8_t *q, *k;
    bf16_t *vt, *oa, *ob, *oc;
    hipMalloc((void**)&q, n); hipMalloc((void**)&k, n); hipMalloc((void**)&vt, nv * 2); hipMalloc((void**)&oa, n * 2); hipMalloc((void**)&ob, n * 2);
    hipMalloc((void**)&oc, n * 2);
    fill_e4m3_kernel<<<2048, 256>>>(q, n, 11u); fill_e4m3_kernel<<<2048, 256>>>(k, n, 12u); fill_kernel<<<2048, 256>>>(vt, nv, 3u);
    hipDeviceSynchronize();
    const float scale = ldexpf(1.0f, -6) / 1.4426950408889634f;  // scale * log2(e) = 2^-6
    AttnOut out{};
    out.p1 = oa, out.ld1 = s.H * 128, out.bstride1 = (int64_t)s.L * s.H * 128;
    // round 5: every operand e4m3 (P rounded in the kernel, V^T handed over as e4m3 bytes in plain key order)
    uint8_t* vt8;
    bf16_t* od;
    hipMalloc((void**)&vt8, nv); hipMalloc((void**)&od, n * 2);
    fill_e4m3_kernel<<<2048, 256>>>(vt8, nv, 13u);
    double us_pv = 0;
    {
      AttnOut o8 = out;
      o8.p1 = od;
      for (int i = 0; i < 3; ++i) launch_attention_ex((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt8, o8, s.B, s.H, s.L, s.L, Lpad, scale, 96, nullptr, 2, nullptr, 0, ATT_NO_EXP2, -1, 0.25f);
      hipDeviceSynchronize();
      if (s.L > 64) {
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) launch_attention_ex((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt8, o8, s.B, s.H, s.L, s.L, Lpad, scale, 96, nullptr, 2, nullptr, 0, ATT_NO_EXP2, -1, 0.25f);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        us_pv = ms / iters * 1e3;
      }
    }
    double us[3];
    for (int w = 0; w < 3; ++w) {  // 8-wave fp8 kernel, round 3's one-wave stream, round 4's lock-step stream
      set_attention_w16(w >= 1);
      set_attention_w16l(w == 2);
      out.p1 = w == 0 ? oa : w == 1 ? ob : oc;
      for (int i = 0; i < 3; ++i) launch_attention_ex((const bf16_t*)q, (const bf16_t*)k, vt, out, s.B, s.H, s.L, s.L, Lpad, scale, 96, nullptr, 1);
      hipDeviceSynchronize();
      hipEventRecord(e0, nullptr);
      for (int i = 0; i < iters; ++i) launch_attention_ex((const bf16_t*)q, (const bf16_t*)k, vt, out, s.B, s.H, s.L, s.L, Lpad, scale, 96, nullptr, 1);
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      us[w] = ms / iters * 1e3;
    }
    std::vector<uint16_t> ha(n), hb(n), hc(n);
    hipMemcpy(ha.data(), oa, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), ob, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), oc, n * 2, hipMemcpyDeviceToHost);
    size_t mis_l = 0, nan_l = 0;
    for (size_t i = 0; i < n; ++i) mis_l += hb[i] != hc[i], nan_l += ((hc[i] & 0x7f80) == 0x7f80 && (hc[i] & 0x7f));
    auto tof = [](uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return (double)f; };
    double num = 0, den = 0, mx = 0;
    size_t nan = 0;
    for (size_t i = 0; i < n; ++i) {
      const double a = tof(hb[i]), b = tof(ha[i]);
      if (!(a == a)) ++nan;
      num += (a - b) * (a - b), den += b * b, mx = std::max(mx, std::fabs(a - b));
    }
    printf("fp8 QK^T  B=%d H=%d L=%d   8-wave %6.1f us   one-wave (w16 QK8) %6.1f us   rel-L2 %.3e  max |diff| %.4g  NaN %zu   lock-step (w16l QK8) %6.1f us  differing from w16 QK8: %zu, NaN %zu   all-e4m3 (w16l PV8) %6.1f us%s\n", s.B, s.H, s.L, us[0], us[1],
           std::sqrt(num / std::max(den, 1e-30)), mx, nan, us[2], mis_l, nan_l, us_pv, hipGetLastError() == hipSuccess ? "" : "  (HIP ERROR)");
    hipFree(q); hipFree(k); hipFree(vt); hipFree(oa); hipFree(ob); hipFree(oc); hipFree(vt8); hipFree(od);
  }
  set_attention_w16(true);
  set_attention_w16l(true);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  if (getenv("FMI_FP8_ONLY")) {
    fp8_section(iters);
    return 0;
  }
  struct Shape { int B, H, L; };
  std::vector<Shape> shapes = {{1, 24, 4608}, {1, 24, 4112}, {2, 24, 4608}, {1, 4, 1000}, {1, 2, 64}, {1, 3, 200}, {1, 2, 128}, {1, 2, 192}, {1, 1, 116}, {1, 2, 127}, {1, 2, 129}, {1, 2, 512}};  // + ragged / tiny shapes for the bit-identity check
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
    bf16_t* o3;
    hipMalloc((void**)&o3, n * 2);
    bf16_t* o4;
    hipMalloc((void**)&o4, n * 2);
    hipMemset(o4, 0xff, n * 2);
    bf16_t* o5;
    hipMalloc((void**)&o5, n * 2);
    hipMemset(o5, 0xff, n * 2);
    bf16_t* o6;
    hipMalloc((void**)&o6, n * 2);
    hipMemset(o6, 0xff, n * 2);
    double tf[6];
    for (int pp = 0; pp < 6; ++pp) {  // single-barrier kernel, the ping-pong kernel, the one-wave-per-SIMD kernel, the 16x16x32 one-wave kernel, its 32x32x16 twin, the lock-step schedule
      set_attention_pingpong(pp != 0);
      set_attention_w4(pp == 2);
      set_attention_w16(pp == 3);
      set_attention_w32(pp == 4);
      set_attention_w16l(pp == 5);
      bf16_t* dst = pp == 0 ? o : pp == 1 ? o2 : pp == 2 ? o3 : pp == 3 ? o4 : pp == 4 ? o5 : o6;
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
    std::vector<uint16_t> ha(n), hb(n), hc(n), hd(n), he(n), hf(n);
    hipMemcpy(hf.data(), o6, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hd.data(), o4, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(he.data(), o5, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(ha.data(), o, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), o2, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), o3, n * 2, hipMemcpyDeviceToHost);
    size_t mis = 0, mis3 = 0, first = n;
    double maxd = 0;
    auto tof = [](uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return (double)f; };
    for (size_t i = 0; i < n; ++i) {
      mis += ha[i] != hb[i];
      if (hb[i] != hc[i]) {
        ++mis3;
        if (first == n) first = i;
        maxd = std::max(maxd, std::fabs(tof(hb[i]) - tof(hc[i])));
      }
    }
    if (mis3) {  // token-major output (B, L, H*128): where does the one-wave kernel first differ?
      const size_t row = first / ((size_t)s.H * 128), col = first % ((size_t)s.H * 128);
      size_t rows_bad = 0, last_row = 0;
      int hist[64] = {};
      for (size_t r = 0; r < (size_t)s.B * s.L; ++r) {
        bool bad = false;
        for (size_t c = 0; c < (size_t)s.H * 128 && !bad; ++c) bad = hb[r * s.H * 128 + c] != hc[r * s.H * 128 + c];
        rows_bad += bad;
        if (bad) last_row = r, ++hist[(r % s.L) & 63];
      }
      printf("   bad rows by (token & 63):");
      for (int i = 0; i < 64; ++i) printf(" %d", hist[i]);
      printf("\n");
      printf("   one-wave vs pp: max |diff| %.4g, first at token %zu head %zu d %zu (pp %.5g, one-wave %.5g); %zu of %zu token rows differ, last %zu\n", maxd, row,
             col / 128, col % 128, tof(hb[first]), tof(hc[first]), rows_bad, (size_t)s.B * s.L, last_row);
    }
    {  // round 4: the lock-step schedule against attention_w16 — default threshold: to rounding; threshold 0 (every tile rescales): bit for bit
      size_t mism = 0, nan = 0;
      double num = 0, den = 0, mx = 0;
      for (size_t i = 0; i < n; ++i) {
        const double a = tof(hf[i]), b = tof(hd[i]);
        mism += hf[i] != hd[i];
        if (!(a == a)) ++nan;
        num += (a - b) * (a - b), den += b * b, mx = std::max(mx, std::fabs(a - b));
      }
      AttnOut ao{};
      ao.p1 = o4, ao.ld1 = s.H * 128, ao.bstride1 = (int64_t)s.L * s.H * 128;
      set_attention_w16l(false), set_attention_w32(false), set_attention_w16(true);
      launch_attention_ex(q, k, vt, ao, s.B, s.H, s.L, s.L, Lpad, scale, 0, nullptr);
      ao.p1 = o6;
      set_attention_w16l(true);
      launch_attention_ex(q, k, vt, ao, s.B, s.H, s.L, s.L, Lpad, scale, 0, nullptr);
      hipDeviceSynchronize();
      std::vector<uint16_t> h0(n), h1(n);
      hipMemcpy(h0.data(), o4, n * 2, hipMemcpyDeviceToHost);
      hipMemcpy(h1.data(), o6, n * 2, hipMemcpyDeviceToHost);
      size_t mism0 = 0;
      for (size_t i = 0; i < n; ++i) mism0 += h0[i] != h1[i];
      printf("   w16l vs w16: rel-L2 %.3e, max |diff| %.4g, NaN %zu, differing elements %zu; at threshold 0: %zu differing     w16l %7.1f TF (%6.1f us)\n",
             std::sqrt(num / std::max(den, 1e-30)), mx, nan, mism, mism0, tf[5], 4.0 * s.B * s.H * (double)s.L * s.L * 128 / tf[5] * 1e-6);
    }
    for (int which = 0; which < 2; ++which) {  // the round-3 kernels equal the others to rounding (Q pre-scaled, row sums of the rounded P), not bit for bit
      const std::vector<uint16_t>& hx = which ? he : hd;
      double num = 0, den = 0, mx = 0;
      size_t nan = 0;
      for (size_t i = 0; i < n; ++i) {
        const double a = tof(hx[i]), b = tof(hb[i]);
        if (!(a == a)) ++nan;
        num += (a - b) * (a - b), den += b * b;
        mx = std::max(mx, std::fabs(a - b));
      }
      printf("   %s vs pp: rel-L2 %.3e, max |diff| %.4g, NaN %zu     %s %7.1f TF (%6.1f us)\n", which ? "w32" : "w16", std::sqrt(num / std::max(den, 1e-30)), mx, nan,
             which ? "w32" : "w16", tf[3 + which], 4.0 * s.B * s.H * (double)s.L * s.L * 128 / tf[3 + which] * 1e-6);
    }
    const double fl = 4.0 * s.B * s.H * (double)s.L * s.L * 128;
    printf("B=%d H=%d L=%d  single-barrier %7.1f TF   ping-pong %7.1f TF (%6.1f us)   one-wave %7.1f TF (%6.1f us)   mismatching elements: pp vs sb %zu, one-wave vs pp %zu%s\n", s.B,
           s.H, s.L, tf[0], tf[1], fl / tf[1] * 1e-6, tf[2], fl / tf[2] * 1e-6, mis, mis3, hipGetLastError() == hipSuccess ? "" : "  (HIP ERROR)");
    hipFree(o2);
    hipFree(o3);
    hipFree(o4);
    hipFree(o5);
    hipFree(o6);
    hipFree(q); hipFree(k); hipFree(vt); hipFree(o);
  }
  fp8_section(iters);
  return 0;
}
