// tools/gemm_bench.hip — standalone micro-benchmark harness for the GEMM kernels (includes the product source).
// Prints TFLOP/s per FLUX shape on random bf16 data (never zero-filled: DVFS, cdna guide rule 25): the double-buffered,
// ping-pong and 4-wave dense kernels, and (FMI_Q4=1) the fused nf4 dequant-GEMM next to "dequant kernel + dense GEMM",
// with the number of output elements that differ between the arms (must be 0).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../diffusion-rs_amd/csrc/gemm_bf16.hip"
#include "../diffusion-rs_amd/csrc/bnb_dequant.hip"

namespace fmi {
static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(fmi_status st, const std::string& m) {
  g_err = m;
  fprintf(stderr, "error: %s\n", m.c_str());
  return (int)st;
}
}  // namespace fmi
using namespace fmi;

__global__ void fill_kernel(bf16_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    float f = ((float)(x & 0xffff) / 32768.0f - 1.0f);  // uniform [-1,1)
    p[i] = f32_to_bf16(f);
  }
}

// random 4-bit codes and absmax ~ U[0.5, 1.5) / 32
__global__ void fill_codes(uint8_t* q, size_t nq, float* am, size_t na) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ 0x9e3779b9u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    q[i] = (uint8_t)(x >> 8);
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2246822519u ^ 0x85ebca6bu;
    x ^= x >> 15; x *= 2654435761u; x ^= x >> 13;
    am[i] = (0.5f + (float)(x & 0xffff) / 65536.0f) / 32.0f;
  }
}
// element e = code(e) * absmax[e / 64], high nibble first (the arithmetic of bnb_dequant.hip, blocksize 64)
__global__ void expand_nf4(const uint8_t* q, const float* am, bf16_t* out, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const unsigned byte = q[e >> 1];
    const unsigned code = (e & 1) ? (byte & 15) : (byte >> 4);
    out[e] = f32_to_bf16(kNF4[code] * am[e >> 6]);
  }
}

__global__ void count_mismatch(const bf16_t* a, const bf16_t* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; const char* name; };
  std::vector<Shape> shapes = {{4608, 21504, 3072, "single qkv+mlp"}, {4608, 3072, 15360, "single proj_out"}, {4096, 9216, 3072, "double qkv img"},
                               {4096, 12288, 3072, "double mlp1 img"}, {4096, 3072, 12288, "double mlp2 img"}, {4096, 3072, 3072, "double proj img"},
                               {512, 9216, 3072, "double qkv txt"}, {8192, 8192, 8192, "8k cube"}};
  if (const char* env = getenv("FMI_SHAPES")) {  // "M,N,K;M,N,K;..." replaces the FLUX list
    shapes.clear();
    int m, n, k, used = 0;
    while (sscanf(env, "%d,%d,%d%n", &m, &n, &k, &used) == 3) {
      shapes.push_back({m, n, k, "custom"});
      env += used;
      if (*env == ';') ++env;
    }
  }
  int iters = argc > 1 ? atoi(argv[1]) : 10;
  const int pad_a = argc > 2 ? atoi(argv[2]) : 0, pad_w = argc > 3 ? atoi(argv[3]) : 0, pad_o = argc > 4 ? atoi(argv[4]) : 0;  // row-stride padding (elements)
  printf("pads: lda +%d, ldw +%d, ldo +%d elements\n", pad_a, pad_w, pad_o);
  size_t maxA = 0, maxW = 0, maxO = 0;
  for (auto& s : shapes) {
    maxA = std::max(maxA, (size_t)s.M * (s.K + 512));
    maxW = std::max(maxW, (size_t)s.N * (s.K + 512));
    maxO = std::max(maxO, (size_t)s.M * (s.N + 512));
  }
  bf16_t *A, *W, *O;
  // FMI_COLD_W=n: rotate over n weight buffers so every launch streams its W from HBM (as in the model, where a weight
  // matrix is used once per denoise step) instead of finding it in the 256 MiB Infinity Cache from the previous launch
  const int ncold = getenv("FMI_COLD_W") ? std::max(1, atoi(getenv("FMI_COLD_W"))) : 1;
  hipMalloc((void**)&A, maxA * 2);
  hipMalloc((void**)&W, maxW * 2 * ncold);
  printf("weight buffers: %d\n", ncold);
  hipMalloc((void**)&O, maxO * 2);
  fill_kernel<<<2048, 256>>>(A, maxA, 1u);
  fill_kernel<<<2048, 256>>>(W, maxW * ncold, 2u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  bf16_t* O2;
  const char* epi_env = getenv("FMI_EPI");  // store (default) | gelu | resid
  const int epi_kind = !epi_env ? 0 : !strcmp(epi_env, "gelu") ? 1 : !strcmp(epi_env, "resid") ? 2 : 0;
  // FMI_EPI=qkv: the model's q|k|v launches — fused RMS + RoPE + head-major relayout epilogue on the first 3*3072 columns
  // (GELU on the rest, as the single blocks' qkv+mlp launch); only the shapes with N >= 9216 take it
  const bool qkv_mode = epi_env && !strcmp(epi_env, "qkv");
  bf16_t *qh[2] = {}, *kh[2] = {}, *vt[2] = {}, *nqk = nullptr;
  float* pe = nullptr;
  const int QD = 3072, QH = 24;
  size_t qkv_elems = 0;
  if (qkv_mode) {
    int maxM = 0;
    for (auto& s : shapes) maxM = std::max(maxM, s.M);
    const int lpad = (maxM + 63) / 64 * 64;
    qkv_elems = (size_t)QH * 128 * lpad;
    for (int i = 0; i < 2; ++i) {
      hipMalloc((void**)&qh[i], qkv_elems * 2), hipMalloc((void**)&kh[i], qkv_elems * 2), hipMalloc((void**)&vt[i], qkv_elems * 2);
      hipMemset(qh[i], 0, qkv_elems * 2), hipMemset(kh[i], 0, qkv_elems * 2), hipMemset(vt[i], 0, qkv_elems * 2);
    }
    hipMalloc((void**)&nqk, 256 * 2);
    fill_kernel<<<1, 256>>>(nqk, 256, 5u);
    std::vector<float> h((size_t)maxM * 128);
    for (size_t i = 0; i < h.size() / 2; ++i) h[2 * i] = cosf(0.001f * i), h[2 * i + 1] = sinf(0.001f * i);
    hipMalloc((void**)&pe, h.size() * 4);
    hipMemcpy(pe, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    set_gemm_w4_qkv_min_n(0);
  }
  const size_t out_es = epi_kind == 2 ? 4 : 2;
  hipFree(O);
  hipMalloc((void**)&O, maxO * out_es);
  hipMalloc((void**)&O2, maxO * out_es);
  float* gate;
  hipMalloc((void**)&gate, 65536 * 4);
  {
    std::vector<float> g(65536, 0.5f);
    hipMemcpy(gate, g.data(), g.size() * 4, hipMemcpyHostToDevice);
  }
  bf16_t* bias;
  hipMalloc((void**)&bias, 65536 * 2);
  fill_kernel<<<64, 256>>>(bias, 65536, 3u);
  printf("epilogue: %s\n", epi_kind == 1 ? "bias + GELU -> bf16" : epi_kind == 2 ? "f32 residual += gate * (acc + bias)" : "store bf16");
  unsigned long long* d_mis;
  hipMalloc((void**)&d_mis, 8);
  for (auto& s : shapes) {
    GemmProblem p{};
    p.A = A, p.W = W, p.out = O, p.M = s.M, p.N = s.N, p.K = s.K, p.lda = s.K + pad_a, p.ldw = s.K + pad_w, p.ldo = s.N + pad_o, p.epi = epi_kind == 1 ? EPI_GELU_BF16 : epi_kind == 2 ? EPI_RESID_GATE_F32 : EPI_STORE_BF16, p.alpha = 1.f;
    if (epi_kind) p.bias = bias;
    if (epi_kind == 2) p.gate = gate;
    const bool qkv = qkv_mode && s.N >= 3 * QD && s.N % 256 == 0 && s.M % 16 == 0;
    if (qkv) {
      p.bias = bias;
      if (s.N > 3 * QD) p.epi = EPI_GELU_FROM_COL, p.gelu_from = 3 * QD;
      p.qk_wq = nqk, p.qk_wk = nqk + 128, p.qk_pe = pe, p.qk_pe_bstride = 0, p.qk_H = QH, p.qk_D = QD, p.qk_rows = s.M, p.qk_row_off = 0, p.qk_Ltot = s.M,
      p.qk_Lpad = (s.M + 63) / 64 * 64;
    }
    double tf[3];
    for (int pp = 0; pp < 3; ++pp) {  // double-buffered kernel, the ping-pong kernel, the 4-wave kernel
      set_gemm_pingpong(pp != 0);
      set_gemm_w4(pp == 2);
      p.out = pp == 1 ? O : O2;
      if (qkv) p.qk_qh = qh[pp == 1], p.qk_kh = kh[pp == 1], p.qk_vt = vt[pp == 1];
      if (epi_kind == 2) hipMemsetAsync(p.out, 0, (size_t)s.M * p.ldo * 4, nullptr);
      for (int i = 0; i < 2; ++i) launch_gemm(&p, 1, nullptr);
      hipDeviceSynchronize();
      hipEventRecord(e0, nullptr);
      for (int i = 0; i < iters; ++i) {
        p.W = W + (size_t)(i % ncold) * maxW;
        launch_gemm(&p, 1, nullptr);
      }
      p.W = W;
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= iters;
      tf[pp] = 2.0 * s.M * s.N * s.K / (ms * 1e-3) / 1e12;
    }
    hipMemset(d_mis, 0, 8);
    count_mismatch<<<1024, 256>>>(O, O2, (size_t)s.M * p.ldo * (out_es / 2), d_mis);  // resid: both accumulated the same number of launches
    unsigned long long mis = 0;
    if (qkv) {
      count_mismatch<<<1024, 256>>>(qh[0], qh[1], qkv_elems, d_mis);
      count_mismatch<<<1024, 256>>>(kh[0], kh[1], qkv_elems, d_mis);
      count_mismatch<<<1024, 256>>>(vt[0], vt[1], qkv_elems, d_mis);
    }
    hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost);
    printf("%-18s M=%5d N=%5d K=%5d  tiles %5d  double-buffered %7.1f TF   ping-pong %7.1f TF (%7.1f us)   4-wave %7.1f TF (%7.1f us)   4-wave vs ping-pong mismatching elements %llu%s\n", s.name, s.M, s.N, s.K,
           ((s.M + 255) / 256) * ((s.N + 255) / 256), tf[0], tf[1], 2.0 * s.M * s.N * s.K / tf[1] * 1e-6, tf[2], 2.0 * s.M * s.N * s.K / tf[2] * 1e-6, mis,
           hipGetLastError() == hipSuccess ? "" : "  (HIP ERROR)");
  }
  if (getenv("FMI_Q4") && epi_kind != 2) {
    // ---- fused nf4 dequant-GEMM: random codes + absmax, reference = stand-alone expansion + the dense kernel of the run above
    uint8_t* Wq;
    float* am;
    hipMalloc((void**)&Wq, maxW / 2);
    hipMalloc((void**)&am, maxW / 64 * 4);
    fill_codes<<<2048, 256>>>(Wq, maxW / 2, am, maxW / 64);
    set_gemm_pingpong(true);
    set_gemm_w4(true);
    for (auto& s : shapes) {
      if (s.M < 256) continue;
      expand_nf4<<<4096, 256>>>(Wq, am, W, (size_t)s.N * s.K);
      GemmProblem p{};
      p.A = A, p.W = W, p.out = O2, p.M = s.M, p.N = s.N, p.K = s.K, p.lda = s.K, p.ldw = s.K, p.ldo = s.N, p.epi = epi_kind == 1 ? EPI_GELU_BF16 : EPI_STORE_BF16, p.alpha = 1.f;
      if (epi_kind) p.bias = bias;
      const bool qkv = qkv_mode && s.N >= 3 * QD && s.N % 256 == 0 && s.M % 16 == 0;
      if (qkv) {
        p.bias = bias;
        if (s.N > 3 * QD) p.epi = EPI_GELU_FROM_COL, p.gelu_from = 3 * QD;
        p.qk_wq = nqk, p.qk_wk = nqk + 128, p.qk_pe = pe, p.qk_pe_bstride = 0, p.qk_H = QH, p.qk_D = QD, p.qk_rows = s.M, p.qk_row_off = 0, p.qk_Ltot = s.M,
        p.qk_Lpad = (s.M + 63) / 64 * 64;
        p.qk_qh = qh[0], p.qk_kh = kh[0], p.qk_vt = vt[0];
      }
      GemmProblem q = p;
      q.out = O, q.W = nullptr, q.Wq = Wq, q.absmax = am, q.q_blocksize = 64, q.q_type = 2;
      if (qkv) q.qk_qh = qh[1], q.qk_kh = kh[1], q.qk_vt = vt[1];
      double us[3];
      for (int arm = 0; arm < 3; ++arm) {  // dense on the expanded weights; fused one-wave kernel; fused 8-wave kernel (VALU expansion)
        GemmProblem& r = arm ? q : p;
        set_gemm_w4q_min_rows(arm == 2 ? (1 << 30) : 256);
        for (int i = 0; i < 2; ++i) launch_gemm(&r, 1, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) launch_gemm(&r, 1, nullptr);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        us[arm] = ms / iters * 1e3;
      }
      set_gemm_w4q_min_rows(256);
      printf("nf4 %-18s 8-wave fused (VALU expansion) %7.1f us  (%.2fx of dense)\n", s.name, us[2], us[0] / us[2]);
      hipMemset(d_mis, 0, 8);
      count_mismatch<<<1024, 256>>>(O, O2, (size_t)s.M * s.N, d_mis);
      if (qkv) {
        count_mismatch<<<1024, 256>>>(qh[0], qh[1], qkv_elems, d_mis);
        count_mismatch<<<1024, 256>>>(kh[0], kh[1], qkv_elems, d_mis);
        count_mismatch<<<1024, 256>>>(vt[0], vt[1], qkv_elems, d_mis);
      }
      unsigned long long mis = 0;
      hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost);
      const double fl = 2.0 * s.M * s.N * s.K;
      printf("nf4 %-18s M=%5d N=%5d K=%5d  dense(on expanded W) %7.1f us %7.1f TF   fused nf4 %7.1f us %7.1f TF  (%.2fx)   mismatching elements %llu%s\n", s.name, s.M, s.N, s.K,
             us[0], fl / us[0] * 1e-6, us[1], fl / us[1] * 1e-6, us[0] / us[1], mis, hipGetLastError() == hipSuccess ? "" : "  (HIP ERROR)");
    }
  }
  if (getenv("FMI_Q8") && epi_kind != 2) {
    // ---- fused LLM.int8 GEMM (weight tiles expanded by the VALU inside the 8-wave kernel): random int8 + SCB, reference =
    // the stand-alone dequant kernel + the dense kernel
    int8_t* W8;
    float* scb;
    hipMalloc((void**)&W8, maxW);
    hipMalloc((void**)&scb, 65536 * 4);
    fill_kernel<<<2048, 256>>>(reinterpret_cast<bf16_t*>(W8), maxW / 2, 9u);  // random bytes
    {
      std::vector<float> h(65536);
      for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f + 0.00001f * (float)(i % 977);
      hipMemcpy(scb, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    set_gemm_pingpong(true);
    set_gemm_w4(true);
    for (auto& s : shapes) {
      launch_dequant_int8_scb_bf16(W8, scb, W, s.K, (int64_t)s.N * s.K, nullptr);
      GemmProblem p{};
      p.A = A, p.W = W, p.out = O2, p.M = s.M, p.N = s.N, p.K = s.K, p.lda = s.K, p.ldw = s.K, p.ldo = s.N, p.epi = epi_kind == 1 ? EPI_GELU_BF16 : EPI_STORE_BF16, p.alpha = 1.f;
      if (epi_kind) p.bias = bias;
      GemmProblem q = p;
      q.out = O, q.W = nullptr, q.Wq = reinterpret_cast<const uint8_t*>(W8), q.absmax = scb, q.q_blocksize = 0, q.q_type = 3;
      double us[2];
      for (int arm = 0; arm < 2; ++arm) {
        GemmProblem& r = arm ? q : p;
        for (int i = 0; i < 2; ++i) launch_gemm(&r, 1, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) launch_gemm(&r, 1, nullptr);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        us[arm] = ms / iters * 1e3;
      }
      hipMemset(d_mis, 0, 8);
      count_mismatch<<<1024, 256>>>(O, O2, (size_t)s.M * s.N, d_mis);
      unsigned long long mis = 0;
      hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost);
      const double fl = 2.0 * s.M * s.N * s.K;
      printf("int8 %-18s M=%5d N=%5d K=%5d  dense(on expanded W) %7.1f us %7.1f TF   fused int8 %7.1f us %7.1f TF  (%.2fx)   mismatching elements %llu%s\n", s.name, s.M, s.N, s.K,
             us[0], fl / us[0] * 1e-6, us[1], fl / us[1] * 1e-6, us[0] / us[1], mis, hipGetLastError() == hipSuccess ? "" : "  (HIP ERROR)");
    }
  }
  return 0;
}
