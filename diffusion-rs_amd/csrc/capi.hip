// capi.hip — error channel, device helpers, host schedule math and the operator-level entry
// points of include/flux_mi355x.h (seam S3 of SURVEY.md §8b).
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace fmi {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(fmi_status st, const std::string& msg) {
  g_err = msg;
  return (int)st;
}
}  // namespace fmi
using namespace fmi;

extern "C" const char* fmi_last_error(void) { return g_err.c_str(); }
extern "C" int fmi_abi_version(void) { return FMI_ABI_VERSION; }

extern "C" int fmi_init(int device_ordinal) {
  int n = 0;
  FMI_HIP_TRY(hipGetDeviceCount(&n));
  if (n <= 0) return fail(FMI_ERR_HIP, "fmi_init: no HIP device visible — this library has no CPU fallback");
  if (device_ordinal < 0 || device_ordinal >= n) return fail(FMI_ERR_INVALID, "fmi_init: device ordinal out of range");
  FMI_HIP_TRY(hipSetDevice(device_ordinal));
  hipDeviceProp_t p;
  FMI_HIP_TRY(hipGetDeviceProperties(&p, device_ordinal));
  if (std::string(p.gcnArchName).find("gfx950") == std::string::npos)
    return fail(FMI_ERR_UNSUPPORTED, std::string("fmi_init: kernels are built for gfx950 only, device is ") + p.gcnArchName);
  FMI_HIP_TRY(hipFree(nullptr));
  return FMI_OK;
}

extern "C" const char* fmi_device_info(void) {
  static thread_local std::string s;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    s = "{\"error\": \"no device\"}";
    return s.c_str();
  }
  size_t fr = 0, tot = 0;
  hipMemGetInfo(&fr, &tot);
  char buf[512];
  snprintf(buf, sizeof(buf),
           "{\"device\": %d, \"name\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"hbm_total\": %zu, \"hbm_free\": %zu, "
           "\"lds_per_cu\": %zu, \"abi\": %d}",
           dev, p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, tot, fr, (size_t)p.maxSharedMemoryPerMultiProcessor,
           FMI_ABI_VERSION);
  s = buf;
  return s.c_str();
}

// ------------------------------------------------------------------ memory / stream helpers
extern "C" int fmi_malloc(void** dptr, size_t bytes) {
  if (!dptr) return fail(FMI_ERR_INVALID, "fmi_malloc: null");
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
  if (e != hipSuccess) return fail(FMI_ERR_NOMEM, std::string("fmi_malloc: ") + hipGetErrorString(e));
  return FMI_OK;
}
extern "C" int fmi_free(void* dptr) {
  FMI_HIP_TRY(hipFree(dptr));
  return FMI_OK;
}
extern "C" int fmi_memcpy(void* dst, const void* src, size_t bytes, void* stream) {
  FMI_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream));
  return FMI_OK;
}
extern "C" int fmi_memset(void* dst, int value, size_t bytes, void* stream) {
  FMI_HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return FMI_OK;
}
extern "C" int fmi_stream_synchronize(void* stream) {
  FMI_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return FMI_OK;
}
extern "C" int fmi_event_create(void** ev) {
  hipEvent_t e;
  FMI_HIP_TRY(hipEventCreate(&e));
  *ev = e;
  return FMI_OK;
}
extern "C" int fmi_event_record(void* ev, void* stream) {
  FMI_HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return FMI_OK;
}
extern "C" int fmi_event_elapsed_ms(void* start, void* stop, float* ms) {
  FMI_HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
  FMI_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return FMI_OK;
}
extern "C" int fmi_event_destroy(void* ev) {
  FMI_HIP_TRY(hipEventDestroy((hipEvent_t)ev));
  return FMI_OK;
}

// ------------------------------------------------------------------ host schedule math (f64)
// calculate_shift, diffusion_rs_core/src/pipelines/flux/sampling.rs:70-80
extern "C" double fmi_calculate_shift(int image_seq_len, int base_seq_len, int max_seq_len, double base_shift, double max_shift) {
  const double m = (max_shift - base_shift) / (double)(max_seq_len - base_seq_len);
  const double b = base_shift - m * (double)base_seq_len;
  return (double)image_seq_len * m + b;
}
// SchedulerConfig::get_timesteps, diffusion_rs_core/src/pipelines/scheduler.rs:22-51
extern "C" int fmi_get_timesteps(const fmi_scheduler_config* cfg, int num_steps, double mu, double* out_host) {
  if (!cfg || !out_host || num_steps <= 0) return fail(FMI_ERR_INVALID, "get_timesteps: bad arguments");
  for (int i = 0; i <= num_steps; ++i) {
    const double sigma = (double)(num_steps - i) / (double)num_steps;
    if (cfg->use_dynamic_shifting) {
      const double e = std::exp(mu);
      out_host[i] = e / (e + std::pow(1.0 / sigma - 1.0, 1.0));  // time_shift(mu, 1., sigma)
    } else {
      out_host[i] = cfg->shift * sigma / (1.0 + (cfg->shift - 1.0) * sigma);
    }
  }
  return FMI_OK;
}

// ------------------------------------------------------------------ operator-level entry points
static int epi_of(fmi_epilogue e) { return e == FMI_EPI_GELU_TANH ? EPI_GELU_BF16 : e == FMI_EPI_SILU ? EPI_SILU_BF16 : EPI_STORE_BF16; }

extern "C" int fmi_linear_bf16(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, fmi_epilogue epi, void* stream) {
  if (!x || !w || !y) return fail(FMI_ERR_INVALID, "linear_bf16: null pointer");
  if (M == 0 || N == 0) return FMI_OK;
  if (N % 4) return fail(FMI_ERR_INVALID, "linear_bf16: N must be a multiple of 4");
  GemmProblem p{};
  p.A = (const bf16_t*)x, p.W = (const bf16_t*)w, p.bias = (const bf16_t*)bias, p.out = y;
  p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldo = N, p.epi = epi_of(epi), p.alpha = 1.f;
  return launch_gemm(&p, 1, (hipStream_t)stream);
}

extern "C" int fmi_linear_bnb4_bf16(const void* x, const uint8_t* packed, const float* absmax, int blocksize, int quant_type, const void* bias,
                                    void* y, int M, int N, int K, fmi_epilogue epi, void* stream) {
  if (!x || !packed || !absmax || !y) return fail(FMI_ERR_INVALID, "linear_bnb4_bf16: null pointer");
  if (quant_type != 1 && quant_type != 2) return fail(FMI_ERR_INVALID, "linear_bnb4_bf16: quant_type must be 1 (fp4) or 2 (nf4)");
  if (M == 0 || N == 0) return FMI_OK;
  if (N % 4) return fail(FMI_ERR_INVALID, "linear_bnb4_bf16: N must be a multiple of 4");
  if (K % blocksize) return fail(FMI_ERR_UNSUPPORTED, "linear_bnb4_bf16: blocksize must divide K");
  GemmProblem p{};
  p.A = (const bf16_t*)x, p.bias = (const bf16_t*)bias, p.out = y;
  p.Wq = packed, p.absmax = absmax, p.q_blocksize = blocksize, p.q_type = quant_type;
  p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldo = N, p.epi = epi_of(epi), p.alpha = 1.f;
  return launch_gemm(&p, 1, (hipStream_t)stream);
}

extern "C" int fmi_linear_int8_bf16(const void* x, const int8_t* weight, const float* scb, const void* bias, void* y, int M, int N, int K,
                                    fmi_epilogue epi, void* stream) {
  if (!x || !weight || !scb || !y) return fail(FMI_ERR_INVALID, "linear_int8_bf16: null pointer");
  if (M == 0 || N == 0) return FMI_OK;
  if (N % 4) return fail(FMI_ERR_INVALID, "linear_int8_bf16: N must be a multiple of 4");
  if (reinterpret_cast<uintptr_t>(weight) & 15) return fail(FMI_ERR_INVALID, "linear_int8_bf16: weight must be 16-byte aligned");
  GemmProblem p{};
  p.A = (const bf16_t*)x, p.bias = (const bf16_t*)bias, p.out = y;
  p.Wq = reinterpret_cast<const uint8_t*>(weight), p.absmax = scb, p.q_blocksize = 0, p.q_type = 3;
  p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldo = N, p.epi = epi_of(epi), p.alpha = 1.f;
  return launch_gemm(&p, 1, (hipStream_t)stream);
}

extern "C" int fmi_quantize_rows_fp8(const void* x, int rows, int K, uint8_t* out, float* scale, void* stream) {
  if (rows == 0) return FMI_OK;
  if (!x || !out || !scale) return fail(FMI_ERR_INVALID, "quantize_rows_fp8: null pointer");
  return launch_quantize_rows_fp8((const bf16_t*)x, K, rows, K, out, scale, (hipStream_t)stream);
}

extern "C" int fmi_linear_fp8(const void* x, const uint8_t* wq, const float* w_scale, const void* bias, void* y, int M, int N, int K,
                              fmi_epilogue epi, void* stream) {
  if (!x || !wq || !w_scale || !y) return fail(FMI_ERR_INVALID, "linear_fp8: null pointer");
  if (M == 0 || N == 0) return FMI_OK;
  if (N % 4 || N <= 128) return fail(FMI_ERR_INVALID, "linear_fp8: N must be a multiple of 4 and > 128");
  hipStream_t s = (hipStream_t)stream;
  uint8_t* xq = nullptr;
  FMI_HIP_TRY(hipMalloc((void**)&xq, (size_t)M * K + (size_t)M * 4 + 256));
  float* xs = reinterpret_cast<float*>(xq + ((size_t)M * K + 255) / 256 * 256);
  int rc = launch_quantize_rows_fp8((const bf16_t*)x, K, M, K, xq, xs, s);
  if (rc == FMI_OK) {
    GemmProblem p{};
    p.A = (const bf16_t*)xq, p.W = (const bf16_t*)wq, p.bias = (const bf16_t*)bias, p.out = y;
    p.M = M, p.N = N, p.K = K, p.lda = K, p.ldw = K, p.ldo = N, p.epi = epi_of(epi), p.alpha = 1.f;
    p.fp8 = 1, p.a_scale = xs, p.w_scale = w_scale;
    rc = launch_gemm(&p, 1, s);
  }
  hipStreamSynchronize(s);
  hipFree(xq);
  return rc;
}

extern "C" int fmi_sdpa_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int d, float scale,
                             int out_token_major, void* stream) {
  if (!q || !k || !v || !o) return fail(FMI_ERR_INVALID, "sdpa_bf16: null pointer");
  if (d != 128) return fail(FMI_ERR_UNSUPPORTED, "sdpa_bf16: head dim must be 128");
  hipStream_t s = (hipStream_t)stream;
  const int Lpad = (Lk + 63) / 64 * 64;
  bf16_t* vt = nullptr;
  const size_t vt_bytes = (size_t)B * H * 128 * Lpad * 2;
  FMI_HIP_TRY(hipMalloc((void**)&vt, vt_bytes));
  int rc = FMI_OK;
  if (hipMemsetAsync(vt, 0, vt_bytes, s) != hipSuccess) rc = fail(FMI_ERR_HIP, "sdpa_bf16: memset failed");
  // V (B,H,Lk,128) head-major == B*H "batches" of one head each
  if (rc == FMI_OK) rc = launch_v_transpose((const bf16_t*)v, 128, (int64_t)Lk * 128, vt, B * H, 1, Lk, 0, Lpad, s);
  if (rc == FMI_OK) rc = launch_attention((const bf16_t*)q, (const bf16_t*)k, vt, (bf16_t*)o, B, H, Lq, Lk, Lpad, scale, out_token_major, s);
  hipStreamSynchronize(s);
  hipFree(vt);
  return rc;
}

// q, k: e4m3 bytes (B,H,L,128) with any scales folded into `scale` by the caller; v, o bf16 as fmi_sdpa_bf16
extern "C" int fmi_sdpa_fp8qk(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int d, float scale,
                             int out_token_major, void* stream) {
  if (!q || !k || !v || !o) return fail(FMI_ERR_INVALID, "sdpa_fp8qk: null pointer");
  if (d != 128) return fail(FMI_ERR_UNSUPPORTED, "sdpa_fp8qk: head dim must be 128");
  hipStream_t s = (hipStream_t)stream;
  const int Lpad = (Lk + 63) / 64 * 64;
  bf16_t* vt = nullptr;
  const size_t vt_bytes = (size_t)B * H * 128 * Lpad * 2;
  FMI_HIP_TRY(hipMalloc((void**)&vt, vt_bytes));
  AttnOut ao{};
  ao.p1 = (bf16_t*)o, ao.ld1 = H * 128, ao.bstride1 = (int64_t)Lq * H * 128, ao.head_major = out_token_major ? 0 : 1;
  int rc = FMI_OK;
  if (hipMemsetAsync(vt, 0, vt_bytes, s) != hipSuccess) rc = fail(FMI_ERR_HIP, "sdpa_fp8qk: memset failed");
  // V (B,H,Lk,128) head-major == B*H "batches" of one head each
  if (rc == FMI_OK) rc = launch_v_transpose((const bf16_t*)v, 128, (int64_t)Lk * 128, vt, B * H, 1, Lk, 0, Lpad, s);
  if (rc == FMI_OK) rc = launch_attention_ex((const bf16_t*)q, (const bf16_t*)k, vt, ao, B, H, Lq, Lk, Lpad, scale, 96, s, 1);
  hipStreamSynchronize(s);
  hipFree(vt);
  return rc;
}

// Process-wide kernel selection for bf16 attention (test / benchmark hook; all three give bit-identical results):
// 2 (default) = one wave per SIMD (attention_w4.h), 1 = 8-wave ping-pong, 0 = 8-wave single barrier.
extern "C" int fmi_set_attention_kernel(int kind) {
  if (kind < 0 || kind > 4) return fail(FMI_ERR_INVALID, "set_attention_kernel: kind must be 0 .. 4");
  set_attention_w32(kind == 4);
  set_attention_w16(kind >= 3);
  set_attention_w4(kind >= 2);
  set_attention_pingpong(kind >= 1);
  return FMI_OK;
}

extern "C" int fmi_layernorm_mod(const float* x, const float* scale, const float* shift, void* out_bf16, int rows, int D, float eps, void* stream) {
  if (!x || !out_bf16) return fail(FMI_ERR_INVALID, "layernorm_mod: null pointer");
  return launch_layernorm_mod(x, scale, shift, 0, 0, (bf16_t*)out_bf16, rows, D, eps, (hipStream_t)stream);
}
