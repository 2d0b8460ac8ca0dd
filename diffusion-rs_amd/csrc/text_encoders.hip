// text_encoders.hip — the two text encoders in front of the FLUX denoise loop:
//   fmi_t5_*   == T5EncoderModel::forward          (diffusion_rs_core/src/models/t5/mod.rs:609-632)
//   fmi_clip_* == ClipTextTransformer::forward     (diffusion_rs_core/src/models/clip/text.rs:303-317)
// They run once per image (~0.1 % of a 50-step generation), so the design goal is exact
// semantics on the existing MFMA GEMM, not a new roofline: every Linear goes through
// launch_gemm (fused [q|k|v] and [wi_0|wi_1] weights, f32 residual stream updated in the GEMM
// epilogue), norms / activations / embedding gathers are single-pass HBM kernels, and the d=64
// attention (T5: additive relative-position bias, no 1/sqrt(d); CLIP: causal) is a small
// lane-per-key kernel — L <= 512 keys, 64 heads: 0.1 TFLOP per prompt, not worth an MFMA pipeline.
#include <math.h>

#include <map>
#include <set>
#include <string>
#include <vector>

#include "common.h"

namespace fmi {
namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------ kernels
// x(r,:) = table[ids[r]] (+ pos[r % T]) as f32.  Embedding::forward (+ ClipTextEmbeddings :63-71)
__global__ void embed_gather_kernel(const int32_t* ids, const bf16_t* table, const bf16_t* pos, int T, int D, int vocab, float* x, int* err) {
  const int r = blockIdx.x;
  const int id = ids[r];
  if (id < 0 || id >= vocab) {
    if (threadIdx.x == 0) atomicExch(err, 1);
    return;
  }
  const bf16_t* src = table + (int64_t)id * D;
  const bf16_t* ps = pos ? pos + (int64_t)(r % T) * D : nullptr;
  for (int i = threadIdx.x; i < D; i += blockDim.x) x[(int64_t)r * D + i] = bf16_to_f32(src[i]) + (ps ? bf16_to_f32(ps[i]) : 0.f);
}

// T5LayerNorm (t5/mod.rs:110-121): x * rsqrt(mean(x^2) + eps) * w ; one wave per row
__global__ void t5_rmsnorm_kernel(const float* x, const bf16_t* w, float eps, int rows, int D, bf16_t* out_bf, float* out_f32) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + (int64_t)r * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i] * xr[i];
  s = wave_sum(s);
  const float inv = 1.0f / sqrtf(s / (float)D + eps);
  for (int i = lane; i < D; i += 64) {
    const float v = xr[i] * inv * bf16_to_f32(w[i]);
    if (out_bf) out_bf[(int64_t)r * D + i] = f32_to_bf16(v);
    if (out_f32) out_f32[(int64_t)r * D + i] = v;
  }
}

// LayerNorm with affine (nn/layer_norm.rs:131-153, eps 1e-5 for CLIP); one wave per row
__global__ void layernorm_affine_kernel(const float* x, const bf16_t* w, const bf16_t* b, float eps, int rows, int D, bf16_t* out_bf, float* out_f32) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + (int64_t)r * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)D;
  float v2 = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float d = xr[i] - mean;
    v2 += d * d;
  }
  const float inv = 1.0f / sqrtf(wave_sum(v2) / (float)D + eps);
  for (int i = lane; i < D; i += 64) {
    const float v = (xr[i] - mean) * inv * bf16_to_f32(w[i]) + bf16_to_f32(b[i]);
    if (out_bf) out_bf[(int64_t)r * D + i] = f32_to_bf16(v);
    if (out_f32) out_f32[(int64_t)r * D + i] = v;
  }
}

// position_bias[h][i][j] = rel[bucket(j - i)][h]; the bucket of every distance is computed on the
// host with the reference's own float formula (t5/mod.rs:340-376) so that no device logf rounding
// can move a boundary.  One block per (i, h).
__global__ void t5_bias_kernel(const bf16_t* rel, const int* bucket_of_dist, int H, int T, float* bias) {
  const int i = blockIdx.x, h = blockIdx.y;
  for (int j = threadIdx.x; j < T; j += blockDim.x) bias[((int64_t)h * T + i) * T + j] = bf16_to_f32(rel[(int64_t)bucket_of_dist[j - i + T - 1] * H + h]);
}

// softmax(q k^T * scale + bias [causal]) v, head dim 64.  qkv: (B*T, 3*I) bf16 token-major rows
// [q | k | v], head h at columns h*64 of each part.  One wave per query row, lane = key index
// (scores) then lane = output channel (P V).  T <= 64 * MAXKB.
constexpr int ENC_MAXKB = 16;
__global__ __launch_bounds__(256) void enc_attention_kernel(const bf16_t* qkv, const float* bias, int B, int T, int H, float scale, int causal, bf16_t* out) {
  extern __shared__ float enc_sm[];  // per wave: 64 floats of q, then T floats of p
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const int I = H * 64, ld = 3 * I;
  float* qs = enc_sm + wave * (64 + T);
  float* ps = qs + 64;
  const bf16_t* base = qkv + (int64_t)b * T * ld + h * 64;
  qs[lane] = bf16_to_f32(base[(int64_t)i * ld + lane]) * scale;
  __builtin_amdgcn_wave_barrier();
  const int nkb = (T + 63) >> 6;
  float sc[ENC_MAXKB];
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < ENC_MAXKB; ++kb) {
    sc[kb] = -INFINITY;
    if (kb < nkb) {
      const int j = kb * 64 + lane;
      if (j < T && (!causal || j <= i)) {
        const uint4* kr = reinterpret_cast<const uint4*>(base + (int64_t)j * ld + I);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint4 u = kr[c];
          const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc += qs[c * 8 + e * 2] * __uint_as_float(w4[e] << 16);
            acc += qs[c * 8 + e * 2 + 1] * __uint_as_float(w4[e] & 0xffff0000u);
          }
        }
        if (bias) acc += bias[((int64_t)h * T + i) * T + j];
        sc[kb] = acc;
        mx = fmaxf(mx, acc);
      }
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < ENC_MAXKB; ++kb)
    if (kb < nkb) {
      const int j = kb * 64 + lane;
      const float p = sc[kb] == -INFINITY ? 0.f : __expf(sc[kb] - mx);
      if (j < T) ps[j] = p;
      sum += p;
    }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  const int jend = causal ? i + 1 : T;
  const bf16_t* vcol = base + 2 * I + lane;
  float acc = 0.f;
  for (int j = 0; j < jend; ++j) acc += ps[j] * bf16_to_f32(vcol[(int64_t)j * ld]);
  out[((int64_t)b * T + i) * I + h * 64 + lane] = f32_to_bf16(acc / sum);
}

// act: 0 relu, 1 gelu-tanh (NewGelu), 2 silu, 3 quick-gelu x*sigmoid(1.702x).
// gated: in (rows, 2F) = [gate-input | linear]  -> out (rows, F) = act(in[:, :F]) * in[:, F:]
// else : in (rows, F) -> out = act(in)
__global__ void enc_act_kernel(const bf16_t* in, int rows, int F, int act, int gated, bf16_t* out) {
  const int64_t n = (int64_t)rows * F;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / F, c = e - r * F;
    const float a = bf16_to_f32(in[gated ? r * 2 * F + c : e]);
    float v;
    if (act == 0)
      v = a > 0.f ? a : 0.f;
    else if (act == 1)
      v = gelu_tanh(a);
    else if (act == 2)
      v = silu(a);
    else
      v = a / (1.0f + __expf(-1.702f * a));
    if (gated) v *= bf16_to_f32(in[r * 2 * F + F + c]);
    out[e] = f32_to_bf16(v);
  }
}

__global__ void fill_f32_kernel(float* p, int n, float v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

// pooled[b] = hidden[b, argmax_t ids[b, t]] (first maximum); clip/text.rs:305-316
__global__ void clip_pool_kernel(const float* hidden, const int32_t* ids, int T, int D, float* pooled) {
  const int b = blockIdx.x;
  __shared__ int best;
  if (threadIdx.x == 0) {
    int bi = 0;
    for (int t = 1; t < T; ++t)
      if (ids[b * T + t] > ids[b * T + bi]) bi = t;
    best = bi;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += blockDim.x) pooled[(int64_t)b * D + i] = hidden[((int64_t)b * T + best) * D + i];
}

// ------------------------------------------------------------------------------ model plumbing
struct Dest {
  bf16_t* ptr;
  int rows, cols;  // cols == 0 -> 1-D
};
struct Registry {
  char* arena = nullptr;
  size_t bytes = 0, used = 0;
  std::map<std::string, Dest> names;
  std::set<std::string> missing;
  std::vector<std::string> missing_list;
  bf16_t* take(size_t count) {
    const size_t off = align_up(used, 256);
    used = off + count * sizeof(bf16_t);
    return arena ? reinterpret_cast<bf16_t*>(arena + off) : nullptr;
  }
  void reg(const std::string& name, bf16_t* p, int rows, int cols) {
    names[name] = Dest{p, rows, cols};
    missing.insert(name);
  }
  int set(const char* who, const char* name, const void* data, fmi_dtype dtype, const int64_t* shape, int rank) {
    if (!name || !data) return fail(FMI_ERR_INVALID, std::string(who) + ": null argument");
    auto it = names.find(name);
    if (it == names.end()) return fail(FMI_ERR_INVALID, std::string(who) + ": unknown tensor name '" + name + "'");
    const Dest& d = it->second;
    bool ok = d.cols ? (rank == 2 && shape[0] == d.rows && shape[1] == d.cols) : (rank == 1 && shape[0] == d.rows);
    if (!ok) return fail(FMI_ERR_INVALID, std::string(who) + ": shape mismatch for " + name + " (expected (" + std::to_string(d.rows) + (d.cols ? "," + std::to_string(d.cols) : "") + "))");
    if (dtype != FMI_F32 && dtype != FMI_F16 && dtype != FMI_BF16) return fail(FMI_ERR_INVALID, std::string(who) + ": dtype must be F32/F16/BF16");
    const int64_t numel = (int64_t)d.rows * (d.cols ? d.cols : 1);
    if (dtype == FMI_BF16) {
      FMI_HIP_TRY(hipMemcpy(d.ptr, data, numel * 2, hipMemcpyDefault));
    } else {
      const size_t esz = dtype == FMI_F32 ? 4 : 2;
      void* tmp = nullptr;
      FMI_HIP_TRY(hipMalloc(&tmp, numel * esz));
      hipError_t e = hipMemcpy(tmp, data, numel * esz, hipMemcpyDefault);
      int rc = e == hipSuccess ? launch_cast_to_bf16(tmp, dtype, d.ptr, numel, nullptr) : fail(FMI_ERR_HIP, hipGetErrorString(e));
      (void)hipDeviceSynchronize();
      (void)hipFree(tmp);
      if (rc) return rc;
    }
    missing.erase(name);
    return FMI_OK;
  }
  int ready(const char* who) {
    if (missing.empty()) return FMI_OK;
    return fail(FMI_ERR_STATE, std::string(who) + ": " + std::to_string(missing.size()) + " tensors not set, first: " + *missing.begin());
  }
};

struct Workspace {
  char* base = nullptr;
  size_t bytes = 0;
  int B = 0, T = 0;
  int reserve(size_t need) {
    if (need <= bytes) return FMI_OK;
    if (base) FMI_HIP_TRY(hipFree(base));
    base = nullptr, bytes = 0;
    FMI_HIP_TRY(hipMalloc((void**)&base, need));
    bytes = need;
    return FMI_OK;
  }
};

GemmProblem lin(const bf16_t* A, int lda, const bf16_t* W, const bf16_t* bias, void* out, int ldo, int M, int N, int K, int epi) {
  GemmProblem p{};
  p.A = A, p.W = W, p.bias = bias, p.out = out, p.M = M, p.N = N, p.K = K, p.lda = lda, p.ldw = K, p.ldo = ldo, p.epi = epi, p.alpha = 1.f;
  return p;
}

int enc_attention(const bf16_t* qkv, const float* bias, int B, int T, int H, float scale, int causal, bf16_t* out, hipStream_t s) {
  if (T > 64 * ENC_MAXKB) return fail(FMI_ERR_UNSUPPORTED, "text encoder attention: sequence longer than " + std::to_string(64 * ENC_MAXKB));
  const size_t sm = 4 * (64 + (size_t)T) * sizeof(float);
  hipLaunchKernelGGL(enc_attention_kernel, dim3((T + 3) / 4, H, B), dim3(256), sm, s, qkv, bias, B, T, H, scale, causal, out);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

// the reference's bucket of a relative position (t5/mod.rs:340-376), rel = j - i
int t5_bucket_of(int rel, int num_buckets_total, int max_distance) {
  const unsigned num_buckets = (unsigned)num_buckets_total / 2, max_exact = num_buckets / 2;
  auto large = [&](unsigned dist) -> unsigned {
    const float b = (logf((float)dist / (float)max_exact) / logf((float)max_distance / (float)max_exact)) * (float)(num_buckets - max_exact);
    return (unsigned)b;
  };
  if (rel > 0) {
    const unsigned d = (unsigned)rel;
    if (d < max_exact) return (int)(d + num_buckets);
    const unsigned v = max_exact + num_buckets + large(d);
    return (int)(v < (unsigned)num_buckets_total - 1 ? v : (unsigned)num_buckets_total - 1);
  }
  const unsigned d = (unsigned)(-rel);
  if (d < max_exact) return (int)d;
  const unsigned v = max_exact + large(d);
  return (int)(v < num_buckets - 1 ? v : num_buckets - 1);
}

int write_out(const float* src_f32, const bf16_t* src_bf, void* out, fmi_dtype dt, int64_t n, hipStream_t s) {
  if (dt == FMI_F32) {
    FMI_HIP_TRY(hipMemcpyAsync(out, src_f32, n * 4, hipMemcpyDeviceToDevice, s));
    return FMI_OK;
  }
  if (dt == FMI_BF16) {
    FMI_HIP_TRY(hipMemcpyAsync(out, src_bf, n * 2, hipMemcpyDeviceToDevice, s));
    return FMI_OK;
  }
  return fail(FMI_ERR_INVALID, "text encoder output dtype must be F32 or BF16");
}

}  // namespace
}  // namespace fmi

using namespace fmi;

// =============================================================================================== T5
struct fmi_t5 {
  fmi_t5_config cfg;
  int device = current_device();
  Registry r;
  Workspace ws;
  bf16_t *shared = nullptr, *rel = nullptr, *final_ln = nullptr;
  struct Layer {
    bf16_t *ln0, *qkv, *o, *ln1, *wi, *wo;
  };
  std::vector<Layer> layers;
  std::vector<int> bucket_host;
};

static void t5_layout(fmi_t5* m) {
  const fmi_t5_config& c = m->cfg;
  const int D = c.d_model, I = c.num_heads * c.d_kv, F = c.d_ff;
  const bool gated = c.feed_forward_act != FMI_T5_RELU;
  Registry& r = m->r;
  r.used = 0;
  r.names.clear();
  r.missing.clear();
  m->shared = r.take((size_t)c.vocab_size * D);
  r.reg("shared.weight", m->shared, c.vocab_size, D);
  m->rel = r.take((size_t)c.relative_attention_num_buckets * c.num_heads);
  r.reg("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", m->rel, c.relative_attention_num_buckets, c.num_heads);
  m->layers.resize(c.num_layers);
  for (int l = 0; l < c.num_layers; ++l) {
    fmi_t5::Layer& L = m->layers[l];
    const std::string p = "encoder.block." + std::to_string(l) + ".layer.";
    L.ln0 = r.take(D);
    r.reg(p + "0.layer_norm.weight", L.ln0, D, 0);
    L.qkv = r.take((size_t)3 * I * D);  // fused [q|k|v]
    const char* nm[3] = {"q", "k", "v"};
    for (int i = 0; i < 3; ++i) r.reg(p + "0.SelfAttention." + nm[i] + ".weight", L.qkv ? L.qkv + (size_t)i * I * D : nullptr, I, D);
    L.o = r.take((size_t)D * I);
    r.reg(p + "0.SelfAttention.o.weight", L.o, D, I);
    L.ln1 = r.take(D);
    r.reg(p + "1.layer_norm.weight", L.ln1, D, 0);
    if (gated) {
      L.wi = r.take((size_t)2 * F * D);  // fused [wi_0 | wi_1]
      r.reg(p + "1.DenseReluDense.wi_0.weight", L.wi, F, D);
      r.reg(p + "1.DenseReluDense.wi_1.weight", L.wi ? L.wi + (size_t)F * D : nullptr, F, D);
    } else {
      L.wi = r.take((size_t)F * D);
      r.reg(p + "1.DenseReluDense.wi.weight", L.wi, F, D);
    }
    L.wo = r.take((size_t)D * F);
    r.reg(p + "1.DenseReluDense.wo.weight", L.wo, D, F);
  }
  m->final_ln = r.take(D);
  r.reg("encoder.final_layer_norm.weight", m->final_ln, D, 0);
}

extern "C" void fmi_t5_default_config(fmi_t5_config* c) {
  if (!c) return;
  // google/t5-v1_1-xxl encoder as shipped in FLUX.1's text_encoder_2/config.json
  c->vocab_size = 32128, c->d_model = 4096, c->d_kv = 64, c->d_ff = 10240, c->num_layers = 24, c->num_heads = 64;
  c->relative_attention_num_buckets = 32, c->relative_attention_max_distance = 128, c->layer_norm_epsilon = 1e-6f;
  c->feed_forward_act = FMI_T5_GATED_GELU;
}

extern "C" int fmi_t5_create(const fmi_t5_config* cfg, fmi_t5** out) {
  if (!cfg || !out) return fail(FMI_ERR_INVALID, "t5_create: null argument");
  if (cfg->d_kv != 64) return fail(FMI_ERR_UNSUPPORTED, "t5_create: d_kv must be 64 (got " + std::to_string(cfg->d_kv) + ")");
  if (cfg->d_model % 64 || cfg->d_ff % 64 || cfg->d_model <= 0 || cfg->num_layers <= 0 || cfg->num_heads <= 0 || cfg->vocab_size <= 0)
    return fail(FMI_ERR_INVALID, "t5_create: d_model and d_ff must be positive multiples of 64");
  if (cfg->feed_forward_act < FMI_T5_RELU || cfg->feed_forward_act > FMI_T5_GATED_SILU) return fail(FMI_ERR_INVALID, "t5_create: bad feed_forward_act");
  if (cfg->relative_attention_num_buckets < 4 || cfg->relative_attention_max_distance < 2) return fail(FMI_ERR_INVALID, "t5_create: bad relative attention config");
  fmi_t5* m = new fmi_t5();
  m->cfg = *cfg;
  t5_layout(m);  // pass 0: size
  m->r.bytes = align_up(m->r.used, 256);
  hipError_t e = hipMalloc((void**)&m->r.arena, m->r.bytes);
  if (e != hipSuccess) {
    delete m;
    return fail(FMI_ERR_HIP, std::string("t5_create: hipMalloc of the weight arena: ") + hipGetErrorString(e));
  }
  t5_layout(m);  // pass 1: pointers
  *out = m;
  return FMI_OK;
}
extern "C" void fmi_t5_destroy(fmi_t5* m) {
  if (!m) return;
  if (m->r.arena) (void)hipFree(m->r.arena);
  if (m->ws.base) (void)hipFree(m->ws.base);
  delete m;
}
extern "C" int fmi_t5_set_tensor(fmi_t5* m, const char* name, const void* data, fmi_dtype dtype, const int64_t* shape, int rank) {
  if (m) FMI_TRY(use_device_ordinal(m->device));
  if (!m) return fail(FMI_ERR_INVALID, "t5_set_tensor: null model");
  return m->r.set("t5_set_tensor", name, data, dtype, shape, rank);
}
// Quantised T5 Linears.  The reference builds every T5 Linear through `linear_no_bias(.., &cfg.quantization_config, ..)`
// (t5/mod.rs:132-133,164-173,258-261) -> BnbLinear (bitsandbytes/mod.rs:111-239), whose forward is "dequantise W, then matmul"
// (:293-312).  The encoder runs ONCE per image (25 ms of 3.2 s), so the codes are expanded once, here, into the linear's slot of
// the bf16 arena with the library's own dequant kernels (kDequantizeBlockwise / dequantize_8bit semantics: bit-exact with the
// stand-alone entry points) and the forward is the dense one: the same numbers BnbLinear::forward produces, no per-call expansion.
namespace {
int t5_linear_dest(fmi_t5* m, const char* who, const char* prefix, int out_features, int in_features, Dest* out, std::string* wname) {
  *wname = std::string(prefix) + ".weight";
  auto it = m->r.names.find(*wname);
  if (it == m->r.names.end() || !it->second.cols) return fail(FMI_ERR_INVALID, std::string(who) + ": unknown linear '" + prefix + "'");
  if (wname->find("SelfAttention.") == std::string::npos && wname->find("DenseReluDense.") == std::string::npos)
    return fail(FMI_ERR_INVALID, std::string(who) + ": '" + prefix + "' is not a Linear of the encoder blocks");
  if (it->second.rows != out_features || it->second.cols != in_features) return fail(FMI_ERR_INVALID, std::string(who) + ": shape mismatch for " + *wname);
  *out = it->second;
  return FMI_OK;
}
}  // namespace
extern "C" int fmi_t5_set_linear_bnb4(fmi_t5* m, const char* prefix, const uint8_t* packed, const float* absmax, int blocksize, int quant_type,
                                      int out_features, int in_features) {
  if (m) FMI_TRY(use_device_ordinal(m->device));
  if (!m || !prefix || !packed || !absmax) return fail(FMI_ERR_INVALID, "t5_set_linear_bnb4: null argument");
  if (quant_type != 1 && quant_type != 2) return fail(FMI_ERR_INVALID, "t5_set_linear_bnb4: quant_type must be 1 (fp4) or 2 (nf4)");
  const int64_t n = (int64_t)out_features * in_features;
  if (blocksize <= 0 || (blocksize & (blocksize - 1)) || n % blocksize || n % 2) return fail(FMI_ERR_UNSUPPORTED, "t5_set_linear_bnb4: blocksize must be a power of two dividing the weight");
  if (n >= (1ll << 31)) return fail(FMI_ERR_UNSUPPORTED, "t5_set_linear_bnb4: linear too large");
  Dest d;
  std::string wname;
  FMI_TRY(t5_linear_dest(m, "t5_set_linear_bnb4", prefix, out_features, in_features, &d, &wname));
  uint8_t* dq = nullptr;
  float* da = nullptr;
  FMI_HIP_TRY(hipMalloc((void**)&dq, (size_t)n / 2));
  hipError_t e = hipMalloc((void**)&da, (size_t)(n / blocksize) * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(dq, packed, (size_t)n / 2, hipMemcpyDefault);
  if (e == hipSuccess) e = hipMemcpy(da, absmax, (size_t)(n / blocksize) * sizeof(float), hipMemcpyDefault);
  if (e == hipSuccess) {
    if (quant_type == 2) dequantize_blockwise_bf16_nf4(nullptr, dq, da, d.ptr, blocksize, (int)n, nullptr);
    else dequantize_blockwise_bf16_fp4(nullptr, dq, da, d.ptr, blocksize, (int)n, nullptr);
    e = hipDeviceSynchronize();
  }
  (void)hipFree(dq);
  if (da) (void)hipFree(da);
  if (e != hipSuccess) return fail(FMI_ERR_HIP, std::string("t5_set_linear_bnb4: ") + hipGetErrorString(e));
  m->r.missing.erase(wname);
  return FMI_OK;
}
extern "C" int fmi_t5_set_linear_int8(fmi_t5* m, const char* prefix, const int8_t* weight, const float* scb, int out_features, int in_features) {
  if (m) FMI_TRY(use_device_ordinal(m->device));
  if (!m || !prefix || !weight || !scb) return fail(FMI_ERR_INVALID, "t5_set_linear_int8: null argument");
  const int64_t n = (int64_t)out_features * in_features;
  Dest d;
  std::string wname;
  FMI_TRY(t5_linear_dest(m, "t5_set_linear_int8", prefix, out_features, in_features, &d, &wname));
  int8_t* dw = nullptr;
  float* ds = nullptr;
  FMI_HIP_TRY(hipMalloc((void**)&dw, (size_t)n));
  hipError_t e = hipMalloc((void**)&ds, (size_t)out_features * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(dw, weight, (size_t)n, hipMemcpyDefault);
  if (e == hipSuccess) e = hipMemcpy(ds, scb, (size_t)out_features * sizeof(float), hipMemcpyDefault);
  int rc = FMI_OK;
  if (e == hipSuccess) {
    rc = launch_dequant_int8_scb_bf16(dw, ds, d.ptr, in_features, n, nullptr);
    e = hipDeviceSynchronize();
  }
  (void)hipFree(dw);
  if (ds) (void)hipFree(ds);
  if (e != hipSuccess) return fail(FMI_ERR_HIP, std::string("t5_set_linear_int8: ") + hipGetErrorString(e));
  if (rc) return rc;
  m->r.missing.erase(wname);
  return FMI_OK;
}
extern "C" int fmi_t5_missing_count(const fmi_t5* m) { return m ? (int)m->r.missing.size() : 0; }
extern "C" const char* fmi_t5_missing_name(fmi_t5* m, int i) {
  if (!m) return nullptr;
  m->r.missing_list.assign(m->r.missing.begin(), m->r.missing.end());
  return (i >= 0 && i < (int)m->r.missing_list.size()) ? m->r.missing_list[i].c_str() : nullptr;
}
extern "C" size_t fmi_t5_size_in_bytes(const fmi_t5* m) { return m ? m->r.bytes + m->ws.bytes : 0; }

extern "C" int fmi_t5_forward(fmi_t5* m, const int32_t* input_ids, int B, int T, void* out, fmi_dtype out_dtype, void* stream) {
  if (m) FMI_TRY(use_device_ordinal(m->device));
  if (!m || !input_ids || !out) return fail(FMI_ERR_INVALID, "t5_forward: null argument");
  if (B <= 0 || T <= 0) return fail(FMI_ERR_INVALID, "t5_forward: empty batch");
  FMI_TRY(m->r.ready("t5_forward"));
  hipStream_t s = (hipStream_t)stream;
  const fmi_t5_config& c = m->cfg;
  const int D = c.d_model, H = c.num_heads, I = H * 64, F = c.d_ff, rows = B * T;
  const bool gated = c.feed_forward_act != FMI_T5_RELU;
  const int FW = gated ? 2 * F : F;
  // workspace carve
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_x = carve((size_t)rows * D * 4), o_n = carve((size_t)rows * D * 2), o_qkv = carve((size_t)rows * 3 * I * 2), o_a = carve((size_t)rows * I * 2);
  const size_t o_h = carve((size_t)rows * FW * 2), o_g = carve((size_t)rows * F * 2), o_bias = carve((size_t)H * T * T * 4), o_ones = carve((size_t)D * 4);
  const size_t o_ids = carve((size_t)rows * 4), o_bkt = carve((size_t)(2 * T - 1) * 4), o_err = carve(4), o_of = carve((size_t)rows * D * 4);
  FMI_TRY(m->ws.reserve(off));
  char* w = m->ws.base;
  float* x = (float*)(w + o_x);
  bf16_t* n = (bf16_t*)(w + o_n);
  bf16_t* qkv = (bf16_t*)(w + o_qkv);
  bf16_t* a = (bf16_t*)(w + o_a);
  bf16_t* h = (bf16_t*)(w + o_h);
  bf16_t* g = (bf16_t*)(w + o_g);
  float* bias = (float*)(w + o_bias);
  float* ones = (float*)(w + o_ones);
  int32_t* ids = (int32_t*)(w + o_ids);
  int* bkt = (int*)(w + o_bkt);
  int* err = (int*)(w + o_err);
  float* of = (float*)(w + o_of);

  FMI_HIP_TRY(hipMemcpyAsync(ids, input_ids, (size_t)rows * 4, hipMemcpyDefault, s));
  FMI_HIP_TRY(hipMemsetAsync(err, 0, 4, s));
  m->bucket_host.resize(2 * T - 1);
  for (int rel = -(T - 1); rel <= T - 1; ++rel) m->bucket_host[rel + T - 1] = t5_bucket_of(rel, c.relative_attention_num_buckets, c.relative_attention_max_distance);
  FMI_HIP_TRY(hipMemcpyAsync(bkt, m->bucket_host.data(), (size_t)(2 * T - 1) * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(fill_f32_kernel, dim3(16), dim3(256), 0, s, ones, D, 1.0f);
  hipLaunchKernelGGL(t5_bias_kernel, dim3(T, H), dim3(256), 0, s, m->rel, bkt, H, T, bias);
  hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(256), 0, s, ids, m->shared, (const bf16_t*)nullptr, T, D, c.vocab_size, x, err);
  FMI_LAUNCH_CHECK();
  const int nb = (rows + 3) / 4;
  for (int l = 0; l < c.num_layers; ++l) {
    const fmi_t5::Layer& L = m->layers[l];
    // T5LayerSelfAttention (t5/mod.rs:412-424)
    hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(nb), dim3(256), 0, s, x, L.ln0, c.layer_norm_epsilon, rows, D, n, (float*)nullptr);
    GemmProblem p = lin(n, D, L.qkv, nullptr, qkv, 3 * I, rows, 3 * I, D, EPI_STORE_BF16);
    FMI_TRY(launch_gemm(&p, 1, s));
    FMI_TRY(enc_attention(qkv, bias, B, T, H, 1.0f, 0, a, s));  // no 1/sqrt(d) in T5 (:317)
    p = lin(a, I, L.o, nullptr, x, D, rows, D, I, EPI_RESID_GATE_F32);
    p.gate = ones;
    FMI_TRY(launch_gemm(&p, 1, s));
    // T5LayerFF (:222-231)
    hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(nb), dim3(256), 0, s, x, L.ln1, c.layer_norm_epsilon, rows, D, n, (float*)nullptr);
    p = lin(n, D, L.wi, nullptr, h, FW, rows, FW, D, EPI_STORE_BF16);
    FMI_TRY(launch_gemm(&p, 1, s));
    const int act = c.feed_forward_act == FMI_T5_RELU ? 0 : (c.feed_forward_act == FMI_T5_GATED_GELU ? 1 : 2);
    hipLaunchKernelGGL(enc_act_kernel, dim3(1024), dim3(256), 0, s, h, rows, F, act, gated ? 1 : 0, g);
    p = lin(g, F, L.wo, nullptr, x, D, rows, D, F, EPI_RESID_GATE_F32);
    p.gate = ones;
    FMI_TRY(launch_gemm(&p, 1, s));
  }
  hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(nb), dim3(256), 0, s, x, m->final_ln, c.layer_norm_epsilon, rows, D, n, of);
  FMI_LAUNCH_CHECK();
  FMI_TRY(write_out(of, n, out, out_dtype, (int64_t)rows * D, s));
  int herr = 0;
  FMI_HIP_TRY(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
  FMI_HIP_TRY(hipStreamSynchronize(s));
  if (herr) return fail(FMI_ERR_INVALID, "t5_forward: token id out of range [0, vocab_size)");
  return FMI_OK;
}

// ============================================================================================= CLIP
struct fmi_clip {
  fmi_clip_config cfg;
  int device = current_device();
  Registry r;
  Workspace ws;
  bf16_t *tok = nullptr, *pos = nullptr, *fw = nullptr, *fb = nullptr;
  struct Layer {
    bf16_t *l1w, *l1b, *qkv, *qkvb, *o, *ob, *l2w, *l2b, *f1, *f1b, *f2, *f2b;
  };
  std::vector<Layer> layers;
};

static void clip_layout(fmi_clip* m) {
  const fmi_clip_config& c = m->cfg;
  const int D = c.projection_dim, F = c.intermediate_size;
  Registry& r = m->r;
  r.used = 0;
  r.names.clear();
  r.missing.clear();
  const std::string tm = "text_model.";
  m->tok = r.take((size_t)c.vocab_size * D);
  r.reg(tm + "embeddings.token_embedding.weight", m->tok, c.vocab_size, D);
  m->pos = r.take((size_t)c.max_position_embeddings * D);
  r.reg(tm + "embeddings.position_embedding.weight", m->pos, c.max_position_embeddings, D);
  m->layers.resize(c.num_hidden_layers);
  for (int l = 0; l < c.num_hidden_layers; ++l) {
    fmi_clip::Layer& L = m->layers[l];
    const std::string p = tm + "encoder.layers." + std::to_string(l) + ".";
    L.l1w = r.take(D), L.l1b = r.take(D);
    r.reg(p + "layer_norm1.weight", L.l1w, D, 0), r.reg(p + "layer_norm1.bias", L.l1b, D, 0);
    L.qkv = r.take((size_t)3 * D * D), L.qkvb = r.take((size_t)3 * D);
    const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
    for (int i = 0; i < 3; ++i) {
      r.reg(p + "self_attn." + nm[i] + ".weight", L.qkv ? L.qkv + (size_t)i * D * D : nullptr, D, D);
      r.reg(p + "self_attn." + nm[i] + ".bias", L.qkvb ? L.qkvb + (size_t)i * D : nullptr, D, 0);
    }
    L.o = r.take((size_t)D * D), L.ob = r.take(D);
    r.reg(p + "self_attn.out_proj.weight", L.o, D, D), r.reg(p + "self_attn.out_proj.bias", L.ob, D, 0);
    L.l2w = r.take(D), L.l2b = r.take(D);
    r.reg(p + "layer_norm2.weight", L.l2w, D, 0), r.reg(p + "layer_norm2.bias", L.l2b, D, 0);
    L.f1 = r.take((size_t)F * D), L.f1b = r.take(F);
    r.reg(p + "mlp.fc1.weight", L.f1, F, D), r.reg(p + "mlp.fc1.bias", L.f1b, F, 0);
    L.f2 = r.take((size_t)D * F), L.f2b = r.take(D);
    r.reg(p + "mlp.fc2.weight", L.f2, D, F), r.reg(p + "mlp.fc2.bias", L.f2b, D, 0);
  }
  m->fw = r.take(D), m->fb = r.take(D);
  r.reg(tm + "final_layer_norm.weight", m->fw, D, 0), r.reg(tm + "final_layer_norm.bias", m->fb, D, 0);
}

extern "C" void fmi_clip_default_config(fmi_clip_config* c) {
  if (!c) return;
  // openai/clip-vit-large-patch14 text tower as shipped in FLUX.1's text_encoder/config.json
  c->vocab_size = 49408, c->projection_dim = 768, c->intermediate_size = 3072, c->max_position_embeddings = 77, c->num_hidden_layers = 12, c->num_attention_heads = 12;
}
extern "C" int fmi_clip_create(const fmi_clip_config* cfg, fmi_clip** out) {
  if (!cfg || !out) return fail(FMI_ERR_INVALID, "clip_create: null argument");
  if (cfg->num_attention_heads <= 0 || cfg->projection_dim != cfg->num_attention_heads * 64)
    return fail(FMI_ERR_UNSUPPORTED, "clip_create: head dim (projection_dim / num_attention_heads) must be 64");
  if (cfg->projection_dim % 64 || cfg->intermediate_size % 64 || cfg->intermediate_size <= 0 || cfg->num_hidden_layers <= 0 || cfg->vocab_size <= 0 || cfg->max_position_embeddings <= 0)
    return fail(FMI_ERR_INVALID, "clip_create: widths must be positive multiples of 64");
  fmi_clip* m = new fmi_clip();
  m->cfg = *cfg;
  clip_layout(m);
  m->r.bytes = align_up(m->r.used, 256);
  hipError_t e = hipMalloc((void**)&m->r.arena, m->r.bytes);
  if (e != hipSuccess) {
    delete m;
    return fail(FMI_ERR_HIP, std::string("clip_create: hipMalloc of the weight arena: ") + hipGetErrorString(e));
  }
  clip_layout(m);
  *out = m;
  return FMI_OK;
}
extern "C" void fmi_clip_destroy(fmi_clip* m) {
  if (!m) return;
  if (m->r.arena) (void)hipFree(m->r.arena);
  if (m->ws.base) (void)hipFree(m->ws.base);
  delete m;
}
extern "C" int fmi_clip_set_tensor(fmi_clip* m, const char* name, const void* data, fmi_dtype dtype, const int64_t* shape, int rank) {
  if (m) FMI_TRY(use_device_ordinal(m->device));
  if (!m) return fail(FMI_ERR_INVALID, "clip_set_tensor: null model");
  return m->r.set("clip_set_tensor", name, data, dtype, shape, rank);
}
extern "C" int fmi_clip_missing_count(const fmi_clip* m) { return m ? (int)m->r.missing.size() : 0; }
extern "C" const char* fmi_clip_missing_name(fmi_clip* m, int i) {
  if (!m) return nullptr;
  m->r.missing_list.assign(m->r.missing.begin(), m->r.missing.end());
  return (i >= 0 && i < (int)m->r.missing_list.size()) ? m->r.missing_list[i].c_str() : nullptr;
}
extern "C" size_t fmi_clip_size_in_bytes(const fmi_clip* m) { return m ? m->r.bytes + m->ws.bytes : 0; }

extern "C" int fmi_clip_forward(fmi_clip* m, const int32_t* input_ids, int B, int T, void* pooled_out, fmi_dtype pooled_dtype, float* hidden_out, void* stream) {
  if (m) FMI_TRY(use_device_ordinal(m->device));
  if (!m || !input_ids || !pooled_out) return fail(FMI_ERR_INVALID, "clip_forward: null argument");
  if (B <= 0 || T <= 0) return fail(FMI_ERR_INVALID, "clip_forward: empty batch");
  const fmi_clip_config& c = m->cfg;
  if (T > c.max_position_embeddings) return fail(FMI_ERR_INVALID, "clip_forward: sequence longer than max_position_embeddings");
  FMI_TRY(m->r.ready("clip_forward"));
  hipStream_t s = (hipStream_t)stream;
  const int D = c.projection_dim, H = c.num_attention_heads, F = c.intermediate_size, rows = B * T;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_x = carve((size_t)rows * D * 4), o_n = carve((size_t)rows * D * 2), o_qkv = carve((size_t)rows * 3 * D * 2), o_a = carve((size_t)rows * D * 2);
  const size_t o_h = carve((size_t)rows * F * 2), o_g = carve((size_t)rows * F * 2), o_ones = carve((size_t)D * 4), o_ids = carve((size_t)rows * 4), o_err = carve(4);
  const size_t o_hf = carve((size_t)rows * D * 4), o_pf = carve((size_t)B * D * 4), o_pb = carve((size_t)B * D * 2);
  FMI_TRY(m->ws.reserve(off));
  char* w = m->ws.base;
  float* x = (float*)(w + o_x);
  bf16_t* n = (bf16_t*)(w + o_n);
  bf16_t* qkv = (bf16_t*)(w + o_qkv);
  bf16_t* a = (bf16_t*)(w + o_a);
  bf16_t* h = (bf16_t*)(w + o_h);
  bf16_t* g = (bf16_t*)(w + o_g);
  float* ones = (float*)(w + o_ones);
  int32_t* ids = (int32_t*)(w + o_ids);
  int* err = (int*)(w + o_err);
  float* hf = (float*)(w + o_hf);
  float* pf = (float*)(w + o_pf);
  bf16_t* pb = (bf16_t*)(w + o_pb);

  FMI_HIP_TRY(hipMemcpyAsync(ids, input_ids, (size_t)rows * 4, hipMemcpyDefault, s));
  FMI_HIP_TRY(hipMemsetAsync(err, 0, 4, s));
  hipLaunchKernelGGL(fill_f32_kernel, dim3(16), dim3(256), 0, s, ones, D, 1.0f);
  hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(256), 0, s, ids, m->tok, m->pos, T, D, c.vocab_size, x, err);
  FMI_LAUNCH_CHECK();
  const int nb = (rows + 3) / 4;
  const float scale = 1.0f / sqrtf(64.0f);  // (head_dim)^-0.5, clip/text.rs:101
  for (int l = 0; l < c.num_hidden_layers; ++l) {
    const fmi_clip::Layer& L = m->layers[l];
    // ClipEncoderLayer::forward (clip/text.rs:228-238)
    hipLaunchKernelGGL(layernorm_affine_kernel, dim3(nb), dim3(256), 0, s, x, L.l1w, L.l1b, 1e-5f, rows, D, n, (float*)nullptr);
    GemmProblem p = lin(n, D, L.qkv, L.qkvb, qkv, 3 * D, rows, 3 * D, D, EPI_STORE_BF16);
    FMI_TRY(launch_gemm(&p, 1, s));
    FMI_TRY(enc_attention(qkv, nullptr, B, T, H, scale, 1, a, s));  // causal mask (:273-291), q * scale (:130)
    p = lin(a, D, L.o, L.ob, x, D, rows, D, D, EPI_RESID_GATE_F32);
    p.gate = ones;
    FMI_TRY(launch_gemm(&p, 1, s));
    hipLaunchKernelGGL(layernorm_affine_kernel, dim3(nb), dim3(256), 0, s, x, L.l2w, L.l2b, 1e-5f, rows, D, n, (float*)nullptr);
    p = lin(n, D, L.f1, L.f1b, h, F, rows, F, D, EPI_STORE_BF16);
    FMI_TRY(launch_gemm(&p, 1, s));
    hipLaunchKernelGGL(enc_act_kernel, dim3(256), dim3(256), 0, s, h, rows, F, 3, 0, g);  // QuickGelu (:13-19)
    p = lin(g, F, L.f2, L.f2b, x, D, rows, D, F, EPI_RESID_GATE_F32);
    p.gate = ones;
    FMI_TRY(launch_gemm(&p, 1, s));
  }
  hipLaunchKernelGGL(layernorm_affine_kernel, dim3(nb), dim3(256), 0, s, x, m->fw, m->fb, 1e-5f, rows, D, (bf16_t*)nullptr, hf);
  hipLaunchKernelGGL(clip_pool_kernel, dim3(B), dim3(256), 0, s, hf, ids, T, D, pf);
  FMI_LAUNCH_CHECK();
  if (pooled_dtype == FMI_BF16) FMI_TRY(launch_cast_to_bf16(pf, FMI_F32, pb, (int64_t)B * D, s));
  FMI_TRY(write_out(pf, pb, pooled_out, pooled_dtype, (int64_t)B * D, s));
  if (hidden_out) FMI_HIP_TRY(hipMemcpyAsync(hidden_out, hf, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, s));
  int herr = 0;
  FMI_HIP_TRY(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
  FMI_HIP_TRY(hipStreamSynchronize(s));
  if (herr) return fail(FMI_ERR_INVALID, "clip_forward: token id out of range [0, vocab_size)");
  return FMI_OK;
}
