// text_oracle.cpp — CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT) of the two text encoders that
// feed the FLUX hot path: T5EncoderModel and ClipTextTransformer.
// Plain f32 restatement of the reference's CPU semantics; every function cites the lines it
// follows under /root/reference/diffusion_rs_core/src/models/.  Built into libflux_oracle.so.
//
// Pin status: the reference has no test or golden vector for either encoder ("parity unpinned" by
// the reference).  The restatement is pinned by independent implementations instead: HuggingFace
// transformers' T5EncoderModel / CLIPTextModel (the code the reference says it follows,
// t5/mod.rs:3-4) run in this container on seeded random weights, outputs committed under
// tests/golden/text_encoders.npz with the generating script (tests/test_oracle_text.py).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "flux_oracle.h"

namespace {
struct Store {
  std::map<std::string, std::vector<float>> t;
  const float* get(const std::string& name, int64_t numel) const {
    auto it = t.find(name);
    if (it == t.end()) {
      fprintf(stderr, "[oracle] missing tensor %s\n", name.c_str());
      return nullptr;
    }
    if ((int64_t)it->second.size() != numel) {
      fprintf(stderr, "[oracle] tensor %s has %zu elements, expected %lld\n", name.c_str(), it->second.size(), (long long)numel);
      return nullptr;
    }
    return it->second.data();
  }
};

// T5LayerNorm::forward (t5/mod.rs:110-121): variance = mean(x^2) in f32, x / sqrt(var + eps), * weight
void t5_layer_norm(const float* x, const float* w, float eps, int rows, int D, float* out) {
#pragma omp parallel for
  for (int r = 0; r < rows; ++r) {
    const float* xr = x + (int64_t)r * D;
    double s = 0.0;
    for (int i = 0; i < D; ++i) s += (double)xr[i] * xr[i];
    const float inv = 1.0f / sqrtf((float)(s / D) + eps);
    for (int i = 0; i < D; ++i) out[(int64_t)r * D + i] = xr[i] * inv * w[i];
  }
}

// NewGelu (nn/activation.rs: Activation::NewGelu -> gelu tanh approximation, core/op.rs:539-582)
inline float new_gelu(float v) { return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * v * (1.0f + 0.044715f * v * v))); }
inline float silu(float v) { return v / (1.0f + expf(-v)); }

// relative position bucket, bidirectional case exactly as T5Attention::forward computes it
// (t5/mod.rs:340-376): note f32::log(x, base), truncation toward zero, and the clamps.
int t5_bucket(int i, int j, int num_buckets_total, int max_distance) {
  const unsigned num_buckets = (unsigned)num_buckets_total / 2;
  const unsigned max_exact = num_buckets / 2;
  auto large = [&](unsigned dist) -> unsigned {
    const float b = (logf((float)dist / (float)max_exact) / logf((float)max_distance / (float)max_exact)) * (float)(num_buckets - max_exact);
    return (unsigned)b;
  };
  if (i < j) {
    const unsigned d = (unsigned)(j - i);
    if (d < max_exact) return (int)(d + num_buckets);
    const unsigned v = max_exact + num_buckets + large(d);
    return (int)(v < (unsigned)num_buckets_total - 1 ? v : (unsigned)num_buckets_total - 1);
  }
  const unsigned d = (unsigned)(i - j);
  if (d < max_exact) return (int)d;
  const unsigned v = max_exact + large(d);
  return (int)(v < num_buckets - 1 ? v : num_buckets - 1);
}

// softmax(q k^T * scale + bias) v for one (batch, head); q,k,v rows have stride `ld` floats.
// bias: (Lq, Lk) additive or null.  Plain f32 like the reference's matmul/softmax/matmul chain.
void attn_head(const float* q, const float* k, const float* v, int ld, int L, int d, float scale, const float* bias, float* out, int ldo) {
  std::vector<float> s(L);
  for (int i = 0; i < L; ++i) {
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) {
      float acc = 0.f;
      for (int c = 0; c < d; ++c) acc += q[(int64_t)i * ld + c] * k[(int64_t)j * ld + c];
      acc = acc * scale + (bias ? bias[(int64_t)i * L + j] : 0.f);
      s[j] = acc;
      mx = acc > mx ? acc : mx;
    }
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
      s[j] = expf(s[j] - mx);
      sum += s[j];
    }
    for (int c = 0; c < d; ++c) {
      float acc = 0.f;
      for (int j = 0; j < L; ++j) acc += s[j] * v[(int64_t)j * ld + c];
      out[(int64_t)i * ldo + c] = acc / sum;
    }
  }
}
}  // namespace

// ------------------------------------------------------------------------------------------ T5
struct orc_t5 {
  int vocab, d_model, d_kv, d_ff, layers, heads, buckets, max_distance, act;  // act: 0 relu (ungated), 1 gated-gelu, 2 gated-silu
  float eps;
  Store s;
};

extern "C" orc_t5* orc_t5_create(int vocab_size, int d_model, int d_kv, int d_ff, int num_layers, int num_heads, int rel_buckets, int rel_max_distance, float eps, int act) {
  orc_t5* m = new orc_t5();
  m->vocab = vocab_size, m->d_model = d_model, m->d_kv = d_kv, m->d_ff = d_ff, m->layers = num_layers, m->heads = num_heads;
  m->buckets = rel_buckets, m->max_distance = rel_max_distance, m->eps = eps, m->act = act;
  return m;
}
extern "C" void orc_t5_destroy(orc_t5* m) { delete m; }
extern "C" int orc_t5_set_tensor(orc_t5* m, const char* name, const float* data, int64_t numel) {
  m->s.t[name] = std::vector<float>(data, data + numel);
  return 0;
}
extern "C" int orc_t5_bucket(int i, int j, int num_buckets_total, int max_distance) { return t5_bucket(i, j, num_buckets_total, max_distance); }

// T5EncoderModel::forward -> T5Stack::forward (t5/mod.rs:589-606,629-631): embedding lookup,
// blocks (self-attention layer then feed-forward layer, no cross attention for the encoder:
// T5Block::forward :527-563 with cross_attn = None, mask = None), final_layer_norm.
// The position bias is built by block 0 and reused by every block (:321-382, :598-603).
extern "C" int orc_t5_forward(orc_t5* m, const int32_t* ids, int B, int T, float* out) {
  const int D = m->d_model, H = m->heads, dk = m->d_kv, I = H * dk, F = m->d_ff;
  const float* emb = m->s.get("shared.weight", (int64_t)m->vocab * D);
  const float* rel = m->s.get("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (int64_t)m->buckets * H);
  if (!emb || !rel) return -1;
  const int rows = B * T;
  std::vector<float> x((int64_t)rows * D), n((int64_t)rows * D), q((int64_t)rows * I), k((int64_t)rows * I), v((int64_t)rows * I), a((int64_t)rows * I),
      y((int64_t)rows * D), h0((int64_t)rows * F), h1((int64_t)rows * F);
  for (int r = 0; r < rows; ++r) {
    const int id = ids[r];
    if (id < 0 || id >= m->vocab) {
      fprintf(stderr, "[oracle] t5 token id %d out of range\n", id);
      return -2;
    }
    memcpy(&x[(int64_t)r * D], emb + (int64_t)id * D, D * sizeof(float));
  }
  // position_bias (1, H, T, T): embedding(bucket(i,j)) permuted (t5/mod.rs:377-381)
  std::vector<float> bias((int64_t)H * T * T);
  for (int i = 0; i < T; ++i)
    for (int j = 0; j < T; ++j) {
      const int b = t5_bucket(i, j, m->buckets, m->max_distance);
      for (int h = 0; h < H; ++h) bias[((int64_t)h * T + i) * T + j] = rel[(int64_t)b * H + h];
    }
  for (int l = 0; l < m->layers; ++l) {
    const std::string p = "encoder.block." + std::to_string(l) + ".layer.";
    // ---- T5LayerSelfAttention::forward (:412-424): x + attn(layer_norm(x))
    const float* ln0 = m->s.get(p + "0.layer_norm.weight", D);
    const float* wq = m->s.get(p + "0.SelfAttention.q.weight", (int64_t)I * D);
    const float* wk = m->s.get(p + "0.SelfAttention.k.weight", (int64_t)I * D);
    const float* wv = m->s.get(p + "0.SelfAttention.v.weight", (int64_t)I * D);
    const float* wo = m->s.get(p + "0.SelfAttention.o.weight", (int64_t)D * I);
    if (!ln0 || !wq || !wk || !wv || !wo) return -1;
    t5_layer_norm(x.data(), ln0, m->eps, rows, D, n.data());
    orc_linear(n.data(), wq, nullptr, rows, I, D, q.data());
    orc_linear(n.data(), wk, nullptr, rows, I, D, k.data());
    orc_linear(n.data(), wv, nullptr, rows, I, D, v.data());
    // T5Attention::forward (:282-392): scores = q k^T (NO 1/sqrt(d) scaling) + position_bias; softmax; @ v
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
      for (int h = 0; h < H; ++h) {
        const int64_t o = (int64_t)b * T * I + (int64_t)h * dk;
        attn_head(q.data() + o, k.data() + o, v.data() + o, I, T, dk, 1.0f, &bias[(int64_t)h * T * T], a.data() + o, I);
      }
    orc_linear(a.data(), wo, nullptr, rows, D, I, y.data());
    for (int64_t i = 0; i < (int64_t)rows * D; ++i) x[i] += y[i];
    // ---- T5LayerFF::forward (:222-231): x + dense(layer_norm(x))
    const float* ln1 = m->s.get(p + "1.layer_norm.weight", D);
    if (!ln1) return -1;
    t5_layer_norm(x.data(), ln1, m->eps, rows, D, n.data());
    if (m->act == 0) {  // T5DenseActDense (:142-149): wo(relu(wi(x)))
      const float* wi = m->s.get(p + "1.DenseReluDense.wi.weight", (int64_t)F * D);
      const float* wo2 = m->s.get(p + "1.DenseReluDense.wo.weight", (int64_t)D * F);
      if (!wi || !wo2) return -1;
      orc_linear(n.data(), wi, nullptr, rows, F, D, h0.data());
      for (auto& e : h0) e = e > 0.f ? e : 0.f;
      orc_linear(h0.data(), wo2, nullptr, rows, D, F, y.data());
    } else {  // T5DenseGatedActDense (:183-191): wo(act(wi_0(x)) * wi_1(x))
      const float* wi0 = m->s.get(p + "1.DenseReluDense.wi_0.weight", (int64_t)F * D);
      const float* wi1 = m->s.get(p + "1.DenseReluDense.wi_1.weight", (int64_t)F * D);
      const float* wo2 = m->s.get(p + "1.DenseReluDense.wo.weight", (int64_t)D * F);
      if (!wi0 || !wi1 || !wo2) return -1;
      orc_linear(n.data(), wi0, nullptr, rows, F, D, h0.data());
      orc_linear(n.data(), wi1, nullptr, rows, F, D, h1.data());
      for (int64_t i = 0; i < (int64_t)rows * F; ++i) h0[i] = (m->act == 1 ? new_gelu(h0[i]) : silu(h0[i])) * h1[i];
      orc_linear(h0.data(), wo2, nullptr, rows, D, F, y.data());
    }
    for (int64_t i = 0; i < (int64_t)rows * D; ++i) x[i] += y[i];
  }
  const float* fl = m->s.get("encoder.final_layer_norm.weight", D);
  if (!fl) return -1;
  t5_layer_norm(x.data(), fl, m->eps, rows, D, out);
  return 0;
}

// ---------------------------------------------------------------------------------------- CLIP
struct orc_clip {
  int vocab, hidden, inter, max_pos, layers, heads;
  Store s;
};

// ClipTextConfig (clip/text.rs:24-33): the reference uses `projection_dim` as the hidden width.
extern "C" orc_clip* orc_clip_create(int vocab_size, int projection_dim, int intermediate_size, int max_position_embeddings, int num_hidden_layers, int num_attention_heads) {
  orc_clip* m = new orc_clip();
  m->vocab = vocab_size, m->hidden = projection_dim, m->inter = intermediate_size, m->max_pos = max_position_embeddings, m->layers = num_hidden_layers, m->heads = num_attention_heads;
  return m;
}
extern "C" void orc_clip_destroy(orc_clip* m) { delete m; }
extern "C" int orc_clip_set_tensor(orc_clip* m, const char* name, const float* data, int64_t numel) {
  m->s.t[name] = std::vector<float>(data, data + numel);
  return 0;
}

// ClipTextTransformer::forward (clip/text.rs:303-317): forward_with_mask(ids, usize::MAX) then the
// row at argmax(token id) of every sequence.  hidden_out (B,T,hidden) optional, pooled (B,hidden).
extern "C" int orc_clip_forward(orc_clip* m, const int32_t* ids, int B, int T, float* hidden_out, float* pooled) {
  const int D = m->hidden, H = m->heads, dh = D / H, F = m->inter;
  const std::string tm = "text_model.";
  const float* tok = m->s.get(tm + "embeddings.token_embedding.weight", (int64_t)m->vocab * D);
  const float* pos = m->s.get(tm + "embeddings.position_embedding.weight", (int64_t)m->max_pos * D);
  if (!tok || !pos || T > m->max_pos) return -1;
  const int rows = B * T;
  std::vector<float> x((int64_t)rows * D), n((int64_t)rows * D), q((int64_t)rows * D), k((int64_t)rows * D), v((int64_t)rows * D), a((int64_t)rows * D), y((int64_t)rows * D),
      h((int64_t)rows * F);
  // ClipTextEmbeddings::forward (:63-71)
  for (int r = 0; r < rows; ++r) {
    const int id = ids[r];
    if (id < 0 || id >= m->vocab) return -2;
    for (int i = 0; i < D; ++i) x[(int64_t)r * D + i] = tok[(int64_t)id * D + i] + pos[(int64_t)(r % T) * D + i];
  }
  // build_causal_attention_mask (:273-291) with mask_after = usize::MAX: f32::MIN where j > i
  std::vector<float> mask((int64_t)T * T);
  for (int i = 0; i < T; ++i)
    for (int j = 0; j < T; ++j) mask[(int64_t)i * T + j] = j > i ? -3.4028234663852886e38f : 0.f;
  const float scale = 1.0f / sqrtf((float)dh);  // (:101)
  for (int l = 0; l < m->layers; ++l) {
    const std::string p = tm + "encoder.layers." + std::to_string(l) + ".";
    auto W = [&](const std::string& nme, int64_t numel) { return m->s.get(p + nme, numel); };
    const float *l1w = W("layer_norm1.weight", D), *l1b = W("layer_norm1.bias", D), *l2w = W("layer_norm2.weight", D), *l2b = W("layer_norm2.bias", D);
    const float *qw = W("self_attn.q_proj.weight", (int64_t)D * D), *qb = W("self_attn.q_proj.bias", D), *kw = W("self_attn.k_proj.weight", (int64_t)D * D), *kb = W("self_attn.k_proj.bias", D);
    const float *vw = W("self_attn.v_proj.weight", (int64_t)D * D), *vb = W("self_attn.v_proj.bias", D), *ow = W("self_attn.out_proj.weight", (int64_t)D * D), *ob = W("self_attn.out_proj.bias", D);
    const float *f1w = W("mlp.fc1.weight", (int64_t)F * D), *f1b = W("mlp.fc1.bias", F), *f2w = W("mlp.fc2.weight", (int64_t)D * F), *f2b = W("mlp.fc2.bias", D);
    if (!l1w || !l1b || !l2w || !l2b || !qw || !qb || !kw || !kb || !vw || !vb || !ow || !ob || !f1w || !f1b || !f2w || !f2b) return -1;
    // ClipEncoderLayer::forward (:228-238)
    orc_layer_norm(x.data(), l1w, l1b, 1e-5f, rows, D, n.data());
    // ClipAttention::forward (:126-168): q scaled BEFORE the matmul, additive mask, softmax, @ v
    orc_linear(n.data(), qw, qb, rows, D, D, q.data());
    for (auto& e : q) e *= scale;
    orc_linear(n.data(), kw, kb, rows, D, D, k.data());
    orc_linear(n.data(), vw, vb, rows, D, D, v.data());
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
      for (int hh = 0; hh < H; ++hh) {
        const int64_t o = (int64_t)b * T * D + (int64_t)hh * dh;
        attn_head(q.data() + o, k.data() + o, v.data() + o, D, T, dh, 1.0f, mask.data(), a.data() + o, D);
      }
    orc_linear(a.data(), ow, ob, rows, D, D, y.data());
    for (int64_t i = 0; i < (int64_t)rows * D; ++i) x[i] += y[i];
    orc_layer_norm(x.data(), l2w, l2b, 1e-5f, rows, D, n.data());
    // ClipMlp::forward (:190-194) with QuickGelu x * sigmoid(1.702 x) (:13-19)
    orc_linear(n.data(), f1w, f1b, rows, F, D, h.data());
    for (auto& e : h) e = e / (1.0f + expf(-1.702f * e));
    orc_linear(h.data(), f2w, f2b, rows, D, F, y.data());
    for (int64_t i = 0; i < (int64_t)rows * D; ++i) x[i] += y[i];
  }
  const float *fw = m->s.get(tm + "final_layer_norm.weight", D), *fb = m->s.get(tm + "final_layer_norm.bias", D);
  if (!fw || !fb) return -1;
  orc_layer_norm(x.data(), fw, fb, 1e-5f, rows, D, n.data());
  if (hidden_out) memcpy(hidden_out, n.data(), (size_t)rows * D * sizeof(float));
  // pooled = hidden state at argmax(input_ids) (first maximum, as Tensor::argmax)
  for (int b = 0; b < B; ++b) {
    int best = 0;
    for (int t = 1; t < T; ++t)
      if (ids[b * T + t] > ids[b * T + best]) best = t;
    memcpy(pooled + (int64_t)b * D, &n[((int64_t)b * T + best) * D], D * sizeof(float));
  }
  return 0;
}
