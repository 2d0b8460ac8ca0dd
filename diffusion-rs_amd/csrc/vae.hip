// vae.hip — AutoencoderKL decoder on MI355X: AutoEncoderKl::decode
// (diffusion_rs_core/src/models/vaes/autoencoder_kl.rs:112-119) -> Decoder::forward
// (vaes/vae.rs:436-456), ResnetBlock (vae.rs:157-172), AttnBlock (vae.rs:95-111), Upsample
// (vae.rs:223-229), GroupNorm (diffusion_rs_common/src/nn/group_norm.rs:39-74).
//
// The reference runs every Conv2d as im2col (2.4 GB at 1024^2) + GEMM + strided copy and every
// GroupNorm as 8 eager f32 passes.  Here activations are NHWC bf16 end to end, convolutions are
// implicit GEMMs on the MFMA main loop of gemm_bf16.hip (taps gathered by the LDS DMA, the
// nearest-2x upsample folded into the gather, bias + residual add in the epilogue), GroupNorm is
// one statistics pass (f32 partials, f64 finalize) + one fused normalise*w+b(+SiLU) pass, and
// the single-head mid-block attention (seq = h*w, dim 512) is four GEMMs and a row softmax.
#include <map>
#include <set>
#include <vector>

#include "common.h"

using namespace fmi;

namespace fmi {

// ---------------------------------------------------------------- layout conversion kernels
// z (B,C,h,w) f32 NCHW -> (B,h,w,Cpad) bf16 NHWC, channels >= C zero
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict z, bf16_t* __restrict out, int C, int Cpad, int HW, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int64_t p = i / Cpad;  // b*HW + pix
    const int64_t b = p / HW, pix = p % HW;
    out[i] = c < C ? f32_to_bf16(z[(b * C + c) * HW + pix]) : (bf16_t)0;
  }
}
// (B,HW,C) bf16 NHWC -> (B,C,HW) f32 NCHW
__global__ void nhwc_to_nchw_f32_kernel(const bf16_t* __restrict x, float* __restrict out, int C, int HW, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % HW;
    const int c = (int)((i / HW) % C);
    const int64_t b = i / ((int64_t)HW * C);
    out[i] = bf16_to_f32(x[(b * HW + pix) * C + c]);
  }
}
// conv weight (Cout,Cin,k,k) bf16 -> (Cout,k,k,Cinpad) bf16 (zero padded channels)
__global__ void conv_weight_relayout_kernel(const bf16_t* __restrict w, bf16_t* __restrict out, int Cin, int Cinpad, int kk, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cinpad);
    const int tap = (int)((i / Cinpad) % kk);
    const int64_t co = i / ((int64_t)Cinpad * kk);
    out[i] = ci < Cin ? w[(co * Cin + ci) * kk + tap] : (bf16_t)0;
  }
}

// ---------------------------------------------------------------- GroupNorm (NHWC)
// pixels per statistics block: about a thousand blocks per sample at every resolution of the decoder (a fixed 1024 pixels left the
// 128 x 128 stages of a 1024 x 1024 decode on 16 of the 256 CUs)
static inline int gn_pix_per_block(int HW) {
  const int p = HW / 1024;
  return p < 32 ? 32 : (p > 1024 ? 1024 : p);
}
static inline int gn_chunks(int HW) { return cdiv(HW, gn_pix_per_block(HW)); }
static inline int gn_max_chunks(int HW) { return std::max(2048, cdiv(HW, 1024)); }  // bound of gn_chunks over every HW' <= HW
// partial[(b*nchunks + chunk)*G + g] = {sum, sumsq} over the chunk's pixels; requires C%8==0, cpg%4==0
// T = bf16_t (activations inside a block) or float (the f32 trunk, round 5): 8 channels per thread either way
template <typename T>
__device__ __forceinline__ void gn_load8(const T* p, float (&v)[8]) {
  if constexpr (sizeof(T) == 4) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  } else {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32(e[i]);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict x, float2* __restrict partial, int HW, int C, int G, int ppb) {
  __shared__ float4 part[256];  // per-thread {sum_lo, sumsq_lo, sum_hi, sumsq_hi} (channels 0-3 / 4-7 of its 8)
  const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int tpp = C >> 3;  // threads per pixel
  const int cpg = C / G;
  const int c8 = threadIdx.x % tpp;
  const int pstep = 256 / tpp;
  const int nact = tpp * pstep;
  const int p0 = chunk * ppb, p1 = min(HW, p0 + ppb);
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  if ((int)threadIdx.x < nact) {
#pragma unroll 4  // four 16-byte loads in flight per thread (one per iteration left the kernel latency-bound at 1.5 TB/s)
    for (int p = p0 + threadIdx.x / tpp; p < p1; p += pstep) {
      float e[8];
      gn_load8(x + ((int64_t)b * HW + p) * C + c8 * 8, e);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = e[i];
        s0 += v;
        q0 += v * v;
      }
#pragma unroll
      for (int i = 4; i < 8; ++i) {
        const float v = e[i];
        s1 += v;
        q1 += v * v;
      }
    }
  }
  part[threadIdx.x] = make_float4(s0, q0, s1, q1);
  __syncthreads();
  // fixed-order combine (no atomics: results are bit-reproducible run to run), in two steps: the pstep threads that share a channel
  // octet, then the octet halves of a group — 32 threads each scanning all 256 partials (the first version) cost more than the loads
  if ((int)threadIdx.x < tpp) {
    float4 a = part[threadIdx.x];
    for (int k = 1; k < pstep; ++k) {
      const float4 v = part[threadIdx.x + k * tpp];
      a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
    }
    part[threadIdx.x] = a;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    float s = 0.f, q = 0.f;
    for (int o = (g * cpg) >> 3; o <= ((g + 1) * cpg - 1) >> 3; ++o) {
      const float4 v = part[o];
      if ((o * 8) / cpg == g) s += v.x, q += v.y;
      if ((o * 8 + 4) / cpg == g) s += v.z, q += v.w;
    }
    partial[((int64_t)b * nchunks + chunk) * G + g] = make_float2(s, q);
  }
}
// stats[b*G+g] = {mean, 1/sqrt(var+eps)}; f64 combine of the f32 partials.  One 256-thread block per (group, sample): at 1024 x 1024 a
// group has 4096 partials, and one thread walking them (the first version) is 4096 dependent loads = 250 us for a 32 KB reduction.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float2* __restrict partial, float2* __restrict stats, int nchunks, int G, double count, float eps) {
  __shared__ double red[2][256];
  const int g = blockIdx.x, b = blockIdx.y;
  double s = 0.0, q = 0.0;
  for (int c = threadIdx.x; c < nchunks; c += 256) {
    const float2 v = partial[((int64_t)b * nchunks + c) * G + g];
    s += v.x;
    q += v.y;
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {  // fixed tree: the same sums in the same order on every run
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = red[0][0] / count;
    double var = red[1][0] / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stats[b * G + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
  }
}
// raw != nullptr: x itself leaves as bf16 too (the operand copy of the f32 trunk that a ResnetBlock's 1 x 1 shortcut convolution reads)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict x, const float2* __restrict stats, const float* __restrict w,
                                                       const float* __restrict bia, bf16_t* __restrict out, bf16_t* __restrict raw_out, int HW, int C, int G,
                                                       int silu_on, int64_t nvec) {
  const int tpp = C >> 3, cpg = C / G;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // a thread keeps its 8 channels for the whole grid-stride walk whenever the stride is a multiple of the threads per pixel (every
  // FLUX width: C / 8 divides 256): the affine weights are then loop invariants instead of four dependent loads per 16 bytes of x
  const bool fixed = stride % tpp == 0;
  float ww[8], bb[8];
  auto load_affine = [&](int c8) {
    const float4 w0 = *reinterpret_cast<const float4*>(w + c8 * 8), w1 = *reinterpret_cast<const float4*>(w + c8 * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(bia + c8 * 8), b1 = *reinterpret_cast<const float4*>(bia + c8 * 8 + 4);
    ww[0] = w0.x, ww[1] = w0.y, ww[2] = w0.z, ww[3] = w0.w, ww[4] = w1.x, ww[5] = w1.y, ww[6] = w1.z, ww[7] = w1.w;
    bb[0] = b0.x, bb[1] = b0.y, bb[2] = b0.z, bb[3] = b0.w, bb[4] = b1.x, bb[5] = b1.y, bb[6] = b1.z, bb[7] = b1.w;
  };
  if (fixed && i0 < nvec) load_affine((int)(i0 % tpp));
#pragma unroll 2
  for (int64_t i = i0; i < nvec; i += stride) {
    const int c8 = (int)(i % tpp);
    const int64_t p = i / tpp;
    const int b = (int)(p / HW);
    float e[8];
    gn_load8(x + i * 8, e);
    const float2 st0 = stats[b * G + (c8 * 8) / cpg], st1 = stats[b * G + (c8 * 8 + 4) / cpg];
    if (!fixed) load_affine(c8);
    if (raw_out)
      *reinterpret_cast<uint4*>(raw_out + i * 8) = make_uint4(pack_bf16x2(e[0], e[1]), pack_bf16x2(e[2], e[3]), pack_bf16x2(e[4], e[5]), pack_bf16x2(e[6], e[7]));
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 st = k < 4 ? st0 : st1;
      float y = (e[k] - st.x) * st.y * ww[k] + bb[k];
      v[k] = silu_on ? silu(y) : y;
    }
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
}

template <typename T>
static int launch_groupnorm_nhwc_t(const T* x, const float* w, const float* b, bf16_t* out, bf16_t* raw_out, int B, int HW, int C, int G, float eps,
                                   int silu_on, float2* partial, float2* stats, hipStream_t s) {
  if (C % 8 || C % G || (C / G) % 4) return fail(FMI_ERR_UNSUPPORTED, "groupnorm: needs C % 8 == 0 and (C/groups) % 4 == 0");
  if (C / 8 > 256) return fail(FMI_ERR_UNSUPPORTED, "groupnorm: C > 2048 not supported");
  const int nchunks = gn_chunks(HW);
  hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(nchunks, B), dim3(256), 0, s, x, partial, HW, C, G, gn_pix_per_block(HW));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, B), dim3(256), 0, s, partial, stats, nchunks, G, (double)HW * (C / G), eps);
  const int64_t nvec = (int64_t)B * HW * (C / 8);
  hipLaunchKernelGGL(gn_apply_kernel<T>, dim3((unsigned)std::min<int64_t>(cdiv64(nvec, 256), 256 * 16)), dim3(256), 0, s, x, stats, w, b, out, raw_out, HW, C,
                     G, silu_on, nvec);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
int launch_groupnorm_nhwc(const bf16_t* x, const float* w, const float* b, bf16_t* out, int B, int HW, int C, int G, float eps, int silu_on,
                          float2* partial, float2* stats, hipStream_t s) {
  return launch_groupnorm_nhwc_t<bf16_t>(x, w, b, out, nullptr, B, HW, C, G, eps, silu_on, partial, stats, s);
}
// bf16 -> f32 widening / f32 -> bf16 rounding of an activation (the f32 trunk's operand copies and the op-level AttnBlock entry)
__global__ void widen_bf16_kernel(const bf16_t* __restrict x, float* __restrict out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float e[8];
    gn_load8(x + i * 8, e);
    *reinterpret_cast<float4*>(out + i * 8) = make_float4(e[0], e[1], e[2], e[3]);
    *reinterpret_cast<float4*>(out + i * 8 + 4) = make_float4(e[4], e[5], e[6], e[7]);
  }
}
__global__ void round_bf16_kernel(const float* __restrict x, bf16_t* __restrict out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float e[8];
    gn_load8(x + i * 8, e);
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(pack_bf16x2(e[0], e[1]), pack_bf16x2(e[2], e[3]), pack_bf16x2(e[4], e[5]), pack_bf16x2(e[6], e[7]));
  }
}

// row softmax, bf16 in place, f32 math (softmax_last_dim, nn/ops.rs:419-448)
__global__ __launch_bounds__(256) void softmax_rows_bf16_kernel(bf16_t* __restrict x, int cols) {
  __shared__ float red[4];
  bf16_t* r = x + (int64_t)blockIdx.x * cols;
  const int nv = cols >> 3;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const uint4 raw = reinterpret_cast<const uint4*>(r)[i];
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
#pragma unroll
    for (int k = 0; k < 8; ++k) mx = fmaxf(mx, bf16_to_f32(e[k]));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const uint4 raw = reinterpret_cast<const uint4*>(r)[i];
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += __expf(bf16_to_f32(e[k]) - mx);
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / sum;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const uint4 raw = reinterpret_cast<const uint4*>(r)[i];
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __expf(bf16_to_f32(e[k]) - mx) * inv;
    reinterpret_cast<uint4*>(r)[i] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
}

GemmProblem conv_problem(const bf16_t* x, const bf16_t* w, const bf16_t* bias, const bf16_t* resid, bf16_t* out, int B, int in_h, int in_w,
                         int cin_pad, int cout, int ks, int up, const bf16_t* zero) {
  GemmProblem p{};
  p.A = x, p.W = w, p.bias = bias, p.out = out, p.resid = resid;
  p.M = up < 0 ? B * (in_h / 2) * (in_w / 2) : B * (in_h << up) * (in_w << up);
  p.N = cout;
  p.K = ks * ks * cin_pad;
  p.lda = cin_pad, p.ldw = p.K, p.ldo = cout;
  p.epi = resid ? EPI_RESID_ADD_BF16 : EPI_STORE_BF16;
  p.alpha = 1.f;
  p.cv_ks = ks, p.cv_h = in_h, p.cv_w = in_w, p.cv_cin = cin_pad, p.cv_up = up, p.cv_zero = zero;
  return p;
}
// the same convolution writing the f32 trunk: trunk = conv + bias (add == false) or trunk += conv + bias (the gated-residual epilogue with a gate of ones)
GemmProblem conv_problem_f32(const bf16_t* x, const bf16_t* w, const bf16_t* bias, float* trunk, bool add, const float* ones, int B, int in_h, int in_w,
                             int cin_pad, int cout, int ks, int up, const bf16_t* zero) {
  GemmProblem p = conv_problem(x, w, bias, nullptr, nullptr, B, in_h, in_w, cin_pad, cout, ks, up, zero);
  p.out = trunk;
  p.epi = add ? EPI_RESID_GATE_F32 : EPI_STORE_F32;
  p.gate = ones;
  return p;
}

}  // namespace fmi

namespace {
int pad64(int c) { return (c + 63) / 64 * 64; }

struct Conv {
  bf16_t* w = nullptr;  // (cout, ks, ks, cin_pad)
  bf16_t* b = nullptr;  // (cout)
  int cin = 0, cin_pad = 0, cout = 0, ks = 0;
};
struct GN {
  float* w = nullptr;
  float* b = nullptr;
  int c = 0;
};
struct Resnet {
  GN n1, n2;
  Conv c1, c2, sc;
  int cin = 0, cout = 0;
};
struct Dest {
  int kind;  // 0 conv weight, 1 bf16 vector (bias), 2 f32 vector (gn), 3 linear weight (cout,cin) -> conv 1x1
  void* ptr;
  Conv* conv;
  int64_t numel;
  std::vector<int64_t> shape;
};
}  // namespace

struct fmi_vae {
  fmi_vae_config cfg;
  int device = current_device();
  std::vector<void*> allocs;
  Conv conv_in, conv_out;
  Resnet mid1, mid2;
  GN attn_gn, norm_out;
  Conv aq, ak, av, ao;
  std::vector<std::vector<Resnet>> up;
  std::vector<Conv> upconv;  // per level (cout == 0 if none)
  // encoder (vae.rs:236-349) + optional quant_conv (autoencoder_kl.rs:67-77)
  Conv e_conv_in, e_conv_out, quant_conv;
  Resnet e_mid1, e_mid2;
  GN e_attn_gn, e_norm_out;
  Conv e_aq, e_ak, e_av, e_ao;
  std::vector<std::vector<Resnet>> down;
  std::vector<Conv> downconv;  // per level (cout == 0 if none)
  bf16_t* zero = nullptr;
  std::map<std::string, Dest> names;
  std::set<std::string> missing;
  std::vector<std::string> missing_list;
  // workspace
  int wB = 0, wh = 0, ww = 0;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  // fx / fx2: the TRUNK (what ResnetBlocks, the AttnBlock and the samplers add to) in f32 since round 5 — its bf16 rounding after every block was what
  // cost the decoder its u8 agreement (tools/vae_rounding_study.py: all activations bf16 99.54 % within 2; trunk f32, everything else bf16: 100 %);
  // bt1 / bt2 / bsc: bf16 activations inside a block (every MFMA operand); ones: the gate of the f32 read-modify-write epilogue
  float *fx = nullptr, *fx2 = nullptr, *ones = nullptr;
  bf16_t *bt1 = nullptr, *bt2 = nullptr, *bsc = nullptr, *scores = nullptr;
  float2 *partial = nullptr, *stats = nullptr;
};

namespace {
template <typename T>
T* valloc(fmi_vae* v, size_t count) {
  void* p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(count * sizeof(T), 256)) != hipSuccess) return nullptr;
  hipMemset(p, 0, std::max<size_t>(count * sizeof(T), 256));
  v->allocs.push_back(p);
  return (T*)p;
}
bool make_conv(fmi_vae* v, Conv& c, const std::string& p, int cin, int cout, int ks, bool linear = false) {
  c.cin = cin, c.cin_pad = pad64(cin), c.cout = cout, c.ks = ks;
  c.w = valloc<bf16_t>(v, (size_t)cout * ks * ks * c.cin_pad);
  c.b = valloc<bf16_t>(v, cout);
  if (!c.w || !c.b) return false;
  Dest dw{linear ? 3 : 0, c.w, &c, (int64_t)cout * cin * ks * ks, {}};
  if (linear)
    dw.shape = {cout, cin};
  else
    dw.shape = {cout, cin, ks, ks};
  v->names[p + ".weight"] = dw;
  v->names[p + ".bias"] = Dest{1, c.b, nullptr, cout, {cout}};
  v->missing.insert(p + ".weight");
  v->missing.insert(p + ".bias");
  return true;
}
bool make_gn(fmi_vae* v, GN& g, const std::string& p, int c) {
  g.c = c;
  g.w = valloc<float>(v, c);
  g.b = valloc<float>(v, c);
  if (!g.w || !g.b) return false;
  v->names[p + ".weight"] = Dest{2, g.w, nullptr, c, {c}};
  v->names[p + ".bias"] = Dest{2, g.b, nullptr, c, {c}};
  v->missing.insert(p + ".weight");
  v->missing.insert(p + ".bias");
  return true;
}
bool make_resnet(fmi_vae* v, Resnet& r, const std::string& p, int cin, int cout) {
  r.cin = cin, r.cout = cout;
  bool ok = make_gn(v, r.n1, p + ".norm1", cin) && make_conv(v, r.c1, p + ".conv1", cin, cout, 3) && make_gn(v, r.n2, p + ".norm2", cout) &&
            make_conv(v, r.c2, p + ".conv2", cout, cout, 3);
  if (ok && cin != cout) ok = make_conv(v, r.sc, p + ".conv_shortcut", cin, cout, 1);
  return ok;
}

int vae_workspace(fmi_vae* v, int B, int h, int w) {
  if (v->ws && v->wB == B && v->wh == h && v->ww == w) return FMI_OK;
  const fmi_vae_config& c = v->cfg;
  // largest NHWC activation: walk the decoder
  size_t max_act = (size_t)B * h * w * pad64(c.latent_channels);
  int H = h, W = w, ch = c.block_out_channels[c.n_blocks - 1];
  max_act = std::max(max_act, (size_t)B * H * W * ch);
  for (int lvl = 0; lvl < c.n_blocks; ++lvl) {
    const int co = c.block_out_channels[c.n_blocks - 1 - lvl];
    max_act = std::max(max_act, (size_t)B * H * W * std::max(ch, co));
    ch = co;
    if (lvl != 3) {
      H *= 2, W *= 2;
      max_act = std::max(max_act, (size_t)B * H * W * ch);
    }
  }
  const size_t act_bytes = (max_act * 2 + 255) / 256 * 256;
  const size_t hw = (size_t)h * w;
  const size_t score_bytes = c.mid_block_add_attention ? (hw * hw * 2 + 255) / 256 * 256 : 256;
  const int G = c.norm_num_groups;
  const size_t part_bytes = ((size_t)B * gn_max_chunks((int)((size_t)H * W)) * G * sizeof(float2) + 255) / 256 * 256;
  const size_t stat_bytes = ((size_t)B * G * sizeof(float2) + 255) / 256 * 256;
  const size_t ones_bytes = 4096 * sizeof(float);
  const size_t total = 7 * act_bytes + score_bytes + part_bytes + stat_bytes + ones_bytes;
  if (v->ws) {
    FMI_HIP_TRY(hipDeviceSynchronize());
    FMI_HIP_TRY(hipFree(v->ws));
    v->ws = nullptr;
  }
  FMI_HIP_TRY(hipMalloc((void**)&v->ws, total));
  char* p = v->ws;
  v->fx = (float*)p, p += 2 * act_bytes;
  v->fx2 = (float*)p, p += 2 * act_bytes;
  v->bt1 = (bf16_t*)p, p += act_bytes;
  v->bt2 = (bf16_t*)p, p += act_bytes;
  v->bsc = (bf16_t*)p, p += act_bytes;
  v->scores = (bf16_t*)p, p += score_bytes;
  v->partial = (float2*)p, p += part_bytes;
  v->stats = (float2*)p, p += stat_bytes;
  v->ones = (float*)p;
  {
    std::vector<float> one(4096, 1.0f);
    FMI_HIP_TRY(hipMemcpy(v->ones, one.data(), ones_bytes, hipMemcpyHostToDevice));
  }
  v->ws_bytes = total;
  v->wB = B, v->wh = h, v->ww = w;
  return FMI_OK;
}

int run_conv(fmi_vae* v, const Conv& c, const bf16_t* x, const bf16_t* resid, bf16_t* out, int B, int H, int W, int up, hipStream_t s) {
  GemmProblem p = conv_problem(x, c.w, c.b, resid, out, B, H, W, c.cin_pad, c.cout, c.ks, up, v->zero);
  return launch_gemm(&p, 1, s);
}
// trunk = conv(x) + bias (add == false) / trunk += conv(x) + bias
int run_conv_trunk(fmi_vae* v, const Conv& c, const bf16_t* x, float* trunk, bool add, int B, int H, int W, int up, hipStream_t s) {
  if (c.cout > 4096) return fail(FMI_ERR_UNSUPPORTED, "vae: more than 4096 channels");
  GemmProblem p = conv_problem_f32(x, c.w, c.b, trunk, add, v->ones, B, H, W, c.cin_pad, c.cout, c.ks, up, v->zero);
  return launch_gemm(&p, 1, s);
}
int run_gn(fmi_vae* v, const GN& g, const bf16_t* x, bf16_t* out, int B, int HW, int silu_on, hipStream_t s) {
  return launch_groupnorm_nhwc(x, g.w, g.b, out, B, HW, g.c, v->cfg.norm_num_groups, 1e-6f, silu_on, v->partial, v->stats, s);
}
int run_gn_trunk(fmi_vae* v, const GN& g, const float* x, bf16_t* out, bf16_t* raw_out, int B, int HW, int silu_on, hipStream_t s) {
  return launch_groupnorm_nhwc_t<float>(x, g.w, g.b, out, raw_out, B, HW, g.c, v->cfg.norm_num_groups, 1e-6f, silu_on, v->partial, v->stats, s);
}
int trunk_to_bf16(fmi_vae* v, bf16_t* out, int64_t n, hipStream_t s) {  // the operand copy a sampler's convolution reads
  if (n % 8) return fail(FMI_ERR_UNSUPPORTED, "vae: activation size must be a multiple of 8");
  hipLaunchKernelGGL(round_bf16_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n / 8, 256), 8192)), dim3(256), 0, s, v->fx, out, n / 8);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
// ResnetBlock::forward (vae.rs:157-172): x <- shortcut(x) + conv2(silu(gn2(conv1(silu(gn1(x)))))), x = the f32 trunk
int run_resnet(fmi_vae* v, const Resnet& r, int B, int H, int W, hipStream_t s) {
  const bool sc = r.cin != r.cout;
  FMI_TRY(run_gn_trunk(v, r.n1, v->fx, v->bt1, sc ? v->bsc : nullptr, B, H * W, 1, s));
  FMI_TRY(run_conv(v, r.c1, v->bt1, nullptr, v->bt2, B, H, W, 0, s));
  FMI_TRY(run_gn(v, r.n2, v->bt2, v->bt1, B, H * W, 1, s));
  if (sc) {  // the 1 x 1 shortcut starts the new trunk (another channel count), conv2 adds to it
    FMI_TRY(run_conv_trunk(v, r.sc, v->bsc, v->fx2, false, B, H, W, 0, s));
    std::swap(v->fx, v->fx2);
  }
  FMI_TRY(run_conv_trunk(v, r.c2, v->bt1, v->fx, true, B, H, W, 0, s));
  return FMI_OK;
}
// AttnBlock::forward (vae.rs:95-111) with the model-dtype sdpa of vae.rs:28-33
struct AttnW {
  const GN& gn;
  const Conv &q, &k, &v, &o;
};
int run_attn(fmi_vae* v, const AttnW& a, int B, int H, int W, hipStream_t s) {
  const int C = a.q.cout, HW = H * W;
  if (HW % 64) return fail(FMI_ERR_UNSUPPORTED, "vae attention: latent h*w must be a multiple of 64");
  FMI_TRY(run_gn_trunk(v, a.gn, v->fx, v->bt1, nullptr, B, HW, 0, s));
  const float scale = (float)(1.0 / sqrt((double)C));
  for (int b = 0; b < B; ++b) {
    const bf16_t* xn = v->bt1 + (size_t)b * HW * C;
    bf16_t* q = v->bt2;                       // (HW, C)
    bf16_t* k = v->bt2 + (size_t)HW * C;      // (HW, C)
    bf16_t* vt = v->bt2 + (size_t)2 * HW * C; // (C, HW)
    bf16_t* o = v->bsc;                       // (HW, C)
    GemmProblem p[2];
    p[0] = conv_problem(xn, a.q.w, a.q.b, nullptr, q, 1, H, W, C, C, 1, 0, v->zero);
    p[1] = conv_problem(xn, a.k.w, a.k.b, nullptr, k, 1, H, W, C, C, 1, 0, v->zero);
    FMI_TRY(launch_gemm(p, 2, s));
    // V^T (C, HW) = Wv (C,C) · xn(HW,C)^T ; the v bias is added after P·V (softmax rows sum to 1)
    GemmProblem pv{};
    pv.A = a.v.w, pv.W = xn, pv.out = vt, pv.M = C, pv.N = HW, pv.K = C, pv.lda = C, pv.ldw = C, pv.ldo = HW, pv.epi = EPI_STORE_BF16, pv.alpha = 1.f;
    FMI_TRY(launch_gemm(&pv, 1, s));
    // scores = (q k^T) * scale  -> bf16 (HW, HW)
    GemmProblem ps{};
    ps.A = q, ps.W = k, ps.out = v->scores, ps.M = HW, ps.N = HW, ps.K = C, ps.lda = C, ps.ldw = C, ps.ldo = HW, ps.epi = EPI_SCALE_BF16, ps.alpha = scale;
    FMI_TRY(launch_gemm(&ps, 1, s));
    hipLaunchKernelGGL(softmax_rows_bf16_kernel, dim3(HW), dim3(256), 0, s, v->scores, HW);
    FMI_LAUNCH_CHECK();
    // o = P · V + b_v
    GemmProblem po{};
    po.A = v->scores, po.W = vt, po.bias = a.v.b, po.out = o, po.M = HW, po.N = C, po.K = HW, po.lda = HW, po.ldw = HW, po.ldo = C, po.epi = EPI_STORE_BF16,
    po.alpha = 1.f;
    FMI_TRY(launch_gemm(&po, 1, s));
    // x[b] += to_out(o)   (in place on the f32 trunk)
    GemmProblem pf = conv_problem_f32(o, a.o.w, a.o.b, v->fx + (size_t)b * HW * C, true, v->ones, 1, H, W, C, C, 1, 0, v->zero);
    FMI_TRY(launch_gemm(&pf, 1, s));
  }
  return FMI_OK;
}
}  // namespace

extern "C" void fmi_vae_default_config(fmi_vae_config* c) {
  c->in_channels = 3, c->out_channels = 3;
  c->block_out_channels[0] = 128, c->block_out_channels[1] = 256, c->block_out_channels[2] = 512, c->block_out_channels[3] = 512;
  c->n_blocks = 4, c->layers_per_block = 2, c->latent_channels = 16, c->norm_num_groups = 32;
  c->mid_block_add_attention = 1, c->use_post_quant_conv = 0;
  c->scaling_factor = 0.3611, c->shift_factor = 0.1159;
  c->use_quant_conv = 0;
}

extern "C" int fmi_vae_create(const fmi_vae_config* cfg, fmi_model_dtype dtype, fmi_vae** out) {
  if (!cfg || !out) return fail(FMI_ERR_INVALID, "vae_create: null argument");
  if (dtype == FMI_MODEL_F16 || dtype == FMI_MODEL_F32) return fail(FMI_ERR_UNSUPPORTED, "vae_create: only the bf16 compute path is implemented");
  if (cfg->n_blocks != 4) return fail(FMI_ERR_UNSUPPORTED, "vae_create: n_blocks must be 4 (the reference hard-codes `i_level != 3`, vae.rs:412)");
  if (cfg->use_post_quant_conv)
    return fail(FMI_ERR_UNSUPPORTED,
                "vae_create: use_post_quant_conv=true is a shape error in the reference (the 1x1 latent conv is applied to the decoded image, "
                "autoencoder_kl.rs:78-88,114-117); FLUX ships false");
  for (int i = 0; i < 4; ++i)
    if (cfg->block_out_channels[i] % 64 || (cfg->block_out_channels[i] / cfg->norm_num_groups) % 4 || cfg->block_out_channels[i] % cfg->norm_num_groups)
      return fail(FMI_ERR_UNSUPPORTED, "vae_create: block_out_channels must be multiples of 64 with (C/groups) % 4 == 0");
  fmi_vae* v = new fmi_vae();
  v->cfg = *cfg;
  const int nb = cfg->n_blocks;
  int block_in = cfg->block_out_channels[nb - 1];
  bool ok = true;
  v->zero = valloc<bf16_t>(v, 256);
  ok = ok && v->zero;
  ok = ok && make_conv(v, v->conv_in, "decoder.conv_in", cfg->latent_channels, block_in, 3);
  ok = ok && make_resnet(v, v->mid1, "decoder.mid_block.resnets.0", block_in, block_in);
  if (cfg->mid_block_add_attention) {
    const std::string p = "decoder.mid_block.attentions.0";
    ok = ok && make_gn(v, v->attn_gn, p + ".group_norm", block_in);
    ok = ok && make_conv(v, v->aq, p + ".to_q", block_in, block_in, 1, true) && make_conv(v, v->ak, p + ".to_k", block_in, block_in, 1, true) &&
         make_conv(v, v->av, p + ".to_v", block_in, block_in, 1, true) && make_conv(v, v->ao, p + ".to_out.0", block_in, block_in, 1, true);
  }
  ok = ok && make_resnet(v, v->mid2, "decoder.mid_block.resnets.1", block_in, block_in);
  v->up.resize(nb);
  v->upconv.resize(nb);
  for (int lvl = 0; lvl < nb && ok; ++lvl) {
    const int block_out = cfg->block_out_channels[nb - 1 - lvl];
    v->up[lvl].resize(cfg->layers_per_block + 1);
    for (int i = 0; i <= cfg->layers_per_block && ok; ++i) {
      ok = make_resnet(v, v->up[lvl][i], "decoder.up_blocks." + std::to_string(lvl) + ".resnets." + std::to_string(i), block_in, block_out);
      block_in = block_out;
    }
    if (lvl != 3 && ok) ok = make_conv(v, v->upconv[lvl], "decoder.up_blocks." + std::to_string(lvl) + ".upsamplers.0.conv", block_in, block_in, 3);
  }
  ok = ok && make_gn(v, v->norm_out, "decoder.conv_norm_out", cfg->block_out_channels[0]);
  ok = ok && make_conv(v, v->conv_out, "decoder.conv_out", cfg->block_out_channels[0], cfg->out_channels, 3);
  // ---- encoder (Encoder::new, vae.rs:249-327)
  {
    int ch = cfg->block_out_channels[0];
    ok = ok && make_conv(v, v->e_conv_in, "encoder.conv_in", cfg->in_channels, ch, 3);
    v->down.resize(nb);
    v->downconv.resize(nb);
    for (int lvl = 0; lvl < nb && ok; ++lvl) {
      const int block_out = cfg->block_out_channels[lvl];
      const std::string p = "encoder.down_blocks." + std::to_string(lvl);
      v->down[lvl].resize(cfg->layers_per_block);
      for (int i = 0; i < cfg->layers_per_block && ok; ++i) {
        ok = make_resnet(v, v->down[lvl][i], p + ".resnets." + std::to_string(i), ch, block_out);
        ch = block_out;
      }
      if (lvl != nb - 1 && ok) ok = make_conv(v, v->downconv[lvl], p + ".downsamplers.0.conv", ch, ch, 3);
    }
    ok = ok && make_resnet(v, v->e_mid1, "encoder.mid_block.resnets.0", ch, ch);
    if (cfg->mid_block_add_attention) {
      const std::string p = "encoder.mid_block.attentions.0";
      ok = ok && make_gn(v, v->e_attn_gn, p + ".group_norm", ch);
      ok = ok && make_conv(v, v->e_aq, p + ".to_q", ch, ch, 1, true) && make_conv(v, v->e_ak, p + ".to_k", ch, ch, 1, true) &&
           make_conv(v, v->e_av, p + ".to_v", ch, ch, 1, true) && make_conv(v, v->e_ao, p + ".to_out.0", ch, ch, 1, true);
    }
    ok = ok && make_resnet(v, v->e_mid2, "encoder.mid_block.resnets.1", ch, ch);
    ok = ok && make_gn(v, v->e_norm_out, "encoder.conv_norm_out", ch);
    ok = ok && make_conv(v, v->e_conv_out, "encoder.conv_out", ch, 2 * cfg->latent_channels, 3);
    if (cfg->use_quant_conv) ok = ok && make_conv(v, v->quant_conv, "quant_conv", 2 * cfg->latent_channels, 2 * cfg->latent_channels, 1);
  }
  if (!ok) {
    fmi_vae_destroy(v);
    return fail(FMI_ERR_NOMEM, "vae_create: device allocation failed");
  }
  *out = v;
  return FMI_OK;
}

extern "C" void fmi_vae_destroy(fmi_vae* v) {
  if (!v) return;
  hipDeviceSynchronize();
  for (void* p : v->allocs) hipFree(p);
  if (v->ws) hipFree(v->ws);
  delete v;
}

extern "C" int fmi_vae_set_tensor(fmi_vae* v, const char* name, const void* data, fmi_dtype dtype, const int64_t* shape, int rank) {
  if (v) FMI_TRY(use_device_ordinal(v->device));
  if (!v || !name || !data) return fail(FMI_ERR_INVALID, "vae_set_tensor: null argument");
  auto it = v->names.find(name);
  if (it == v->names.end()) return fail(FMI_ERR_INVALID, std::string("vae_set_tensor: unknown tensor name '") + name + "'");
  const Dest& d = it->second;
  bool ok = rank == (int)d.shape.size();
  for (int i = 0; ok && i < rank; ++i) ok = shape[i] == d.shape[i];
  // Linear weights of the attention block may also arrive already unsqueezed to (C,C,1,1)
  if (!ok && d.kind == 3 && rank == 4 && shape[0] == d.shape[0] && shape[1] == d.shape[1] && shape[2] == 1 && shape[3] == 1) ok = true;
  if (!ok) return fail(FMI_ERR_INVALID, std::string("vae_set_tensor: shape mismatch for ") + name);
  if (dtype != FMI_F32 && dtype != FMI_F16 && dtype != FMI_BF16) return fail(FMI_ERR_INVALID, "vae_set_tensor: dtype must be F32/F16/BF16");
  const size_t esz = dtype == FMI_F32 ? 4 : 2;
  void* tmp = nullptr;
  FMI_HIP_TRY(hipMalloc(&tmp, d.numel * esz));
  int rc = FMI_OK;
  if (hipMemcpy(tmp, data, d.numel * esz, hipMemcpyDefault) != hipSuccess) rc = fail(FMI_ERR_HIP, "vae_set_tensor: copy failed");
  if (rc == FMI_OK) {
    if (d.kind == 2) {
      rc = launch_cast_to_f32(tmp, dtype, (float*)d.ptr, d.numel, nullptr);
    } else if (d.kind == 1) {
      rc = launch_cast_to_bf16(tmp, dtype, (bf16_t*)d.ptr, d.numel, nullptr);
    } else {
      bf16_t* wb = nullptr;
      if (hipMalloc((void**)&wb, d.numel * 2) != hipSuccess) rc = fail(FMI_ERR_NOMEM, "vae_set_tensor: alloc failed");
      if (rc == FMI_OK) rc = launch_cast_to_bf16(tmp, dtype, wb, d.numel, nullptr);
      if (rc == FMI_OK) {
        const Conv& c = *d.conv;
        const int64_t n = (int64_t)c.cout * c.ks * c.ks * c.cin_pad;
        hipLaunchKernelGGL(conv_weight_relayout_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, nullptr, wb, c.w, c.cin,
                           c.cin_pad, c.ks * c.ks, n);
      }
      hipDeviceSynchronize();
      if (wb) hipFree(wb);
    }
  }
  hipDeviceSynchronize();
  hipFree(tmp);
  if (rc == FMI_OK) v->missing.erase(name);
  return rc;
}

extern "C" int fmi_vae_missing_count(const fmi_vae* v) { return v ? (int)v->missing.size() : 0; }
extern "C" const char* fmi_vae_missing_name(const fmi_vae* v, int i) {
  if (!v || i < 0 || i >= (int)v->missing.size()) return nullptr;
  auto* vv = const_cast<fmi_vae*>(v);
  vv->missing_list.assign(v->missing.begin(), v->missing.end());
  return vv->missing_list[i].c_str();
}
extern "C" double fmi_vae_scale_factor(const fmi_vae* v) { return v ? v->cfg.scaling_factor : 0.0; }
extern "C" double fmi_vae_shift_factor(const fmi_vae* v) { return v ? v->cfg.shift_factor : 0.0; }

namespace {
bool is_encoder_name(const std::string& n) { return n.rfind("encoder.", 0) == 0 || n.rfind("quant_conv.", 0) == 0; }
// decode needs only the decoder's tensors and encode only the encoder's (+ quant_conv)
int check_part(fmi_vae* v, bool decoder) {
  int n = 0;
  const std::string* first = nullptr;
  for (const auto& m : v->missing)
    if (is_encoder_name(m) != decoder) {
      if (!first) first = &m;
      ++n;
    }
  if (n) return fail(FMI_ERR_STATE, std::string("vae ") + (decoder ? "decoder" : "encoder") + ": " + std::to_string(n) + " tensors not set, first: " + *first);
  return FMI_OK;
}

// NHWC bf16 moments (B, hw, 2L) -> z (B, L, hw) f32 NCHW = mean + exp(0.5 logvar) * noise, and
// optionally the moments as f32 NCHW.  DiagonalGaussian::forward (vae.rs:470-480).
__global__ void diag_gaussian_kernel(const bf16_t* __restrict mom, const float* __restrict noise, float* __restrict z, float* __restrict mom_out, int L, int hw,
                                     int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(e % hw);
    const int c = (int)((e / hw) % L);
    const int64_t b = e / ((int64_t)hw * L);
    const bf16_t* m = mom + (b * hw + p) * 2 * L;
    const float mean = bf16_to_f32(m[c]), logvar = bf16_to_f32(m[L + c]);
    z[e] = noise ? mean + __expf(0.5f * logvar) * noise[e] : mean;
    if (mom_out) {
      mom_out[(b * 2 * L + c) * hw + p] = mean;
      mom_out[(b * 2 * L + L + c) * hw + p] = logvar;
    }
  }
}
}  // namespace

// == AutoEncoderKl::encode (autoencoder_kl.rs:103-110) = Encoder::forward (vae.rs:330-349),
// optional quant_conv, DiagonalGaussian.  H, W must be multiples of 8.
extern "C" int fmi_vae_encode(fmi_vae* v, const float* image, int B, int H, int W, const float* noise, float* z_out, float* moments_out, void* stream) {
  if (v) FMI_TRY(use_device_ordinal(v->device));
  if (!v || !image || !z_out) return fail(FMI_ERR_INVALID, "vae_encode: null argument");
  if (B <= 0 || H <= 0 || W <= 0 || H % 8 || W % 8) return fail(FMI_ERR_INVALID, "vae_encode: H and W must be positive multiples of 8");
  FMI_TRY(check_part(v, false));
  hipStream_t s = (hipStream_t)stream;
  const fmi_vae_config& c = v->cfg;
  FMI_TRY(vae_workspace(v, B, H / 8, W / 8));
  {
    const int cp = v->e_conv_in.cin_pad;
    const int64_t n = (int64_t)B * H * W * cp;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, s, image, v->bt1, c.in_channels, cp, H * W, n);
    FMI_LAUNCH_CHECK();
  }
  FMI_TRY(run_conv_trunk(v, v->e_conv_in, v->bt1, v->fx, false, B, H, W, 0, s));
  for (int lvl = 0; lvl < c.n_blocks; ++lvl) {
    for (auto& r : v->down[lvl]) FMI_TRY(run_resnet(v, r, B, H, W, s));
    if (v->downconv[lvl].cout) {  // Downsample::forward (vae.rs:194-201): stride 2, zero column/row on the right/bottom
      FMI_TRY(trunk_to_bf16(v, v->bt1, (int64_t)B * H * W * v->downconv[lvl].cin_pad, s));
      FMI_TRY(run_conv_trunk(v, v->downconv[lvl], v->bt1, v->fx2, false, B, H, W, -1, s));
      std::swap(v->fx, v->fx2);
      H /= 2, W /= 2;
    }
  }
  FMI_TRY(run_resnet(v, v->e_mid1, B, H, W, s));
  if (c.mid_block_add_attention) FMI_TRY(run_attn(v, AttnW{v->e_attn_gn, v->e_aq, v->e_ak, v->e_av, v->e_ao}, B, H, W, s));
  FMI_TRY(run_resnet(v, v->e_mid2, B, H, W, s));
  FMI_TRY(run_gn_trunk(v, v->e_norm_out, v->fx, v->bt1, nullptr, B, H * W, 1, s));
  FMI_TRY(run_conv(v, v->e_conv_out, v->bt1, nullptr, v->bt2, B, H, W, 0, s));
  const bf16_t* mom = v->bt2;
  if (c.use_quant_conv) {
    // conv_out leaves 2L channels per pixel; the 1x1 conv reads cin_pad = 64-channel rows: re-pad through bt1
    const int L2 = 2 * c.latent_channels, cp = v->quant_conv.cin_pad;
    FMI_HIP_TRY(hipMemsetAsync(v->bt1, 0, (size_t)B * H * W * cp * 2, s));
    FMI_HIP_TRY(hipMemcpy2DAsync(v->bt1, (size_t)cp * 2, v->bt2, (size_t)L2 * 2, (size_t)L2 * 2, (size_t)B * H * W, hipMemcpyDeviceToDevice, s));
    FMI_TRY(run_conv(v, v->quant_conv, v->bt1, nullptr, v->bsc, B, H, W, 0, s));
    mom = v->bsc;
  }
  const int64_t n = (int64_t)B * c.latent_channels * H * W;
  hipLaunchKernelGGL(diag_gaussian_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, s, mom, noise, z_out, moments_out, c.latent_channels,
                     H * W, n);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

extern "C" int fmi_vae_decode(fmi_vae* v, const float* z, int B, int h, int w, float* image_out, void* stream) {
  if (v) FMI_TRY(use_device_ordinal(v->device));
  if (!v || !z || !image_out) return fail(FMI_ERR_INVALID, "vae_decode: null argument");
  if (B <= 0 || h <= 0 || w <= 0) return fail(FMI_ERR_INVALID, "vae_decode: empty input");
  FMI_TRY(check_part(v, true));
  hipStream_t s = (hipStream_t)stream;
  FMI_TRY(vae_workspace(v, B, h, w));
  const fmi_vae_config& c = v->cfg;
  int H = h, W = w;
  {
    const int cp = v->conv_in.cin_pad;
    const int64_t n = (int64_t)B * H * W * cp;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, s, z, v->bt1, c.latent_channels, cp,
                       H * W, n);
    FMI_LAUNCH_CHECK();
  }
  FMI_TRY(run_conv_trunk(v, v->conv_in, v->bt1, v->fx, false, B, H, W, 0, s));  // vae.rs:438
  FMI_TRY(run_resnet(v, v->mid1, B, H, W, s));
  if (c.mid_block_add_attention) FMI_TRY(run_attn(v, AttnW{v->attn_gn, v->aq, v->ak, v->av, v->ao}, B, H, W, s));
  FMI_TRY(run_resnet(v, v->mid2, B, H, W, s));
  for (int lvl = 0; lvl < c.n_blocks; ++lvl) {
    for (auto& r : v->up[lvl]) FMI_TRY(run_resnet(v, r, B, H, W, s));
    if (v->upconv[lvl].cout) {  // Upsample::forward: nearest 2x folded into the conv gather (vae.rs:223-229)
      FMI_TRY(trunk_to_bf16(v, v->bt1, (int64_t)B * H * W * v->upconv[lvl].cin_pad, s));
      FMI_TRY(run_conv_trunk(v, v->upconv[lvl], v->bt1, v->fx2, false, B, H, W, 1, s));
      std::swap(v->fx, v->fx2);
      H *= 2, W *= 2;
    }
  }
  FMI_TRY(run_gn_trunk(v, v->norm_out, v->fx, v->bt1, nullptr, B, H * W, 1, s));
  FMI_TRY(run_conv(v, v->conv_out, v->bt1, nullptr, v->bt2, B, H, W, 0, s));
  {
    const int64_t n = (int64_t)B * c.out_channels * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, s, v->bt2, image_out, c.out_channels,
                       H * W, n);
    FMI_LAUNCH_CHECK();
  }
  return FMI_OK;
}

// ---------------------------------------------------------------- op-level entry points
// AttnBlock::forward (vae.rs:95-111) of the decoder's mid block as one op: x (B,H,W,C) bf16 NHWC with C = block_out_channels.last
// -> out of the same shape (GroupNorm, q/k/v, softmax, to_out, + x).  Exists so the block can be checked alone at production size.
extern "C" int fmi_vae_mid_attention(fmi_vae* v, const void* x_bf16_nhwc, int B, int H, int W, void* out_bf16_nhwc, void* stream) {
  if (v) FMI_TRY(use_device_ordinal(v->device));
  if (!v || !x_bf16_nhwc || !out_bf16_nhwc) return fail(FMI_ERR_INVALID, "vae_mid_attention: null argument");
  if (B <= 0 || H <= 0 || W <= 0) return fail(FMI_ERR_INVALID, "vae_mid_attention: empty input");
  if (!v->cfg.mid_block_add_attention) return fail(FMI_ERR_UNSUPPORTED, "vae_mid_attention: the config has no mid-block attention");
  FMI_TRY(check_part(v, true));
  hipStream_t s = (hipStream_t)stream;
  FMI_TRY(vae_workspace(v, B, H, W));
  const int64_t n = (int64_t)B * H * W * v->aq.cout;
  if (n % 8) return fail(FMI_ERR_UNSUPPORTED, "vae_mid_attention: activation size must be a multiple of 8");
  hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(n / 8, 256), 8192)), dim3(256), 0, s, (const bf16_t*)x_bf16_nhwc, v->fx, n / 8);
  FMI_LAUNCH_CHECK();
  FMI_TRY(run_attn(v, AttnW{v->attn_gn, v->aq, v->ak, v->av, v->ao}, B, H, W, s));
  return trunk_to_bf16(v, (bf16_t*)out_bf16_nhwc, n, s);
}

extern "C" int fmi_groupnorm_nhwc(const void* x_bf16, const float* weight, const float* bias, void* out_bf16, int B, int HW, int C, int groups, float eps,
                                  int fuse_silu, void* stream) {
  if (!x_bf16 || !weight || !bias || !out_bf16) return fail(FMI_ERR_INVALID, "groupnorm_nhwc: null pointer");
  hipStream_t s = (hipStream_t)stream;
  // the partial sums and statistics live in the library's per-stream op scratch: stream-ordered, nothing allocated or waited for per call
  const int nchunks = gn_chunks(HW);
  OpScratch scratch(s);  // (holds the cache's lock until the three kernels below are enqueued)
  FMI_TRY(scratch.get(((size_t)B * nchunks * groups + (size_t)B * groups) * sizeof(float2)));
  float2* tmp = static_cast<float2*>(scratch.p);
  return launch_groupnorm_nhwc((const bf16_t*)x_bf16, weight, bias, (bf16_t*)out_bf16, B, HW, C, groups, eps, fuse_silu, tmp,
                               tmp + (size_t)B * nchunks * groups, s);
}

extern "C" int fmi_conv2d_nhwc(const void* x_bf16, const void* w_bf16, const void* bias_bf16, const void* residual_bf16, void* out_bf16, int B, int in_h,
                               int in_w, int Cin, int Cout, int ksize, int upsample2x, void* stream) {
  if (!x_bf16 || !w_bf16 || !out_bf16) return fail(FMI_ERR_INVALID, "conv2d_nhwc: null pointer");
  if (Cin % 64) return fail(FMI_ERR_INVALID, "conv2d_nhwc: Cin must be a multiple of 64 (zero-pad the channels)");
  if (ksize != 1 && ksize != 3) return fail(FMI_ERR_UNSUPPORTED, "conv2d_nhwc: ksize must be 1 or 3");
  static bf16_t* zero = nullptr;
  if (!zero) {
    FMI_HIP_TRY(hipMalloc((void**)&zero, 512));
    FMI_HIP_TRY(hipMemset(zero, 0, 512));
  }
  GemmProblem p = conv_problem((const bf16_t*)x_bf16, (const bf16_t*)w_bf16, (const bf16_t*)bias_bf16, (const bf16_t*)residual_bf16, (bf16_t*)out_bf16, B,
                               in_h, in_w, Cin, Cout, ksize, upsample2x ? 1 : 0, zero);
  return launch_gemm(&p, 1, (hipStream_t)stream);
}
