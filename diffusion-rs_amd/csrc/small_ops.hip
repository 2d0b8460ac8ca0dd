// small_ops.hip — the non-GEMM odds and ends of Flux::forward / FluxPipeline::forward:
//   gemv            : M<=8 row Linear (MlpEmbedder, Modulation1/2, LastLayer.ada_ln) — pure weight
//                     streaming, one wave per output row, 16-B loads  (model.rs:178-183,244-299,695-698)
//   timestep_embedding (model.rs:104-122), Euler update (pipelines/sampling.rs:43),
//   pack/unpack latents (pipelines/flux/sampling.rs:26-48,61-68), u8 post-process (flux/mod.rs:332),
//   Philox N(0,1) latents (seedable replacement of get_noise, flux/sampling.rs:5-14), dtype casts.
#include "common.h"

namespace fmi {

constexpr int GEMV_MAXM = 8;
constexpr int GEMV_ROWS_PER_WAVE = 4;

// y[m][n] (+)= sum_k act(x[m][k]) * W[n][k] + bias[n].  x staged in LDS as f32.
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict x, const bf16_t* __restrict W, const bf16_t* __restrict bias,
                                                   float* __restrict y, int M, int N, int K, int silu_in, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // M*K
  for (int i = threadIdx.x; i < M * K; i += 256) {
    float v = x[i];
    xs[i] = silu_in ? silu(v) : v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbase = (blockIdx.x * 4 + wave) * GEMV_ROWS_PER_WAVE;
#pragma unroll 1
  for (int rr = 0; rr < GEMV_ROWS_PER_WAVE; ++rr) {
    const int n = nbase + rr;
    if (n >= N) break;
    const bf16_t* wr = W + (int64_t)n * K;
    float acc[GEMV_MAXM];
#pragma unroll
    for (int m = 0; m < GEMV_MAXM; ++m) acc[m] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      const uint4 raw = *reinterpret_cast<const uint4*>(wr + k);
      const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
      float wv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[i] = bf16_to_f32(e[i]);
#pragma unroll
      for (int m = 0; m < GEMV_MAXM; ++m) {
        if (m < M) {
          const float4 a = *reinterpret_cast<const float4*>(xs + m * K + k);
          const float4 c = *reinterpret_cast<const float4*>(xs + m * K + k + 4);
          acc[m] += (a.x * wv[0] + a.y * wv[1]) + (a.z * wv[2] + a.w * wv[3]) + (c.x * wv[4] + c.y * wv[5]) + (c.z * wv[6] + c.w * wv[7]);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < GEMV_MAXM; ++m) {
      if (m < M) {
        float s = wave_sum(acc[m]);
        if (lane == 0) {
          if (bias) s += bf16_to_f32(bias[n]);
          float* o = y + (int64_t)m * N + n;
          *o = accumulate ? *o + s : s;
        }
      }
    }
  }
}

int launch_gemv(const float* x, const bf16_t* W, const bf16_t* bias, float* y, int M, int N, int K, int silu_in, int accumulate,
                hipStream_t stream) {
  if (M <= 0 || N <= 0) return FMI_OK;
  if (K % 8) return fail(FMI_ERR_INVALID, "gemv: K must be a multiple of 8");
  if ((size_t)K * sizeof(float) > 64 * 1024) return fail(FMI_ERR_UNSUPPORTED, "gemv: K too large for LDS staging");
  // The x rows of a pass are staged in LDS as f32 (<= 64 KiB) and a wave keeps GEMV_MAXM accumulators: more rows (8 samples at
  // D = 3072 are 96 KiB) run as several passes over W — each output row depends on its own x row only, so the split changes nothing.
  const int rows_per_pass = std::max(1, std::min<int>(GEMV_MAXM, (int)((64 * 1024) / ((size_t)K * sizeof(float)))));
  const int rows_per_block = 4 * GEMV_ROWS_PER_WAVE;
  for (int m0 = 0; m0 < M; m0 += rows_per_pass) {
    const int mm = std::min(rows_per_pass, M - m0);
    hipLaunchKernelGGL(gemv_kernel, dim3(cdiv(N, rows_per_block)), dim3(256), (size_t)mm * K * sizeof(float), stream, x + (size_t)m0 * K, W, bias,
                       y + (size_t)m0 * N, mm, N, K, silu_in, accumulate);
    FMI_LAUNCH_CHECK();
  }
  return FMI_OK;
}

// timestep_embedding (model.rs:104-122): t*1000; freqs = exp(-ln(1e4) * i/half) in f32; [cos, sin]
__global__ void timestep_embedding_kernel(const float* __restrict t, int dim, float* __restrict out) {
  const int b = blockIdx.x, half = dim / 2;
  const float ts = t[b] * 1000.0f;
  const float c = (float)(-9.210340371976184 / (double)half);  // -ln(10000)/half rounded to f32 (affine scalar)
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float fr = expf((float)i * c);
    const float a = ts * fr;
    float sn, cs;
    sincosf(a, &sn, &cs);
    out[(int64_t)b * dim + i] = cs;
    out[(int64_t)b * dim + half + i] = sn;
  }
}
int launch_timestep_embedding(const float* t, int B, int dim, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(128), 0, stream, t, dim, out);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

template <typename F>
__global__ void map_kernel(int64_t n, F f) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) f(i);
}
static inline dim3 map_grid(int64_t n) { return dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 256 * 8)); }

int launch_cast_to_bf16(const void* src, fmi_dtype dt, bf16_t* dst, int64_t n, hipStream_t stream) {
  if (n <= 0) return FMI_OK;
  if (dt == FMI_BF16) {
    FMI_HIP_TRY(hipMemcpyAsync(dst, src, n * 2, hipMemcpyDefault, stream));
  } else if (dt == FMI_F32) {
    const float* s = (const float*)src;
    hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) { dst[i] = f32_to_bf16(s[i]); });
  } else if (dt == FMI_F16) {
    const uint16_t* s = (const uint16_t*)src;
    hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) { dst[i] = f32_to_bf16(f16_to_f32(s[i])); });
  } else {
    return fail(FMI_ERR_INVALID, "cast_to_bf16: unsupported source dtype");
  }
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
// dst bf16 = silu(src f32): the MFMA form of the modulation projection lin(silu(vec)) (model.rs:244-259)
int launch_silu_to_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t stream) {
  if (n <= 0) return FMI_OK;
  hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) { dst[i] = f32_to_bf16(silu(src[i])); });
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
int launch_cast_to_f32(const void* src, fmi_dtype dt, float* dst, int64_t n, hipStream_t stream) {
  if (n <= 0) return FMI_OK;
  if (dt == FMI_F32) {
    FMI_HIP_TRY(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDefault, stream));
  } else if (dt == FMI_BF16) {
    const bf16_t* s = (const bf16_t*)src;
    hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) { dst[i] = bf16_to_f32(s[i]); });
  } else if (dt == FMI_F16) {
    const uint16_t* s = (const uint16_t*)src;
    hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) { dst[i] = f16_to_f32(s[i]); });
  } else {
    return fail(FMI_ERR_INVALID, "cast_to_f32: unsupported source dtype");
  }
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

// img = img + pred * dt   (pipelines/sampling.rs:43; latent kept in f32)
int launch_euler_update(float* img, const float* pred, float dt, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) { img[i] = img[i] + pred[i] * dt; });
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

// y[r][n] = (y[r][n] + g[r % B][n]) + v[r % B][n]: the step-invariant terms of `vec` added in the order the accumulating GEMVs added them
int launch_add2_rows(float* y, const float* g, const float* v, int R, int B, int N, hipStream_t stream) {
  const int64_t n = (int64_t)R * N;
  if (n <= 0) return FMI_OK;
  if (g && v)
    hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) {
      const int64_t o = ((i / N) % B) * N + i % N;
      y[i] = (y[i] + g[o]) + v[o];
    });
  else
    hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, stream, n, [=] __device__(int64_t i) { y[i] = y[i] + v[((i / N) % B) * N + i % N]; });
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

// dst (B, rows, D) <- src (B, rows_src_per_b, D)[:, row_off:row_off+rows, :]  (or the reverse by swapping roles)
int launch_split_rows_f32(const float* src, float* dst, int B, int rows_src_per_b, int row_off, int rows, int D, hipStream_t stream) {
  for (int b = 0; b < B; ++b)
    FMI_HIP_TRY(hipMemcpyAsync(dst + (int64_t)b * rows * D, src + ((int64_t)b * rows_src_per_b + row_off) * D, (size_t)rows * D * 4,
                               hipMemcpyDeviceToDevice, stream));
  return FMI_OK;
}

namespace {
__global__ void splitk_resid_gate_kernel(const float4* __restrict__ parts, int S, const bf16_t* __restrict__ bias, const float* __restrict__ gate,
                                         int rows_per_batch, int gate_bstride, float* __restrict__ out, int ldo, int M, int N4) {
  const int64_t total = (int64_t)M * N4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / N4), n = (int)(i % N4) * 4;
    float4 v = parts[i];
    for (int s = 1; s < S; ++s) {  // fixed order: the result does not depend on the launch
      const float4 w = parts[(int64_t)s * total + i];
      v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
    }
    if (bias) v.x += bf16_to_f32(bias[n]), v.y += bf16_to_f32(bias[n + 1]), v.z += bf16_to_f32(bias[n + 2]), v.w += bf16_to_f32(bias[n + 3]);
    const float* g = gate + (rows_per_batch > 0 ? (int64_t)(m / rows_per_batch) * gate_bstride : 0) + n;
    float4* o = reinterpret_cast<float4*>(out + (int64_t)m * ldo + n);
    float4 x = *o;
    x.x += g[0] * v.x, x.y += g[1] * v.y, x.z += g[2] * v.z, x.w += g[3] * v.w;
    *o = x;
  }
}
}  // namespace
int launch_splitk_resid_gate(const float* parts, int S, const bf16_t* bias, const float* gate, int rows_per_batch, int gate_bstride, float* out, int ldo,
                             int M, int N, hipStream_t stream) {
  if (N % 4 || ldo % 4) return fail(FMI_ERR_INVALID, "splitk reduce: N and ldo must be multiples of 4");
  const int64_t total = (int64_t)M * (N / 4);
  splitk_resid_gate_kernel<<<(int)std::min<int64_t>((total + 255) / 256, 2048), 256, 0, stream>>>(reinterpret_cast<const float4*>(parts), S, bias, gate, rows_per_batch,
                                                                                                 gate_bstride, out, ldo, M, N / 4);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

}  // namespace fmi

using namespace fmi;

// ------------------------------------------------------------------ C-ABI: pipeline glue
extern "C" int fmi_pack_latents(const float* latent, int B, int C, int h, int w, float* img_out, float* img_ids_out, void* stream) {
  if (h % 2 || w % 2) return fail(FMI_ERR_INVALID, "pack_latents: h and w must be even");
  const int h2 = h / 2, w2 = w / 2, C4 = C * 4;
  const int64_t n = (int64_t)B * h2 * w2 * C4;
  hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, (hipStream_t)stream, n, [=] __device__(int64_t i) {
    const int e = (int)(i % C4);
    const int64_t tok = i / C4;
    const int j = (int)(tok % w2), ii = (int)((tok / w2) % h2), b = (int)(tok / ((int64_t)w2 * h2));
    const int c = e >> 2, ph = (e >> 1) & 1, pw = e & 1;
    img_out[i] = latent[(((int64_t)b * C + c) * h + (2 * ii + ph)) * w + (2 * j + pw)];
    if (img_ids_out && e < 3) img_ids_out[tok * 3 + e] = e == 0 ? 0.f : (e == 1 ? (float)ii : (float)j);
  });
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

extern "C" int fmi_unpack_latents(const float* img, int B, int C, int h, int w, double scale_factor, double shift_factor, float* z_out,
                                  void* stream) {
  if (h % 2 || w % 2) return fail(FMI_ERR_INVALID, "unpack_latents: h and w must be even");
  const int h2 = h / 2, w2 = w / 2, C4 = C * 4;
  const int64_t n = (int64_t)B * C * h * w;
  // (img / scale_factor) + shift_factor : two affine ops with the scalars rounded to f32 (flux/mod.rs:329)
  const float inv = (float)(1.0 / scale_factor), sh = (float)shift_factor;
  hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, (hipStream_t)stream, n, [=] __device__(int64_t i) {
    const int x = (int)(i % w), y = (int)((i / w) % h), c = (int)((i / ((int64_t)w * h)) % C), b = (int)(i / ((int64_t)w * h * C));
    const int64_t tok = ((int64_t)b * h2 + (y >> 1)) * w2 + (x >> 1);
    const float v = img[tok * C4 + (c * 2 + (y & 1)) * 2 + (x & 1)];
    z_out[i] = __fadd_rn(__fmul_rn(v, inv), sh);  // two roundings, like the two affine ops
  });
  (void)w2;
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

extern "C" int fmi_postprocess_u8(const float* image, int B, int C, int H, int W, int interleave, uint8_t* out, void* stream) {
  const int64_t n = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, (hipStream_t)stream, n, [=] __device__(int64_t i) {
    float v = image[i];
    v = fminf(fmaxf(v, -1.f), 1.f);
    v = (v + 1.0f) * 127.5f;
    // Rust `as u8`: truncate toward zero, saturate, NaN -> 0 (cpu_backend/mod.rs:2571-2574)
    uint8_t u = !(v == v) ? 0 : (v <= 0.f ? 0 : (v >= 255.f ? 255 : (uint8_t)v));
    if (interleave) {
      const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
      out[(((int64_t)b * H + y) * W + x) * C + c] = u;
    } else {
      out[i] = u;
    }
  });
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

// Philox4x32-10 + Box-Muller.  counter = (i/4, sample, 0, 0), key = seed; 4 normals per counter.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
  c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
  k[0] += 0x9E3779B9u;
  k[1] += 0xBB67AE85u;
}
// counter = (quad lo, quad hi, sample lo, sample hi), key = (seed lo, seed hi): 4 words per counter
__device__ __forceinline__ void philox_quad(int64_t qd, uint64_t sample, uint64_t seed, uint32_t (&c)[4]) {
  c[0] = (uint32_t)qd, c[1] = (uint32_t)((uint64_t)qd >> 32), c[2] = (uint32_t)sample, c[3] = (uint32_t)(sample >> 32);
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
}
// the raw stream fmi_randn draws from (checked bit for bit against the oracle / Random123 known answers)
extern "C" int fmi_philox_u32(uint32_t* out, int64_t n_per_sample, int B, uint64_t seed, uint64_t first_sample, void* stream) {
  if (!out || n_per_sample < 0 || B < 0) return fail(FMI_ERR_INVALID, "philox_u32: bad arguments");
  const int64_t quads = (n_per_sample + 3) / 4;
  const int64_t n = quads * B;
  if (n == 0) return FMI_OK;
  auto body = [=] __device__(int64_t i) {
    const int64_t qd = i % quads;
    const int b = (int)(i / quads);
    uint32_t c[4];
    philox_quad(qd, first_sample + (uint64_t)b, seed, c);
    for (int e = 0; e < 4; ++e) {
      const int64_t idx = qd * 4 + e;
      if (idx < n_per_sample) out[(int64_t)b * n_per_sample + idx] = c[e];
    }
  };
  hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, (hipStream_t)stream, n, body);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
extern "C" int fmi_randn(float* out, int64_t n_per_sample, int B, uint64_t seed, uint64_t first_sample, void* stream) {
  if (!out || n_per_sample < 0 || B < 0) return fail(FMI_ERR_INVALID, "randn: bad arguments");
  const int64_t quads = (n_per_sample + 3) / 4;
  const int64_t n = quads * B;
  if (n == 0) return FMI_OK;
  auto body = [=] __device__(int64_t i) {
    const int64_t qd = i % quads;
    const int b = (int)(i / quads);
    uint32_t c[4];
    philox_quad(qd, first_sample + (uint64_t)b, seed, c);
    float z[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float u1 = ((float)(c[2 * p] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(c[2 * p + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float rad = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * p] = rad * cs;
      z[2 * p + 1] = rad * sn;
    }
    for (int e = 0; e < 4; ++e) {
      const int64_t idx = qd * 4 + e;
      if (idx < n_per_sample) out[(int64_t)b * n_per_sample + idx] = z[e];
    }
  };
  hipLaunchKernelGGL(map_kernel, map_grid(n), dim3(256), 0, (hipStream_t)stream, n, body);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
