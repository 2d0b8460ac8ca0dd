// norm_rope.hip — HBM-bound row kernels of the DiT block:
//   layernorm_mod   : LayerNorm(eps 1e-6, no affine) fused with ModulationOut::scale_shift
//                     (model.rs:33-38, 218-221; nn/ops.rs:1020-1041 statistics formula)
//   qk_norm_rope    : QkNorm (RMSNorm slow path, nn/layer_norm.rs:136-153, eps 1e-6) fused with
//                     apply_rope (model.rs:86-95) and the (B,L,H,d)->(B,H,L,d) relayout of
//                     SelfAttention::qkv (model.rs:414-425)
//   rope_table      : rope()/EmbedNd (model.rs:65-84,142-157) computed once per image, in f32
// Each replaces ~10 eager elementwise kernels of the reference with one pass over the data,
// 16-byte vector loads/stores, f32 statistics.
#include "common.h"

namespace fmi {

// one 256-thread block per row; x f32, out bf16.  REG = true (D <= 4096): the row stays in registers between the
// statistics and the normalisation (same per-thread element order as the two-pass form: identical results).
// A second row set (x2 .. out2; rows1 = rows of the first) rides in the same launch: the image and text streams of a double block are
// normalised together (the text stream's 512 rows were a 4.5 us launch of their own, 38 times per step).
template <bool REG>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const float* __restrict x1, const float* __restrict scale1,
                                                            const float* __restrict shift1, int mod_bstride, int rows_per_batch1,
                                                            bf16_t* __restrict out1, int D, float eps, int rows1, const float* __restrict x2,
                                                            const float* __restrict scale2, const float* __restrict shift2, int rows_per_batch2,
                                                            bf16_t* __restrict out2) {
  __shared__ float red[2][4];
  const bool second = (int)blockIdx.x >= rows1;
  const int row = second ? blockIdx.x - rows1 : blockIdx.x;
  const float* x = second ? x2 : x1;
  const float* scale = second ? scale2 : scale1;
  const float* shift = second ? shift2 : shift1;
  const int rows_per_batch = second ? rows_per_batch2 : rows_per_batch1;
  bf16_t* out = second ? out2 : out1;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
  const int nv = D >> 2;
  constexpr int MAXV = 4;
  float4 keep[MAXV], ksc[MAXV], ksh[MAXV];
  float s = 0.f, s2 = 0.f;
  const int batch = rows_per_batch > 0 ? row / rows_per_batch : 0;
  const float4* sc = scale ? reinterpret_cast<const float4*>(scale + (int64_t)batch * mod_bstride) : nullptr;
  const float4* sh = shift ? reinterpret_cast<const float4*>(shift + (int64_t)batch * mod_bstride) : nullptr;
  if constexpr (REG) {
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
      const int i = threadIdx.x + c * 256;
      if (i < nv) {
        const float4 v = xr[i];
        keep[c] = v;
        // the modulation vectors are requested with the row, not after the reduction (their latency hides behind it)
        if (sc) ksc[c] = sc[i];
        if (sh) ksh[c] = sh[i];
        s += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    }
  } else {
    for (int i = threadIdx.x; i < nv; i += 256) {
      const float4 v = xr[i];
      s += (v.x + v.y) + (v.z + v.w);
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  }
  s = wave_sum(s);
  s2 = wave_sum(s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = s;
    red[1][w] = s2;
  }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float mean = s / (float)D;
  const float var = s2 / (float)D - mean * mean;  // E[x^2]-mean^2, as nn/ops.rs:1029-1031
  const float inv_std = 1.0f / sqrtf(var + eps);
  uint2* o = reinterpret_cast<uint2*>(out + (int64_t)row * D);
  auto emit = [&](int i, const float4 v, const float4 ksc_i, const float4 ksh_i) {
    float a = (v.x - mean) * inv_std, b = (v.y - mean) * inv_std, c = (v.z - mean) * inv_std, d = (v.w - mean) * inv_std;
    if (sc) {
      const float4 k = ksc_i;
      a *= (k.x + 1.0f);
      b *= (k.y + 1.0f);
      c *= (k.z + 1.0f);
      d *= (k.w + 1.0f);
    }
    if (sh) {
      const float4 k = ksh_i;
      a += k.x;
      b += k.y;
      c += k.z;
      d += k.w;
    }
    o[i] = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
  };
  if constexpr (REG) {
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
      const int i = threadIdx.x + c * 256;
      if (i < nv) emit(i, keep[c], ksc[c], ksh[c]);
    }
  } else {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < nv; i += 256) emit(i, xr[i], sc ? sc[i] : z, sh ? sh[i] : z);
  }
}

int launch_layernorm_mod2(const float* x, const float* scale, const float* shift, int mod_bstride, int rows_per_batch, bf16_t* out, int rows,
                          const float* x2, const float* scale2, const float* shift2, int rows_per_batch2, bf16_t* out2, int rows2, int D, float eps,
                          hipStream_t stream) {
  if (rows + rows2 <= 0) return FMI_OK;
  if (D % 4) return fail(FMI_ERR_INVALID, "layernorm_mod: D must be a multiple of 4");
  if (D <= 4096)
    hipLaunchKernelGGL(layernorm_mod_kernel<true>, dim3(rows + rows2), dim3(256), 0, stream, x, scale, shift, mod_bstride, rows_per_batch, out, D, eps, rows,
                       x2, scale2, shift2, rows_per_batch2, out2);
  else
    hipLaunchKernelGGL(layernorm_mod_kernel<false>, dim3(rows + rows2), dim3(256), 0, stream, x, scale, shift, mod_bstride, rows_per_batch, out, D, eps, rows,
                       x2, scale2, shift2, rows_per_batch2, out2);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
int launch_layernorm_mod(const float* x, const float* scale, const float* shift, int mod_bstride, int rows_per_batch, bf16_t* out,
                         int rows, int D, float eps, hipStream_t stream) {
  return launch_layernorm_mod2(x, scale, shift, mod_bstride, rows_per_batch, out, rows, nullptr, nullptr, nullptr, 0, nullptr, 0, D, eps, stream);
}

// 16 lanes per (row, head) vector of 128; each lane owns 8 elements = 4 rope pairs.
// grid.x covers B*rows*H/16 groups of 16 head-rows; q and k handled by the same lane.
// F32OUT (the op-level seam fmi_rmsnorm_rope with an f32 result): the same arithmetic, the final rounding to bf16 left out — what the
// tight (1e-6) comparison with the oracle's rms_norm + apply_rope reads; qo / ko then point to float.
template <bool F32OUT>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const bf16_t* __restrict q, const bf16_t* __restrict k, int ld,
                                                           int64_t in_bstride, const bf16_t* __restrict wq,
                                                           const bf16_t* __restrict wk, const float* __restrict pe,
                                                           int64_t pe_bstride, void* __restrict qo, void* __restrict ko, int B,
                                                           int H, int rows, int row_off, int Ltot) {
  const int64_t gid = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // head-row index
  const int sub = threadIdx.x & 15;
  const int64_t total = (int64_t)B * rows * H;
  if (gid >= total) return;
  const int h = (int)(gid % H);
  const int64_t br = gid / H;
  const int r = (int)(br % rows), b = (int)(br / rows);
  const int pos = row_off + r;
  // rope factors for pairs 4*sub .. 4*sub+3 : {cos, sin} f32
  const float4* pp = reinterpret_cast<const float4*>(pe + (int64_t)b * pe_bstride + ((int64_t)pos * 64 + 4 * sub) * 2);
  const float4 c01 = pp[0], c23 = pp[1];
  const float cs[4] = {c01.x, c01.z, c23.x, c23.z};
  const float sn[4] = {c01.y, c01.w, c23.y, c23.w};
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const bf16_t* src = (which ? k : q) + (int64_t)b * in_bstride + (int64_t)r * ld + h * 128 + sub * 8;
    const bf16_t* wv = (which ? wk : wq) + sub * 8;
    const int64_t doff = (((int64_t)b * H + h) * Ltot + pos) * 128 + sub * 8;
    const uint4 raw = *reinterpret_cast<const uint4*>(src);
    const uint4 wraw = *reinterpret_cast<const uint4*>(wv);
    const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
    const bf16_t* we = reinterpret_cast<const bf16_t*>(&wraw);
    float v[8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = bf16_to_f32(e[i]);
      ss += v[i] * v[i];
    }
    ss = row16_sum(ss);
    const float inv = rms_inv128(ss);
    uint32_t o[4];
    float of[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float x0 = v[2 * p] * inv * bf16_to_f32(we[2 * p]);
      const float x1 = v[2 * p + 1] * inv * bf16_to_f32(we[2 * p + 1]);
      of[2 * p] = cs[p] * x0 - sn[p] * x1, of[2 * p + 1] = sn[p] * x0 + cs[p] * x1;
      o[p] = pack_bf16x2(of[2 * p], of[2 * p + 1]);
    }
    if (F32OUT) {
      float4* d4 = reinterpret_cast<float4*>(static_cast<float*>(which ? ko : qo) + doff);
      d4[0] = make_float4(of[0], of[1], of[2], of[3]);
      d4[1] = make_float4(of[4], of[5], of[6], of[7]);
    } else {
      *reinterpret_cast<uint4*>(static_cast<bf16_t*>(which ? ko : qo) + doff) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

int launch_qk_norm_rope(const bf16_t* q, const bf16_t* k, int ld, int64_t in_bstride, const bf16_t* wq, const bf16_t* wk,
                        const float* pe, int64_t pe_bstride, bf16_t* qo, bf16_t* ko, int B, int H, int rows, int row_off, int Ltot,
                        hipStream_t stream) {
  if (rows <= 0) return FMI_OK;
  if (ld % 8) return fail(FMI_ERR_INVALID, "qk_norm_rope: ld must be a multiple of 8");
  const int64_t total = (int64_t)B * rows * H;
  hipLaunchKernelGGL(qk_norm_rope_kernel<false>, dim3((unsigned)cdiv64(total, 16)), dim3(256), 0, stream, q, k, ld, in_bstride, wq, wk, pe,
                     pe_bstride, (void*)qo, (void*)ko, B, H, rows, row_off, Ltot);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}
int launch_qk_norm_rope_f32(const bf16_t* q, const bf16_t* k, int ld, int64_t in_bstride, const bf16_t* wq, const bf16_t* wk,
                            const float* pe, int64_t pe_bstride, float* qo, float* ko, int B, int H, int rows, int row_off, int Ltot,
                            hipStream_t stream) {
  if (rows <= 0) return FMI_OK;
  if (ld % 8) return fail(FMI_ERR_INVALID, "qk_norm_rope: ld must be a multiple of 8");
  const int64_t total = (int64_t)B * rows * H;
  hipLaunchKernelGGL(qk_norm_rope_kernel<true>, dim3((unsigned)cdiv64(total, 16)), dim3(256), 0, stream, q, k, ld, in_bstride, wq, wk, pe,
                     pe_bstride, (void*)qo, (void*)ko, B, H, rows, row_off, Ltot);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

// pe[b][l][i] = {cos(pos*inv_freq_i), sin(...)} for l in [0,T) from txt_ids and [T,T+S) from img_ids.
// inv_freq exactly as model.rs:71-74: 1f32 / (theta^(2j/dim) computed in f64) as f32.
__global__ void rope_table_kernel(const float* __restrict txt_ids, const float* __restrict img_ids, int T, int S, int a0, int a1, int a2,
                                  int theta, float* __restrict pe) {
  const int L = T + S;
  const int half = (a0 + a1 + a2) / 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= (int64_t)L * half) return;
  const int l = (int)(idx / half), i = (int)(idx % half);
  int axis, j, dim;
  if (i < a0 / 2) {
    axis = 0, j = i, dim = a0;
  } else if (i < (a0 + a1) / 2) {
    axis = 1, j = i - a0 / 2, dim = a1;
  } else {
    axis = 2, j = i - (a0 + a1) / 2, dim = a2;
  }
  const float* ids = l < T ? txt_ids + ((int64_t)b * T + l) * 3 : img_ids + ((int64_t)b * S + (l - T)) * 3;
  const float pos = ids[axis];
  const float inv_freq = 1.0f / (float)pow((double)theta, (double)(2 * j) / (double)dim);
  const float f = pos * inv_freq;
  float sn, cs;
  sincosf(f, &sn, &cs);
  float2* o = reinterpret_cast<float2*>(pe + ((int64_t)b * L * half + idx) * 2);
  *o = make_float2(cs, sn);
}

int launch_rope_table(const float* txt_ids, const float* img_ids, int B, int T, int S, const int* axes, int theta, float* pe,
                      hipStream_t stream) {
  const int half = (axes[0] + axes[1] + axes[2]) / 2;
  const int64_t n = (int64_t)(T + S) * half;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)cdiv64(n, 256), B), dim3(256), 0, stream, txt_ids, img_ids, T, S, axes[0], axes[1],
                     axes[2], theta, pe);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

}  // namespace fmi
