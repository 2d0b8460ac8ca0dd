// bnb_dequant.hip — bitsandbytes blockwise dequantisation, the reference's only native library,
// rebuilt for gfx950 with the SAME extern "C" entry points (CUstream -> hipStream_t):
//   diffusion_rs_backend/kernels/bitsandbytes/dequant.cu:94-232, FFI decls bitsandbytes/ffi.rs:5-114.
//
// The reference kernel is a cub BlockLoad/BlockStore WARP_TRANSPOSE tile of 64 threads x 8 bytes.
// Here each lane reads 16 packed bytes (one dwordx4) and writes 32 outputs with 16-byte stores:
// pure streaming, coalesced 1 KiB per wave-instruction on the read side, HBM-bound
// (algorithmic bytes per output element: 0.5 B in + sizeof(T) out + 4/blocksize B absmax).
// Semantics are bit-exact with dequant.cu: high nibble first, value = LUT[nibble] * absmax
// (nf4) / tree(nibble) * absmax * sign (fp4) / code[byte] * absmax (int8) in f32, one RNE
// rounding to the output type.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace fmi {

__device__ __constant__ float kNF4d[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                           -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                           0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                                           0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

__device__ __forceinline__ float dq_fp4_tree(unsigned v, float am) {
  // dequant.cu:12-37: <abs value> * absmax * sign, in this order
  const float tab[8] = {0.0f, 5.208333333e-03f, 0.66666667f, 1.0f, 0.33333333f, 0.5f, 0.16666667f, 0.25f};
  const float sign = (v & 8) ? -1.0f : 1.0f;
  return tab[v & 7] * am * sign;
}

template <typename T>
__device__ __forceinline__ T cvt_out(float v);
template <>
__device__ __forceinline__ float cvt_out<float>(float v) {
  return v;
}
struct half_bits {
  uint16_t v;
};
struct bf16_bits {
  uint16_t v;
};
template <>
__device__ __forceinline__ half_bits cvt_out<half_bits>(float v) {
  // The reference rounds the f32 product to f32 and THEN to f16 (two roundings, dequant.cu:136-151).
  // Without this barrier the backend folds fmul+fptrunc into v_fma_mixlo_f16 (one rounding) and
  // 1 value in ~20k differs by an f16 ulp.
  asm volatile("" : "+v"(v));
  return half_bits{f32_to_f16(v)};
}
template <>
__device__ __forceinline__ bf16_bits cvt_out<bf16_bits>(float v) {
  return bf16_bits{f32_to_bf16(v)};
}

// QT: 1 = fp4, 2 = nf4.  One thread = 16 packed bytes = 32 outputs.
template <typename T, int QT>
__global__ __launch_bounds__(256) void dequant4_kernel(const uint8_t* __restrict A, const float* __restrict absmax, T* __restrict out,
                                                       int half_block, int n) {
  const int64_t nbytes = ((int64_t)n + 1) / 2;
  const int64_t b0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (b0 >= nbytes) return;
  uint8_t q[16];
  if (b0 + 16 <= nbytes && (reinterpret_cast<uintptr_t>(A + b0) & 15) == 0) {
    const uint4 raw = *reinterpret_cast<const uint4*>(A + b0);
    __builtin_memcpy(q, &raw, 16);
  } else {
    for (int i = 0; i < 16; ++i) q[i] = b0 + i < nbytes ? A[b0 + i] : 0;
  }
  T vals[32];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    // absmax index = byte / (blocksize/2)  (dequant.cu:125 with the launcher's blocksize/2, :167)
    const float am = absmax[(b0 + i) / half_block];
    float hi, lo;
    if (QT == 2) {
      hi = kNF4d[q[i] >> 4] * am;
      lo = kNF4d[q[i] & 15] * am;
    } else {
      hi = dq_fp4_tree(q[i] >> 4, am);
      lo = dq_fp4_tree(q[i] & 15, am);
    }
    vals[2 * i] = cvt_out<T>(hi);
    vals[2 * i + 1] = cvt_out<T>(lo);
  }
  const int64_t o0 = b0 * 2;
  if (o0 + 32 <= n && (reinterpret_cast<uintptr_t>(out + o0) & 15) == 0) {
    constexpr int NV = 32 * sizeof(T) / 16;
    uint4* dst = reinterpret_cast<uint4*>(out + o0);
    const uint4* src = reinterpret_cast<const uint4*>(vals);
#pragma unroll
    for (int i = 0; i < NV; ++i) dst[i] = src[i];
  } else {
    for (int i = 0; i < 32 && o0 + i < n; ++i) out[o0 + i] = vals[i];
  }
}

// Streaming form of the 4-bit expansion for the denoise loop (flux_model.hip: densify() expands a matrix right before its GEMM,
// 152 times per step): bf16 out, power-of-two blocksize, n % 8 == 0.  A lane takes ONE packed dword (8 weights) per iteration and
// stores 16 bytes, so a wave-instruction reads 256 contiguous bytes and writes 1 KiB contiguous (the general kernel above writes
// 16-byte pieces at a 64-byte lane stride and divides by the blocksize per element).  Same arithmetic, so the same bits:
// value = LUT[nibble] * absmax (nf4) / tree(nibble) * absmax * sign (fp4), one RNE rounding — the table holds both nibbles of a byte.
template <int QT>
__global__ __launch_bounds__(256) void dequant4_stream_bf16_kernel(const uint32_t* __restrict A, const float* __restrict absmax, uint4* __restrict out,
                                                                   int dwords_per_block_shift, int64_t ndwords) {
  __shared__ float2 lut[256];  // byte -> (value of the high nibble, value of the low nibble) for absmax = 1
  {
    const unsigned b = threadIdx.x;
    lut[b] = QT == 2 ? make_float2(kNF4d[b >> 4], kNF4d[b & 15]) : make_float2(dq_fp4_tree(b >> 4, 1.0f), dq_fp4_tree(b & 15, 1.0f));
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ndwords; g += stride) {
    const uint32_t w = A[g];
    const float am = absmax[g >> dwords_per_block_shift];
    uint32_t o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float2 v = lut[(w >> (8 * b)) & 0xffu];
      float hi, lo;
      if (QT == 2) hi = v.x * am, lo = v.y * am;
      else hi = dq_fp4_tree(((w >> (8 * b)) >> 4) & 15u, am), lo = dq_fp4_tree((w >> (8 * b)) & 15u, am);  // (abs * absmax) * sign: the order matters for the bits
      o[b] = (uint32_t)f32_to_bf16(hi) | ((uint32_t)f32_to_bf16(lo) << 16);
    }
    out[g] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// General8bit: out[i] = code[A[i]] * absmax[i / blocksize]  (dequant.cu:132-137). 16 outputs per thread.
template <typename T>
__global__ __launch_bounds__(256) void dequant8_kernel(const float* __restrict code, const uint8_t* __restrict A,
                                                       const float* __restrict absmax, T* __restrict out, int blocksize, int n) {
  __shared__ float lut[256];
  lut[threadIdx.x] = code[threadIdx.x];
  __syncthreads();
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i0 >= n) return;
  for (int i = 0; i < 16 && i0 + i < n; ++i) out[i0 + i] = cvt_out<T>(lut[A[i0 + i]] * absmax[(i0 + i) / blocksize]);
}

// LLM.int8: out[idx] = (float(w[idx]) * SCB[idx / col]) / 127  (dequant.cu:205-214)
template <typename T>
__global__ __launch_bounds__(256) void dequant_int8_scb_kernel(const int8_t* __restrict w, const float* __restrict scb, T* __restrict out,
                                                               int col, int n) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  for (int i = 0; i < 8 && i0 + i < n; ++i) out[i0 + i] = cvt_out<T>(((float)w[i0 + i] * scb[(i0 + i) / col]) / 127.f);
}

// Streaming form of the LLM.int8 expansion for the denoise loop (densify() expands a matrix right before its GEMM when the launch
// has more than 256 rows): bf16 out, col % 8 == 0.  A lane takes 8 weights (one 8-byte load) and stores 16 bytes — 1 KiB contiguous
// per wave-instruction — with the row's SCB read once and x / 127 as div127() (gemm_bf16.hip: correctly rounded, so the same bits as
// the general kernel's IEEE division; that kernel divides twice per element and reaches ~1 TB/s).
__device__ __forceinline__ float dq_div127(float x) {
  const float r = 1.0f / 127.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-127.0f, q, x), r, q);
}
__global__ __launch_bounds__(256) void dequant_int8_stream_bf16_kernel(const uint2* __restrict w, const float* __restrict scb, uint4* __restrict out,
                                                                       uint32_t k8_per_row, uint32_t n8) {
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < n8; g += stride) {
    const uint2 p = w[g];
    const float s = scb[g / k8_per_row];
    const uint32_t d[2] = {p.x, p.y};
    uint32_t o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t dw = d[b >> 1];
      const int i0 = (int)(dw << (24 - 16 * (b & 1))) >> 24, i1 = (int)(dw << (16 - 16 * (b & 1))) >> 24;
      o[b] = (uint32_t)f32_to_bf16(dq_div127((float)i0 * s)) | ((uint32_t)f32_to_bf16(dq_div127((float)i1 * s)) << 16);
    }
    out[g] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

template <typename T, int QT>
void launch_dq4(const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) {
  if (n <= 0) return;
  if (std::is_same<T, bf16_bits>::value && n >= (1 << 20) && n % 8 == 0 && blocksize >= 8 && (blocksize & (blocksize - 1)) == 0 &&
      (reinterpret_cast<uintptr_t>(A) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int64_t ndw = n / 8;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(ndw, 256 * 4), 8192);
    hipLaunchKernelGGL((dequant4_stream_bf16_kernel<QT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(A), absmax,
                       reinterpret_cast<uint4*>(out), __builtin_ctz(blocksize / 8), ndw);
    return;
  }
  const int64_t nbytes = ((int64_t)n + 1) / 2;
  const unsigned grid = (unsigned)cdiv64(nbytes, 256 * 16);
  hipLaunchKernelGGL((dequant4_kernel<T, QT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, A, absmax, (T*)out, blocksize / 2, n);
}
template <typename T>
void launch_dq8(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL((dequant8_kernel<T>), dim3((unsigned)cdiv64(n, 256 * 16)), dim3(256), 0, (hipStream_t)stream, code, A, absmax, (T*)out,
                     blocksize, n);
}
int launch_dequant_int8_scb_bf16(const int8_t* w, const float* scb, bf16_t* out, int col, int64_t n, hipStream_t stream);
template <typename T>
void launch_scb(const int8_t* w, const float* scb, void* out, int col, int n) {
  if (n <= 0) return;
  if (std::is_same<T, bf16_bits>::value && n >= (1 << 20)) {  // large bf16 expansions: the streaming kernel when eligible (same bits)
    launch_dequant_int8_scb_bf16(w, scb, (bf16_t*)out, col, n, nullptr);
    return;
  }
  // legacy default stream, like the reference (dequant.cu:221)
  hipLaunchKernelGGL((dequant_int8_scb_kernel<T>), dim3((unsigned)cdiv64(n, 256 * 8)), dim3(256), 0, (hipStream_t) nullptr, w, scb, (T*)out, col, n);
}

// stream-ordered form for the model's own use (LLM.int8 weights are expanded right before their GEMM)
int launch_dequant_int8_scb_bf16(const int8_t* w, const float* scb, bf16_t* out, int col, int64_t n, hipStream_t stream) {
  if (n <= 0) return FMI_OK;
  if (n >= (1ll << 31)) return fail(FMI_ERR_UNSUPPORTED, "dequant_int8_scb: more than 2^31 elements");
  if (n >= (1 << 20) && col % 8 == 0 && (reinterpret_cast<uintptr_t>(w) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const uint32_t n8 = (uint32_t)(n / 8);
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n8, 256 * 4), 8192);
    hipLaunchKernelGGL(dequant_int8_stream_bf16_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint2*>(w), scb, reinterpret_cast<uint4*>(out),
                       (uint32_t)(col / 8), n8);
    FMI_LAUNCH_CHECK();
    return FMI_OK;
  }
  hipLaunchKernelGGL((dequant_int8_scb_kernel<bf16_bits>), dim3((unsigned)cdiv64(n, 256 * 8)), dim3(256), 0, stream, w, scb, (bf16_bits*)out, col, (int)n);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

}  // namespace fmi

using namespace fmi;

extern "C" {
void dequantize_blockwise_f32_int8(const float* code, const uint8_t* A, const float* absmax, float* out, int blocksize, int n, void* stream) { launch_dq8<float>(code, A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_f32_fp4(const float*, const uint8_t* A, const float* absmax, float* out, int blocksize, int n, void* stream) { launch_dq4<float, 1>(A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_f32_nf4(const float*, const uint8_t* A, const float* absmax, float* out, int blocksize, int n, void* stream) { launch_dq4<float, 2>(A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_f16_int8(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) { launch_dq8<half_bits>(code, A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_f16_fp4(const float*, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) { launch_dq4<half_bits, 1>(A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_f16_nf4(const float*, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) { launch_dq4<half_bits, 2>(A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_bf16_int8(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) { launch_dq8<bf16_bits>(code, A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_bf16_fp4(const float*, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) { launch_dq4<bf16_bits, 1>(A, absmax, out, blocksize, n, stream); }
void dequantize_blockwise_bf16_nf4(const float*, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream) { launch_dq4<bf16_bits, 2>(A, absmax, out, blocksize, n, stream); }
void dequantize_8bit_kernel_f32(const int8_t* weight, const float* scb, float* out, int, int col, int n) { launch_scb<float>(weight, scb, out, col, n); }
void dequantize_8bit_kernel_f16(const int8_t* weight, const float* scb, void* out, int, int col, int n) { launch_scb<half_bits>(weight, scb, out, col, n); }
void dequantize_8bit_kernel_bf16(const int8_t* weight, const float* scb, void* out, int, int col, int n) { launch_scb<bf16_bits>(weight, scb, out, col, n); }
}
