// common.h — shared device/host helpers for libflux_mi355x (gfx950 / CDNA4 only).
#pragma once
// FMI_ALT_KERNELS = 1 (the TEST build, libflux_mi355x_alt.so: `make alt`): the superseded kernels are compiled in as well — the 8-wave single-barrier
// attention, attention_w4 outside the key-split launches, attention_w16 / attention_w32 (bf16 and fp8-QK), the dense 4-wave GEMM and the double-buffered
// 256 x 256 GEMM — and fmi_set_attention_kernel / FMI_GEMM_W4 select them: the bit-identity cross-checks of tests/ run against that build.  The product
// library (0, the default) carries what the product runs: kernels 5 and 1 of fmi_set_attention_kernel, gemm_pp_kernel + the 128-wide / conv / 4-bit kernels.
#ifndef FMI_ALT_KERNELS
#define FMI_ALT_KERNELS 0
#endif
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/flux_mi355x.h"

namespace fmi {

// ---------------------------------------------------------------- error channel
void set_error(const std::string& msg);
int fail(fmi_status st, const std::string& msg);
#define FMI_HIP_TRY(expr)                                                                       \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return ::fmi::fail(FMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + \
                                          __FILE__ + ":" + std::to_string(__LINE__));           \
  } while (0)
#define FMI_TRY(expr)        \
  do {                       \
    int _rc = (expr);        \
    if (_rc != FMI_OK) return _rc; \
  } while (0)
#define FMI_LAUNCH_CHECK() FMI_HIP_TRY(hipGetLastError())
// Kernels whose hand-scheduled instruction streams address LDS from byte 0 (the generated attention streams' ring-slot xor, the fused
// 4-bit GEMM's table look-ups) need their one __shared__ array to be the workgroup's WHOLE static LDS allocation.  That is a property
// of the code object, so it is checked on the host, once per kernel instantiation, before its first launch: the kernel's static LDS size
// must equal the array's size (any further __shared__ object would make it larger).  A violation is a refused launch with
// FMI_ERR_STATE — not a device-side trap that takes the process down.
inline int lds_sole_owner(const void* kernel, size_t expect_bytes, const char* what) {
  hipFuncAttributes a{};
  hipError_t e = hipFuncGetAttributes(&a, kernel);
  if (e != hipSuccess) return fail(FMI_ERR_HIP, std::string("hipFuncGetAttributes(") + what + "): " + hipGetErrorString(e));
  if ((size_t)a.sharedSizeBytes != expect_bytes)
    return fail(FMI_ERR_STATE, std::string(what) + ": static LDS is " + std::to_string((size_t)a.sharedSizeBytes) + " bytes, the kernel's ring is " +
                                   std::to_string(expect_bytes) + " — another __shared__ object would move it off LDS byte 0; launch refused");
  return FMI_OK;
}
#define FMI_LDS_GUARD(kernel, bytes)                                                    \
  do {                                                                                  \
    static const int once_ = ::fmi::lds_sole_owner((const void*)(kernel), (bytes), #kernel); \
    if (once_ != FMI_OK) return ::fmi::lds_sole_owner((const void*)(kernel), (bytes), #kernel); \
  } while (0)
// guard + launch as ONE statement (usable as the body of an unbraced if / else)
#define FMI_LAUNCH_LDS(kernel, bytes, ...)        \
  do {                                            \
    FMI_LDS_GUARD(kernel, bytes);                 \
    hipLaunchKernelGGL(kernel, __VA_ARGS__);      \
  } while (0)
// A handle remembers the device it was created on; every entry point makes it the calling thread's current device
// (hipSetDevice is per thread, and the Python front door calls from whichever thread holds the pipeline lock).
inline int current_device() {
  int d = 0;
  hipGetDevice(&d);
  return d;
}
inline int use_device_ordinal(int dev) {
  int cur = -1;
  FMI_HIP_TRY(hipGetDevice(&cur));
  if (cur != dev) FMI_HIP_TRY(hipSetDevice(dev));
  return FMI_OK;
}

// ---------------------------------------------------------------- bf16 helpers (device)
typedef uint16_t bf16_t;  // raw bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even (same rule as half::bf16::from_f32) on the hardware converter:
// the fptrunc selects v_cvt_pk_bf16_f32 on gfx950 (a software RNE + NaN branch costs ~10
// instructions and an exec-mask diamond per element — it was 2/3 of the attention loop's VALU work).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  uint32_t u;
  __builtin_memcpy(&u, &r, 4);
  return u;
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float f16_to_f32(uint16_t h) {
  _Float16 v;
  __builtin_memcpy(&v, &h, 2);
  return (float)v;
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
  _Float16 v = (_Float16)f;  // v_cvt_f16_f32: RNE
  uint16_t h;
  __builtin_memcpy(&h, &v, 2);
  return h;
}

// GELU tanh approximation (core/op.rs:539-582, f32 arm) and SiLU (op.rs:699-721)
__device__ __forceinline__ float gelu_tanh(float v) {
  // 0.5 v (1 + tanh(u)), u = k v (1 + 0.044715 v^2), written as v sigmoid(2u) = v - v / (2^(v (c1 + c2 v^2)) + 1) with
  // c1 = 2 k log2(e), c2 = 0.044715 c1: 6 VALU + v_exp_f32 + v_rcp_f32 (1 ulp each) per element instead of 11 + 2 for the
  // tanh form — the GELU of a 256 x 256 tile is 128 of these per lane in the GEMM epilogue.  Saturates cleanly: 2^(+big) = inf
  // -> v, 2^(-big) = 0 -> 0.  The differences to the f32 arm of the reference formula are rounding (far below the bf16 result).
  const float c1 = 2.0f * 0.79788456080286535587989211986876373f * 1.44269504088896340736f;
  const float c2 = c1 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(v * fmaf(c2, v * v, c1));
  return fmaf(-v, __builtin_amdgcn_rcpf(e + 1.0f), v);
}
// The same GELU on two values per instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32; exp2 and rcp stay scalar): component by
// component the operations of gelu_tanh — mul, fma, mul, exp2, add, rcp, fma — so the results are the same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_tanh4(float (&v)[4]) {
  const float c1 = 2.0f * 0.79788456080286535587989211986876373f * 1.44269504088896340736f;
  const float c2 = c1 * 0.044715f;
  const f32x2 k1 = {c1, c1}, k2 = {c2, c2}, one = {1.0f, 1.0f};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2 x = {v[2 * h], v[2 * h + 1]};
    f32x2 t = x * x;
    t = __builtin_elementwise_fma(k2, t, k1);
    t = x * t;
    f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + one;
    const f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    const f32x2 o = __builtin_elementwise_fma(-x, r, x);
    v[2 * h] = o.x, v[2 * h + 1] = o.y;
  }
}
// Sum over the 16 lanes of a DPP row, every lane receiving the total, on the VALU (quad_perm, row_half_mirror, row_mirror) instead
// of four dependent ds_bpermute round trips: the same additions in the same order as the __shfl_xor(1, 2, 4, 8) tree — after the
// two quad steps a quad's lanes agree, so mirroring inside 8 / 16 lanes fetches what lane ^ 4 / lane ^ 8 holds — hence the same bits.
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl_tag) {
    constexpr int ctrl = decltype(ctrl_tag)::value;
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]  = lane ^ 1
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]  = lane ^ 2
  v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror      ~ lane ^ 4 (quads agree)
  v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror           ~ lane ^ 8 (halves agree)
  return v;
}
// 1 / sqrt(mean of squares over the 128 elements of a head + 1e-6): the QkNorm factor (model.rs:186-209) on v_rsq_f32 (1 ulp) —
// the IEEE sqrt + division pair is ~25 instructions per head row; shared by the stand-alone kernel and the GEMM's fused relayout
// so that both produce the same bits
__device__ __forceinline__ float rms_inv128(float ss) { return __builtin_amdgcn_rsqf(ss * (1.0f / 128.0f) + 1e-6f); }
// v sigmoid(v) on v_exp_f32 + v_rcp_f32 (1 ulp each; the IEEE division it replaces is ~10 instructions, and the VAE's GroupNorm + SiLU
// passes over up to 134 M elements are ALU-bound)
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f)); }

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Barrier that guards LDS-DMA data.  hipcc does not reliably drain global_load_lds before a
// __syncthreads() (in the 2x-unrolled attention loop it emitted only lgkmcnt(0) before one of the
// two barriers -> rare stale tiles, caught by the determinism property test), so the wait is explicit:
// every wave first retires ITS OWN DMA pieces, then the barrier makes all pieces visible to all.
__device__ __forceinline__ void dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

constexpr int NUM_XCD = 8;
// Bijective XCD-aware remap: consecutive logical tiles land on the same XCD (= same L2).
// (block b is dispatched to XCD b % 8; cdna guide T1 "XCD swizzle must be bijective")
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
  int q = nwg / NUM_XCD, r = nwg % NUM_XCD;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- kernel launchers (host)
// GEMM problem: y = epi(x W^T + bias).  All leading dims in elements.
enum GemmEpi : int {
  EPI_STORE_BF16 = 0,      // out bf16 = acc + bias
  EPI_GELU_BF16 = 1,       // out bf16 = gelu(acc + bias)
  EPI_RESID_GATE_F32 = 2,  // resid f32 (in/out) += gate[n] * (acc + bias)
  EPI_GELU_FROM_COL = 3,   // out bf16 = (n >= gelu_from ? gelu : id)(acc + bias)
  EPI_STORE_F32 = 4,       // out f32 = acc*alpha + bias
  EPI_SCALE_BF16 = 5,      // out bf16 = acc*alpha + bias
  EPI_RESID_ADD_BF16 = 6,  // out bf16 = resid_bf16 + acc + bias   (VAE)
  EPI_SILU_BF16 = 7
};
struct GemmProblem {
  const bf16_t* A;  // (M, K) activations, row stride lda
  const bf16_t* W;  // (N, K) weights, row stride ldw
  const bf16_t* bias;  // (N) bf16 or null
  void* out;        // bf16 or f32 (M, N), row stride ldo
  const float* gate;  // (N) f32 for EPI_RESID_GATE_F32 (per batch: gate + batch*gate_bstride)
  const void* resid;  // bf16 residual for EPI_RESID_ADD_BF16
  int M, N, K;
  int lda, ldw, ldo;
  int epi;
  int gelu_from;
  float alpha;
  int rows_per_batch;  // for per-batch gate vectors: batch = m / rows_per_batch (0 = single)
  int gate_bstride;
  // 4-bit weights (optional): W is ignored, Wq packed u8 (N, K/2), absmax (N*K/blocksize)
  const uint8_t* Wq;
  const float* absmax;
  int q_blocksize;
  int q_type;  // 0 none, 1 fp4, 2 nf4
  // implicit-GEMM convolution (optional, cv_ks != 0): A is an NHWC image (B, cv_h, cv_w, cv_cin),
  // M = B * (cv_h<<cv_up) * (cv_w<<cv_up) output pixels, K = cv_ks^2 * cv_cin with k = (tap, cin),
  // W is (N, cv_ks, cv_ks, cv_cin); cv_up = 1 folds a nearest-2x upsample into the gather;
  // cv_up = -1 is the stride-2 Downsample of the VAE encoder (M = B * (cv_h/2) * (cv_w/2),
  // zero padding on the right/bottom edge only, vaes/vae.rs:194-201).
  int cv_ks, cv_h, cv_w, cv_cin, cv_up;
  const bf16_t* cv_zero;  // >= 128 B of zeros for padding taps
  // Fused [q|k|v] relayout epilogue (optional, qk_qh != null; 256-wide N tiles only): output columns
  // [0, 3*qk_D) are q | k | v of a fused projection, head h at columns h*128 of each part.  Instead of
  // being stored to `out`, a q / k tile goes through QkNorm (RMS, eps 1e-6, weight qk_wq / qk_wk) + RoPE
  // into the head-major (B,H,qk_Ltot,128) buffers and a v tile into the transposed, kv-permuted
  // (B,H,128,qk_Lpad) layout the attention kernel reads — exactly what qk_norm_rope_kernel and
  // v_transpose_kernel produce from the stored bf16 projection, minus two HBM round trips per layer.
  // Rows m of this problem are tokens: batch m / qk_rows, position qk_row_off + m % qk_rows.
  // Columns >= 3*qk_D (single-stream block: the MLP part) take the normal path.
  bf16_t *qk_qh, *qk_kh, *qk_vt;
  const bf16_t *qk_wq, *qk_wk;
  const float* qk_pe;  // (B or 1, qk_Ltot, 64, {cos, sin}) f32
  int64_t qk_pe_bstride;
  int qk_H, qk_D, qk_rows, qk_row_off, qk_Ltot, qk_Lpad;
  // fp8 attention operands (optional, qk_q8 > 0): q / k leave the relayout epilogue as OCP e4m3 bytes, value * qk_q8 /
  // value * qk_k8 (static scales chosen so that sqrt(128) * max|norm weight| maps to 448), rows of 128 B in the same
  // head-major order; qk_qh / qk_kh then point to byte buffers.  v is unchanged (bf16).
  float qk_q8, qk_k8;
  // 8-bit operands (optional, fp8 != 0: 1 = OCP e4m3, 2 = int8; dense 256-wide N tiles only): A and W point to bytes,
  // K / lda / ldw count elements (= bytes, K % 128 == 0), and the f32 accumulator is multiplied by
  // a_scale[m] * w_scale[n] (per-token / per-output-channel dequantisation) before the epilogue.
  int fp8;
  const float* a_scale;
  const float* w_scale;
  // int8 only, optional (both or neither): the activation rows carry an offset segment (fp8.hip: the post-GELU form of the int8 recipe) —
  // a_off[m] * w_sum[n] is added behind the scaling: y = acc * (a_scale[m] * w_scale[n]) + a_off[m] * w_sum[n] (+ bias ... in the epilogue)
  const float* a_off;
  const float* w_sum;
};
int launch_gemm(const GemmProblem* probs, int nprob, hipStream_t stream);
// Process-wide kernel-selection hooks (tests, ablations; the alternatives are bit-identical):
void set_gemm_w4(bool on);             // residual-update launches (N > 128, no fused relayout) on the 4-wave 128x128-per-wave kernel (default off)
void set_gemm_pingpong(bool on);       // dense N > 128 launches: the ping-pong kernel (default) or the double-buffered one
void set_gemm_w4q_min_rows(int rows);  // 4-bit weights: M from which the one-wave-per-SIMD fused dequant-GEMM runs (default 256)
void set_gemm_w4_qkv_min_n(int n);     // dense fused-relayout launches at least this wide run on the 4-wave kernel (default: never)

// attention output routing: query rows [0,rows0) -> p0, the rest -> p1 (token-major, head h at
// column h*128); or head-major (B,H,Lq,128) in p1.
struct AttnOut {
  bf16_t* p0;
  int rows0;
  int ld0;
  int64_t bstride0;
  bf16_t* p1;
  int ld1;
  int64_t bstride1;
  int head_major;
};
// qk_fp8 != 0: q and k are e4m3 bytes (B,H,L,128), QK^T runs on the fp8 MFMA (scale must already hold 1 / (q scale * k scale))
constexpr int ATT_NO_EXP2 = 1 << 30;
int launch_attention_ex(const bf16_t* q, const bf16_t* k, const bf16_t* vt, const AttnOut& out, int B, int H, int Lq, int Lk,
                        int Lkpad, float scale, int rescale_thr_x16, hipStream_t stream, int qk_fp8 = 0, float* lse = nullptr, int nsplit = 0,
                        int score_exp2 = ATT_NO_EXP2, int kind = -1, float v_inv = 1.f);
// qk_fp8 == 2 (round 5): vt is e4m3 as well — (B, H, 128, Lkpad) BYTES, keys in plain order, value * v_scale, zero beyond Lk — and P is rounded to e4m3:
// both products on the fp8 MFMA (attention_w16l_kernel<.., true, true>); v_inv = 1 / v_scale.  Only the lock-step stream has this form: the call
// needs a power-of-two score factor and more than one KV tile, anything else is an error (there is no kernel to fall back to).
// kind: 0..5 = this kernel (fmi_set_attention_kernel's numbering: a model handle's own choice, fmi_flux_set_attention_kernel), -1 = the process-wide switches
// score_exp2 (fp8 QK^T only): the caller KNOWS that scale * log2(e) == 2^score_exp2 exactly and says so as an integer (the model's fp8
// mode constructs its q scale that way) -> the one-wave stream, which carries the factor in the MFMA's E8M0 block scale.  ATT_NO_EXP2 =
// unknown: the launcher recognises an exact power of two itself, anything else runs on the 8-wave kernel and is COUNTED
// (attention_fp8_fallbacks(), visible in fmi_device_info) — a silent slide to the slower, numerically different kernel was ADVICE r3's finding.
unsigned long long attention_fp8_fallbacks();
// the op-level entries' scratch: a grow-only block per (device, stream) the library holds (capi.hip: ScratchCache) — no allocation, no wait in the steady
// state.  The guard keeps the cache locked from get() until it goes out of scope at the end of the entry, so the op's launches are enqueued as one unit.
struct OpScratch {
  explicit OpScratch(hipStream_t stream);
  ~OpScratch();
  OpScratch(const OpScratch&) = delete;
  OpScratch& operator=(const OpScratch&) = delete;
  int get(size_t bytes);
  void* p = nullptr;

 private:
  hipStream_t s_;
  bool locked_ = false;
};
bool alt_kernels_built();  // attention.hip: was THIS library linked from the test build's objects (the flag differs per object: only attention.o / gemm_bf16.o)
void set_attention_pingpong(bool on);  // 8-wave kernels: ping-pong (default) or the single-barrier one
void set_attention_w4(bool on);        // bf16 operands: one-wave-per-SIMD kernel (default) or the 8-wave ones
void set_attention_w16(bool on);       // bf16 operands: the 16x16x32-MFMA one-wave kernel in front of the others (default on)
void set_attention_w16l(bool on);      // bf16 operands: the lock-step schedule of the 16x16x32 kernel, in front of all (default on)
void set_attention_w32(bool on);       // bf16 operands: the same design on the 32x32x16 MFMA, in front of all (default off)
// flash attention, d = 128.  q,k: (BH, L, 128) bf16; vt: (BH, 128, Lpad) bf16 with the kv axis
// permuted inside each group of 16 (see attention.hip); out token-major (B, L, H*128) or (BH,L,128)
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out, int B, int H,
                     int Lq, int Lk, int Lkpad, float scale, int out_token_major, hipStream_t stream);
// (rows, 128)-per-head V -> V^T layout consumed by launch_attention.  v: token-major with row
// stride ldv (elements) and head h at column h*128; rows [0,rows) of batch b land at kv = row_off+r
int launch_v_transpose(const bf16_t* v, int ldv, int64_t v_bstride, bf16_t* vt, int B, int H, int rows,
                       int row_off, int Lpad, hipStream_t stream);
int launch_vt_zero_pad(bf16_t* vt, int BH, int L, int Lpad, hipStream_t stream);
// V (BH, Lk, 128) bf16 -> V^T (BH, 128, Lpad) e4m3 bytes = e4m3(clamp(v * v_scale, +-448)), plain key order, zero from Lk on (the qk_fp8 == 2 operand)
int launch_v_transpose_fp8(const bf16_t* v, uint8_t* vt8, int BH, int Lk, int Lpad, float v_scale, hipStream_t stream);
// sequence-parallel exchange buffers (seq_parallel.hip): q|k|vt of the local tokens <-> all tokens of H/N heads
size_t sp_qkv_bytes_per_peer(int Hr, int Ll);
size_t sp_o_bytes_per_peer(int Hr, int Ll);
int launch_sp_pack_qkv(const bf16_t* q, const bf16_t* k, const bf16_t* vt, void* send, int H, int Tl, int Sl, int N, hipStream_t s);
int launch_sp_unpack_qkv(const void* recv, bf16_t* qf, bf16_t* kf, bf16_t* vtf, int H, int Tl, int Sl, int N, hipStream_t s);
int launch_sp_pack_o(const bf16_t* o, void* send, int H, int Tl, int Sl, int N, hipStream_t s);
int launch_sp_unpack_o(const void* recv, const AttnOut& out, int H, int Ll, int N, hipStream_t s);
// key-split attention: parts (S, L, W) bf16 normalised partial outputs, lse (S, W/128 heads, L) f32 (log2 domain) ->
// out (L, W) = sum_s 2^(lse_s - max) parts_s / sum_s 2^(lse_s - max)
int launch_sp_merge_splits(const bf16_t* parts, const float* lse, int S, bf16_t* out, int heads, int L, hipStream_t s);
// q/k RMSNorm (eps 1e-6, weight (128)) + RoPE, token-major in (row stride ld, head h at col h*128)
// -> head-major (B,H,Ltot,128) at row offset row_off.  pe: (B or 1, Ltot, 64, 2) f32 {cos,sin}
int launch_qk_norm_rope(const bf16_t* q, const bf16_t* k, int ld, int64_t in_bstride, const bf16_t* wq,
                        const bf16_t* wk, const float* pe, int64_t pe_bstride, bf16_t* qo, bf16_t* ko,
                        int B, int H, int rows, int row_off, int Ltot, hipStream_t stream);
int launch_qk_norm_rope_f32(const bf16_t* q, const bf16_t* k, int ld, int64_t in_bstride, const bf16_t* wq, const bf16_t* wk, const float* pe,
                            int64_t pe_bstride, float* qo, float* ko, int B, int H, int rows, int row_off, int Ltot, hipStream_t stream);
int launch_rope_table(const float* txt_ids, const float* img_ids, int B, int T, int S, const int* axes,
                      int theta, float* pe, hipStream_t stream);
// LayerNorm(no affine) * (1+scale) + shift; x f32 (rows, D) -> bf16. scale/shift per batch
// (vector + batch*mod_bstride), batch = row / rows_per_batch.
// two row sets in one launch (the image and text streams of a double block); rows2 = 0: one set
int launch_layernorm_mod2(const float* x, const float* scale, const float* shift, int mod_bstride, int rows_per_batch, bf16_t* out, int rows,
                          const float* x2, const float* scale2, const float* shift2, int rows_per_batch2, bf16_t* out2, int rows2, int D, float eps,
                          hipStream_t stream);
int launch_layernorm_mod_fp8_2(const float* x, const float* scale, const float* shift, int mod_bstride, int rows_per_batch, uint8_t* out,
                               float* out_scale, int rows, const float* x2, const float* scale2, const float* shift2, int rows_per_batch2, uint8_t* out2,
                               float* out_scale2, int rows2, int D, float eps, hipStream_t stream, int kind = 1, const float* smooth = nullptr,
                               const float* smooth2 = nullptr);
int launch_layernorm_mod(const float* x, const float* scale, const float* shift, int mod_bstride,
                         int rows_per_batch, bf16_t* out, int rows, int D, float eps, hipStream_t stream);
// y(M,N) f32 (+)= act_in(x(M,K) f32) W(N,K)^T bf16 + bias bf16 ; M <= 8
// fp8 path (fp8.hip).  Row-wise dynamic quantisation: scale[r] = max(absmax(x[r,:]), 1e-30) / 448,
// out[r,k] = e4m3_rne(x[r,k] * (448 / max(absmax, 1e-30))); x bf16 with row stride ld, out (rows, K) dense.
// kind 2: the int8 form (scale = absmax / 127, codes = clamp(rint(x * 127 / absmax), -127, 127)) — fp8.hip's header
// vec (optional, K floats, 16-byte aligned): every column is multiplied by vec[k] in f32 before the recipe (the smoothed int8 recipe: 1 / s for activations, s for weights)
int launch_quantize_rows_fp8(const bf16_t* x, int ld, int rows, int K, uint8_t* out, float* scale, hipStream_t stream, int kind = 1, const float* vec = nullptr);
// int8, post-GELU form (fp8.hip's header): columns [0, d0) symmetric, columns [d0, K) on 256 levels over their [min, max], one step per row;
// offset[r] = lo + 128 * scale[r].  d0 % 8 == 0, 0 <= d0 < K.
int launch_quantize_rows_i8_asym(const bf16_t* x, int ld, int rows, int K, int d0, uint8_t* out, float* scale, float* offset, hipStream_t stream, const float* vec = nullptr);
// w_sum[n] = w_scale[n] * float(sum_{k >= d0} wq[n,k]) over int8 codes (N, K) row-major: the column term of the offset segment
int launch_rowsum_i8(const int8_t* wq, const float* w_scale, int N, int K, int d0, float* w_sum, hipStream_t stream);
// launch_layernorm_mod with the row quantisation fused (values quantised from f32, not via bf16)
int launch_layernorm_mod_fp8(const float* x, const float* scale, const float* shift, int mod_bstride, int rows_per_batch,
                             uint8_t* out, float* out_scale, int rows, int D, float eps, hipStream_t stream, int kind = 1, const float* smooth = nullptr);
// calibration of the smoothed int8 recipe (fp8.hip): column absmax of a bf16 matrix folded into amax (atomic max), and the factors s, 1 / s from the two statistics
int launch_col_absmax(const bf16_t* x, int ld, int rows, int K, float* amax, hipStream_t stream);
void smooth_factors_host(const float* act_amax, const float* w_amax, int K, float* s_out, float* inv_out);  // host arrays; median-floored SmoothQuant (fp8.hip)
int launch_gemv(const float* x, const bf16_t* W, const bf16_t* bias, float* y, int M, int N, int K,
                int silu_in, int accumulate, hipStream_t stream);
// bf16 out = w_i8 * SCB[row] / 127 (dequant.cu:205-214) on `stream`
int launch_dequant_int8_scb_bf16(const int8_t* w, const float* scb, bf16_t* out, int col, int64_t n, hipStream_t stream);
int launch_timestep_embedding(const float* t, int B, int dim, float* out, hipStream_t stream);
int launch_cast_to_bf16(const void* src, fmi_dtype dt, bf16_t* dst, int64_t n, hipStream_t stream);
int launch_cast_to_f32(const void* src, fmi_dtype dt, float* dst, int64_t n, hipStream_t stream);
int launch_silu_to_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t stream);
int launch_euler_update(float* img, const float* pred, float dt, int64_t n, hipStream_t stream);
int launch_add2_rows(float* y, const float* g, const float* v, int R, int B, int N, hipStream_t stream);  // y[r] = (y[r] + g[r % B]) + v[r % B] (g may be null)
// split-K reduce: out[m][n] += gate_b[n] * (part_0 + part_1 + ... + part_{S-1} + bias[n]), the parts added in index order
// (f32 (S, M, N) contiguous); the EPI_RESID_GATE_F32 epilogue for a launch whose K range was cut into S problems
int launch_splitk_resid_gate(const float* parts, int S, const bf16_t* bias, const float* gate, int rows_per_batch, int gate_bstride, float* out, int ldo,
                             int M, int N, hipStream_t stream);
int launch_split_rows_f32(const float* src, float* dst, int B, int rows_src_per_b, int row_off, int rows, int D, hipStream_t stream);

}  // namespace fmi
