// gemm_bf16.hip — grouped bf16 GEMM  y = epi(x · Wᵀ + bias)  on v_mfma_f32_16x16x32_bf16 (fp8: 32x32x64_f8f6f4),
// plus the implicit-GEMM 3x3 / 1x1 convolution of the VAE decoder on the same main loop.
//
// Replaces every Linear on the FLUX hot path: UnquantLinear::forward
// (diffusion_rs_backend/src/unquantized/mod.rs:34-77: cuBLASLt TN batched matmul with bias as C)
// and, with QUANT, BnbLinear::forward (bitsandbytes/mod.rs:301-312) without the dense
// dequantised round trip through HBM; with CONV, Conv2d::forward (nn/conv.rs:212-230 ->
// im2col + GEMM + strided copy, cuda_backend/mod.rs:1544-1599) without materialising im2col.
//
// CDNA4 design (not a port of the reference's cuBLAS call):
//   * 256 x (128|256) x 64 macro tile, 8 waves (2 along M x 4 along N), each wave owns
//     128 x (32|64) of C as 8 x 2 NJ accumulators of the 16x16x32 MFMA (this part sustains that form 14 % better than
//     32x32x16, tools/mfma_peak; the results are bit-identical).  The epilogues are written over accumulator views
//     (Acc32 / Acc16) so that the fp8 kernel and the fused nf4 kernel, which stay on 32 x 32 tiles, share them.
//   * A and W tiles are DMA'd HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip),
//     double buffered.  The LDS image is lane-linear, so the bank-conflict XOR swizzle (16-B
//     slot c -> c ^ ((row>>1)&7) inside each 128-B row) is applied on the per-lane *source*
//     address and again on the ds_read_b128 address.
//   * Operands are swapped (W rows feed the MFMA A operand, x rows feed B) so every lane ends
//     up with 4 consecutive output columns of one row: 8-byte bf16 / 16-byte f32 stores and
//     vector loads of bias / gate in the epilogue.
//   * One barrier per K tile; the DMA of tile k+1 is in flight while tile k is multiplied.
//   * Grouped launch (img + txt streams of a double block in one grid) and an XCD-aware
//     bijective block remap so tiles sharing a W panel sit on the same XCD's L2.
//   * Epilogues fuse bias, GELU(tanh), gate*y + residual (f32 residual stream), per-column-range
//     GELU (single-stream block's [q|k|v|mlp] fused projection), scale, bf16 residual add.
//   * QUANT: nf4 / fp4 weight tiles are read packed (32 B per row per K tile), expanded with the
//     16-entry LUT * absmax in registers and written to the same swizzled LDS image — the
//     "dequant as an LDS stage" of BASELINE.json's north star.
//   * CONV: the A-tile row is an output pixel, the K tile a (tap, 64-channel) slice of an NHWC
//     image; padding taps read a zero line, a nearest-2x upsample is folded into the gather.
#include <cstdlib>
#include <type_traits>

#include <atomic>

#include "common.h"

namespace fmi {

// Tile-rows per band of the tile order (see the kernels).  8 is the default (8 x 4 patches per XCD; measured best on the shapes whose
// tile-row count is a multiple of 8); launch_gemm picks another height per problem when 8 would leave a ragged last band
// (pick_tile_band: M = 4608 has 18 tile rows -> 6 + 6 + 6 instead of 8 + 8 + 2, -11 % L2-miss bytes).
constexpr int TILE_BAND = 8;
constexpr int BM = 256, BK = 64;
constexpr int GEMM_THREADS = 512;
constexpr int A_TILE_BYTES = BM * BK * 2;  // 32 KiB
constexpr int MAX_PROBLEMS = 8;

struct GemmBatch {
  GemmProblem p[MAX_PROBLEMS];
  int tile_start[MAX_PROBLEMS + 1];
  int band[MAX_PROBLEMS];  // tile-rows per band of problem i's tile order (pick_tile_band)
  int nprob;
};

// Logical tile t of a problem -> (tm, tn).  Tiles are numbered band by band (gh tile-rows each), column by column inside a band, so
// the ~32 consecutive tiles an XCD runs at any time form a compact gh x (32 / gh) patch: gh + 32 / gh distinct A / W panels per K
// step instead of 33 (L2 hits).  Bijective for any gh >= 1.
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int gh, int& tm, int& tn) {
  const int band = t / (gh * tiles_n);
  const int band_h = min(gh, tiles_m - band * gh);
  const int tin = t - band * gh * tiles_n;
  tn = tin / band_h;
  tm = band * gh + tin % band_h;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __constant__ float kNF4[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                          -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                          0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                                          0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

// fp4 tree of dequant.cu:12-37 as (value * absmax) * sign, same operation order
__device__ __forceinline__ float dq_fp4(unsigned v, float am) {
  const float tab[8] = {0.0f, 5.208333333e-03f, 0.66666667f, 1.0f, 0.33333333f, 0.5f, 0.16666667f, 0.25f};
  float sign = (v & 8) ? -1.0f : 1.0f;
  return tab[v & 7] * am * sign;
}

// Stage ROWS x 64 bf16 (rows r0.., cols k0..k0+63 of a row-major matrix with `ld`) into LDS.
// Rows past `rmax` are clamped (duplicates of the last row; their results are never stored).
template <int ROWS>
__device__ __forceinline__ void stage_tile_dma(const bf16_t* __restrict g, int ld, int r0, int rmax, int k0, char* lds_tile, int wave, int lane) {
  const int r8 = lane >> 3, cs = lane & 7;
  constexpr int CPW = ROWS / 64;  // 1-KiB chunks (8 rows) per wave
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int chunk = wave * CPW + i;
    const int row = chunk * 8 + r8;  // tile-local row
    const int src_slot = cs ^ ((row >> 1) & 7);
    int grow = r0 + row;
    grow = grow > rmax ? rmax : grow;
    const bf16_t* src = g + (int64_t)grow * ld + k0 + src_slot * 8;
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lds_tile + chunk * 1024), 16, 0, 0);
  }
}

// x / 127 for the LLM.int8 expansion, correctly rounded without the IEEE division sequence: q = x * RN(1/127), one Newton
// step on the residual.  Equal to x / 127.0f for EVERY finite f32 x with |x| >= 2^-118 (checked exhaustively over all 2^32
// bit patterns on the host, tests/test_oracle_kats.py keeps a sampled version), so the fused path reproduces
// dequantize_8bit (dequant.cu:205-214: w * SCB / 127) bit for bit at 3 instructions instead of ~10.
__device__ __forceinline__ float div127(float x) {
  const float r = 1.0f / 127.0f;
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-127.0f, q, x), r, q);
}

// Quantised weight tile, expanded by the VALU into the swizzled bf16 LDS image: ROWS x 64 k; thread t expands half a row
// (4-bit: 16 packed bytes; LLM.int8, q_type 3: 32 int8 with the row's SCB).
template <int ROWS>
__device__ __forceinline__ void stage_tile_q4(const GemmProblem& P, int n0, int k0, char* lds_tile, int tid) {
  const int row = tid >> 1, half = tid & 1;
  if (row >= ROWS) return;
  int n = n0 + row;
  n = n > P.N - 1 ? P.N - 1 : n;
  if (P.q_type == 3) {
    const uint4* src = reinterpret_cast<const uint4*>(P.Wq + (int64_t)n * P.K + k0 + half * 32);
    const uint4 p0 = src[0], p1 = src[1];
    const float scb = P.absmax[n];
    const uint32_t w8[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const int sw8 = (row >> 1) & 7;
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // 4 chunks of 8 weights (two dwords each)
      uint32_t out[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t d = w8[2 * c + (b >> 1)];
        const int i0 = (int)(d << (24 - 16 * (b & 1))) >> 24, i1 = (int)(d << (16 - 16 * (b & 1))) >> 24;  // sign-extended bytes 2(b&1), 2(b&1)+1
        out[b] = pack_bf16x2(div127((float)i0 * scb), div127((float)i1 * scb));
      }
      const int slot = (half * 4 + c) ^ sw8;
      *reinterpret_cast<uint4*>(lds_tile + row * 128 + slot * 16) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    return;
  }
  const int64_t e0 = (int64_t)n * P.K + k0 + half * 32;  // first element index of this thread's 32 weights
  const uint4 pk = *reinterpret_cast<const uint4*>(P.Wq + (e0 >> 1));
  const float am = P.absmax[e0 / P.q_blocksize];
  const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
  const int sw = (row >> 1) & 7;
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // 4 chunks of 8 weights (4 bytes each)
    uint32_t out[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const unsigned byte = (w[c] >> (8 * b)) & 0xffu;
      float hi, lo;  // high nibble first (dequant.cu:142-151)
      if (P.q_type == 2) {
        hi = kNF4[byte >> 4] * am;
        lo = kNF4[byte & 15] * am;
      } else {
        hi = dq_fp4(byte >> 4, am);
        lo = dq_fp4(byte & 15, am);
      }
      out[b] = pack_bf16x2(hi, lo);
    }
    const int slot = (half * 4 + c) ^ sw;
    *reinterpret_cast<uint4*>(lds_tile + row * 128 + slot * 16) = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

// Shared epilogue.  Lane holds, for accumulator (i, j): row m = m0 + wm*128 + i*32 + (lane&31) and
// columns n = n0 + wn*32*NJ + j*32 + 8q + 4(lane>>5) + {0..3} in registers 4q..4q+3.
// `smem` (>= 8 * 8192*NJ bytes) is free for staging once every wave has passed the leading barrier.
// Accumulator views: which (row, 4 consecutive columns) of its wave's 128 x (32 NJ) sub-tile a lane owns, so that the epilogues
// below are written once for both MFMA shapes.  each<HALF>(f) calls f(r, c, v, slot) for every group the lane owns — r = row inside
// the sub-tile, c = index of the 4-column group (column = 4 c), v = the four f32 — for all rows (HALF = -1) or one 64-row half.
// A lane owns only kSlots distinct column groups (the same ones in every row block): `slot` numbers them (a compile-time value
// after unrolling), cols(f) calls f(slot, c) once for each — what the epilogues use to fetch the bias ONCE, in one round trip,
// instead of one dependent global load in front of every group (measured: 32 serialised L2 round trips per lane and tile).
//   Acc32: v_mfma_f32_32x32x16_bf16 / 32x32x64_f8 with swapped operands: acc[i][j] is rows 32 i + (lane & 31), columns
//          32 j + 8 q + 4 (lane >> 5) + {0..3} in registers 4 q .. 4 q + 3.
//   Acc16: v_mfma_f32_16x16x32_bf16: acc[tm][tn] is rows 16 tm + (lane & 15), columns 16 tn + 4 (lane >> 4) + {0..3}.
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4_nt;
template <int NJ>
struct Acc32 {
  f32x16 (&a)[4][NJ];
  int lane;
  static constexpr bool kFence = NJ == 4;
  static constexpr int kSlots = 4 * NJ;
  template <class F>
  __device__ __forceinline__ void cols(F&& f) const {
    const int hl = lane >> 5;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) f(j * 4 + q, j * 8 + q * 2 + hl);
  }
  template <int HALF, class F>
  __device__ __forceinline__ void each(F&& f) const {
    const int hl = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (HALF >= 0 && (i >> 1) != HALF) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = a[i][j][q * 4 + e];
          f(i * 32 + l31, j * 8 + q * 2 + hl, v, j * 4 + q);
          // 4-wave kernels: the accumulators sit in AGPRs; without a fence the scheduler hoists all 256 reads and spills
          if constexpr (kFence) __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
};
template <int NJ>
struct Acc16 {
  f32x4 (&a)[8][2 * NJ];
  int lane;
  static constexpr bool kFence = NJ == 4;
  static constexpr int kSlots = 2 * NJ;
  template <class F>
  __device__ __forceinline__ void cols(F&& f) const {
    const int l4 = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 2 * NJ; ++tn) f(tn, tn * 4 + l4);
  }
  template <int HALF, class F>
  __device__ __forceinline__ void each(F&& f) const {
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < 8; ++tm) {
      if (HALF >= 0 && (tm >> 2) != HALF) continue;
#pragma unroll
      for (int tn = 0; tn < 2 * NJ; ++tn) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[tm][tn][e];
        f(tm * 16 + l15, tn * 4 + l4, v, tn);
        if constexpr (kFence) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
};

// Fused [q|k|v] relayout of one 256 x 256 tile (two heads of q, k or v); see GemmProblem::qk_*.
// The arithmetic (bf16 rounding of the projection first, then f32 RMS / RoPE in the same operation
// order, 16 lanes x 8 elements per head row) is that of qk_norm_rope_kernel, so both paths produce
// the same bits.  Requires qk_rows, qk_row_off and M to be multiples of 16 (host-checked).
// Two halves, so that the 4-wave kernels (each wave = two of the eight 128 x 64 sub-tiles) run them as
//   barrier; stage(sub-tile 0); stage(sub-tile 1); barrier; emit(row group 0); emit(row group 1)
// over the same LDS image and the same stores: `wave` is the index 0..7 in the 2 x 4 layout.
template <class ACC>
__device__ __forceinline__ void qkv_relayout_stage(const GemmProblem& P, const ACC& acc, char* smem, int n0, int wave, int lane) {
  const int wm = wave >> 2, wn = wave & 3;
  const int part = n0 / P.qk_D;                     // 0 q, 1 k, 2 v
  // the lane's kSlots bias groups, fetched once (see the accumulator views)
  uint2 bc[ACC::kSlots];
  if (P.bias) acc.cols([&](int slot, int c) { bc[slot] = *reinterpret_cast<const uint2*>(P.bias + n0 + wn * 64 + 4 * c); });
  auto biased = [&](int slot, float (&v)[4]) {
    if (P.bias) {
      const uint2 b = bc[slot];
      v[0] += bf16_to_f32((bf16_t)(b.x & 0xffff));
      v[1] += bf16_to_f32((bf16_t)(b.x >> 16));
      v[2] += bf16_to_f32((bf16_t)(b.y & 0xffff));
      v[3] += bf16_to_f32((bf16_t)(b.y >> 16));
    }
  };
  if (part < 2) {
    // ---- stage the bf16 tile in the wave-private swizzled regions of the normal store path
    char* cw = smem + wave * 16384;
    acc.template each<-1>([&](int r, int c, float (&v)[4], int slot) {
      biased(slot, v);
      *reinterpret_cast<uint2*>(cw + r * 128 + ((c ^ (r & 15)) << 3)) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    });
  } else {
    // ---- v: stage TRANSPOSED, [d column 0..255][tile-local token 0..255] bf16 (512-B rows), with the
    // attention kernel's kv permutation (swap bits 2,3 inside groups of 16) applied to the token
    // index on the way in, so that a row leaves as plain 16-B pieces of consecutive stored positions
    bf16_t* tp = reinterpret_cast<bf16_t*>(smem);
    acc.template each<-1>([&](int r, int c, float (&v)[4], int slot) {
      const int tl = wm * 128 + r;
      const int tpos = (tl & ~12) | ((tl & 4) << 1) | ((tl & 8) >> 1);
      const int dc = wn * 64 + 4 * c;
      biased(slot, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) tp[(dc + e) * 256 + tpos] = f32_to_bf16(v[e]);
    });
  }
}

__device__ __forceinline__ void qkv_relayout_emit(const GemmProblem& P, char* smem, int m0, int n0, int wave, int lane) {
  const int hl = lane >> 5;
  const int part = n0 / P.qk_D;                     // 0 q, 1 k, 2 v
  const int head0 = (n0 - part * P.qk_D) >> 7;      // first of the tile's two heads
  if (part < 2) {
    // ---- 512 (row, head) vectors of 128: 16 lanes x 8 elements each; row group `wave` is rows 32*wave .. +31
    const int sub = lane & 15, grp = lane >> 4;
    const bf16_t* wsel = part == 0 ? P.qk_wq : P.qk_wk;
    bf16_t* osel = part == 0 ? P.qk_qh : P.qk_kh;
    const uint4 wraw = *reinterpret_cast<const uint4*>(wsel + sub * 8);
    const bf16_t* we = reinterpret_cast<const bf16_t*>(&wraw);
    // GRP rows at a time: their RoPE table entries are requested first (clamped row, no branch between the loads), then the
    // rows are normalised, rotated and stored — one L2 round trip per GRP rows; written row by row the loads of row i + 1 cannot
    // move above the store of row i (they may alias for all the compiler knows) and every row pays its own round trip
    // sample and position of a row: one division per tile (a tile rarely spans more than two samples), not one per row
    const int b0 = m0 / P.qk_rows;
    auto locate = [&](int m, int& b, int& pos) {
      b = b0;
      int rem = m - b0 * P.qk_rows;
      while (rem >= P.qk_rows) rem -= P.qk_rows, ++b;
      pos = P.qk_row_off + rem;
    };
    constexpr int GRP = 4;
#pragma unroll 1
    for (int it0 = 0; it0 < 16; it0 += GRP) {
      float4 pc01[GRP], pc23[GRP];
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int mc = min(m0 + ((wave * 64 + (it0 + u) * 4 + grp) >> 1), P.M - 1);
        int b, pos;
        locate(mc, b, pos);
        const float4* pp = reinterpret_cast<const float4*>(P.qk_pe + (int64_t)b * P.qk_pe_bstride + ((int64_t)pos * 64 + 4 * sub) * 2);
        pc01[u] = pp[0], pc23[u] = pp[1];
      }
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const int item = wave * 64 + (it0 + u) * 4 + grp;
        const int r = item >> 1, hh = item & 1;
        const int m = m0 + r;
        // source: staging region of wave (r>>7, 2*hh + (sub>>3)), row r&127, 16 B = 8-B slots 2*(sub&7), +1
        const int rr = r & 127;
        const char* reg = smem + ((r >> 7) * 4 + 2 * hh + (sub >> 3)) * 16384 + rr * 128;
        const int sp = ((2 * (sub & 7)) ^ (rr & 15)) & ~1;
        uint4 raw = *reinterpret_cast<const uint4*>(reg + (sp << 3));
        if (rr & 1) raw = make_uint4(raw.z, raw.w, raw.x, raw.y);  // odd rows hold the slot pair swapped
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&raw);
        float v[8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[i] = bf16_to_f32(e[i]);
          ss += v[i] * v[i];
        }
        ss = row16_sum(ss);
        if (m < P.M) {
          int b, pos;
          locate(m, b, pos);
          const float4 c01 = pc01[u], c23 = pc23[u];
          const float cs[4] = {c01.x, c01.z, c23.x, c23.z};
          const float sn[4] = {c01.y, c01.w, c23.y, c23.w};
          const float inv = rms_inv128(ss);
          float rr8[8];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float x0 = v[2 * p] * inv * bf16_to_f32(we[2 * p]);
            const float x1 = v[2 * p + 1] * inv * bf16_to_f32(we[2 * p + 1]);
            rr8[2 * p] = cs[p] * x0 - sn[p] * x1;
            rr8[2 * p + 1] = sn[p] * x0 + cs[p] * x1;
          }
          const int64_t row = ((int64_t)b * P.qk_H + head0 + hh) * P.qk_Ltot + pos;
          if (P.qk_q8 > 0.f) {  // fp8 attention operands: e4m3(value * static scale), 8 bytes per lane
            const float sc8 = part == 0 ? P.qk_q8 : P.qk_k8;
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(rr8[0] * sc8, rr8[1] * sc8, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(rr8[2] * sc8, rr8[3] * sc8, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(rr8[4] * sc8, rr8[5] * sc8, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(rr8[6] * sc8, rr8[7] * sc8, hi, true);
            *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(osel) + row * 128 + sub * 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
          } else {
            bf16_t* dst = osel + row * 128 + sub * 8;
            *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(rr8[0], rr8[1]), pack_bf16x2(rr8[2], rr8[3]), pack_bf16x2(rr8[4], rr8[5]), pack_bf16x2(rr8[6], rr8[7]));
          }
        }
      }
    }
  } else {
    const bf16_t* tp = reinterpret_cast<const bf16_t*>(smem);
    const int g8 = lane & 31;  // 8 stored positions (16 B) of a row per lane
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int dc = wave * 32 + it * 2 + hl;
      const int m16 = m0 + 16 * (g8 >> 1);  // the 16 tokens this piece's group comes from
      if (m16 < P.M) {
        const uint4 x = *reinterpret_cast<const uint4*>(tp + dc * 256 + g8 * 8);
        const int b = m16 / P.qk_rows, kv16 = P.qk_row_off + (m16 - b * P.qk_rows);
        bf16_t* dst = P.qk_vt + (((int64_t)b * P.qk_H + head0 + (dc >> 7)) * 128 + (dc & 127)) * P.qk_Lpad + kv16 + (g8 & 1) * 8;
        *reinterpret_cast<uint4*>(dst) = x;
      }
    }
  }
}

template <class ACC>
__device__ __forceinline__ void qkv_relayout_epilogue(const GemmProblem& P, const ACC& acc, char* smem, int m0, int n0, int wave, int lane) {
  __syncthreads();  // every wave is done with the operand tiles
  qkv_relayout_stage(P, acc, smem, n0, wave, lane);
  __syncthreads();
  qkv_relayout_emit(P, smem, m0, n0, wave, lane);
}

// WN = waves along N (4: the 8-wave kernels, wave = wm*4 + wn, 128 x 32*NJ per wave; 2: the 4-wave kernel, 128 x 128 per wave)
// ACT: -1 = everything decided at run time inside the loops (the 8-wave double-buffered kernels); otherwise the kernel is
// instantiated per kind and the host picks it: 0 = no activation / no alpha, 1 = GELU, 2 = GELU from a column on,
// 3 = the remaining kinds at run time (alpha scale, SiLU).  The loops below are fully unrolled (the accumulator array
// must stay in registers), so a run-time switch inside them multiplies the code of every activation by 32-64
// iterations: with 128-wide wave tiles the staged path alone was ~90 KB of instructions and ran out of the instruction
// cache (measured on the 4-wave kernel: 45 000 clocks for 64 LDS writes).
template <int NJ, int WN, int ACT, class ACC>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmProblem& P, const ACC& acc, char* smem, int m0, int n0, int wave, int lane) {
  constexpr int BN = WN * 32 * NJ;
  const int wm = wave / WN, wn = wave % WN;
  // ---- epilogue: which rows / 4-column groups a lane holds is the accumulator view's business (Acc32 / Acc16)
  const int epi = P.epi;
  const float alpha = P.alpha;
  auto add_bias = [](const uint2 b, float (&v)[4]) {
    v[0] += bf16_to_f32((bf16_t)(b.x & 0xffff));
    v[1] += bf16_to_f32((bf16_t)(b.x >> 16));
    v[2] += bf16_to_f32((bf16_t)(b.y & 0xffff));
    v[3] += bf16_to_f32((bf16_t)(b.y >> 16));
  };
  auto scale = [&](float (&v)[4]) {
    if constexpr (ACT == 3 || ACT == -1) {
      if (epi == EPI_STORE_F32 || epi == EPI_SCALE_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= alpha;
      }
    }
  };
  const int tile_gelu = n0 >= P.gelu_from ? 1 : (n0 + BN <= P.gelu_from ? -1 : 0);  // all / none / mixed (ACT == 2)
  auto activate = [&](int n, float (&v)[4]) {
    if constexpr (ACT == -1) {
      if (epi == EPI_GELU_BF16 || (epi == EPI_GELU_FROM_COL && n >= P.gelu_from)) {
        gelu_tanh4(v);
      } else if (epi == EPI_SILU_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
      }
    } else if constexpr (ACT == 1) {
      gelu_tanh4(v);
    } else if constexpr (ACT == 2) {
      // whole tiles lie on one side of gelu_from in the model's launches (a multiple of the tile width): decided per tile, the
      // per-lane comparison only where a tile straddles it
      if (tile_gelu > 0 || (tile_gelu == 0 && n >= P.gelu_from)) gelu_tanh4(v);
    } else if constexpr (ACT == 3) {
      if (epi == EPI_SILU_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu(v[e]);
      }
    }
  };
  // alpha, bias, activation on 4 consecutive columns starting at n (direct path: the bias is read per group)
  auto finish = [&](int n, float (&v)[4], bool full) {
    scale(v);
    if (P.bias) {
      if (full) {
        add_bias(*reinterpret_cast<const uint2*>(P.bias + n), v);
      } else {
        for (int e = 0; e < 4 && n + e < P.N; ++e) v[e] += bf16_to_f32(P.bias[n + e]);
      }
    }
    activate(n, v);
  };
  const bool f32_out = (epi == EPI_RESID_GATE_F32 || epi == EPI_STORE_F32);
  // Staged path: the C tile goes through LDS (free after the K loop) and leaves as whole 128-B
  // (bf16) / 256-B (f32) row segments with 16-B stores.  Direct per-lane stores touch 32 partial
  // cache lines per instruction and cost ~30 us per tile with nothing else resident on the CU.
  const bool staged = (P.ldo % 8 == 0) && (n0 + BN <= P.N) && ((reinterpret_cast<uintptr_t>(P.out) & 15) == 0) && epi != EPI_RESID_ADD_BF16 &&
                      (P.bias == nullptr || (reinterpret_cast<uintptr_t>(P.bias) & 7) == 0);
  if (staged) {
    __syncthreads();  // every wave is done with the operand tiles
    char* cw = smem + wave * (8192 * NJ);  // wave-private staging region
    const int ncol0 = n0 + wn * 32 * NJ;
    // the lane's kSlots bias groups in one round trip (see the accumulator views)
    // — zeros without a bias, so that the add is unconditional; with few slots (16 x 16 accumulators) also widened to f32 once
    // (no unpacking per group), with more (32 x 32) kept packed: the wider cache would spill next to 128 live accumulators
    constexpr bool WIDE = ACC::kSlots <= 4;
    uint2 bc[WIDE ? 1 : ACC::kSlots];
    float bf[WIDE ? ACC::kSlots : 1][4];
    acc.cols([&](int slot, int c) {
      uint2 b = make_uint2(0u, 0u);
      if (P.bias) b = *reinterpret_cast<const uint2*>(P.bias + ncol0 + 4 * c);
      if constexpr (WIDE) {
        bf[slot][0] = bf16_to_f32((bf16_t)(b.x & 0xffff)), bf[slot][1] = bf16_to_f32((bf16_t)(b.x >> 16));
        bf[slot][2] = bf16_to_f32((bf16_t)(b.y & 0xffff)), bf[slot][3] = bf16_to_f32((bf16_t)(b.y >> 16));
      } else {
        bc[slot] = b;
      }
    });
    auto finish_s = [&](int slot, int n, float (&v)[4]) {
      scale(v);
      if constexpr (WIDE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf[slot][e];
      } else {
        add_bias(bc[slot], v);
      }
      activate(n, v);
    };
    if (!f32_out) {
      constexpr int RB = 64 * NJ;   // bytes per staged row (32*NJ bf16)
      constexpr int NS8 = 8 * NJ;   // 8-byte slots per row
      acc.template each<-1>([&](int r, int c, float (&v)[4], int slot) {
        finish_s(slot, ncol0 + 4 * c, v);
        *reinterpret_cast<uint2*>(cw + r * RB + ((c ^ (r & (NS8 - 1))) << 3)) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int LPR = RB / 16, RPI = 64 / LPR;  // lanes per row, rows per wave-instruction
      bf16_t* ob = reinterpret_cast<bf16_t*>(P.out);
#pragma unroll
      for (int it = 0; it < 128 / RPI; ++it) {
        const int r = it * RPI + lane / LPR, ch = lane % LPR;
        const int sp = ((2 * ch) ^ (r & (NS8 - 1))) & ~1;
        uint4 d = *reinterpret_cast<const uint4*>(cw + r * RB + (sp << 3));
        if (r & 1) d = make_uint4(d.z, d.w, d.x, d.y);  // odd rows hold the slot pair swapped
        const int m = m0 + wm * 128 + r;
        // non-temporal: the tile's 128 KiB of output would otherwise push operand panels of the patch out of the XCD's L2 (measured: -1.5 %
        // per bf16-out launch stand-alone, -3 % on the one-round ones; -0.3 ms per denoise step with the consumers' reads included)
        if (m < P.M) __builtin_nontemporal_store(*reinterpret_cast<const i32x4_nt*>(&d), reinterpret_cast<i32x4_nt*>(ob + (int64_t)m * P.ldo + ncol0 + ch * 8));
        if constexpr (NJ == 4) {
          if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the scheduler's hoisting (register pressure -> spills)
        }
      }
    } else {
      constexpr int RBF = 128 * NJ;  // bytes per staged row (32*NJ f32)
      constexpr int NS16 = 8 * NJ;   // 16-byte slots per row
      constexpr int LPR = RBF / 16, RPI = 64 / LPR;
      float* of = reinterpret_cast<float*>(P.out);
      // (explicit instantiation instead of a pragma: with 128-wide wave tiles the optimizer refuses to unroll a loop
      // this large, and a dynamically indexed accumulator array lives in scratch)
      auto f32_pass = [&](auto pass_tag) {
        constexpr int pass = decltype(pass_tag)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc.template each<pass>([&](int rg, int c, float (&v)[4], int slot) {
          finish_s(slot, ncol0 + 4 * c, v);
          const int r = rg - pass * 64;  // row inside this 64-row pass
          *reinterpret_cast<float4*>(cw + r * RBF + ((c ^ (r & (NS16 - 1))) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int NIT = 64 / RPI;
        const int ch = lane % LPR, n = ncol0 + ch * 4;
        if (epi == EPI_RESID_GATE_F32) {
          // residual read-modify-write: the loads of GRP rows are issued back to back, then consumed — one HBM / L2 round
          // trip per GRP rows instead of one per row (the compiler otherwise waits vmcnt(0) in front of every store: measured
          // ~16 us per tile, all of it latency)
          constexpr int GRP = NIT < 8 ? NIT : 8;
#pragma unroll
          for (int g0 = 0; g0 < NIT; g0 += GRP) {
            float4 xr[GRP], gg[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
              const int r = (g0 + u) * RPI + lane / LPR;
              const int m = min(m0 + wm * 128 + pass * 64 + r, P.M - 1);  // clamped, not predicated: no branch between the loads
              const float* gate = P.gate + (P.rows_per_batch > 0 ? (int64_t)(m / P.rows_per_batch) * P.gate_bstride : 0);
              gg[u] = *reinterpret_cast<const float4*>(gate + n);
              xr[u] = *reinterpret_cast<const float4*>(of + (int64_t)m * P.ldo + n);
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
              const int r = (g0 + u) * RPI + lane / LPR;
              const int m = m0 + wm * 128 + pass * 64 + r;
              const float4 v = *reinterpret_cast<const float4*>(cw + r * RBF + ((ch ^ (r & (NS16 - 1))) << 4));
              if (m < P.M) {
                float4 x = xr[u];
                x.x += gg[u].x * v.x;
                x.y += gg[u].y * v.y;
                x.z += gg[u].z * v.z;
                x.w += gg[u].w * v.w;
                *reinterpret_cast<float4*>(of + (int64_t)m * P.ldo + n) = x;
              }
            }
          }
        } else {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int r = it * RPI + lane / LPR;
            const float4 v = *reinterpret_cast<const float4*>(cw + r * RBF + ((ch ^ (r & (NS16 - 1))) << 4));
            const int m = m0 + wm * 128 + pass * 64 + r;
            if (m < P.M) *reinterpret_cast<float4*>(of + (int64_t)m * P.ldo + n) = v;
          }
        }
      };
      f32_pass(std::integral_constant<int, 0>{});
      f32_pass(std::integral_constant<int, 1>{});
    }
    return;
  }
  // ---- direct path (ragged N tile, unaligned output, bf16 residual add)
  acc.template each<-1>([&](int r, int c, float (&v)[4], int) {
    const int m = m0 + wm * 128 + r;
    const int n = n0 + wn * 32 * NJ + 4 * c;
    if (m >= P.M || n >= P.N) return;
    const float* gate = P.gate;
    if (epi == EPI_RESID_GATE_F32 && P.rows_per_batch > 0) gate += (int64_t)(m / P.rows_per_batch) * P.gate_bstride;
    const bool full = (n + 3 < P.N) && (P.bias == nullptr || (reinterpret_cast<uintptr_t>(P.bias + n) & 7) == 0);
    finish(n, v, full);
    if (epi == EPI_RESID_GATE_F32) {
      float* o = reinterpret_cast<float*>(P.out) + (int64_t)m * P.ldo + n;
      for (int e = 0; e < 4 && n + e < P.N; ++e) o[e] += gate[n + e] * v[e];
    } else if (epi == EPI_STORE_F32) {
      float* o = reinterpret_cast<float*>(P.out) + (int64_t)m * P.ldo + n;
      for (int e = 0; e < 4 && n + e < P.N; ++e) o[e] = v[e];
    } else {
      bf16_t* o = reinterpret_cast<bf16_t*>(P.out) + (int64_t)m * P.ldo + n;
      if (epi == EPI_RESID_ADD_BF16) {
        const bf16_t* rs = reinterpret_cast<const bf16_t*>(P.resid) + (int64_t)m * P.ldo + n;
        for (int e = 0; e < 4 && n + e < P.N; ++e) v[e] += bf16_to_f32(rs[e]);
      }
      for (int e = 0; e < 4 && n + e < P.N; ++e) o[e] = f32_to_bf16(v[e]);
    }
  });
}

template <int NJ, int WN = 4, int ACT = -1, class ACC>
__device__ __forceinline__ void gemm_epilogue(const GemmProblem& P, const ACC& acc, char* smem, int m0, int n0, int wave, int lane) {
  if constexpr (NJ == 2 && WN == 4) {
    if (P.qk_qh != nullptr && n0 < 3 * P.qk_D) {
      qkv_relayout_epilogue(P, acc, smem, m0, n0, wave, lane);
      return;
    }
  }
  gemm_epilogue_impl<NJ, WN, ACT>(P, acc, smem, m0, n0, wave, lane);
}
// the activation kind a launch group needs; all problems of a group must agree (-1 otherwise: the per-kind kernel
// instantiations apply ONE activation, a mixed group would silently lose a GELU — launch_gemm rejects it)
inline int epilogue_kind(const GemmProblem* p, int n) {
  auto kind = [](int epi) { return (epi == EPI_STORE_BF16 || epi == EPI_RESID_GATE_F32) ? 0 : epi == EPI_GELU_BF16 ? 1 : epi == EPI_GELU_FROM_COL ? 2 : 3; };
  const int k = kind(p[0].epi);
  for (int i = 1; i < n; ++i)
    if (kind(p[i].epi) != k) return -1;
  return k;
}

// MODE 0: dense GEMM, 1: 4-bit weights, 2: implicit-GEMM convolution (NHWC)
template <int MODE, int NJ>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_kernel(const GemmBatch batch) {
  constexpr int BN = 128 * NJ;
  constexpr int W_TILE_BYTES = BN * BK * 2;
  constexpr int BUF_BYTES = A_TILE_BYTES + W_TILE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- which problem / tile
  const int total = batch.tile_start[batch.nprob];
  const int lid = xcd_remap(blockIdx.x, total);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_PROBLEMS; ++i)
    if (i < batch.nprob && lid >= batch.tile_start[i]) pi = i;
  const GemmProblem& P = batch.p[pi];
  const int t = lid - batch.tile_start[pi];
  const int tiles_m = (P.M + BM - 1) / BM;
  const int tiles_n = (P.N + BN - 1) / BN;
  int tm, tn;
  tile_coords(t, tiles_m, tiles_n, batch.band[pi], tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = P.K / BK;

  auto bufA = [&](int b) -> char* { return smem + b * BUF_BYTES; };
  auto bufW = [&](int b) -> char* { return smem + b * BUF_BYTES + A_TILE_BYTES; };

  // v_mfma_f32_16x16x32_bf16 (see gemm_pp_kernel): the wave's 128 x 32 NJ is 8 x 2 NJ accumulators of 16 x 16
  f32x4 acc[8][2 * NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // per-lane swizzled column-slot byte offsets for the 2 k-steps (32 wide) of a tile: slot 4 s + (lane >> 4) of row lane & 15
  const int sw = ((lane & 15) >> 1) & 7;
  int koff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) koff[s] = ((s * 4 + (lane >> 4)) ^ sw) << 4;
  const int a_row_off = (wm * 128 + (lane & 15)) * 128;
  const int w_row_off = (wn * 32 * NJ + (lane & 15)) * 128;

  // CONV: this lane stages rows chunk*8 + (lane>>3), chunk = wave*4 + i; precompute their pixels.
  // cv_up > 0: nearest-2x upsample folded in; cv_up < 0: the VAE encoder's Downsample (stride 2,
  // one zero column / row on the right / bottom only): input pixel = 2*out + tap, no centring
  int cv_y[4], cv_x[4], cv_pix[4];
  const int cv_sh = MODE == 2 && P.cv_up > 0 ? P.cv_up : 0;        // shift of the (virtually upsampled) input coordinates
  const int cv_half = MODE == 2 && P.cv_up >= 0 ? P.cv_ks >> 1 : 0;
  const int cv_ow = P.cv_w << cv_sh, cv_oh = P.cv_h << cv_sh;        // bounds of the (virtually upsampled) input
  const uint32_t cv_cin2 = (uint32_t)P.cv_cin * 2;                  // bytes per pixel
  if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + (wave * 4 + i) * 8 + (lane >> 3);
      m = m > P.M - 1 ? P.M - 1 : m;
      const bool down = P.cv_up < 0;
      const int ow = down ? P.cv_w >> 1 : cv_ow, oh = down ? P.cv_h >> 1 : cv_oh;
      const int x = m % ow, y = (m / ow) % oh, b = m / (ow * oh);
      cv_x[i] = down ? 2 * x : x;
      cv_y[i] = down ? 2 * y : y;
      cv_pix[i] = b * P.cv_h * P.cv_w;  // pixels, not elements: < 2^31 at any batch the pipeline runs
    }
  }
  // the 16-byte slot of a row this lane stages: (lane & 7) ^ ((row >> 1) & 7), row = (wave * 4 + i) * 8 + (lane >> 3) -> depends on i & 1 only
  const int cv_slot[2] = {((lane & 7) ^ ((lane >> 4) & 7)) << 4, ((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4};
  // source of piece i of the K tile (tap offset dy, dx; 64-channel slice at c0): one v_mad_u64_u32 per row, the zero line for padding taps
  auto conv_src = [&](int i, int dy, int dx, int c0) -> const char* {
    const int yy = cv_y[i] + dy, xx = cv_x[i] + dx;
    const bool in = (unsigned)yy < (unsigned)cv_oh && (unsigned)xx < (unsigned)cv_ow;
    const uint32_t pix = (uint32_t)(cv_pix[i] + (yy >> cv_sh) * P.cv_w + (xx >> cv_sh));
    const char* src = reinterpret_cast<const char*>(P.A) + ((uint64_t)pix * cv_cin2 + (uint32_t)(c0 * 2 + cv_slot[i & 1]));
    const char* zero = reinterpret_cast<const char*>(P.cv_zero) + cv_slot[i & 1];  // zero padding (conv2d pad = k/2)
    return in ? src : zero;
  };
  // uniform walk over (tap, slice), kept one K tile AHEAD of the tile being multiplied (the DMA pieces of tile kt + 1 are issued among
  // the MFMAs of tile kt, as in the dense kernel)
  int cvn_c0 = 0, cvn_dy = -cv_half, cvn_dx = -cv_half;
  auto conv_advance = [&]() {
    cvn_c0 += 64;
    if (cvn_c0 == P.cv_cin) {
      cvn_c0 = 0;
      if (++cvn_dx == P.cv_ks - cv_half) cvn_dx = -cv_half, ++cvn_dy;
    }
  };
  // per-lane DMA source pointers, hoisted out of the K loop: this lane stages tile rows
  // chunk*8 + (lane>>3) (1-KiB chunk = 8 rows), 16-B slot (lane&7) ^ ((row>>1)&7) of each row
  constexpr int CPWN = BN / 64;
  const bf16_t* a_src[4];
  const bf16_t* w_src[CPWN];
  if (MODE != 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + (lane >> 3);
      const int g = min(m0 + row, P.M - 1);
      a_src[i] = P.A + (int64_t)g * P.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
  }
  if (MODE != 1) {
#pragma unroll
    for (int i = 0; i < CPWN; ++i) {
      const int row = (wave * CPWN + i) * 8 + (lane >> 3);
      const int g = min(n0 + row, P.N - 1);
      w_src[i] = P.W + (int64_t)g * P.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
  }
  auto stage_a = [&](int kt, char* dst) {  // (CONV: tile 0 only — the walk's state is tile 0's)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 2)
        __builtin_amdgcn_global_load_lds((glb_void*)conv_src(i, cvn_dy, cvn_dx, cvn_c0), (lds_void*)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_void*)(a_src[i] + kt * BK), (lds_void*)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
    }
  };
  auto stage_w = [&](int kt, char* dst) {
    if (MODE == 1) {
      stage_tile_q4<BN>(P, n0, kt * BK, dst, tid);
    } else {
#pragma unroll
      for (int i = 0; i < CPWN; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(w_src[i] + kt * BK), (lds_void*)(dst + (wave * CPWN + i) * 1024), 16, 0, 0);
    }
  };

  stage_a(0, bufA(0));
  stage_w(0, bufW(0));

  // One 1-KiB piece (8 rows) of tile `kt` into buffer `buf`: pieces 0..3 are A, the rest W.
  auto dma_piece = [&](int kt, int d, int buf) {
    if (d < 4 && MODE == 2)
      __builtin_amdgcn_global_load_lds((glb_void*)conv_src(d, cvn_dy, cvn_dx, cvn_c0), (lds_void*)(bufA(buf) + (wave * 4 + d) * 1024), 16, 0, 0);
    else if (d < 4)
      __builtin_amdgcn_global_load_lds((glb_void*)(a_src[d] + kt * BK), (lds_void*)(bufA(buf) + (wave * 4 + d) * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_void*)(w_src[d - 4] + kt * BK), (lds_void*)(bufW(buf) + (wave * CPWN + d - 4) * 1024), 16, 0, 0);
  };
  // Dense GEMM: the DMA pieces of tile kt+1 are issued one at a time BETWEEN the MFMAs of the
  // first two k-steps of tile kt (an LDS-DMA costs ~60 cycles among MFMAs but 100-185 in a burst,
  // and a burst leaves the matrix pipe of the SIMD idle because both of its waves burst together);
  // the last two k-steps give the pieces time to land before the next barrier.
  constexpr bool INTERLEAVE = (MODE == 0 || MODE == 2);
  auto ktile = [&](int kt, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int cur = kt & 1;
    dma_barrier();  // tile kt landed for every wave; everyone finished reading buf[cur^1]
    if (MODE == 2 && MORE) conv_advance();
    if (MORE && !INTERLEAVE) {
      stage_a(kt + 1, bufA(cur ^ 1));
      stage_w(kt + 1, bufW(cur ^ 1));
    }
    const char* la = bufA(cur) + a_row_off;
    const char* lw = bufW(cur) + w_row_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t xf[8], wf[2 * NJ];
#pragma unroll
      for (int j = 0; j < 2 * NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(lw + j * 16 * 128 + koff[s]);
#pragma unroll
      for (int i = 0; i < 8; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(la + i * 16 * 128 + koff[s]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 2 * NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        if (MORE && INTERLEAVE && s * 8 + i < 4 + CPWN) dma_piece(kt + 1, s * 8 + i, cur ^ 1);
      }
    }
  };
  for (int kt = 0; kt < nk - 1; ++kt) ktile(kt, std::true_type{});
  ktile(nk - 1, std::false_type{});

  gemm_epilogue<NJ>(P, Acc16<NJ>{acc, lane}, smem, m0, n0, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// Dense 256x256x64 kernel: ping-pong wave groups, register-resident fragments, 5-tile LDS ring.
//
// Why (ablations on the double-buffered kernel above and on this one, profiles/ + DESIGN.md):
//   * an LDS-DMA piece needs ~1.1-1.3 us from issue to landed under load, but double buffering
//     gives a tile only one K-tile period of lead: every K tile ends in a wait for data;
//   * a CU gets at most one 1-KiB global_load_lds through every ~45 clocks (64 KiB per 1.2 us: the L2 /
//     Infinity-Cache delivery rate with 256 CUs pulling, tools/load_rate.hip) and a
//     wave that issues one while that queue is full stalls IN ORDER — the MFMAs behind it wait too.
// So: (1) all 160 KiB of LDS hold 5 operand tiles (A ring of 2, W ring of 3) and a tile's
// fragments are copied to registers in one burst, so its slot is recycled after a fraction of a
// period and every DMA piece has 3 slots (1.5 periods) to land; (2) the wave that issues DMA is
// never the wave that feeds the matrix pipe.
//
// The 8 waves form two groups (waves 0-3 = rows 0..127, waves 4-7 = rows 128..255; one wave of
// each group per SIMD).  Time is cut into barrier-delimited slots; in every slot one group LOADs
// (24 ds_read_b128 per wave: its fragments of tile t -> VGPRs, then its 8 DMA pieces, then waits
// for the pieces it issued one period ago) while the other COMPUTEs (32 back-to-back MFMAs out of
// registers, nothing else).  Group 1 runs one slot behind group 0:
//     slot      2t        2t+1        2t+2        2t+3
//     group 0   LOAD(t)   COMPUTE(t)  LOAD(t+1)   COMPUTE(t+1)
//     group 1   COMP(t-1) LOAD(t)     COMPUTE(t)  LOAD(t+1)
// DMA issued by a wave of group g during LOAD(t) (its 1-KiB chunks wave*4 + i, i < 4):
//     A rows of the OTHER group, tile t+1+g -> A slot (t+1+g)&1   (last read one slot earlier)
//     W rows (its half),         tile t+2   -> W slot (t+2)%3     (last read by group 1 in slot 2t-1)
// `s_waitcnt vmcnt(8)` at the end of LOAD(t) retires what the wave issued in LOAD(t-1); the
// closing barrier publishes it to the group that LOADs next.  Barriers are raw s_barrier +
// lgkmcnt(0): a __syncthreads() would make hipcc drain vmcnt as well and collapse the pipeline,
// and sched_barrier(0) keeps the MFMAs (not memory operations) from being hoisted across slots.
// Accumulation order per output element is identical to gemm_bf16_kernel: bit-identical results.
//
// FP8 = true: the same pipeline on OCP e4m3 operands (GemmProblem::fp8).  A 128-byte tile row holds 128 k
// instead of 64, so the LDS image, the swizzle, the DMA pieces and the 24 ds_read_b128 per LOAD are
// unchanged; two 16-byte fragments are paired into the 32-byte operand of v_mfma_f32_32x32x64_f8f6f4
// (16 per COMPUTE instead of 32 of the bf16 instruction, the same matrix-pipe time for twice the k).
// Which k a (lane, byte) of the operand carries is irrelevant as long as A and W use the same map,
// which they do: both tiles have the same LDS layout and are read with the same offsets.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) int i32x16_t;
template <int Q8, int ACT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_pp_kernel(const GemmBatch batch) {
  constexpr bool FP8 = Q8 != 0;  // 8-bit operands: 1 = OCP e4m3, 2 = int8 (GemmProblem::fp8)
  constexpr bool I8 = Q8 == 2;
  constexpr int NJ = 2, BN = 256;
  constexpr int ES = FP8 ? 1 : 2;  // operand element size
  constexpr int A_RING = 0, W_RING = 2 * A_TILE_BYTES, TILE = A_TILE_BYTES;  // 2 x 32 KiB + 3 x 32 KiB
  __shared__ __attribute__((aligned(16))) char smem[5 * TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wn = wave & 3;  // group = row half (wm)

  const int total = batch.tile_start[batch.nprob];
  const int lid = xcd_remap(blockIdx.x, total);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_PROBLEMS; ++i)
    if (i < batch.nprob && lid >= batch.tile_start[i]) pi = i;
  const GemmProblem& P = batch.p[pi];
  const int t_in = lid - batch.tile_start[pi];
  const int tiles_m = (P.M + BM - 1) / BM;
  const int tiles_n = (P.N + BN - 1) / BN;
  int tm, tn;  // same band / patch order as gemm_bf16_kernel
  tile_coords(t_in, tiles_m, tiles_n, batch.band[pi], tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = P.K * ES / (BK * 2);  // 128-byte K tiles

  // fp8: v_mfma_f32_32x32x64_f8f6f4, accumulators acc[4][NJ] of 32 x 32 (Acc32).  bf16: v_mfma_f32_16x16x32_bf16, accumulators
  // acc16[8][2 NJ] of 16 x 16 (Acc16) — on this power-capped part the 16 x 16 x 32 form sustains 14 % more than 32 x 32 x 16
  // (tools/mfma_peak); same LDS image, the fragment of a 16-row block is rows lane & 15 at k slot 4 s + (lane >> 4).
  // int8: v_mfma_i32_32x32x32_i8, two per 32-byte fragment pair (each 16-byte fragment is one instruction's 32-k operand), accumulators
  // acci of the same shape and lane layout as acc; the sums are exact integers, converted once behind the K loop.
  f32x16 acc[FP8 ? 4 : 1][NJ];
  i32x16_t acci[I8 ? 4 : 1][NJ];
  f32x4 acc16[FP8 ? 1 : 8][2 * NJ];
  if constexpr (I8) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acci[i][j][r] = 0;
  } else if constexpr (FP8) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  }

  const int rl = FP8 ? (lane & 31) : (lane & 15);  // fragment row inside its 32- / 16-row block
  const int sw = (rl >> 1) & 7;
  int koff[4];  // fp8: 16-byte k slots 2 s + (lane >> 5), s = 0..3; bf16: slots 4 s + (lane >> 4), s = 0..1
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = FP8 ? ((s * 2 + (lane >> 5)) ^ sw) << 4 : (((s & 1) * 4 + (lane >> 4)) ^ sw) << 4;
  const int a_row_off = (g * 128 + rl) * 128;
  const int w_row_off = (wn * 64 + rl) * 128;

  // chunks (8 rows, 1 KiB) this wave stages: A chunks of the other group's rows, W chunks wave*4+i
  const int a_chunk0 = (wave ^ 4) * 4, w_chunk0 = wave * 4;
  // per-lane BYTE offsets (32-bit: the operands are < 4 GiB) from the uniform tile base, so the DMA
  // uses the saddr + voffset form: 8 VGPRs instead of 16 for pointers (the kernel sits at the
  // 256-register limit and a spilled pointer costs a vmcnt(0) reload — a full pipeline drain)
  uint32_t a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = (a_chunk0 + i) * 8 + (lane >> 3), rw = (w_chunk0 + i) * 8 + (lane >> 3);
    // relative to the tile's first row (the uniform base below carries m0 / n0 in 64 bits): an operand may exceed 4 GiB — the fused
    // modulation matrix of FLUX.1 is 6.5 GB — but a tile's 256 rows never do
    a_off[i] = (uint32_t)((int64_t)(min(m0 + ra, P.M - 1) - m0) * P.lda * ES + (((lane & 7) ^ ((ra >> 1) & 7)) << 4));
    w_off[i] = (uint32_t)((int64_t)(min(n0 + rw, P.N - 1) - n0) * P.ldw * ES + (((lane & 7) ^ ((rw >> 1) & 7)) << 4));
  }
  const char* const a_base = reinterpret_cast<const char*>(P.A) + (int64_t)m0 * P.lda * ES;
  const char* const w_base = reinterpret_cast<const char*>(P.W) + (int64_t)n0 * P.ldw * ES;
  auto dma_a = [&](int kt, int i) {
    const char* base = a_base + (int64_t)kt * (BK * 2);  // uniform
    __builtin_amdgcn_global_load_lds((glb_void*)(base + a_off[i]), (lds_void*)(smem + A_RING + (kt & 1) * TILE + (a_chunk0 + i) * 1024), 16, 0, 0);
  };
  auto dma_w = [&](int kt, int slot, int i) {
    const char* base = w_base + (int64_t)kt * (BK * 2);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + w_off[i]), (lds_void*)(smem + W_RING + slot * TILE + (w_chunk0 + i) * 1024), 16, 0, 0);
  };
  auto slot_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS reads are done: the slots I read may be refilled
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: what the steady-state rule would have issued before LOAD(0): A(0) (both
  // halves), W(0), W(1), and the rows group 0 reads of A(1) (staged by group 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_a(0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_w(0, 0, i);
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_w(1, 1, i);
    if (g == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_a(1, i);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  slot_barrier();
  if (g == 1) slot_barrier();  // group 1 starts one slot late

  bf16x8_t xf[2][8], wf[2][2 * NJ];  // bf16: fragments of the 2 k-steps of 32: 8 row blocks of A, 2 NJ of W (16 rows each)
  i32x8_t xq[2][4], wq[2][NJ];    // fp8: fragments of the 2 k-steps (two 16-byte reads each)
  int wr = 0;  // W slot of tile t = t % 3
  int wi = 2;  // W slot of tile t + 2
  auto ktile = [&](int t, auto main_tag) {
    constexpr bool MAIN = decltype(main_tag)::value;  // steady state: both issues are in range
    // ---- LOAD(t): fragments -> registers, then this wave's DMA pieces (issuing them first was
    // measured 8-10 % slower: the ds_reads queue up behind DMA instructions blocked on a full queue)
    const bool a_ok = MAIN || (t + 1 + g < nk);
    const bool w_ok = MAIN || (t + 2 < nk);
    auto issue_dma = [&]() {
      if (a_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(t + 1 + g, i);
      }
      if (w_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_w(t + 2, wi, i);
      }
    };
    {
      const char* la = smem + A_RING + (t & 1) * TILE + a_row_off;
      const char* lw = smem + W_RING + wr * TILE + w_row_off;
      if constexpr (FP8) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            wq[s][j] = __builtin_shufflevector(*reinterpret_cast<const i32x4_t*>(lw + j * 32 * 128 + koff[2 * s]),
                                               *reinterpret_cast<const i32x4_t*>(lw + j * 32 * 128 + koff[2 * s + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            xq[s][i] = __builtin_shufflevector(*reinterpret_cast<const i32x4_t*>(la + i * 32 * 128 + koff[2 * s]),
                                               *reinterpret_cast<const i32x4_t*>(la + i * 32 * 128 + koff[2 * s + 1]), 0, 1, 2, 3, 4, 5, 6, 7);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int j = 0; j < 2 * NJ; ++j) wf[s][j] = *reinterpret_cast<const bf16x8_t*>(lw + j * 16 * 128 + koff[s]);
#pragma unroll
          for (int i = 0; i < 8; ++i) xf[s][i] = *reinterpret_cast<const bf16x8_t*>(la + i * 16 * 128 + koff[s]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_dma();
    __builtin_amdgcn_sched_barrier(0);
    if (MAIN) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {  // tail: wait for everything but what was issued just now
      if (a_ok && w_ok)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (a_ok || w_ok)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    slot_barrier();
    // ---- COMPUTE(t): the matrix pipe only
    if constexpr (I8) {
      // two instructions per 32-byte fragment pair, the halves in separate sweeps over the 8 accumulators (no instruction follows one on its own accumulator)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acci[i][j] = h == 0 ? __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_shufflevector(wq[s][j], wq[s][j], 0, 1, 2, 3),
                                                                         __builtin_shufflevector(xq[s][i], xq[s][i], 0, 1, 2, 3), acci[i][j], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_shufflevector(wq[s][j], wq[s][j], 4, 5, 6, 7),
                                                                         __builtin_shufflevector(xq[s][i], xq[s][i], 4, 5, 6, 7), acci[i][j], 0, 0, 0);
    } else if constexpr (FP8) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq[s][j], xq[s][i], acc[i][j], 0, 0, 0, 0, 0, 0);  // e4m3 x e4m3, unscaled
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 2 * NJ; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], xf[s][i], acc16[i][j], 0, 0, 0);
    }
    slot_barrier();
    wr = wr == 2 ? 0 : wr + 1;
    wi = wi == 2 ? 0 : wi + 1;
  };
  const int nmain = nk > 2 ? nk - 2 : 0;  // t + 2 < nk  (and t + 1 + g < nk)
  for (int t = 0; t < nmain; ++t) ktile(t, std::true_type{});
  for (int t = nmain; t < nk; ++t) ktile(t, std::false_type{});
  if (g == 0) slot_barrier();  // group 0 finished one slot early

  if constexpr (I8) {  // the exact integer sums as f32 (v_cvt_f32_i32: round to nearest even beyond 2^24, as the oracle's (float) cast)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)acci[i][j][r];
  }
  if constexpr (FP8) {  // dequantise: per-token scale of the row, per-channel scale of the 4 consecutive columns
    float sa[4], oa[4];
    const bool off = I8 && P.a_off != nullptr;  // int8, post-GELU form of the row codes: + a_off[m] * w_sum[n] (uniform per problem)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(m0 + g * 128 + i * 32 + (lane & 31), P.M - 1);
      sa[i] = P.a_scale[r];
      oa[i] = off ? P.a_off[r] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
        float sn[4], wsn[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sn[c] = P.w_scale[min(n + c, P.N - 1)];
          wsn[c] = off ? P.w_sum[min(n + c, P.N - 1)] : 0.f;
        }
        if (off) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][j][4 * q + c] = acc[i][j][4 * q + c] * (sa[i] * sn[c]) + oa[i] * wsn[c];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][j][4 * q + c] *= sa[i] * sn[c];
        }
      }
  }
  // the lane id is laundered (as in w4_epilogue): what the epilogue derives from it is then computed here, not hoisted above the
  // K loop, held across 256 registers of loop state, spilled and reloaded (a scratch reload waits vmcnt(0): the stores serialise)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  if constexpr (FP8) gemm_epilogue<NJ, 4, ACT>(P, Acc32<NJ>{acc, lane_e}, smem, m0, n0, wave, lane_e);
  else gemm_epilogue<NJ, 4, ACT>(P, Acc16<NJ>{acc16, lane_e}, smem, m0, n0, wave, lane_e);
}


// ---------------------------------------------------------------------------------------------
// Dense 256x256x64 kernel, 4 waves (one per SIMD), 128 x 128 of C per wave in 256 accumulator registers.
//
// Why: the 8-wave kernels give each wave 128 x 64 of C, so per K tile the CU reads (128 + 64) * 64 * 2 B * 8
// = 192 KiB of fragments out of LDS on top of the 64 KiB the DMA writes into it: 256 KiB at the LDS's
// 128 B/clk = 2048 clocks, the same as the 2048 MFMA clocks of the tile — the LDS port, not the matrix pipe
// or the memory side, is what the ping-pong kernel's 2830 clocks per K tile run into.  A 128 x 128 wave tile
// reads (128 + 128) * 64 * 2 B * 4 = 128 KiB: 25 % less LDS traffic for the same FLOPs.  It needs 256
// accumulator registers per wave, i.e. the 512-register budget of ONE wave per SIMD, so there is no second
// wave to hide latencies behind: everything is software-pipelined inside the wave —
//   * fragments of k-step s+1 are read (inline-asm ds_read_b128, one per MFMA) while the 16 MFMAs of step s
//     issue; two fragment buffers (64 VGPRs);
//   * one barrier per K tile, between the MFMA blocks of steps 2 and 3, after this tile's last fragment read:
//     it frees the tile's LDS slots and publishes tile t+1; the 16 DMA pieces of A(t+2) / W(t+3) and the
//     step-0 reads of tile t+1 are then issued one per MFMA of step 3.
// LDS: A ring of 2 + W ring of 3 tiles of 32 KiB (same image, swizzle and DMA pieces as the other kernels).
// Accumulation order per output element is that of the other kernels: bit-identical results.
constexpr int W4_THREADS = 256;
// Tail of the one-wave-per-SIMD kernels (gemm_w4_kernel, gemm_w4q_kernel): wave (wm, wn) of the 2 x 2 layout holds 128 x 128.
// It leaves as two 128 x 64 halves through the 8-wave epilogue: half r plays wave (wm, 2*wn + r) of the 2 x 4 layout (same
// staging regions, same stores, same arithmetic).  Real loops, not two inlined copies: the unrolled epilogue is large, and
// twice that code (or its 128-wide instantiation) runs out of the instruction cache and, with the accumulators filling the
// AGPRs, spills (measured 3-4x slower).
// The lane id is laundered: everything the epilogue derives from it is then computed after the K loop instead of being
// hoisted above it, kept live across 256 arch VGPRs of loop state, spilled, and reloaded (each reload = vmcnt(0) = the
// wave's stores serialised: measured 30 us per tile instead of 12).
template <int ACT>
__device__ __forceinline__ void w4_epilogue(const GemmProblem& P, f32x16 (&acc)[4][4], char* smem, int m0, int n0, int wave, int wm, int wn, int lane) {
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  if (P.qk_qh != nullptr && n0 < 3 * P.qk_D) {
    // fused q|k|v relayout: both halves staged into the one LDS image, then this wave emits row groups 2*wave, 2*wave + 1
    __syncthreads();
    {
      f32x16 hacc[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) hacc[i][0] = acc[i][0], hacc[i][1] = acc[i][1];
#pragma clang loop unroll(disable)
      for (int r = 0; r < 2; ++r) {
        qkv_relayout_stage(P, Acc32<2>{hacc, lane_e}, smem, n0, wm * 4 + wn * 2 + r, lane_e);
#pragma unroll
        for (int i = 0; i < 4; ++i) hacc[i][0] = acc[i][2], hacc[i][1] = acc[i][3];
        asm volatile("" : "+v"(lane_e));
      }
    }
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int r = 0; r < 2; ++r) {
      qkv_relayout_emit(P, smem, m0, n0, wave * 2 + r, lane_e);
      asm volatile("" : "+v"(lane_e));
    }
    return;
  }
  f32x16 hacc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) hacc[i][0] = acc[i][0], hacc[i][1] = acc[i][1];
#pragma clang loop unroll(disable)
  for (int r = 0; r < 2; ++r) {
    gemm_epilogue_impl<2, 4, ACT>(P, Acc32<2>{hacc, lane_e}, smem, m0, n0, wm * 4 + wn * 2 + r, lane_e);
#pragma unroll
    for (int i = 0; i < 4; ++i) hacc[i][0] = acc[i][2], hacc[i][1] = acc[i][3];
    asm volatile("" : "+v"(lane_e));  // keep the second round's address math out of the first
  }
}

#if FMI_ALT_KERNELS  // the dense 4-wave kernel (FMI_GEMM_W4=1: off by default since round 2) lives in the test build; its epilogue tail above serves gemm_w4q_kernel
// The same tail for 16 x 16 accumulators (gemm_w4_kernel): acc[tm][tn], the half r = column tiles 4 r .. 4 r + 3.
template <int ACT>
__device__ __forceinline__ void w4_epilogue16(const GemmProblem& P, f32x4 (&acc)[8][8], char* smem, int m0, int n0, int wave, int wm, int wn, int lane) {
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  f32x4 hacc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) hacc[i][j] = acc[i][j];
  if (P.qk_qh != nullptr && n0 < 3 * P.qk_D) {
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int r = 0; r < 2; ++r) {
      qkv_relayout_stage(P, Acc16<2>{hacc, lane_e}, smem, n0, wm * 4 + wn * 2 + r, lane_e);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) hacc[i][j] = acc[i][4 + j];
      asm volatile("" : "+v"(lane_e));
    }
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int r = 0; r < 2; ++r) {
      qkv_relayout_emit(P, smem, m0, n0, wave * 2 + r, lane_e);
      asm volatile("" : "+v"(lane_e));
    }
    return;
  }
#pragma clang loop unroll(disable)
  for (int r = 0; r < 2; ++r) {
    gemm_epilogue_impl<2, 4, ACT>(P, Acc16<2>{hacc, lane_e}, smem, m0, n0, wm * 4 + wn * 2 + r, lane_e);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) hacc[i][j] = acc[i][4 + j];
    asm volatile("" : "+v"(lane_e));  // keep the second round's address math out of the first
  }
}

template <bool FP8, int ACT>
__global__ __launch_bounds__(W4_THREADS, 1) void gemm_w4_kernel(const GemmBatch batch) {
  constexpr int NJ = 4, BN = 256;
  constexpr int A_RING = 0, W_RING = 2 * A_TILE_BYTES, TILE = A_TILE_BYTES;
  constexpr int ES = FP8 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) char smem[5 * TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int total = batch.tile_start[batch.nprob];
  const int lid = xcd_remap(blockIdx.x, total);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_PROBLEMS; ++i)
    if (i < batch.nprob && lid >= batch.tile_start[i]) pi = i;
  const GemmProblem& P = batch.p[pi];
  const int t_in = lid - batch.tile_start[pi];
  const int tiles_m = (P.M + BM - 1) / BM;
  const int tiles_n = (P.N + BN - 1) / BN;
  int tm, tn;
  tile_coords(t_in, tiles_m, tiles_n, batch.band[pi], tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = P.K * ES / (BK * 2);

  // v_mfma_f32_16x16x32_bf16: the wave's 128 x 128 is 8 x 8 accumulators of 16 x 16 (the 16 x 16 x 32 form sustains 14 % more than
  // 32 x 32 x 16 on this power-capped part, tools/mfma_peak; same results bit for bit, same LDS image)
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  const int sw = ((lane & 15) >> 1) & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  uint32_t koff[2];  // k-step s of 32: 16-byte slot 4 s + (lane >> 4) of the row
#pragma unroll
  for (int s = 0; s < 2; ++s) koff[s] = ((s * 4 + (lane >> 4)) ^ sw) << 4;
  const uint32_t a_row = lds0 + A_RING + (wm * 128 + (lane & 15)) * 128;
  const uint32_t w_row = lds0 + W_RING + (wn * 128 + (lane & 15)) * 128;

  // DMA pieces of this wave: 1-KiB chunks wave*8 + i of the A tile and of the W tile
  uint32_t a_off[8], w_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (wave * 8 + i) * 8 + (lane >> 3);
    a_off[i] = (uint32_t)((int64_t)(min(m0 + r, P.M - 1) - m0) * P.lda * ES + (((lane & 7) ^ ((r >> 1) & 7)) << 4));  // tile-relative, see gemm_pp_kernel
    w_off[i] = (uint32_t)((int64_t)(min(n0 + r, P.N - 1) - n0) * P.ldw * ES + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
  }
  const char* const a_base = reinterpret_cast<const char*>(P.A) + (int64_t)m0 * P.lda * ES;
  const char* const w_base = reinterpret_cast<const char*>(P.W) + (int64_t)n0 * P.ldw * ES;
  // LDS-DMA in the scalar-base form (SGPR pair + 32-bit lane offset), written as asm: from the builtin hipcc forms a 64-bit
  // per-lane address with a v_lshl_add_u64 in front of every piece inside this loop (+1-2 % on the K loop, tools/gemm_bench).
  // m0 = LDS destination of the 1-KiB piece; one wait state between the m0 write and the load.
  auto dma_a = [&](int kt, int i) {
    const char* base = a_base + (int64_t)kt * (BK * 2);
    const uint32_t l = lds0 + A_RING + (kt & 1) * TILE + (wave * 8 + i) * 1024;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(a_off[i]), "s"(base), "s"(l) : "memory");
  };
  auto dma_w = [&](int kt, int slot, int i) {
    const char* base = w_base + (int64_t)kt * (BK * 2);
    const uint32_t l = lds0 + W_RING + slot * TILE + (wave * 8 + i) * 1024;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(w_off[i]), "s"(base), "s"(l) : "memory");
  };
  auto sync_all = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue, in the order the steady-state vmcnt arithmetic expects: A(0), W(0), W(1), A(1)
  const int klast = nk - 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_a(0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_w(0, 0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_w(min(1, klast), 1, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_a(min(1, klast), i);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  sync_all();

  typedef __attribute__((ext_vector_type(4))) int frag_t;  // 16 bytes: 8 bf16 of one row
  frag_t xf[2][8], wf[2][8];  // fragment buffer s holds k-step s of a tile: 8 row blocks of A, 8 of W (16 rows each)
  // LDS read addresses: one VGPR per (operand, k-step), rebased once per tile; the row block is the immediate offset (2 KiB apart)
  uint32_t a_ad[2], w_ad[2];
  auto rebase = [&](int aslot, int wslot_) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a_ad[s] = a_row + koff[s] + aslot * TILE;
      w_ad[s] = w_row + koff[s] + wslot_ * TILE;
    }
  };
#define FMI_W4_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto read_frag = [&](int s, int which) {  // k-step s (= buffer s); which: 0..7 = A row block, 8..15 = W row block
    switch (which) {
      case 0: FMI_W4_RD(xf[s][0], a_ad[s], 0); break;
      case 1: FMI_W4_RD(xf[s][1], a_ad[s], 2048); break;
      case 2: FMI_W4_RD(xf[s][2], a_ad[s], 4096); break;
      case 3: FMI_W4_RD(xf[s][3], a_ad[s], 6144); break;
      case 4: FMI_W4_RD(xf[s][4], a_ad[s], 8192); break;
      case 5: FMI_W4_RD(xf[s][5], a_ad[s], 10240); break;
      case 6: FMI_W4_RD(xf[s][6], a_ad[s], 12288); break;
      case 7: FMI_W4_RD(xf[s][7], a_ad[s], 14336); break;
      case 8: FMI_W4_RD(wf[s][0], w_ad[s], 0); break;
      case 9: FMI_W4_RD(wf[s][1], w_ad[s], 2048); break;
      case 10: FMI_W4_RD(wf[s][2], w_ad[s], 4096); break;
      case 11: FMI_W4_RD(wf[s][3], w_ad[s], 6144); break;
      case 12: FMI_W4_RD(wf[s][4], w_ad[s], 8192); break;
      case 13: FMI_W4_RD(wf[s][5], w_ad[s], 10240); break;
      case 14: FMI_W4_RD(wf[s][6], w_ad[s], 12288); break;
      default: FMI_W4_RD(wf[s][7], w_ad[s], 14336); break;
    }
  };
  // LDS reads retire in order: with at most N reads of the other buffer pending, buffer s has landed.
  // The "+v" ties make the fragments depend on the wait so no MFMA is scheduled above it.
#define FMI_W4_WAIT(N, s)                                                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                                                        \
               : "+v"(xf[s][0]), "+v"(xf[s][1]), "+v"(xf[s][2]), "+v"(xf[s][3]), "+v"(xf[s][4]), "+v"(xf[s][5]), "+v"(xf[s][6]), "+v"(xf[s][7]), \
                 "+v"(wf[s][0]), "+v"(wf[s][1]), "+v"(wf[s][2]), "+v"(wf[s][3]), "+v"(wf[s][4]), "+v"(wf[s][5]), "+v"(wf[s][6]), "+v"(wf[s][7]))
  // (inline asm pins the 64 accumulators to the AGPR half; with the builtin hipcc spread them over both halves and moved ~500
  // registers per K tile between them.  An accumulator is touched once per k-step, 64 MFMAs apart: no dependent-issue hazard.)
  auto mfma = [&](int s, int i, int j) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wf[s][j]), "v"(xf[s][i]));
  };

  // k-step 0 of tile 0
  rebase(0, 0);
#pragma unroll
  for (int w = 0; w < 16; ++w) read_frag(0, w);
  int wslot = 0;  // W slot of tile t = t % 3
  // Branch-free body: past the end of K the DMA re-fetches the last tile into slots nobody reads again, and the
  // fragment reads of the "next tile" fetch garbage nobody multiplies; every count below is the same for all t.
  //   k-step 0 (64 MFMAs): the 16 reads of this tile's k-step 1 (every 3rd MFMA); W(t+2) -> slot (t+2)%3, free since
  //                        barrier(t-1) (every 8th MFMA)
  //   k-step 1 (64 MFMAs): its fragments are in = all of this tile's reads -> vmcnt(8): A(t+1), W(t+1) landed -> BARRIER(t);
  //                        then the 16 reads of tile t+1's k-step 0 (every 3rd MFMA, done 16 MFMAs before they are needed)
  //                        and A(t+2) -> slot t&1 (every 8th MFMA)
  for (int t = 0; t < nk; ++t) {
    const int wnext = wslot == 2 ? 0 : wslot + 1;   // W slot of tile t+1
    const int wnn = wnext == 2 ? 0 : wnext + 1;     // W slot of tile t+2
    const int kt2 = min(t + 2, klast);
    FMI_W4_WAIT(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 64; ++q) {
      mfma(0, q >> 3, q & 7);
      if (q % 3 == 1 && q / 3 < 16) read_frag(1, q / 3);
      if ((q & 7) == 3) dma_w(kt2, wnn, q >> 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    FMI_W4_WAIT(0, 1);  // the tile's last reads are in: its LDS slots are free
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // oldest first: ..., A(t+1) x8, W(t+2) x8
    sync_all();
    rebase((t + 1) & 1, wnext);
#pragma unroll
    for (int q = 0; q < 64; ++q) {
      mfma(1, q >> 3, q & 7);
      if (q % 3 == 1 && q / 3 < 16) read_frag(0, q / 3);
      if ((q & 7) == 5) dma_a(kt2, q >> 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    wslot = wnext;
  }
  // the trailing dummy DMA / reads must not land in the epilogue's staging; the last MFMAs drain before the epilogue reads the
  // accumulators (hipcc knows nothing of an asm MFMA's latency: every tile passes through a volatile asm behind the nops)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
#undef FMI_W4_RD
#undef FMI_W4_WAIT
  w4_epilogue16<ACT>(P, acc, smem, m0, n0, wave, wm, wn, lane);
}

#endif  // FMI_ALT_KERNELS

}  // namespace fmi
#include "gemm_w4q.h"
namespace fmi {

// Kernel-selection switches: PROCESS-WIDE test / ablation hooks (documented as such in include/flux_mi355x.h), shared by every
// handle and thread.  They select between kernels that produce identical bits, so a concurrent change can move time, not results;
// they are atomics so that a setter racing a launch is at least well-defined.
static std::atomic<bool> g_pingpong{true};
// Default OFF since both kernels moved to v_mfma_f32_16x16x32_bf16 (round 2): with twice the MFMA instructions per K tile the
// one-wave-per-SIMD stream no longer beats two waves per SIMD (tools/gemm_bench, f32 residual epilogue: 4608x3072x15360 1280 vs
// 1385 TF, 4096x3072x12288 1259 vs 1353 TF; it was +5-8 % with 32x32x16).  Bit-identical to the ping-pong kernel;
// FMI_GEMM_W4=1 in the environment (or set_gemm_w4(true)) sends the launches `w4_pays` selects below (the residual-update
// GEMMs: proj, mlp2, linear2) to it.
static std::atomic<bool> g_w4{[] {
  const char* e = getenv("FMI_GEMM_W4");
  return e ? atoi(e) != 0 : false;
}()};
static std::atomic<int> g_w4q_min_rows{256};
// never for dense weights (measured: slower, see launch_gemm); the packed 4-bit kernel always fuses.  FMI_GEMM_W4_QKV_MIN_N = the bound for experiments
static std::atomic<int> g_w4_qkv_min_n{[]() { const char* e = getenv("FMI_GEMM_W4_QKV_MIN_N"); return e && atoi(e) > 0 ? atoi(e) : 1 << 30; }()};
void set_gemm_w4_qkv_min_n(int n) { g_w4_qkv_min_n = n; }
void set_gemm_w4q_min_rows(int rows) { g_w4q_min_rows = rows; }
void set_gemm_w4(bool on) { g_w4 = on; }
void set_gemm_pingpong(bool on) { g_pingpong = on; }

// Band height of a problem's tile order.  An XCD runs ~32 consecutive tiles at a time; in a band of r tile-rows they span 32 / r
// columns, i.e. r + 32 / r operand panels per K step for 32 tiles — the L2-miss bytes of the launch are proportional to the
// tile-weighted mean of that figure (measured: FETCH_SIZE of 4608 x 21504 x 3072 1.07 GB at 8 + 8 + 2 rows, 0.95 GB at 6 + 6 + 6;
// the model says 12.67 vs 11.33, profiles/r03_band_probe.txt).  Multiples of 8 rows keep 8 (measured best on M = 4096 in round 1);
// otherwise the height in 5 .. 10 with the least modelled traffic, the ragged last band included.  FMI_GEMM_BAND=<n> pins one height
// (A/B runs).  Any height gives the same results: the map is a bijection of the tiles.
static int pick_tile_band(int tiles_m) {
  static const int pinned = [] {
    const char* e = getenv("FMI_GEMM_BAND");
    return e ? atoi(e) : 0;
  }();
  if (pinned > 0) return pinned;
  if (tiles_m <= TILE_BAND || tiles_m % TILE_BAND == 0) return TILE_BAND;
  auto panels = [](int r) { return (double)r + 32.0 / r; };
  int best = TILE_BAND;
  double best_cost = 1e30;
  for (int h = 5; h <= 10; ++h) {
    const int rag = tiles_m % h;
    const double cost = (tiles_m - rag) * panels(h) + (rag ? rag * panels(rag) : 0.0);
    if (cost < best_cost - 1e-9) best_cost = cost, best = h;
  }
  return best;
}

int launch_gemm(const GemmProblem* probs, int nprob, hipStream_t stream) {
  if (nprob <= 0) return FMI_OK;
  if (nprob > MAX_PROBLEMS) return fail(FMI_ERR_INVALID, "launch_gemm: too many grouped problems");
  GemmBatch b;
  b.nprob = nprob;
  int total = 0;
  const bool quant = probs[0].q_type != 0;
  const bool conv = probs[0].cv_ks != 0;
  const bool fp8 = probs[0].fp8 != 0;
  int max_n = 0;
  for (int i = 0; i < nprob; ++i) max_n = std::max(max_n, probs[i].N);
  int bn = max_n <= 128 ? 128 : 256;
  if (conv && bn == 256) {
    // a convolution whose 256 x 256 tiles leave a quarter of the CUs or more without one (the decoder's 512 -> 512 layers on 128 x 128
    // pixels: 128 tiles) runs on 256 x 128 tiles instead
    int64_t tiles = 0;
    for (int i = 0; i < nprob; ++i) tiles += (int64_t)cdiv(probs[i].M, BM) * cdiv(probs[i].N, 256);
    if (tiles <= 192) bn = 128;
  }
  for (int i = 0; i < nprob; ++i) {
    const GemmProblem& p = probs[i];
    if (p.M <= 0 || p.N <= 0) return fail(FMI_ERR_INVALID, "launch_gemm: empty problem");
    if (p.fp8 != probs[0].fp8 || p.fp8 < 0 || p.fp8 > 2) return fail(FMI_ERR_INVALID, "launch_gemm: cannot mix e4m3 / int8 / bf16 problems in one group");
    if (fp8 && (p.q_type || p.cv_ks || bn != 256 || p.K % 128 || p.lda % 16 || p.ldw % 16 || !p.a_scale || !p.w_scale))
      return fail(FMI_ERR_INVALID, "launch_gemm: fp8 needs dense operands, N > 128, K / lda / ldw multiples of 128 / 16 / 16 and both scale vectors");
    if ((p.a_off != nullptr) != (p.w_sum != nullptr) || (p.a_off && p.fp8 != 2))
      return fail(FMI_ERR_INVALID, "launch_gemm: the offset term (a_off, w_sum) belongs to int8 problems and needs both vectors");
    if (p.K <= 0 || p.K % BK != 0) return fail(FMI_ERR_INVALID, "launch_gemm: K must be a positive multiple of 64, got " + std::to_string(p.K));
    if ((p.q_type != 0) != quant || (p.cv_ks != 0) != conv) return fail(FMI_ERR_INVALID, "launch_gemm: cannot mix dense / 4-bit / conv problems in one group");
    if (quant && conv) return fail(FMI_ERR_UNSUPPORTED, "launch_gemm: quantised convolution");
    if ((!conv && p.lda % 8) || (!quant && p.ldw % 8)) return fail(FMI_ERR_INVALID, "launch_gemm: lda/ldw must be multiples of 8 elements (16-byte rows)");
    if (conv && (p.cv_cin % 64 || p.K != p.cv_ks * p.cv_ks * p.cv_cin || !p.cv_zero)) return fail(FMI_ERR_INVALID, "launch_gemm: bad conv descriptor (Cin % 64, K = k*k*Cin)");
    if (quant && p.q_type != 3 && (p.q_blocksize % 64 != 0 || p.q_blocksize <= 0)) return fail(FMI_ERR_INVALID, "launch_gemm: 4-bit blocksize must be a multiple of 64");
    if (p.qk_qh) {
      if (conv || bn != 256 || p.qk_D % 256 || p.qk_H * 128 != p.qk_D || 3 * p.qk_D > p.N || p.qk_rows <= 0 || p.qk_rows % 16 || p.qk_row_off % 16 || p.M % 16 ||
          p.qk_Lpad % 64 || !p.qk_kh || !p.qk_vt || !p.qk_wq || !p.qk_wk || !p.qk_pe || (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 7)))
        return fail(FMI_ERR_INVALID, "launch_gemm: bad fused qkv relayout descriptor (needs 256-wide tiles, D % 256 == 0, rows/row_off/M % 16 == 0)");
    }
    b.p[i] = p;
    b.tile_start[i] = total;
    b.band[i] = pick_tile_band(cdiv(p.M, BM));
    total += cdiv(p.M, BM) * cdiv(p.N, bn);
  }
  for (int i = nprob; i <= MAX_PROBLEMS; ++i) b.tile_start[i] = total;
  for (int i = nprob; i < MAX_PROBLEMS; ++i) b.band[i] = TILE_BAND;
  const dim3 grid(total), blk(GEMM_THREADS);
#define FMI_GEMM_LAUNCH(MODE)                                                               \
  do {                                                                                      \
    if (bn == 128)                                                                          \
      hipLaunchKernelGGL((gemm_bf16_kernel<MODE, 1>), grid, blk, 0, stream, b);             \
    else                                                                                    \
      hipLaunchKernelGGL((gemm_bf16_kernel<MODE, 2>), grid, blk, 0, stream, b);             \
  } while (0)
  const int act = epilogue_kind(probs, nprob);
  if (act < 0) return fail(FMI_ERR_INVALID, "launch_gemm: the problems of a grouped launch must share the activation kind");
  // 4-wave kernel: its K loop is 7-10 % faster, its two-round epilogue slower — measured break-even (tools/gemm_bench,
  // FMI_EPI=store|gelu|resid on the FLUX shapes): the f32 residual read-modify-write launches at every K (proj -5 %,
  // mlp2 -10 %, linear2 -7 %), everything else from K = 8192 on (mlp1 + GELU at K = 3072 is a wash).
  // The launches with the fused q|k|v relayout epilogue stay on the 8-wave kernel: the 4-wave kernels have the epilogue
  // too (w4_epilogue; the packed 4-bit kernel needs it), but with one workgroup per CU nothing overlaps its two LDS round
  // trips, and it costs more than the K loop gains (tools/gemm_bench FMI_EPI=qkv, 4608 x 21504 x 3072: 8-wave 588 us,
  // 4-wave 637 us; 4096 x 9216 x 3072: 258 vs 305 us).  set_gemm_w4_qkv_min_n lowers the bound for experiments.
  bool w4_pays = !fp8 && !conv && !quant;
  // 4-bit weights: the one-wave-per-SIMD fused kernel from g_w4q_min_rows rows on (below it the GEMM is bound by the packed
  // weight stream and the two-workgroups-per-CU kernel with the VGPR expand hides latency better)
  bool w4q_ok = quant;
  for (int i = 0; quant && i < nprob; ++i) {
    const GemmProblem& p = probs[i];
    const int kb = p.q_blocksize / BK;  // K tiles per absmax block
    if (p.q_type == 3 || p.M < g_w4q_min_rows || p.K % p.q_blocksize || (kb & (kb - 1)) || (int64_t)p.N * p.K / 2 >= (1ll << 32)) w4q_ok = false;
  }
  for (int i = 0; i < nprob; ++i) {
    const GemmProblem& p = probs[i];
    if (p.epi != EPI_RESID_GATE_F32 && p.K < 8192 && !(p.qk_qh && p.N >= g_w4_qkv_min_n)) w4_pays = false;
  }
#define FMI_ACT_LAUNCH(KERNEL, FP8FLAG, THREADS)                                                    \
  do {                                                                                              \
    if (act == 0) hipLaunchKernelGGL((KERNEL<FP8FLAG, 0>), grid, dim3(THREADS), 0, stream, b);      \
    else if (act == 1) hipLaunchKernelGGL((KERNEL<FP8FLAG, 1>), grid, dim3(THREADS), 0, stream, b); \
    else if (act == 2) hipLaunchKernelGGL((KERNEL<FP8FLAG, 2>), grid, dim3(THREADS), 0, stream, b); \
    else hipLaunchKernelGGL((KERNEL<FP8FLAG, 3>), grid, dim3(THREADS), 0, stream, b);               \
  } while (0)
  if (fp8 && probs[0].fp8 == 2)
    FMI_ACT_LAUNCH(gemm_pp_kernel, 2, GEMM_THREADS);
  else if (fp8)
    FMI_ACT_LAUNCH(gemm_pp_kernel, 1, GEMM_THREADS);
  else if (conv)
    FMI_GEMM_LAUNCH(2);
  else if (quant && bn == 256 && w4q_ok) {
    // fused dequant-GEMM, one wave per SIMD (gemm_w4q.h)
    const dim3 g4(total), b4(W4_THREADS);
    FMI_LDS_GUARD((gemm_w4q_kernel<0>), W4Q_LUT_BYTES + 4 * A_TILE_BYTES);  // (the four instantiations share the LDS layout)
    if (act == 0) hipLaunchKernelGGL((gemm_w4q_kernel<0>), g4, b4, 0, stream, b);
    else if (act == 1) hipLaunchKernelGGL((gemm_w4q_kernel<1>), g4, b4, 0, stream, b);
    else if (act == 2) hipLaunchKernelGGL((gemm_w4q_kernel<2>), g4, b4, 0, stream, b);
    else hipLaunchKernelGGL((gemm_w4q_kernel<3>), g4, b4, 0, stream, b);
  } else if (quant)
    FMI_GEMM_LAUNCH(1);
#if FMI_ALT_KERNELS
  else if (bn == 256 && g_w4 && w4_pays)
    FMI_ACT_LAUNCH(gemm_w4_kernel, false, W4_THREADS);
  else if (bn == 256 && !g_pingpong)
    FMI_GEMM_LAUNCH(0);  // (the double-buffered 256 x 256 kernel: gemm_pp_kernel's bit-identical predecessor)
#else
  else if (bn == 256 && ((g_w4 && w4_pays) || !g_pingpong))
    return fail(FMI_ERR_UNSUPPORTED, "launch_gemm: the dense 4-wave / double-buffered 256-wide kernels live in the test build (libflux_mi355x_alt.so: make alt)");
#endif
  else if (bn == 256)
    FMI_ACT_LAUNCH(gemm_pp_kernel, 0, GEMM_THREADS);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<0, 1>), grid, blk, 0, stream, b);  // N <= 128: the 256 x 128 double-buffered kernel
#undef FMI_GEMM_LAUNCH
#undef FMI_ACT_LAUNCH
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

}  // namespace fmi
