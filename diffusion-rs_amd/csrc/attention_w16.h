// attention_w16.h — joint attention on v_mfma_f32_16x16x32_bf16, one wave per SIMD (included by attention.hip; same
// translation unit).  Round 3's kernel; replaces attention_w4_kernel as the default bf16 path.
//
// Replaces scaled_dot_product_attention (diffusion_rs_core/src/models/flux/model.rs:40-50) -> backend::ops::sdpa
// (diffusion_rs_backend/src/ops.rs:247-262: softmax((q k^T) * scale) v, f32, scores materialised).
//
// The machine mapping is attention_w4_kernel's (4 waves, one per SIMD, 64 query rows per wave as two blocks that run half a KV
// tile apart; K / V^T tiles by LDS-DMA into 4-deep rings, one barrier per tile; transposed products S^T = K Q^T,
// O^T = V^T P^T with P fed from the S registers through the k-permutation baked into V^T); what changed, and why, is written
// up in tools/gen_attention_w16.py, which generates the WHOLE KV stream (first tile to last) as one asm statement:
//   * both products on the 16x16x32 MFMA (a lane owns 4 keys x 1 query of a 16 x 16 score tile);
//   * Q pre-multiplied by scale * log2(e) (rounded to bf16 once, here, when the fragments are loaded) and -m accumulated by
//     the first d-step of the score product, so p = exp2(s') with no per-score fma;
//   * no cross-lane traffic on the common softmax path; row sums from the bf16-rounded probabilities (v_dot2c_f32_bf16).
// This file is the frame around that statement: Q fragments, the K ring's DMA offsets (this kernel's own swizzle: slot p of row
// r holds global slot p ^ f(r), f(r) = (r & 7) | ((r >> 4) & 1) << 3), the first DMA pieces, and the epilogue (row sums
// reduced over the four lane groups, O = O^T / l staged through LDS into whole 256-byte rows).
//
// Numerics vs the 8-wave kernels: same f32 accumulation inside a product; the scores differ by the rounding of q * scale *
// log2(e) to bf16 (relative 2^-9 per element of q) and the row sum by the rounding of p — both far inside the stated
// tolerance against the f32 oracle (tests/test_gpu_ops.py: rel-L2 <= 6e-3), not bit-identical to attention_pp_kernel.
#pragma once
#ifndef FMI_AW16_LOOP_INC  // (tools/run_attn_w16_ablations.sh points this at a timing-experiment variant of the generated stream)
#define FMI_AW16_LOOP_INC "attention_w16_loop.inc"
#endif
#include FMI_AW16_LOOP_INC
#ifndef FMI_AW16F8_LOOP_INC  // the fp8-QK^T stream (AW16_MODE=fp8qk of the same generator)
#define FMI_AW16F8_LOOP_INC "attention_w16f8_loop.inc"
#endif
#include FMI_AW16F8_LOOP_INC

namespace fmi {

constexpr int AW16_THREADS = 256;

// QK8 = true (the model's fp8 mode): Q and K point to OCP e4m3 bytes, rows of 128 B, with their static scales folded
// into scale_log2e, which the launcher guarantees to be an exact power of two 2^-n (the host picks the q scale accordingly): the
// score product is one v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile whose E8M0 block scale carries 2^-n.  P, V^T and the
// second product are the bf16 ones.
template <int THR_X16, bool QK8 = false>
__global__ __launch_bounds__(AW16_THREADS, 1) void attention_w16_kernel(const bf16_t* __restrict Q, const bf16_t* __restrict K, const bf16_t* __restrict Vt,
                                                                        AttnOut out, int H, int Lq, int Lk, int Lkpad, float scale_log2e) {
  constexpr int TILE = 16384, VT_RING = 4 * TILE;
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE];  // K ring [4][64 x 128] at 0, V^T ring [4][128 x 64] at 64 KiB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (Lq + ATT_QBLK - 1) / ATT_QBLK;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqb;
  const int b_ = bh / H, h = bh % H;
  const int q0 = (lid % nqb) * ATT_QBLK + wave * 64;
  const int g = lane >> 4, n16 = lane & 15;
  constexpr int KROW = QK8 ? 128 : 256, TILE_K = 64 * KROW;  // bytes of a K row / of a K tile (HBM and LDS)
  const char* Kb = reinterpret_cast<const char*>(K) + (int64_t)bh * Lk * KROW;
  const bf16_t* Vb = Vt + (int64_t)bh * HD * Lkpad;
  const int ntiles = (Lk + ATT_KV - 1) / ATT_KV;  // >= 2 (the launcher sends single-tile problems to the 8-wave kernel)

  typedef __attribute__((ext_vector_type(4))) int frag_t;
  typedef float f32x32 __attribute__((ext_vector_type(32)));
  typedef int i32x32 __attribute__((ext_vector_type(32)));
  typedef int i32x16 __attribute__((ext_vector_type(16)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));

  // ---- LDS-DMA: 16 one-KiB chunks per tile and operand, 4 per wave.  Destination is lane-linear, the swizzle sits in the source
  // offsets (loop invariants); a tile index past the end is clamped in the stream (the last tile is fetched again: identical bytes).
  i32x16 R0, R1;
  i32x8 R2;
  frag_t R3;  // the ones fragment: A operand whose row 0 is bf16 1.0 (V^T extended by a row of ones -> the row sums)
  const int k_last_rows = Lk - (ntiles - 1) * ATT_KV;  // keys in the last tile (1..64): rows beyond are fetched from the last key
  uint32_t k_voff[4], v_voff[4], k_voffc[4];
  constexpr int KP = QK8 ? 2 : 4;  // 1-KiB DMA pieces of a K tile per wave
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vr = (wave * 4 + i) * 8 + (lane >> 3);
    v_voff[i] = (uint32_t)(vr * Lkpad * 2 + (((lane & 7) ^ ((vr >> 1) & 7)) << 4));
    if constexpr (QK8) {  // piece = 8 rows of 128 B; slot p of row r holds global slot p ^ f8(r)
      const int kr = (wave * 2 + (i & 1)) * 8 + (lane >> 3);
      const int fk = ((kr & 7) >> 1) | (((kr >> 4) & 1) << 2);
      k_voff[i] = (uint32_t)(kr * 128 + (((lane & 7) ^ fk) << 4));
      k_voffc[i] = kr >= k_last_rows ? (uint32_t)((k_last_rows - 1) * 128 + (((lane & 7) ^ fk) << 4)) : k_voff[i];
    } else {      // piece = 4 rows of 256 B; slot p of row r holds global slot p ^ f(r)
      const int kr = (wave * 4 + i) * 4 + (lane >> 4);
      const int fk = (kr & 7) | (((kr >> 4) & 1) << 3);
      k_voff[i] = (uint32_t)(kr * 256 + (((lane & 15) ^ fk) << 4));
      k_voffc[i] = kr >= k_last_rows ? (uint32_t)((k_last_rows - 1) * 256 + (((lane & 15) ^ fk) << 4)) : k_voff[i];
    }
  }
  auto stage_k = [&](int tile, int i) __attribute__((always_inline)) {
    const char* base = Kb + (int64_t)tile * TILE_K;
    const uint32_t off = (tile == ntiles - 1) ? k_voffc[i] : k_voff[i];
    __builtin_amdgcn_global_load_lds((glb_void*)(base + off), (lds_void*)(smem + (tile & 3) * TILE_K + (wave * KP + i) * 1024), 16, 0, 0);
  };
  auto stage_v = [&](int tile, int i) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(Vb) + (int64_t)tile * (ATT_KV * 2);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + v_voff[i]), (lds_void*)(smem + VT_RING + (tile & 3) * TILE + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  // (smem sits at LDS byte 0 — the ring-slot xor rely on it: it is the kernel's only __shared__ object, which the host checks before the first launch, FMI_LDS_GUARD)
  // Fragment read addresses.  K fragment (key block a, d-step s): lane (g, m) reads row 32 (a >> 1) + 8 (a & 1) + (m & 7) + 16 (m >> 3)
  // — the key whose score the V^T k-permutation expects in row m of block a — global slot 4 s + g, at KAD[s] + the block's
  // immediate offset.  V^T fragment (d block dt, k-step kk): row 16 dt + m, slot 4 kk + g, at VAD[kk] + 2048 dt.
  {
    const int m = n16, krow = (m & 7) + 16 * (m >> 3);
    if constexpr (QK8) {  // fp8 K fragment (key block a): the row's bytes 32 g .. + 31 = global slots 2 g, 2 g + 1; f8(row) = ((m & 7) >> 1) | (m >> 3) << 2
      const int f8 = ((m & 7) >> 1) | ((m >> 3) << 2);
      R0[0] = krow * 128 + (((2 * g) ^ f8) << 4);
      R0[1] = R0[0] ^ 16;
      R0[2] = R0[3] = 0;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) R0[s] = krow * 256 + (((4 * s + g) ^ m) << 4);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) R0[4 + kk] = VT_RING + m * 128 + (((4 * kk + g) ^ ((m >> 1) & 7)) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) R0[6 + i] = (int)k_voff[i], R0[10 + i] = (int)v_voff[i];
    R0[14] = (int)k_voffc[0], R0[15] = (int)k_voffc[1];
    R1[0] = (int)k_voffc[2], R1[1] = (int)k_voffc[3];
    R1[2] = 16 * (g >> 1) + 4 * (g & 1);  // LKEY: the lane's part of a score's key index
    R1[3] = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) R1[4 + i] = 0;  // NM = 0: the first tile's fold is m = 0 (it always takes the rescale block)
#pragma unroll
    for (int i = 0; i < 4; ++i) R2[i] = 0, R2[4 + i] = __float_as_int(-1e30f);  // NM[12..15], M = -1e30
#pragma unroll
    for (int i = 0; i < 4; ++i) R3[i] = n16 == 0 ? 0x3f803f80 : 0;
  }

  // ---- prologue: K(0..2), V^T(0..1) in flight; everything landed and published before the first read
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < ntiles) {
#pragma unroll
      for (int i = 0; i < KP; ++i) stage_k(t, i);
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < ntiles) {
#pragma unroll
      for (int i = 0; i < 4; ++i) stage_v(t, i);
    }
  __builtin_amdgcn_sched_barrier(0);
  // (after the first DMA pieces have been issued: the loads and the scale-and-round of Q run while those are in flight)
  // ---- Q fragments (MFMA B operand, rows = d): QF[b][c][s] = bf16(Q[q0 + 32 b + 16 c + n][32 s + 8 g .. + 7] * scale * log2(e));
  // fp8: QF8[b][c] = the 32 bytes Q8[q][32 g .. + 31] as they are (the scale rides in the MFMA's block scale)
  i32x32 QA[2];
#pragma unroll
  for (int r = 0; r < 32; ++r) QA[1][r] = 0;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int qr = min(q0 + 32 * b + 16 * c + n16, Lq - 1);
      if constexpr (QK8) {
        const char* qp = reinterpret_cast<const char*>(Q) + ((int64_t)bh * Lq + qr) * 128 + 32 * g;
        const uint4 lo = *reinterpret_cast<const uint4*>(qp), hi = *reinterpret_cast<const uint4*>(qp + 16);
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) QA[0][(2 * b + c) * 8 + e] = (int)w[e];
      } else {
        const bf16_t* qp = Q + ((int64_t)bh * Lq + qr) * HD + 8 * g;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint4 raw = *reinterpret_cast<const uint4*>(qp + 32 * s);
          const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16) * scale_log2e, hi = __uint_as_float(w[e] & 0xffff0000u) * scale_log2e;
            QA[b][(c * 4 + s) * 4 + e] = (int)pack_bf16x2(lo, hi);
          }
        }
      }
    }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- the KV stream: one generated asm statement (tools/gen_attention_w16.py), every array pinned to the registers its text names
  f32x32 O[4];
  i32x32 SP0, SP1, FP;  // S^T (v[0:63]), P + fragment buffers (v[64:127]): written before read inside the statement
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    O[0][r] = O[1][r] = O[2][r] = O[3][r] = 0.f;
    SP0[r] = SP1[r] = 0;
    FP[r] = 0;
  }
  i32x32 FB = FP;
  frag_t R4;  // fp8: E8M0 block scales of the score product (byte = biased exponent): 2^-n on the K side, 1 on the Q side
  R4[0] = (int)(((__float_as_uint(scale_log2e) >> 23) & 0xffu) * 0x01010101u), R4[1] = 0x7f7f7f7f, R4[2] = R4[3] = 0;
  i32x32 KF = FP;  // fp8 mode: the four 32-byte K fragment buffers (a[208:239]); named here so that the registers belong to the kernel
  f32x16 OL;  // ones-row accumulators: OL[(2 b + c) * 4] in lanes 0..15 = the row sum of query 32 b + 16 c + n
#pragma unroll
  for (int r = 0; r < 16; ++r) OL[r] = 0.f;
  {
    const uint64_t kb64 = (uint64_t)(uintptr_t)Kb, vb64 = (uint64_t)(uintptr_t)Vb;
    const uint32_t kb_lo = __builtin_amdgcn_readfirstlane((uint32_t)kb64), kb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(kb64 >> 32));
    const uint32_t vb_lo = __builtin_amdgcn_readfirstlane((uint32_t)vb64), vb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(vb64 >> 32));
    const float thr = (float)THR_X16 * 0.0625f;
#define FMI_AW16_OPERANDS                                                                                                                      \
    : "+{a[0:31]}"(O[0]), "+{a[32:63]}"(O[1]), "+{a[64:95]}"(O[2]), "+{a[96:127]}"(O[3]), "+{v[0:31]}"(SP0), "+{v[32:63]}"(SP1), "+{v[64:95]}"(FP), \
      "+{v[96:127]}"(FB), "+{v[128:143]}"(R0), "+{v[144:159]}"(R1), "+{v[160:167]}"(R2), "+{v[168:171]}"(R3), "+{v[172:175]}"(R4),               \
      "+{a[192:207]}"(OL), "+{a[208:239]}"(KF)                                                                                                       \
    : "{a[128:159]}"(QA[0]), "{a[160:191]}"(QA[1]), [kb_lo] "s"(kb_lo), [kb_hi] "s"(kb_hi), [vb_lo] "s"(vb_lo), [vb_hi] "s"(vb_hi),                \
      [ntm1] "s"(ntiles - 1), [thr] "s"(thr), [woffk] "s"(wave * KP * 1024), [woffv] "s"(VT_RING + wave * 4096), [rag] "s"(k_last_rows)             \
    : "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201",  \
      "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "s80", "s81", "s82", "s83", "s84", "s85",      \
      "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "vcc", "scc", "memory"
    if constexpr (QK8) asm volatile(FMI_AW16F8_LOOP_ASM FMI_AW16_OPERANDS);
    else asm volatile(FMI_AW16_LOOP_ASM FMI_AW16_OPERANDS);
#undef FMI_AW16_OPERANDS
  }

  // ---- epilogue.  Lane (g, n) holds O^T[d = 16 dt + 4 g + i][query 32 b + 16 c + n] in O[..][((8 b + dt) * 2 + c) * 4 + i] and its
  // the row sums in OL (lanes 0..15); the statement ends drained (nothing in flight).
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the rings
  char* stg = smem + wave * TILE;  // 64 rows x 256 B, 16-byte slot s of row r at s ^ (r & 15)
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float l = __shfl(OL[(2 * b + c) * 4], n16, 64);  // row 0 of the ones product lives in lane group 0
      const float inv = 1.0f / l;
      const int r = 32 * b + 16 * c + n16;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const int idx = ((8 * b + dt) * 2 + c) * 4;
        const f32x32& acc = O[idx >> 5];
        const int o = idx & 31;
        const int d = 16 * dt + 4 * g;
        const uint2 v = make_uint2(pack_bf16x2(acc[o] * inv, acc[o + 1] * inv), pack_bf16x2(acc[o + 2] * inv, acc[o + 3] * inv));
        *reinterpret_cast<uint2*>(stg + r * 256 + ((((d * 2) >> 4) ^ (r & 15)) << 4) + ((d * 2) & 15)) = v;
      }
    }
  __syncthreads();  // (each wave reads back only its own region; the barrier also orders the LDS writes before the reads)
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = it * 4 + (lane >> 4), c = lane & 15;
    const int q = q0 + r;
    const uint4 v = *reinterpret_cast<const uint4*>(stg + r * 256 + ((c ^ (r & 15)) << 4));
    if (q < Lq) {
      bf16_t* op;
      if (out.head_major) op = out.p1 + ((int64_t)bh * Lq + q) * HD;
      else if (q < out.rows0) op = out.p0 + (int64_t)b_ * out.bstride0 + (int64_t)q * out.ld0 + h * HD;
      else op = out.p1 + (int64_t)b_ * out.bstride1 + (int64_t)(q - out.rows0) * out.ld1 + h * HD;
      *reinterpret_cast<uint4*>(op + c * 8) = v;
    }
  }
}

}  // namespace fmi
