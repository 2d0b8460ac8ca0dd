// attention_w32.h — joint attention on v_mfma_f32_32x32x16_bf16, one wave per SIMD (included by attention.hip; same
// translation unit).  Round 3's kernel, the sibling of attention_w16.h (read that header and tools/gen_attention_w32.py first:
// Q pre-multiplied by scale * log2(e) and -m accumulated by the first d-step of the score product, exp2 in place, no cross-lane
// traffic on the common softmax path, row sums from a ones-row MFMA, the whole KV stream one generated asm statement) on the
// 32 x 32 shape: a 32-clock MFMA hides ~19 clocks of the softmax's instructions, a 16-clock one ~8 (tools/gen_issue_model.py),
// and half as many MFMAs and waits are issued — with one wave per SIMD the kernel is bound by what sits between the MFMAs.
//
// Replaces scaled_dot_product_attention (diffusion_rs_core/src/models/flux/model.rs:40-50) -> backend::ops::sdpa
// (diffusion_rs_backend/src/ops.rs:247-262: softmax((q k^T) * scale) v, f32, scores materialised).
//
// Layouts are attention_w4_kernel's (block b = 32 queries, a lane owns one query and 32 of a tile's 64 keys; K ring slot c of
// row r at c ^ (r & 15); V^T with the k-permutation of attention.hip: vt_perm).  Not bit-identical to the 8-wave kernels: the
// scores differ by the rounding of q * scale * log2(e) to bf16 and the row sum is that of the rounded p.
#pragma once
#ifndef FMI_AW32_LOOP_INC  // (tools/run_attn_w16_ablations.sh points this at a timing-experiment variant of the generated stream)
#define FMI_AW32_LOOP_INC "attention_w32_loop.inc"
#endif
#include FMI_AW32_LOOP_INC

namespace fmi {

constexpr int AW32_THREADS = 256;

template <int THR_X16>
__global__ __launch_bounds__(AW32_THREADS, 1) void attention_w32_kernel(const bf16_t* __restrict Q, const bf16_t* __restrict K, const bf16_t* __restrict Vt,
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
  const int hl = lane >> 5, l31 = lane & 31;
  const bf16_t* Kb = K + (int64_t)bh * Lk * HD;
  const bf16_t* Vb = Vt + (int64_t)bh * HD * Lkpad;
  const int ntiles = (Lk + ATT_KV - 1) / ATT_KV;  // >= 2 (the launcher sends single-tile problems to the 8-wave kernel)

  typedef float f32x32 __attribute__((ext_vector_type(32)));
  typedef int i32x32 __attribute__((ext_vector_type(32)));
  typedef int i32x16 __attribute__((ext_vector_type(16)));

  // ---- Q fragments (MFMA B operand): QF[b][s] = bf16(Q[q0 + 32 b + l31][16 s + 8 hl .. + 7] * scale * log2(e))
  i32x32 QA[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int qr = min(q0 + 32 * b + l31, Lq - 1);
    const bf16_t* qp = Q + ((int64_t)bh * Lq + qr) * HD + 8 * hl;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const uint4 raw = *reinterpret_cast<const uint4*>(qp + 16 * s);
      const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(w[e] << 16) * scale_log2e, hi = __uint_as_float(w[e] & 0xffff0000u) * scale_log2e;
        QA[b][s * 4 + e] = (int)pack_bf16x2(lo, hi);
      }
    }
  }

  // ---- LDS-DMA: 16 one-KiB chunks per tile and operand, 4 per wave.  Destination is lane-linear, the swizzle sits in the source
  // offsets (loop invariants); a tile index past the end is clamped in the stream (the last tile is fetched again: identical bytes).
  i32x16 R0, R1;
  i32x32 NMR;
  const int k_last_rows = Lk - (ntiles - 1) * ATT_KV;  // keys in the last tile (1..64): rows beyond are fetched from the last key
  uint32_t k_voff[4], v_voff[4], k_voffc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int chunk = wave * 4 + i;
    const int kr = chunk * 4 + (lane >> 4), vr = chunk * 8 + (lane >> 3);
    k_voff[i] = (uint32_t)(kr * 256 + (((lane & 15) ^ (kr & 15)) << 4));
    k_voffc[i] = kr >= k_last_rows ? (uint32_t)((k_last_rows - 1) * 256 + (((lane & 15) ^ (kr & 15)) << 4)) : k_voff[i];
    v_voff[i] = (uint32_t)(vr * Lkpad * 2 + (((lane & 7) ^ ((vr >> 1) & 7)) << 4));
  }
  auto stage_k = [&](int tile, int i) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(Kb) + (int64_t)tile * (ATT_KV * 256);
    const uint32_t off = (tile == ntiles - 1) ? k_voffc[i] : k_voff[i];
    __builtin_amdgcn_global_load_lds((glb_void*)(base + off), (lds_void*)(smem + (tile & 3) * TILE + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  auto stage_v = [&](int tile, int i) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(Vb) + (int64_t)tile * (ATT_KV * 2);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + v_voff[i]), (lds_void*)(smem + VT_RING + (tile & 3) * TILE + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  // (smem sits at LDS byte 0 — the ring-slot xor rely on it: it is the kernel's only __shared__ object, which the host checks before the first launch, FMI_LDS_GUARD)
  // Fragment read addresses: K fragment (key half u, d-step s) at KAD[s] + 8192 u, V^T fragment (d block dt, k-step c) at VAD[c] + 4096 dt
  {
#pragma unroll
    for (int s = 0; s < 8; ++s) R0[s] = (l31 * 256 + ((hl ^ (lane & 15)) << 4)) ^ (s << 5);
#pragma unroll
    for (int c = 0; c < 4; ++c) R0[8 + c] = VT_RING + ((l31 * 128 + ((hl ^ ((l31 >> 1) & 7)) << 4)) ^ (c << 5));
#pragma unroll
    for (int i = 0; i < 4; ++i) R0[12 + i] = (int)k_voff[i], R1[i] = (int)v_voff[i], R1[4 + i] = (int)k_voffc[i];
    R1[8] = 4 * hl;  // LKEY: the lane's part of a score's key index
    R1[9] = 0;
    R1[10] = R1[11] = __float_as_int(-1e30f);  // M
#pragma unroll
    for (int i = 0; i < 4; ++i) R1[12 + i] = l31 == 0 ? 0x3f803f80 : 0;  // the ones fragment: row 0 = bf16 1.0
#pragma unroll
    for (int i = 0; i < 32; ++i) NMR[i] = 0;  // NM = 0: the first tile's fold is m = 0 (it always takes the rescale block)
  }

  // ---- prologue: K(0..2), V^T(0..1) in flight; everything landed and published before the first read
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < ntiles) {
#pragma unroll
      for (int i = 0; i < 4; ++i) stage_k(t, i);
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < ntiles) {
#pragma unroll
      for (int i = 0; i < 4; ++i) stage_v(t, i);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- the KV stream: one generated asm statement (tools/gen_attention_w32.py), every array pinned to the registers its text names
  f32x32 O[4], OL;
  i32x32 SP0, SP1, FP;  // S^T (v[0:63]), P (v[64:95]) and the fragment buffers (v[96:127]): written before read inside the statement
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    O[0][r] = O[1][r] = O[2][r] = O[3][r] = OL[r] = 0.f;
    SP0[r] = SP1[r] = 0;
    FP[r] = 0;
  }
  i32x32 FB = FP;
  {
    const uint64_t kb64 = (uint64_t)(uintptr_t)Kb, vb64 = (uint64_t)(uintptr_t)Vb;
    const uint32_t kb_lo = __builtin_amdgcn_readfirstlane((uint32_t)kb64), kb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(kb64 >> 32));
    const uint32_t vb_lo = __builtin_amdgcn_readfirstlane((uint32_t)vb64), vb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(vb64 >> 32));
    const float thr = (float)THR_X16 * 0.0625f;
    asm volatile(FMI_AW32_LOOP_ASM
                 : "+{a[0:31]}"(O[0]), "+{a[32:63]}"(O[1]), "+{a[64:95]}"(O[2]), "+{a[96:127]}"(O[3]), "+{a[192:223]}"(OL), "+{v[0:31]}"(SP0), "+{v[32:63]}"(SP1),
                   "+{v[64:95]}"(FP), "+{v[96:127]}"(FB), "+{v[128:143]}"(R0), "+{v[144:159]}"(R1), "+{v[160:191]}"(NMR)
                 : "{a[128:159]}"(QA[0]), "{a[160:191]}"(QA[1]), [kb_lo] "s"(kb_lo), [kb_hi] "s"(kb_hi), [vb_lo] "s"(vb_lo), [vb_hi] "s"(vb_hi),
                   [ntm1] "s"(ntiles - 1), [thr] "s"(thr), [woff] "s"(wave * 4096), [rag] "s"(k_last_rows)
                 : "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",
                   "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87",
                   "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "vcc", "scc", "memory");
  }

  // ---- epilogue.  Lane (hl, q) holds O^T[d = 32 dt + 8 (r >> 2) + 4 hl + (r & 3)][query 32 b + q] in O[2 b + (dt >> 1)][16 (dt & 1) + r];
  // the row sum of query 32 b + q is register 16 b of OL in lane q (row 0 of the ones product); the statement ends drained.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the rings
  char* stg = smem + wave * TILE;  // 64 rows x 256 B, 16-byte slot s of row r at s ^ (r & 15)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const float l = __shfl(OL[16 * b], l31, 64);
    const float inv = 1.0f / l;
    const int r = 32 * b + l31;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x32& acc = O[2 * b + (dt >> 1)];
        const int o = 16 * (dt & 1) + 4 * g4;
        const int d = dt * 32 + g4 * 8 + 4 * hl;
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
