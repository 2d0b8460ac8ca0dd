// attention.hip — fused joint attention for the FLUX DiT, head dim 128, bf16 in / f32 softmax.
//
// Replaces scaled_dot_product_attention (diffusion_rs_core/src/models/flux/model.rs:40-50) ->
// backend::ops::sdpa (diffusion_rs_backend/src/ops.rs:247-262), which on CUDA/CPU upcasts to f32
// and materialises the full (B,24,L,L) score tensor (2 GB at L=4608) through three eager
// kernels.  Here scores never leave the CU.
//
// CDNA4 design:
//   * One workgroup = 8 waves = 256 query rows of one (batch, head); each wave owns 32 query rows.
//   * Everything is computed TRANSPOSED so that a lane owns one query row:
//       Sᵀ(kv,q)  = K · Qᵀ      (MFMA A operand = K rows from LDS, B operand = Q rows in VGPRs)
//       Oᵀ(d ,q)  = Vᵀ · Pᵀ     (A = Vᵀ rows from LDS,             B = P of this lane's row)
//     With the 32x32x16 MFMA the D layout puts column j = lane&31 in the lane, so the running max,
//     the running sum and the O rescale are lane-local; the only cross-lane traffic per KV tile
//     is one 32-lane swap of the row max.
//   * The k-index of the second MFMA is free to permute as long as A and B agree, so P is fed
//     straight from the S accumulator registers (8 consecutive regs -> one B operand) and the
//     matching permutation (swap bits 2<->3 of kv within each group of 16) is baked into the Vᵀ
//     layout that launch_v_transpose writes.  No ds_bpermute / permlane shuffles of P at all.
//   * K tile (64 kv x 128 d) and Vᵀ tile (128 d x 64 kv) are DMA'd HBM->LDS with
//     global_load_lds_dwordx4, double buffered (64 KiB LDS), XOR-swizzled through the source
//     address so the ds_read_b128 operand reads are bank-conflict free.
//   * Online softmax in the exp2 domain with deferred rescale (skip the O rescale while the row
//     max grows by < THR; P stays bounded by 2^THR, cdna guide T13).
#include <cstdlib>
#include <type_traits>

#include <atomic>

#include "common.h"

namespace fmi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int ATT_THREADS = 512;
constexpr int ATT_QBLK = 256;  // query rows per workgroup
constexpr int ATT_KV = 64;     // kv rows per tile
constexpr int HD = 128;
constexpr int ATT_PF = 4;  // operand reads in flight in the ping-pong kernel's MFMA phase (4/6/8 measured within 2 %)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

#if FMI_ALT_KERNELS  // the single-barrier 8-wave kernel (round 1): superseded by the ping-pong form below, kept in the test build as its bit-identical twin
template <int THR_X16>  // rescale threshold in 1/16 units of log2 (0 = always rescale)
__global__ __launch_bounds__(ATT_THREADS, 2) void attention_kernel(const bf16_t* __restrict Q, const bf16_t* __restrict K,
                                                                    const bf16_t* __restrict Vt, AttnOut out, int H, int Lq, int Lk,
                                                                    int Lkpad, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 16384];  // [buf][K 16K | Vt 16K]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, XCD-aware: block b runs on XCD b % 8, so give each XCD a contiguous range of
  // (head, q-block) ids — all q-blocks of a head then share that head's K / Vt in ONE L2 instead
  // of pulling it into all eight.
  const int nqb = (Lq + ATT_QBLK - 1) / ATT_QBLK;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqb;
  const int b = bh / H, h = bh % H;
  const int q0 = (lid % nqb) * ATT_QBLK + wave * 32;
  const int hl = lane >> 5;   // half of the wave
  const int l31 = lane & 31;  // query row within the wave / operand row

  const bf16_t* Kb = K + (int64_t)bh * Lk * HD;
  const bf16_t* Vb = Vt + (int64_t)bh * HD * Lkpad;

  // ---- Q fragments (B operand): lane holds Q[q0+l31][16s + 8hl .. +7], s = 0..7
  bf16x8_t qf[8];
  {
    int qr = q0 + l31;
    qr = qr > Lq - 1 ? Lq - 1 : qr;
    const bf16_t* qp = Q + ((int64_t)bh * Lq + qr) * HD + 8 * hl;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * s);
  }

  f32x16 ot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // per-lane LDS read offsets
  // K tile: row-major [64][128] bf16 (256 B rows), slot c -> c ^ (row & 15)
  int k_off[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) k_off[s] = l31 * 256 + (((s * 2 + hl) ^ (lane & 15)) << 4);
  // Vt tile: [128][64] bf16 (128 B rows), slot c -> c ^ ((row>>1)&7)
  int v_off[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v_off[c] = l31 * 128 + ((c ^ ((l31 >> 1) & 7)) << 4);

  // LDS: K ring [2][64x128] at 0 / 16K, Vt ring [2][128x64] at 32K / 48K
  auto stage_k = [&](int tile, int buf) {
    char* kd = smem + buf * 16384;
    const int kv0 = tile * ATT_KV;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = wave * 2 + i;  // 16 chunks of 4 K rows
      const int row = chunk * 4 + (lane >> 4);
      int kr = kv0 + row;
      kr = kr > Lk - 1 ? Lk - 1 : kr;
      const bf16_t* src = Kb + (int64_t)kr * HD + (((lane & 15) ^ (row & 15)) << 3);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(kd + chunk * 1024), 16, 0, 0);
    }
  };
  auto stage_v = [&](int tile, int buf) {
    char* vd = smem + 32768 + buf * 16384;
    const int kv0 = tile * ATT_KV;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = wave * 2 + i;  // 16 chunks of 8 Vt rows
      const int row = chunk * 8 + (lane >> 3);
      const bf16_t* src = Vb + (int64_t)row * Lkpad + kv0 + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(vd + chunk * 1024), 16, 0, 0);
    }
  };
  // Sᵀ = K Qᵀ for one 64-kv tile: two 32-kv sub-tiles, 16 MFMAs
  auto qk = [&](f32x16 (&st)[2], int buf) {
    const char* kl = smem + buf * 16384;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[u][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kl + u * 32 * 256 + k_off[s]);
        st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[u], 0, 0, 0);
      }
    }
  };

  const int ntiles = (Lk + ATT_KV - 1) / ATT_KV;
  // Software pipeline: iteration t issues the 16 QKᵀ MFMAs of tile t+1 BEFORE the softmax of tile t,
  // so the matrix pipe works through them while this wave's VALU does max/exp2/sum/convert (MFMA
  // and VALU are separate pipes; in the straight-line order every wave of the workgroup did its
  // softmax at the same time with the matrix pipe idle).  K and Vt therefore run on separate
  // double-buffered rings: at iteration t the K ring holds tiles t+1 / t+2, the Vt ring t / t+1.
  auto body = [&](int t, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
    dma_barrier();  // K(t+1), Vt(t) landed; every wave finished iteration t-1
    if (t + 2 < ntiles) stage_k(t + 2, t & 1);
    if (t + 1 < ntiles) stage_v(t + 1, (t + 1) & 1);
    if (t + 1 < ntiles) qk(sn, (t + 1) & 1);
    // ---- mask the ragged tail (kv >= Lk); kv_local = 32u + (r&3) + 8(r>>2) + 4hl
    if ((t + 1) * ATT_KV > Lk) {
      const int kvb = t * ATT_KV + 4 * hl;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kvb + 32 * u + (r & 3) + 8 * (r >> 2) >= Lk) sc[u][r] = -1e30f;
    }
    // ---- online softmax (exp2 domain), lane-local row
    float pmax = sc[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) pmax = fmaxf(pmax, sc[u][r]);
    pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
    const float ps = pmax * scale_log2e;
    if (__any(ps - m_run > (float)THR_X16 * 0.0625f)) {
      const float mn = fmaxf(m_run, ps);
      const float alpha = fast_exp2(m_run - mn);
      m_run = mn;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
    }
    float lsum = 0.f;
    bf16x8_t pf[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint32_t pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = fast_exp2(sc[u][r] * scale_log2e - m_run);
        const float p1 = fast_exp2(sc[u][r + 1] * scale_log2e - m_run);
        lsum += p0 + p1;
        pk[r >> 1] = pack_bf16x2(p0, p1);
      }
      uint4 lo = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      uint4 hi = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      __builtin_memcpy(&pf[2 * u], &lo, 16);
      __builtin_memcpy(&pf[2 * u + 1], &hi, 16);
    }
    l_run += lsum;
    // ---- Oᵀ += Vᵀ Pᵀ : k-slot (hl,e) of step (u,w) <-> Vt position 32u + 16w + 8hl + e
    const char* vl = smem + 32768 + (t & 1) * 16384;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {  // c = 2u + w
        const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(vl + dt * 32 * 128 + v_off[c * 2 + hl]);
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[c], ot[dt], 0, 0, 0);
      }
    }
  };

  f32x16 sa[2], sb[2];
  stage_k(0, 0);
  stage_v(0, 0);
  if (ntiles > 1) stage_k(1, 1);
  dma_barrier();
  qk(sa, 0);
  for (int t = 0; t < ntiles; t += 2) {
    body(t, sa, sb);
    if (t + 1 < ntiles) body(t + 1, sb, sa);
  }

  // ---- epilogue: O[q][d] = Oᵀ / l ; d = 32dt + (r&3) + 8(r>>2) + 4hl
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < Lq) {
    bf16_t* op;
    if (out.head_major)
      op = out.p1 + ((int64_t)bh * Lq + q) * HD;
    else if (q < out.rows0)
      op = out.p0 + (int64_t)b * out.bstride0 + (int64_t)q * out.ld0 + h * HD;
    else
      op = out.p1 + (int64_t)b * out.bstride1 + (int64_t)(q - out.rows0) * out.ld1 + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + g * 8 + 4 * hl;
        const uint2 v = make_uint2(pack_bf16x2(ot[dt][4 * g] * inv, ot[dt][4 * g + 1] * inv),
                                   pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv));
        *reinterpret_cast<uint2*>(op + d) = v;
      }
  }
}

#endif  // FMI_ALT_KERNELS

// ---------------------------------------------------------------------------------------------
// Ping-pong variant of the kernel above (same math, same accumulation order: bit-identical output).
//
// Per KV tile a wave has ~1040 clocks of softmax VALU work (exp2 is quarter rate) and 1024 clocks
// of MFMA work (16 QK^T + 16 PV).  With one workgroup barrier per tile both waves of a SIMD did
// their softmax at the same time (matrix pipe idle) and then their MFMAs at the same time (VALU
// idle): measured 46 % MFMA utilisation.  Here the 8 waves form two groups (waves 0-3 / 4-7, one
// of each per SIMD) that alternate barrier-delimited slots:
//     slot      2t        2t+1       2t+2       2t+3
//     group 0   V(t)      M(t)       V(t+1)     M(t+1)
//     group 1   M(t-1)    V(t)       M(t)       V(t+1)
//   V(t) = this wave's DMA pieces of K(t+3) / Vt(t+2), then the online softmax of S(t) -> P(t)
//   M(t) = PV(t) then QK^T(t+1) -> S(t+1) (the ONE S buffer: nothing overlaps inside a wave any
//          more), 32 (ds_read_b128, MFMA) steps as one software pipeline with ATT_PF operand reads
//          in flight — only this wave feeds the SIMD's matrix pipe in its slot, so an LDS latency
//          (~150 clk) in front of every 32-clk MFMA would idle it; ends with vmcnt(4) (everything
//          the wave issued before V(t) has landed; the closing barrier publishes it)
// so on every SIMD one wave keeps the VALU busy while the other keeps the matrix pipe busy.
// K and Vt live in 4-deep LDS rings (128 KiB): a piece has >= 4 slots (two tile periods) to land.
// QK8 = true (fp8 mode of the model): Q and K arrive as OCP e4m3 bytes (rows of 128 B, static scales folded
// into scale_log2e by the caller) and S^T = K Q^T runs on v_mfma_f32_32x32x64_f8f6f4: 4 MFMAs of 64 clocks per tile instead
// of 16 of 32.  The K tile is 64 x 128 B (one DMA piece per wave, 16-byte slots XOR-swizzled with row & 7); a lane's
// operand is the 32 contiguous bytes of its row half, fetched by two ds_read_b128 at the start of the M phase, ahead of
// the hand-pipelined PV reads.  Softmax, P (bf16) and PV are unchanged.
typedef __attribute__((ext_vector_type(4))) int att_i32x4;
typedef __attribute__((ext_vector_type(8))) int att_i32x8;
template <int THR_X16, bool QK8>
__global__ __launch_bounds__(ATT_THREADS, 2) void attention_pp_kernel(const bf16_t* __restrict Q, const bf16_t* __restrict K,
                                                                       const bf16_t* __restrict Vt, AttnOut out, int H, int Lq, int Lk,
                                                                       int Lkpad, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) char smem[8 * 16384];  // K ring [4][64x128] at 0, Vt ring [4][128x64] at 64K
  constexpr int VT_RING = 4 * 16384;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2;
  const int nqb = (Lq + ATT_QBLK - 1) / ATT_QBLK;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqb;
  const int b = bh / H, h = bh % H;
  const int q0 = (lid % nqb) * ATT_QBLK + wave * 32;
  const int hl = lane >> 5;
  const int l31 = lane & 31;

  const bf16_t* Kb = K + (int64_t)bh * Lk * HD;
  const bf16_t* Vb = Vt + (int64_t)bh * HD * Lkpad;

  bf16x8_t qf[8];
  att_i32x8 qq[2];  // QK8: the lane's 32 bytes of each 64-wide d step
  {
    int qr = q0 + l31;
    qr = qr > Lq - 1 ? Lq - 1 : qr;
    if constexpr (QK8) {
      const uint8_t* qp = reinterpret_cast<const uint8_t*>(Q) + ((int64_t)bh * Lq + qr) * HD + 32 * hl;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        qq[s] = __builtin_shufflevector(*reinterpret_cast<const att_i32x4*>(qp + 64 * s), *reinterpret_cast<const att_i32x4*>(qp + 64 * s + 16), 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      const bf16_t* qp = Q + ((int64_t)bh * Lq + qr) * HD + 8 * hl;
#pragma unroll
      for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * s);
    }
  }
  f32x16 ot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // LDS operand offsets.  The swizzles are XORs of disjoint bit fields, so the offset of k-step s /
  // slot c is the step-0 offset XOR an immediate: 2 registers instead of 16.
  //   K tile  [64][128] bf16: row l31, 16-B slot (2s + hl) ^ (lane & 15)  ==  k_off0 ^ (s << 5)
  //   Vt tile [128][64] bf16: row l31, 16-B slot (2c' + hl) ^ ((l31>>1)&7) ==  v_off0 ^ (c' << 5)
  const int k_off0 = l31 * 256 + ((hl ^ (lane & 15)) << 4);
  //   K tile  [64][128] e4m3 (QK8): row l31, 16-B slots (4s + 2hl + j) ^ (l31 & 7)  ==  k8_off0 ^ (s << 6) ^ (j << 4)
  const int k8_off0 = l31 * 128 + (((2 * hl) ^ (l31 & 7)) << 4);
  const int v_off0 = l31 * 128 + ((hl ^ ((l31 >> 1) & 7)) << 4);

  auto stage_k = [&](int tile) {
    char* kd = smem + (tile & 3) * 16384;
    const int kv0 = tile * ATT_KV;
    if constexpr (QK8) {  // 64 rows x 128 B = 8 pieces of 8 rows: one per wave
      const int row = wave * 8 + (lane >> 3);
      int kr = kv0 + row;
      kr = kr > Lk - 1 ? Lk - 1 : kr;
      const uint8_t* src = reinterpret_cast<const uint8_t*>(K) + ((int64_t)bh * Lk + kr) * HD + (((lane & 7) ^ (row & 7)) << 4);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(kd + wave * 1024), 16, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = wave * 2 + i;
      const int row = chunk * 4 + (lane >> 4);
      int kr = kv0 + row;
      kr = kr > Lk - 1 ? Lk - 1 : kr;
      const bf16_t* src = Kb + (int64_t)kr * HD + (((lane & 15) ^ (row & 15)) << 3);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(kd + chunk * 1024), 16, 0, 0);
    }
  };
  auto stage_v = [&](int tile) {
    char* vd = smem + VT_RING + (tile & 3) * 16384;
    const int kv0 = tile * ATT_KV;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = wave * 2 + i;
      const int row = chunk * 8 + (lane >> 3);
      const bf16_t* src = Vb + (int64_t)row * Lkpad + kv0 + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(vd + chunk * 1024), 16, 0, 0);
    }
  };
  auto slot_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int ntiles = (Lk + ATT_KV - 1) / ATT_KV;
  // ---- prologue: K(0..2), Vt(0..1) in flight; S(0) computed by everyone
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < ntiles) stage_k(t);
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < ntiles) stage_v(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  slot_barrier();
  f32x16 sc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[u][r] = 0.f;
    if constexpr (QK8) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const att_i32x8 kf = __builtin_shufflevector(*reinterpret_cast<const att_i32x4*>(smem + u * 32 * 128 + (k8_off0 ^ (s << 6))),
                                                     *reinterpret_cast<const att_i32x4*>(smem + u * 32 * 128 + (k8_off0 ^ (s << 6) ^ 16)), 0, 1, 2, 3, 4, 5, 6, 7);
        sc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qq[s], sc[u], 0, 0, 0, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(smem + u * 32 * 256 + (k_off0 ^ (s << 5)));
        sc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc[u], 0, 0, 0);
      }
    }
  }
  if (g == 1) slot_barrier();  // group 1 runs one slot behind

  bf16x8_t pf[4];
  auto body = [&](int t, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;  // a tile t+1 exists
    // ================= V(t): DMA issue + online softmax of S(t) -> pf
    if (t + 3 < ntiles) stage_k(t + 3);
    if (t + 2 < ntiles) stage_v(t + 2);
    __builtin_amdgcn_sched_barrier(0);
    if ((t + 1) * ATT_KV > Lk) {
      const int kvb = t * ATT_KV + 4 * hl;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kvb + 32 * u + (r & 3) + 8 * (r >> 2) >= Lk) sc[u][r] = -1e30f;
    }
    float pmax = sc[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) pmax = fmaxf(pmax, sc[u][r]);
    pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
    const float ps = pmax * scale_log2e;
    if (__any(ps - m_run > (float)THR_X16 * 0.0625f)) {
      const float mn = fmaxf(m_run, ps);
      const float alpha = fast_exp2(m_run - mn);
      m_run = mn;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
    }
    float lsum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint32_t pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = fast_exp2(sc[u][r] * scale_log2e - m_run);
        const float p1 = fast_exp2(sc[u][r + 1] * scale_log2e - m_run);
        lsum += p0 + p1;
        pk[r >> 1] = pack_bf16x2(p0, p1);
      }
      uint4 lo = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      uint4 hi = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      __builtin_memcpy(&pf[2 * u], &lo, 16);
      __builtin_memcpy(&pf[2 * u + 1], &hi, 16);
    }
    l_run += lsum;
    slot_barrier();
    // ================= M(t): PV(t) (steps 0..15), then QK^T(t+1) (steps 16..31) into sc
    {
      const char* vl = smem + VT_RING + (t & 3) * 16384;
      const char* kl = smem + ((t + 1) & 3) * 16384;
      constexpr int NSTEP = (MORE && !QK8) ? 32 : 16;
      bf16x8_t fr[ATT_PF];
      // QK8: the four K(t+1) operands (2 key blocks x 2 d steps) are requested first; they are older than every PV read, so
      // the in-order lgkmcnt arithmetic below is unchanged, and they have the whole PV phase to land
      att_i32x8 kq[2][2];
      if constexpr (QK8 && MORE) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            kq[u][s2] = __builtin_shufflevector(*reinterpret_cast<const att_i32x4*>(kl + u * 32 * 128 + (k8_off0 ^ (s2 << 6))),
                                                *reinterpret_cast<const att_i32x4*>(kl + u * 32 * 128 + (k8_off0 ^ (s2 << 6) ^ 16)), 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_sched_barrier(0);
      }
      // Step order: consecutive MFMAs never hit the same accumulator (a dependent 32x32 MFMA cannot
      // issue until the previous one has drained: 4- and 8-long chains ran the phase at half rate).
      // PV steps walk dt fastest (4 accumulators), QK steps walk u fastest (2 accumulators); the
      // k-order inside every accumulator is unchanged, so results stay bit-identical.
      // The reads are inline asm with hand-placed `s_waitcnt lgkmcnt(ATT_PF-1)`: hipcc's own waitcnt
      // insertion waits for lgkmcnt(0) — i.e. for the read it has just issued — every few steps,
      // which exposes a full LDS latency per ATT_PF MFMAs (measured 62 clocks per MFMA instead of 32).
      // LDS reads retire in order, so before MFMA i at most (reads issued) - (i+1) may be pending.
      const uint32_t vbase = (uint32_t)(uintptr_t)(lds_void*)vl, kbase = (uint32_t)(uintptr_t)(lds_void*)kl;
      auto load = [&](int i, bf16x8_t& dst) {  // i is a compile-time constant after unrolling
        const uint32_t a = i < 16 ? vbase + (i & 3) * 32 * 128 + (v_off0 ^ ((i >> 2) << 5)) : kbase + ((i - 16) & 1) * 32 * 256 + (k_off0 ^ (((i - 16) >> 1) << 5));
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(a));
      };
#pragma unroll
      for (int i = 0; i < ATT_PF; ++i) load(i, fr[i]);
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) {
        {
          const int pending = (i + ATT_PF < NSTEP ? ATT_PF : NSTEP - i) - 1;  // reads issued after read i
          bf16x8_t& f = fr[i % ATT_PF];
          // the "+v" ties the wait to the fragment: the MFMA below cannot be scheduled above it
          if (pending >= 7) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(f));
          else if (pending == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f));
          else if (pending == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f));
          else if (pending == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f));
          else if (pending == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f));
          else if (pending == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f));
          else if (pending == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f));
        }
        if (i < 16) {
          ot[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % ATT_PF], pf[i >> 2], ot[i & 3], 0, 0, 0);
        } else {
          const int u = (i - 16) & 1, sq = (i - 16) >> 1;
          if (sq == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[u][r] = 0.f;
          }
          sc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % ATT_PF], qf[sq], sc[u], 0, 0, 0);
        }
        if (i + ATT_PF < NSTEP) load(i + ATT_PF, fr[i % ATT_PF]);
        __builtin_amdgcn_sched_barrier(0);  // keep the read-ahead distance: one read issued per MFMA
      }
      if constexpr (QK8 && MORE) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[u][r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int u = 0; u < 2; ++u) sc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kq[u][s2], qq[s2], sc[u], 0, 0, 0, 0, 0, 0);
      }
    }
    // everything older than the pieces issued in this tile's V phase must have landed
    if (t + 3 < ntiles) {
      if constexpr (QK8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // K(t+3) is one piece per wave here
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    else if (t + 2 < ntiles)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    slot_barrier();
  };
  for (int t = 0; t < ntiles - 1; ++t) body(t, std::true_type{});
  body(ntiles - 1, std::false_type{});
  if (g == 0) slot_barrier();  // group 0 finished one slot early

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < Lq) {
    bf16_t* op;
    if (out.head_major)
      op = out.p1 + ((int64_t)bh * Lq + q) * HD;
    else if (q < out.rows0)
      op = out.p0 + (int64_t)b * out.bstride0 + (int64_t)q * out.ld0 + h * HD;
    else
      op = out.p1 + (int64_t)b * out.bstride1 + (int64_t)(q - out.rows0) * out.ld1 + h * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = dt * 32 + gq * 8 + 4 * hl;
        const uint2 v = make_uint2(pack_bf16x2(ot[dt][4 * gq] * inv, ot[dt][4 * gq + 1] * inv),
                                   pack_bf16x2(ot[dt][4 * gq + 2] * inv, ot[dt][4 * gq + 3] * inv));
        *reinterpret_cast<uint2*>(op + d) = v;
      }
  }
}

}  // namespace fmi
#include "attention_w4.h"  // (product: the key-split launches of the sequence-parallel split-K mode)
#if FMI_ALT_KERNELS
#include "attention_w16.h"
#include "attention_w32.h"
#endif
#include "attention_w16l.h"
namespace fmi {

bool alt_kernels_built() { return FMI_ALT_KERNELS != 0; }
static std::atomic<bool> g_att_pingpong{true};  // process-wide test hooks, like the GEMM switches (gemm_bf16.hip)
void set_attention_pingpong(bool on) { g_att_pingpong = on; }
// bf16 operands: the one-wave-per-SIMD kernel (attention_w4.h); FMI_ATT_W4=0 / set_attention_w4(false) -> the 8-wave ping-pong kernel
static std::atomic<bool> g_att_w4{[] {
  const char* e = getenv("FMI_ATT_W4");
  return e ? atoi(e) != 0 : true;
}()};
void set_attention_w4(bool on) { g_att_w4 = on; }
// bf16 operands, round 3: the 16x16x32-MFMA one-wave kernel (attention_w16.h) in front of all of them; FMI_ATT_W16=0 /
// set_attention_w16(false) falls back to the selection above.  Not bit-identical to the others (Q pre-scaled, row sums of the rounded P).
static std::atomic<bool> g_att_w16{[] {
  const char* e = getenv("FMI_ATT_W16");
  return e ? atoi(e) != 0 : true;
}()};
void set_attention_w16(bool on) { g_att_w16 = on; }
// the same design on the 32x32x16 MFMA (attention_w32.h), bit-identical to attention_w16: off by default (in the denoise loop the
// 16x16x32 form measured 13.17 vs 13.32 ms of attention per step, profiles/r03_attention_ab.txt); FMI_ATT_W32=1 / set_attention_w32(true)
static std::atomic<bool> g_att_w32{[] {
  const char* e = getenv("FMI_ATT_W32");
  return e ? atoi(e) != 0 : false;
}()};
void set_attention_w32(bool on) { g_att_w32 = on; }
// bf16 operands, round 4: the lock-step schedule of the 16x16x32 kernel (attention_w16l.h) in front of all of them — the default;
// FMI_ATT_W16L=0 / set_attention_w16l(false) falls back to the selection above.  Bit-identical to attention_w16 / _w32 at rescale
// threshold 0, equal to rounding at the default threshold.
static std::atomic<bool> g_att_w16l{[] {
  const char* e = getenv("FMI_ATT_W16L");
  return e ? atoi(e) != 0 : true;
}()};
void set_attention_w16l(bool on) { g_att_w16l = on; }

static std::atomic<unsigned long long> g_fp8_fallbacks{0};  // fp8-QK^T launches that did not get the one-wave stream
unsigned long long attention_fp8_fallbacks() { return g_fp8_fallbacks.load(std::memory_order_relaxed); }

int launch_attention_ex(const bf16_t* q, const bf16_t* k, const bf16_t* vt, const AttnOut& out, int B, int H, int Lq, int Lk,
                        int Lkpad, float scale, int rescale_thr_x16, hipStream_t stream, int qk_fp8, float* lse, int k_hstride, int score_exp2,
                        int kind, float v_inv) {  // (k_hstride: see the lse branch)
  // which kernel: the caller's choice (a model handle's fmi_flux_set_attention_kernel, kind 0..5) or, kind < 0, the process-wide switches
  const bool g_att_w16l = kind >= 0 ? kind == 5 : (bool)fmi::g_att_w16l, g_att_w32 = kind >= 0 ? kind == 4 : (bool)fmi::g_att_w32,
             g_att_w16 = kind >= 0 ? kind >= 3 : (bool)fmi::g_att_w16, g_att_w4 = kind >= 0 ? kind >= 2 : (bool)fmi::g_att_w4,
             g_att_pingpong = kind >= 0 ? kind >= 1 : (bool)fmi::g_att_pingpong;
  if (Lq <= 0 || Lk <= 0) return fail(FMI_ERR_INVALID, "attention: empty sequence");
  if (Lkpad % ATT_KV != 0 || Lkpad < Lk) return fail(FMI_ERR_INVALID, "attention: Lkpad must be a multiple of 64 and >= Lk");
#if !FMI_ALT_KERNELS
  // the product build carries the lock-step kernel (5) and the 8-wave ping-pong kernel (1: single-tile problems, fp8 factors that are no power of two)
  if (!g_att_pingpong || g_att_w32 || (g_att_w16 != g_att_w16l) || (g_att_w4 != g_att_w16l))
    return fail(FMI_ERR_UNSUPPORTED, "attention: this build carries kernels 5 and 1 only (the others live in the test build, libflux_mi355x_alt.so: make alt)");
#endif
  dim3 grid(cdiv(Lq, ATT_QBLK) * B * H);
  const float sl = scale * 1.4426950408889634f;
  if (lse) {  // key-split launch (k_hstride = number of key ranges): only the one-wave kernel writes the log-sum-exp
    const int nsplit = k_hstride;
    if (qk_fp8 || B != 1 || out.head_major || out.rows0 != 0 || nsplit < 2 || cdiv(Lk, ATT_KV) < 2 * nsplit)
      return fail(FMI_ERR_UNSUPPORTED, "attention: a key-split launch needs bf16 operands, B = 1, one token-major output and >= 2 KV tiles per part");
    const dim3 gs(grid.x * nsplit);
    if (rescale_thr_x16 == 0)
      FMI_LAUNCH_LDS((attention_w4_kernel<0, true>), 8 * 16384, gs, dim3(AW4_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl, Lk, lse, nsplit);
    else
      FMI_LAUNCH_LDS((attention_w4_kernel<96, true>), 8 * 16384, gs, dim3(AW4_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl, Lk, lse, nsplit);
    FMI_LAUNCH_CHECK();
    return FMI_OK;
  }
  if (qk_fp8) {
    // round 3: the one-wave kernel's fp8-QK^T stream (attention_w16.h, QK8) carries scale * log2(e) / (sq * sk) as an E8M0 block
    // scale of the score MFMA, so it serves the calls whose factor is a power of two 2^-n, n = 0 .. 126 — the model's fp8 mode
    // picks its q scale that way (flux_model.hip: fp8_q_scale_pow2); anything else runs on the 8-wave kernel below
    // The exponent comes from the caller as an integer when it has one (score_exp2: the model, fmi_sdpa_fp8qk_ws).  A bare float
    // (fmi_sdpa_fp8qk) qualifies if scale * log2(e), rounded to f32 here, is a power of two or ONE ulp beside one: a caller who builds
    // scale = 2^n / log2(e) in f32 lands there by construction — anything further off was not constructed as one.
    int n2 = 1;
    if (score_exp2 != ATT_NO_EXP2) n2 = score_exp2;
    else if (sl > 0.f) {
      int e;
      const float mant = frexpf(sl, &e);  // in [0.5, 1)
      if (mant == 0.5f || mant == nextafterf(0.5f, 1.0f)) n2 = e - 1;
      else if (mant == nextafterf(1.0f, 0.0f)) n2 = e;
    }
    const bool pow2 = n2 <= 0 && n2 >= -126;
    // counted (fmi_device_info: fp8_attention_fallbacks): launches that WOULD take a one-wave stream — that kernel family is selected and
    // there is more than one KV tile — but whose factor is not a power of two.  A handle or user that picked kernel 0..2 chose the 8-wave
    // kernel; a single-tile problem has no stream to fall back from.
    if (qk_fp8 == 2) {  // round 5: e4m3 P and V^T as well — the lock-step stream's third form, and the only kernel that reads this V^T
      if (!(g_att_w16l && g_att_w16) || Lk <= ATT_KV || !pow2)
        return fail(FMI_ERR_UNSUPPORTED, "attention: e4m3 P / V needs the lock-step kernel (kind 5), more than one KV tile and a power-of-two score factor");
      const float sl2 = ldexpf(1.0f, n2);
      if (rescale_thr_x16 == 0)
        FMI_LAUNCH_LDS((attention_w16l_kernel<0, true, true>), 8 * 16384, grid, dim3(AW16L_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl2, v_inv);
      else
        FMI_LAUNCH_LDS((attention_w16l_kernel<96, true, true>), 8 * 16384, grid, dim3(AW16L_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl2, v_inv);
      FMI_LAUNCH_CHECK();
      return FMI_OK;
    }
    if (g_att_w16 && Lk > ATT_KV && !pow2) g_fp8_fallbacks.fetch_add(1, std::memory_order_relaxed);
    if (g_att_w16l && g_att_w16 && Lk > ATT_KV && pow2) {  // round 4: the lock-step schedule's fp8-QK^T stream
      const float sl2 = ldexpf(1.0f, n2);
      if (rescale_thr_x16 == 0)
        FMI_LAUNCH_LDS((attention_w16l_kernel<0, true>), 8 * 16384, grid, dim3(AW16L_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl2);
      else
        FMI_LAUNCH_LDS((attention_w16l_kernel<96, true>), 8 * 16384, grid, dim3(AW16L_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl2);
      FMI_LAUNCH_CHECK();
      return FMI_OK;
    }
#if FMI_ALT_KERNELS
    if (g_att_w16 && Lk > ATT_KV && pow2) {
      const float sl2 = ldexpf(1.0f, n2);
      if (rescale_thr_x16 == 0)
        FMI_LAUNCH_LDS((attention_w16_kernel<0, true>), 8 * 16384, grid, dim3(AW16_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl2);
      else
        FMI_LAUNCH_LDS((attention_w16_kernel<96, true>), 8 * 16384, grid, dim3(AW16_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl2);
      FMI_LAUNCH_CHECK();
      return FMI_OK;
    }
#endif
    if (rescale_thr_x16 == 0)
      hipLaunchKernelGGL((attention_pp_kernel<0, true>), grid, dim3(ATT_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
    else
      hipLaunchKernelGGL((attention_pp_kernel<96, true>), grid, dim3(ATT_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
  } else if (g_att_w16l && Lk > ATT_KV) {  // (a single KV tile has no steady state to pipeline: the 8-wave kernel serves it)
    if (rescale_thr_x16 == 0)
      FMI_LAUNCH_LDS((attention_w16l_kernel<0>), 8 * 16384, grid, dim3(AW16L_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
    else
      FMI_LAUNCH_LDS((attention_w16l_kernel<96>), 8 * 16384, grid, dim3(AW16L_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
#if FMI_ALT_KERNELS
  } else if (g_att_w32 && Lk > ATT_KV) {
    if (rescale_thr_x16 == 0)
      FMI_LAUNCH_LDS((attention_w32_kernel<0>), 8 * 16384, grid, dim3(AW32_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
    else
      FMI_LAUNCH_LDS((attention_w32_kernel<96>), 8 * 16384, grid, dim3(AW32_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
  } else if (g_att_w16 && Lk > ATT_KV) {
    if (rescale_thr_x16 == 0)
      FMI_LAUNCH_LDS((attention_w16_kernel<0>), 8 * 16384, grid, dim3(AW16_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
    else
      FMI_LAUNCH_LDS((attention_w16_kernel<96>), 8 * 16384, grid, dim3(AW16_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
  } else if (g_att_w4 && Lk > ATT_KV) {
    if (rescale_thr_x16 == 0)
      FMI_LAUNCH_LDS((attention_w4_kernel<0>), 8 * 16384, grid, dim3(AW4_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl, 0, nullptr, 1);
    else
      FMI_LAUNCH_LDS((attention_w4_kernel<96>), 8 * 16384, grid, dim3(AW4_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl, 0, nullptr, 1);
#endif
  } else if (g_att_pingpong) {
    if (rescale_thr_x16 == 0)
      hipLaunchKernelGGL((attention_pp_kernel<0, false>), grid, dim3(ATT_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
    else
      hipLaunchKernelGGL((attention_pp_kernel<96, false>), grid, dim3(ATT_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
  }
#if FMI_ALT_KERNELS
  else if (rescale_thr_x16 == 0)
    hipLaunchKernelGGL(attention_kernel<0>, grid, dim3(ATT_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
  else
    hipLaunchKernelGGL(attention_kernel<96>, grid, dim3(ATT_THREADS), 0, stream, q, k, vt, out, H, Lq, Lk, Lkpad, sl);
#endif
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int H, int Lq, int Lk, int Lkpad,
                     float scale, int out_token_major, hipStream_t stream) {
  AttnOut out{};
  out.p0 = nullptr;
  out.rows0 = 0;
  out.p1 = o;
  out.ld1 = H * HD;
  out.bstride1 = (int64_t)Lq * H * HD;
  out.head_major = out_token_major ? 0 : 1;
  return launch_attention_ex(q, k, vt, out, B, H, Lq, Lk, Lkpad, scale, 96, stream);
}

// ------------------------------------------------------------------------------------------
// V -> Vᵀ relayout.  vt[(bh*128 + d) * Lpad + pos] = V[b][row(pos) - row_off][h*128 + d] with
// pos <-> kv: swap bits 2 and 3 of the index inside each group of 16 (see header comment).
// Grid: (Lpad/64 groups restricted to the touched range, H, B); block 256.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int vt_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

__global__ __launch_bounds__(256) void v_transpose_kernel(const bf16_t* __restrict v, int ldv, int64_t v_bstride, bf16_t* __restrict vt,
                                                          int H, int rows, int row_off, int Lpad, int g0) {
  __shared__ bf16_t tile[64][HD + 2];
  const int g = g0 + blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int pos0 = g * 64;
  // load: 64 positions x 128 d; thread -> (pos = tid>>2, 32 d each)
  {
    const int p = tid >> 2, dq = (tid & 3) * 32;
    const int kv = vt_perm(pos0 + p);  // original kv index stored at this position
    const int r = kv - row_off;
    if (r >= 0 && r < rows) {
      const bf16_t* src = v + (int64_t)b * v_bstride + (int64_t)r * ldv + h * HD + dq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 x = *reinterpret_cast<const uint4*>(src + i * 8);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&x);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[p][dq + i * 8 + j] = e[j];
      }
    }
  }
  __syncthreads();
  // store: thread -> (d = tid>>1, 32 positions)
  const int d = tid >> 1, ph = (tid & 1) * 32;
  bf16_t* dst = vt + ((int64_t)(b * H + h) * HD + d) * Lpad + pos0 + ph;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16_t e[8];
    bool all = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = ph + i * 8 + j;
      const int r = vt_perm(pos0 + p) - row_off;
      const bool ok = (r >= 0 && r < rows);
      all = all && ok;
      e[j] = tile[p][d];
    }
    if (all) {
      uint4 x;
      __builtin_memcpy(&x, e, 16);
      *reinterpret_cast<uint4*>(dst + i * 8) = x;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = vt_perm(pos0 + ph + i * 8 + j) - row_off;
        if (r >= 0 && r < rows) dst[i * 8 + j] = e[j];
      }
    }
  }
}

int launch_v_transpose(const bf16_t* v, int ldv, int64_t v_bstride, bf16_t* vt, int B, int H, int rows, int row_off, int Lpad,
                       hipStream_t stream) {
  if (rows <= 0) return FMI_OK;
  if (ldv % 8) return fail(FMI_ERR_INVALID, "v_transpose: ldv must be a multiple of 8");
  const int g0 = row_off / 64, g1 = (row_off + rows - 1) / 64;
  hipLaunchKernelGGL(v_transpose_kernel, dim3(g1 - g0 + 1, H, B), dim3(256), 0, stream, v, ldv, v_bstride, vt, H, rows, row_off, Lpad, g0);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

// V (BH, Lk, 128) bf16 -> V^T (BH, 128, Lpad) e4m3: block = 64 keys of one head; thread -> (key = tid >> 2, 32 d) in, (d = tid >> 1, 32 keys) out
__global__ __launch_bounds__(256) void v_transpose_fp8_kernel(const bf16_t* __restrict v, uint8_t* __restrict vt8, int Lk, int Lpad, float v_scale) {
  __shared__ uint8_t tile[128][64 + 4];
  const int bh = blockIdx.y, pos0 = blockIdx.x * 64, tid = threadIdx.x;
  {
    const int p = tid >> 2, dq = (tid & 3) * 32;
    const int kv = pos0 + p;
    if (kv < Lk) {
      const bf16_t* src = v + ((int64_t)bh * Lk + kv) * HD + dq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 x = *reinterpret_cast<const uint4*>(src + i * 8);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&x);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float a = fminf(fmaxf(bf16_to_f32(e[j]) * v_scale, -448.f), 448.f), b = fminf(fmaxf(bf16_to_f32(e[j + 1]) * v_scale, -448.f), 448.f);
          const int c = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
          tile[dq + i * 8 + j][p] = (uint8_t)(c & 0xff);
          tile[dq + i * 8 + j + 1][p] = (uint8_t)((c >> 8) & 0xff);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) tile[dq + j][p] = 0;
    }
  }
  __syncthreads();
  const int d = tid >> 1, ph = (tid & 1) * 32;
  uint8_t* dst = vt8 + ((int64_t)bh * HD + d) * Lpad + pos0 + ph;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint4 x;
    uint8_t e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) e[j] = tile[d][ph + i * 16 + j];
    __builtin_memcpy(&x, e, 16);
    *reinterpret_cast<uint4*>(dst + i * 16) = x;
  }
}
int launch_v_transpose_fp8(const bf16_t* v, uint8_t* vt8, int BH, int Lk, int Lpad, float v_scale, hipStream_t stream) {
  if (BH <= 0 || Lk <= 0) return FMI_OK;
  if (Lpad % 64 || Lpad < Lk) return fail(FMI_ERR_INVALID, "v_transpose_fp8: Lpad must be a multiple of 64 and >= Lk");
  hipLaunchKernelGGL(v_transpose_fp8_kernel, dim3(Lpad / 64, BH), dim3(256), 0, stream, v, vt8, Lk, Lpad, v_scale);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

__global__ void vt_zero_pad_kernel(bf16_t* vt, int L, int Lpad) {
  const int row = blockIdx.x;  // bh*128 + d
  for (int p = L + threadIdx.x; p < Lpad; p += blockDim.x) {
    // positions whose ORIGINAL kv index is >= L must be zero
    vt[(int64_t)row * Lpad + p] = 0;
  }
}
// Zero every position of the last (partial) 64-group whose source kv index is >= L.
__global__ void vt_zero_tail_kernel(bf16_t* vt, int L, int Lpad) {
  const int row = blockIdx.x;
  const int p0 = (L / 16) * 16;
  for (int p = p0 + threadIdx.x; p < Lpad; p += blockDim.x)
    if (vt_perm(p) >= L) vt[(int64_t)row * Lpad + p] = 0;
}
int launch_vt_zero_pad(bf16_t* vt, int BH, int L, int Lpad, hipStream_t stream) {
  if (Lpad == L) return FMI_OK;
  hipLaunchKernelGGL(vt_zero_tail_kernel, dim3(BH * HD), dim3(64), 0, stream, vt, L, Lpad);
  FMI_LAUNCH_CHECK();
  return FMI_OK;
}

}  // namespace fmi
