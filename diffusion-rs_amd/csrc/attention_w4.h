// attention_w4.h — joint attention, one wave per SIMD (included by attention.hip; same translation unit).
//
// Same math as attention_pp_kernel (transposed flash attention: S^T = K Q^T, O^T = V^T P^T, a lane owns query
// columns, exp2-domain online softmax with deferred rescale, P fed to the second MFMA straight from the S
// registers through the k-permutation baked into V^T) and the same accumulation order per output element, so the
// result is bit-identical — but a different machine mapping:
//
//   * 4 waves, one per SIMD, each with the whole 512-register file: a wave owns 64 query rows (two 32-row blocks
//     b = 0, 1) of the workgroup's 256.  (The two blocks run half a tile apart, so a K / V^T fragment is still read once
//     per block: the LDS fragment traffic per FLOP is that of the 8-wave kernels.  Halving it was measured to be worth
//     3 %, DESIGN 4.4: the kernel is bound by instruction issue, not by LDS.)
//   * No second wave on the SIMD to overlap with, so the overlap is built inside the wave's own instruction stream
//     (MI355X_MICROARCH.md: one wave per SIMD hides ~5 single-issue instructions behind each 32x32x16 MFMA): a KV
//     tile is two phases of 32 MFMAs,
//         A(t):  MFMA  QK^T(b=1, t)  and  PV(b=1, t-1)       VALU  softmax of S(b=0, t)  -> P(0, t)
//         B(t):  MFMA  PV(b=0, t)    and  QK^T(b=0, t+1)     VALU  softmax of S(b=1, t)  -> P(1, t)
//     so the MFMAs of a phase never depend on the softmax running beside them (it belongs to the other query block),
//     and every gap between two MFMAs carries one fragment read plus a slice of the softmax: 8 steps of two
//     three-input max, the row-max exchange / rescale decision, then 16 pairs of (fma, fma | exp2, exp2 | add, pack, add).
//     In the generated steady-state loop the slices are levelled to ~6 instructions per gap (an MFMA occupies the pipe
//     for 32 clocks: a gap that needs more stalls it) and two pairs are in flight, because a v_exp_f32 result read two
//     or three instructions later is stale in half of the lanes (tools/gen_attention_w4_loop.py).
//   * The (fragment read -> MFMA) stream is continuous across phases: the read for MFMA i + 7 is issued behind MFMA i,
//     whichever phase it belongs to, so no phase starts with an exposed LDS latency.  It lands in the buffer MFMA i - 1
//     consumed, never in MFMA i's own operand (nothing orders the LDS return behind a queued MFMA's operand read).
//   * K (64 keys x 128 d) and V^T (128 d x 64 keys) tiles arrive by LDS-DMA into 4-deep rings (128 KiB), K three
//     tiles ahead, V^T two; ONE barrier per KV tile, in the middle of A(t): it publishes K(t+1) / V^T(t) half a phase
//     before their first read and, with four slots, a slot is rewritten a whole tile after its last read; waits are
//     counted (vmcnt(8)).
//   * Register files are assigned by hand (inline-asm MFMAs): O^T and the Q fragments in the accumulator half
//     (128 + 64 AGPRs), S^T, P and the streamed fragments in the architectural VGPRs the softmax VALU works on.
//   * O leaves through LDS as whole 256-byte rows (16-byte stores, four rows per wave-instruction) instead of
//     8-byte pieces at a row stride.
#pragma once
#include "attention_w4_loop.inc"

namespace fmi {

constexpr int AW4_THREADS = 256;

struct Aw4Phase {  // one phase of the MFMA stream: PV of query block b_pv and / or QK^T of query block b_qk
  bool has_pv, has_qk;
  int b_pv, b_qk;
  __device__ int n() const { return (has_pv && has_qk) ? 32 : (has_pv || has_qk) ? 16 : 0; }
};

// LSE = true (key-split launches of the sequence-parallel latency mode, flux_model.hip: attention_sp): K may be a range of a
// longer per-head sequence (`k_hstride` rows per head) and the kernel also writes the row's log2-sum-exp, m + log2(l), to
// lse[bh * Lq + q], from which sp_merge_splits combines the partial outputs.  The default instantiation is unchanged.
template <int THR_X16, bool LSE = false>
__global__ __launch_bounds__(AW4_THREADS, 1) void attention_w4_kernel(const bf16_t* __restrict Q, const bf16_t* __restrict K, const bf16_t* __restrict Vt,
                                                                      AttnOut out, int H, int Lq, int Lk, int Lkpad, float scale_log2e, int k_hstride = 0,
                                                                      float* __restrict lse = nullptr, int nsplit = 1) {
  constexpr int TILE = 16384, VT_RING = 4 * TILE, PF = 8;  // PF fragment buffers, PF - 1 reads in flight; PF divides the phase lengths, so buffer i % PF lines up across phases
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE];  // K ring [4][64 x 128] at 0, V^T ring [4][128 x 64] at 64 KiB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (Lq + ATT_QBLK - 1) / ATT_QBLK;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  if (LSE) {
    // key-split launch: the grid is nsplit x (heads x query blocks); part `sp` walks the KV tiles [nt * sp / nsplit, nt * (sp + 1) / nsplit)
    // of the full sequence (Lk = its length on entry, k_hstride = rows per head of K), writes its normalised output to slice sp of
    // out.p1 (slices of Lq * out.ld1 elements) and its log-sum-exp to slice sp of lse
    const int per = gridDim.x / nsplit, sp = lid / per, nt = (Lk + ATT_KV - 1) / ATT_KV;
    lid -= sp * per;
    const int t0 = (int)((int64_t)nt * sp / nsplit), t1 = (int)((int64_t)nt * (sp + 1) / nsplit);
    const int k0 = t0 * ATT_KV, k1 = min(Lk, t1 * ATT_KV);
    K += (int64_t)k0 * HD, Vt += k0, Lk = k1 - k0;
    out.p1 += (int64_t)sp * Lq * out.ld1;
    lse += (int64_t)sp * (per / nqb) * Lq;
  }
  const int bh = lid / nqb;
  const int b_ = bh / H, h = bh % H;
  const int q0 = (lid % nqb) * ATT_QBLK + wave * 64;
  const int hl = lane >> 5, l31 = lane & 31;
  const bf16_t* Kb = K + (int64_t)bh * (LSE ? k_hstride : Lk) * HD;
  const bf16_t* Vb = Vt + (int64_t)bh * HD * Lkpad;
  const int ntiles = (Lk + ATT_KV - 1) / ATT_KV;  // >= 2 (the launcher sends single-tile problems to the 8-wave kernel)

  // ---- Q fragments (MFMA B operand): block b, d-step s -> Q[q0 + 32 b + l31][16 s + 8 hl .. + 7]
  typedef __attribute__((ext_vector_type(4))) int frag_t;
  frag_t qf[2][8];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int qr = min(q0 + 32 * b + l31, Lq - 1);
    const bf16_t* qp = Q + ((int64_t)bh * Lq + qr) * HD + 8 * hl;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[b][s] = *reinterpret_cast<const frag_t*>(qp + 16 * s);
  }

  // ---- LDS-DMA: 16 one-KiB chunks per tile and operand, 4 per wave (same image and swizzle as attention_pp_kernel).
  // Per-lane byte offsets inside a tile are loop invariants; a tile index past the end is clamped by the caller (the last
  // tile is fetched again into a free slot: branch-free issue, constant vmcnt arithmetic).
  uint32_t k_voff[4], v_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int chunk = wave * 4 + i;
    const int kr = chunk * 4 + (lane >> 4), vr = chunk * 8 + (lane >> 3);
    k_voff[i] = (uint32_t)(kr * 256 + (((lane & 15) ^ (kr & 15)) << 4));
    v_voff[i] = (uint32_t)(vr * Lkpad * 2 + (((lane & 7) ^ ((vr >> 1) & 7)) << 4));
  }
  const int k_last_rows = Lk - (ntiles - 1) * ATT_KV;  // keys in the last tile (1..64): rows beyond are clamped to the last key
  auto stage_k = [&](int tile, int i) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(Kb) + (int64_t)tile * (ATT_KV * 256);
    uint32_t off = k_voff[i];
    if (tile == ntiles - 1 && k_last_rows < ATT_KV) {
      const int kr = (wave * 4 + i) * 4 + (lane >> 4);
      if (kr >= k_last_rows) off = (uint32_t)((k_last_rows - 1) * 256 + (((lane & 15) ^ (kr & 15)) << 4));
    }
    __builtin_amdgcn_global_load_lds((glb_void*)(base + off), (lds_void*)(smem + (tile & 3) * TILE + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  auto stage_v = [&](int tile, int i) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(Vb) + (int64_t)tile * (ATT_KV * 2);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + v_voff[i]), (lds_void*)(smem + VT_RING + (tile & 3) * TILE + (wave * 4 + i) * 1024), 16, 0, 0);
  };
  // Operand read addresses (image of attention_pp_kernel): K fragment (key half u, d-step s) at k_ad[s] + 8192 u, V^T fragment
  // (d block dt, k-step c) at v_ad[c] + 4096 dt — one VGPR per swizzle phase, the block offset is the instruction's immediate.
  // B(t) and A(t+1) read the same ring slots (K(t+1), V^T(t)), so one set serves a loop iteration and is advanced by one
  // slot (2 VALU per register) behind A's last own read.
  // (smem sits at LDS byte 0 — the ring wrap rely on it: it is the kernel's only __shared__ object, which the host checks before the first launch, FMI_LDS_GUARD)
  uint32_t k_ad[8], v_ad[4];
#pragma unroll
  for (int s = 0; s < 8; ++s) k_ad[s] = (l31 * 256 + ((hl ^ (lane & 15)) << 4)) ^ (s << 5);
#pragma unroll
  for (int c = 0; c < 4; ++c) v_ad[c] = VT_RING + ((l31 * 128 + ((hl ^ ((l31 >> 1) & 7)) << 4)) ^ (c << 5));
  // ring slots are bits 14..15 of the address: slot s -> s + 1 (mod 4) is ONE xor per register with a wave-uniform mask,
  // 1 << 14 out of an even slot, 3 << 14 out of an odd one (`from` = the slot being left)
  auto advance_k = [&](int from) __attribute__((always_inline)) {
    const uint32_t mk = (from & 1) ? 3u * TILE : 1u * TILE;
#pragma unroll
    for (int s = 0; s < 8; ++s) k_ad[s] ^= mk;
  };
  auto advance_v = [&](int from) __attribute__((always_inline)) {
    const uint32_t mk = (from & 1) ? 3u * TILE : 1u * TILE;
#pragma unroll
    for (int c = 0; c < 4; ++c) v_ad[c] ^= mk;
  };

  f32x16 ot[2][4];  // O^T accumulators: block b, 32-wide d block
  f32x16 sc[2][2];  // S^T: block b, 32-key half of the tile
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[b][i][r] = 0.f;
  frag_t pf[2][4];  // P of block b as four bf16x8 B operands (k-steps of 16 keys)
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

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

  // ---- the (fragment read -> MFMA) stream.  MFMA i of a two-product phase: even i = PV step i/2 (d block j & 3, k-step
  // j >> 2), odd i = QK^T step i/2 (key half j & 1, d-step j >> 1): consecutive MFMAs never write the same accumulator and
  // the k-order inside every accumulator is that of attention_pp_kernel.  One-product phases (first / last tile) have 16.
  frag_t fr[PF];
  // (reads, waits and MFMAs share asm statements — see mfma_step)
  // ---- softmax of block b, cut into steps that ride in the MFMA gaps (arithmetic and operation order of attention_pp_kernel)
  float pmax = 0.f, lsum = 0.f;
  auto sm_mask = [&](int b, int t) __attribute__((always_inline)) {  // ragged last tile: keys >= Lk
    if ((t + 1) * ATT_KV > Lk) {
      // S^T(b) was finished by an MFMA two gaps ago: let it drain before VALU touches it (tied, so the selects stay below)
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sc[b][0]), "+v"(sc[b][1]));
      const int kvb = t * ATT_KV + 4 * hl;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kvb + 32 * u + (r & 3) + 8 * (r >> 2) >= Lk) sc[b][u][r] = -1e30f;
    }
  };
  auto sm_max = [&](int b, int g) __attribute__((always_inline)) {  // g = 0..7: four more scores into the running maximum
    // (asm: a C++ fmaxf on an MFMA result is preceded by a canonicalising v_max per element; two v_max3_f32 in ONE statement
    // because hipcc pads an s_nop between dependent asm statements.  S^T(b) was finished >= 3 MFMAs ago, see sm_gap.)
    const int u = g >> 2, r = 4 * (g & 3);
    if (g == 0)
      asm volatile("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "=&v"(pmax) : "v"(sc[b][0][0]), "v"(sc[b][0][1]), "v"(sc[b][0][2]), "v"(sc[b][0][3]));
    else
      asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "+v"(pmax) : "v"(sc[b][u][r]), "v"(sc[b][u][r + 1]), "v"(sc[b][u][r + 2]), "v"(sc[b][u][r + 3]));
  };
  auto sm_decide = [&](int b) __attribute__((always_inline)) {  // row maximum across the two halves of the wave, deferred-rescale decision
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(pmax), __float_as_uint(pmax), false, false);
    pmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    const float ps = pmax * scale_log2e;
    if (__any(ps - m_run[b] > (float)THR_X16 * 0.0625f)) {
      const float mn = fmaxf(m_run[b], ps);
      const float alpha = fast_exp2(m_run[b] - mn);
      m_run[b] = mn;
      l_run[b] *= alpha;
      // (volatile: hipcc otherwise hoists the 64 accumulator reads of this rarely taken block above the branch, into every tile)
      asm volatile("" : "+a"(ot[b][0]), "+a"(ot[b][1]), "+a"(ot[b][2]), "+a"(ot[b][3]));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[b][i][r] *= alpha;  // (O^T(b) was last written by the previous phase's PV: >= 9 MFMAs ago)
    }
    lsum = 0.f;
  };
  // k = 0..16: step k turns scores (2k, 2k+1) into probabilities (fma, exp2) and finishes pair k-1 (row sum, bf16 pack) —
  // one asm statement per step: it pins the seven instructions to their gap (hipcc would sink them to P's first use),
  // and the one-pair skew keeps a transcendental's result out of the very next instruction (gfx950 trans-use hazard).
  // Same operations in the same order as attention_pp_kernel: p = exp2(fma(s, scale, -m)); lsum += p0 + p1; pack(p0, p1).
  float e0 = 0.f, e1 = 0.f;
  auto sm_exp = [&](int b, int k) __attribute__((always_inline)) {
    const int u = (k & 15) >> 3, r = 2 * (k & 7);
    const int kp = k - 1, up = (kp & 15) >> 3;  // the pair being finished -> dword kp & 3 of P fragment 2 up + ((kp & 7) >> 2)
    float t0, t1, sum;
    if (k == 0) {
      asm volatile("v_fma_f32 %0, %2, %4, -%5\n\tv_fma_f32 %1, %3, %4, -%5\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1"
                   : "=&v"(e0), "=&v"(e1)
                   : "v"(sc[b][0][0]), "v"(sc[b][0][1]), "s"(scale_log2e), "v"(m_run[b]));
    } else if (k < 16) {
      asm volatile(
          "v_fma_f32 %[t0], %[s0], %[sl], -%[m]\n\tv_fma_f32 %[t1], %[s1], %[sl], -%[m]\n\tv_add_f32 %[sum], %[q0], %[q1]\n\t"
          "v_cvt_pk_bf16_f32 %[pk], %[q0], %[q1]\n\tv_add_f32 %[ls], %[ls], %[sum]\n\tv_exp_f32 %[q0], %[t0]\n\tv_exp_f32 %[q1], %[t1]"
          : [q0] "+v"(e0), [q1] "+v"(e1), [ls] "+v"(lsum), [pk] "=&v"(pf[b][2 * up + ((kp & 7) >> 2)][kp & 3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [sum] "=&v"(sum)
          : [s0] "v"(sc[b][u][r]), [s1] "v"(sc[b][u][r + 1]), [sl] "s"(scale_log2e), [m] "v"(m_run[b]));
    } else {
      asm volatile("v_add_f32 %[sum], %[q0], %[q1]\n\tv_cvt_pk_bf16_f32 %[pk], %[q0], %[q1]\n\tv_add_f32 %[ls], %[ls], %[sum]"
                   : [ls] "+v"(lsum), [pk] "=&v"(pf[b][2 * up + ((kp & 7) >> 2)][kp & 3]), [sum] "=&v"(sum)
                   : [q0] "v"(e0), [q1] "v"(e1));
      l_run[b] += lsum;
    }
  };
  // softmax slice of gap g (0..31): gap 0 masks (ragged tail only), gaps 1..8 one max step (4 scores) each — the first
  // scores read were finished by the previous phase's MFMA 29, three MFMAs before gap 1 —, gap 9 decides, gaps 10..31 carry
  // the 17 exp steps
  auto sm_gap = [&](int g, int b, int t) __attribute__((always_inline)) {
    if (g == 0) sm_mask(b, t);
    if (g >= 1 && g <= 8) sm_max(b, g - 1);
    if (g == 9) sm_decide(b);
    if (g >= 10) {
      const int k0 = (g - 10) * 17 / 22, k1 = (g - 9) * 17 / 22;
      if (k1 > k0) sm_exp(b, k0);
    }
  };

  // The MFMAs are inline asm so that the register FILE of every operand is fixed: O^T and the Q fragments live in the
  // accumulator half (AGPRs), S^T, P and the streamed K / V^T fragments in the architectural VGPRs the softmax VALU works on.
  // (With builtins hipcc put S^T into AGPRs and moved it back and forth: ~280 v_accvgpr moves per KV tile.)  hipcc pads no
  // hazards around asm: every consumer of an MFMA result sits >= 3 MFMAs behind its producer by construction of the
  // schedule (noted at each site); the first d-step of S^T starts from the constant 0.
  // One gap's MFMA, optionally preceded by an lgkmcnt wait (LDS reads retire in order) and followed by the fragment read
  // that refills the buffer the MFMA has just consumed — ONE asm statement, so the MFMA stays below the wait without a
  // register tie and hipcc pads no s_nop between the three.  rd_imm < 0: no read.
#define FMI_AW4_G3(WAIT, CSTR, OUT, B, IMM)                                                                                                  \
  asm volatile(WAIT "v_mfma_f32_32x32x16_bf16 %0, %2, %3, " CSTR "\n\tds_read_b128 %1, %4 offset:" #IMM : OUT, "=&v"(g) : "v"(f), B, "v"(ra))
#define FMI_AW4_G2(WAIT, CSTR, OUT, B) asm volatile(WAIT "v_mfma_f32_32x32x16_bf16 %0, %1, %2, " CSTR : OUT : "v"(f), B)
#define FMI_AW4_G2D(WAIT, CSTR, OUT, B) asm volatile(WAIT "v_mfma_f32_32x32x16_bf16 %0, %1, %2, " CSTR "\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : OUT : "v"(f), B)
#define FMI_AW4_GAP(WAIT, CSTR, OUT, B)                        \
  switch (rd_imm) {                                            \
    case 0: FMI_AW4_G3(WAIT, CSTR, OUT, B, 0); break;          \
    case 4096: FMI_AW4_G3(WAIT, CSTR, OUT, B, 4096); break;    \
    case 8192: FMI_AW4_G3(WAIT, CSTR, OUT, B, 8192); break;    \
    case 12288: FMI_AW4_G3(WAIT, CSTR, OUT, B, 12288); break;  \
    case -2: FMI_AW4_G2D(WAIT, CSTR, OUT, B); break;           \
    default: FMI_AW4_G2(WAIT, CSTR, OUT, B); break;            \
  }
#define FMI_AW4_MFMA(CSTR, OUT, B)                                          \
  switch (wait) {                                                           \
    case 4: FMI_AW4_GAP("s_waitcnt lgkmcnt(4)\n\t", CSTR, OUT, B) break;    \
    case 3: FMI_AW4_GAP("s_waitcnt lgkmcnt(3)\n\t", CSTR, OUT, B) break;    \
    case 2: FMI_AW4_GAP("s_waitcnt lgkmcnt(2)\n\t", CSTR, OUT, B) break;    \
    case 1: FMI_AW4_GAP("s_waitcnt lgkmcnt(1)\n\t", CSTR, OUT, B) break;    \
    case 0: FMI_AW4_GAP("s_waitcnt lgkmcnt(0)\n\t", CSTR, OUT, B) break;    \
    default: FMI_AW4_GAP("", CSTR, OUT, B) break;                           \
  }
  // The MFMAs are inline asm so that the register FILE of every operand is fixed: O^T and the Q fragments live in the
  // accumulator half (AGPRs), S^T, P and the streamed K / V^T fragments in the architectural VGPRs the softmax VALU works on.
  // (With builtins hipcc put S^T into AGPRs and moved it back and forth: ~280 v_accvgpr moves per KV tile.)  hipcc pads no
  // hazards around asm: every consumer of an MFMA result sits >= 3 MFMAs behind its producer by construction of the
  // schedule (noted at each site); the first d-step of S^T starts from the constant 0.  The read in a gap refills the
  // buffer of the PREVIOUS MFMA (see `phase`).
  auto mfma_step = [&](const Aw4Phase& p, int i, frag_t& f, frag_t& g, int wait, uint32_t ra, int rd_imm) __attribute__((always_inline)) {
    const bool pv = p.has_pv && (!p.has_qk || (i & 1) == 0);
    const int j = (p.has_pv && p.has_qk) ? (i >> 1) : i;
    if (pv) {
      FMI_AW4_MFMA("%0", "+a"(ot[p.b_pv][j & 3]), "v"(pf[p.b_pv][j >> 2]))
    } else {
      const int u = j & 1, s = j >> 1;
      if (s == 0) { FMI_AW4_MFMA("0", "=v"(sc[p.b_qk][u]), "a"(qf[p.b_qk][s])) }
      else { FMI_AW4_MFMA("%0", "+v"(sc[p.b_qk][u]), "a"(qf[p.b_qk][s])) }
    }
  };
  // address register and immediate of the fragment of MFMA i of a phase with the given product mix
  auto frag_reg = [&](bool has_pv, bool has_qk, int i) __attribute__((always_inline)) -> uint32_t {
    const bool pv = has_pv && (!has_qk || (i & 1) == 0);
    const int j = (has_pv && has_qk) ? (i >> 1) : i;
    return pv ? v_ad[j >> 2] : k_ad[j >> 1];
  };
  auto frag_imm = [&](bool has_pv, bool has_qk, int i) __attribute__((always_inline)) -> int {
    const bool pv = has_pv && (!has_qk || (i & 1) == 0);
    const int j = (has_pv && has_qk) ? (i >> 1) : i;
    return pv ? (j & 3) * 4096 : (j & 1) * 8192;
  };
  // One phase; its product mix (has_pv, has_qk, blocks) is a literal at every call site, so the gaps below specialise at
  // compile time.  `nx` = the phase that follows (its first PF - 1 fragment reads are issued behind this phase's last MFMAs;
  // its QK^T part may be a run-time choice).  sm: softmax of block b_sm / tile t_sm in the gaps.  bar_gap >= 0: the tile
  // barrier sits in that gap, after waiting until at most bar_vm of this wave's DMA pieces are outstanding.  dma: 1 = the
  // pieces of V^T(dma_tile) in the second half (behind the barrier), 2 = those of K(dma_tile), one every 8 gaps.
  // adv: bit 0 / 1 = advance the K / V^T address registers by one ring slot behind this phase's last own read (adv_u = the
  // tile index u of the phase A(u) doing it).
  auto phase = [&](bool has_pv, bool has_qk, int b_pv, int b_qk, const Aw4Phase& nx, bool has_sm, int b_sm, int t_sm, int bar_gap, int bar_vm, int dma,
                   int dma_tile, int adv, int adv_u, bool tail_drain = false) __attribute__((always_inline)) {
    const Aw4Phase p{has_pv, has_qk, b_pv, b_qk};
    const int n = p.n(), nn = nx.n();
#pragma clang loop unroll(full)
    for (int i = 0; i < 32; ++i) {
      if (i < n) {
        // every fourth gap waits for the fragments of gaps i .. i+3: at most PF - 4 younger reads may be pending
        // The read behind MFMA i fetches fragment i + RD of the stream (RD = PF - 1) into the buffer MFMA i - 1 consumed — never
        // into MFMA i's own A operand: an MFMA can still be waiting for the matrix pipe when the LDS data returns, and nothing
        // orders that register write behind the operand read (with other work on the CU the MFMA then multiplied the fragment
        // meant for MFMA i + 8: tools/attn_race.hip).  MFMA i has issued, so MFMA i - 1 has left the front of the pipe.
        constexpr int RD = PF - 1;
        const int ahead = n - i - 4 + nn;  // stream reads issued beyond gap i + 3
        const int wait = (i & 3) == 0 ? min(RD - 4, max(ahead, 0)) : -1;
        if (adv && i + RD == n) {  // all own reads are out: move the address registers to the next iteration's slots
          if (adv & 1) advance_k(adv_u);  // A(u): K leaves slot u, V^T slot u - 1
          if (adv & 2) advance_v(adv_u - 1);
        }
        frag_t& dst = fr[(i + RD) % PF];
        if (i + RD < n) {
          mfma_step(p, i, fr[i % PF], dst, wait, frag_reg(has_pv, has_qk, i + RD), frag_imm(has_pv, has_qk, i + RD));
        } else if (i + RD - n < nn) {  // the next phase's first fragments (its product mix may be a run-time choice)
          const int k = i + RD - n;
          if (nx.has_pv && nx.has_qk) mfma_step(p, i, fr[i % PF], dst, wait, frag_reg(true, true, k), frag_imm(true, true, k));
          else if (nx.has_pv) mfma_step(p, i, fr[i % PF], dst, wait, frag_reg(true, false, k), frag_imm(true, false, k));
          else mfma_step(p, i, fr[i % PF], dst, wait, frag_reg(false, true, k), frag_imm(false, true, k));
        } else {
          // no read; rd_imm -2 on the phase's last MFMA = drain inside the statement (used in front of the pinned loop statement:
          // hipcc may copy S^T / O^T right behind this statement and knows nothing of the MFMAs in flight)
          mfma_step(p, i, fr[i % PF], dst, wait, 0u, (tail_drain && i == n - 1) ? -2 : -1);
        }
        const int g0 = n == 32 ? i : 2 * i;  // a 16-MFMA phase carries two softmax slices per gap
        // (its first max step would sit one MFMA behind the previous phase's last write of S^T: drain once, untied asm order)
        if (has_sm && n == 16 && i == 0) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sc[b_sm][0]), "+v"(sc[b_sm][1]));
        if (has_sm) {
          sm_gap(g0, b_sm, t_sm);
          if (n == 16) sm_gap(g0 + 1, b_sm, t_sm);
        }
        if (bar_gap == i) {
          if (bar_vm >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        }
        if (dma == 2 && (n == 32 ? (i & 7) == 3 : (i & 3) == 1)) stage_k(min(dma_tile, ntiles - 1), n == 32 ? i >> 3 : i >> 2);
        if (dma == 1 && (n == 32 ? (i >= 16 && (i & 3) == 3) : (i >= 8 && (i & 1) == 1))) stage_v(min(dma_tile, ntiles - 1), n == 32 ? (i - 16) >> 2 : (i - 8) >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // The stream:  pre = QK^T(0, 0);  A(t) = PV(1, t-1) + QK^T(1, t) beside softmax(0, t);  B(t) = PV(0, t) + QK^T(0, t+1) beside
  // softmax(1, t);  post = PV(1, last).  A(0) has no PV part, B(last) no QK^T part.
  // DMA in program order: prologue K(0..2), V^T(0..1); then A(0): V^T(2); B(0): K(3); A(1): V^T(3); B(1): K(4); ... (tile
  // indices clamped to the last tile, so every phase issues its four pieces).  The barrier in A(t), t >= 1, needs K(t+1),
  // V^T(t): everything but this wave's 8 newest pieces, V^T(t+1) [A(t-1)] and K(t+2) [B(t-1)].
  const Aw4Phase none{false, false, 0, 0};
  {
#pragma unroll
    for (int i = 0; i < PF - 1; ++i) {
      switch (i) {  // the first PF - 1 fragments of pre: K(0), key half i & 1, d-step i >> 1
        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(fr[0]) : "v"(k_ad[0])); break;
        case 1: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fr[1]) : "v"(k_ad[0])); break;
        case 2: asm volatile("ds_read_b128 %0, %1" : "=v"(fr[2]) : "v"(k_ad[1])); break;
        case 3: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fr[3]) : "v"(k_ad[1])); break;
        case 4: asm volatile("ds_read_b128 %0, %1" : "=v"(fr[4]) : "v"(k_ad[2])); break;
        case 5: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fr[5]) : "v"(k_ad[2])); break;
        case 6: asm volatile("ds_read_b128 %0, %1" : "=v"(fr[6]) : "v"(k_ad[3])); break;
        default: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fr[7]) : "v"(k_ad[3])); break;
      }
    }
    phase(false, true, 0, 0, Aw4Phase{false, true, 1, 1}, false, 0, 0, -1, 0, 0, 0, 0, 0);       // pre: QK^T(0, 0) from K(0)
    // A(0): QK^T(1, 0); V^T(2); then K -> slot 1.  No fragment prefetch for the next phase and the MFMAs drained inside its
    // last statement: the pinned loop statement follows, see there.
    phase(false, true, 1, 1, none, true, 0, 0, 8, 0, 1, 2, 1, 0, true);
  }
  // B(t), A(t+1): both read K(t+1) and V^T(t).  The last pair is peeled so that the phase following A is a literal
  // inside the loop (a run-time choice there made hipcc shuffle S^T between two register assignments every iteration).
  // The steady state — B(t), A(t+1) for t = 0 .. ntiles-3, i.e. the two calls
  //     phase(true, true, 0, 0, Aw4Phase{true, true, 1, 1}, true, 1, t, -1, 0, 2, t + 3, 0, 0);
  //     phase(true, true, 1, 1, Aw4Phase{true, true, 0, 0}, true, 0, t + 1, 16, 8, 1, t + 3, 3, t + 1);
  // — is ONE asm statement with hand-assigned registers, generated by tools/gen_attention_w4_loop.py into
  // attention_w4_loop.inc (same instructions, same order per gap; nothing padded or placed by hipcc in between: as a
  // chain of per-gap statements the loop carried 52 compiler-inserted s_nop and ~70 stray address / control
  // instructions per pair of phases).  The operands below pin every array to the registers the generated text names.
  // hipcc knows nothing of the fragment reads / MFMAs in flight and copies registers in front of and behind a statement with
  // pinned operands, so the statement is entered and left with nothing in flight (A(0) above; the statement fetches the
  // first fragments of B itself and ends with a drain) and runs on every path (zero iterations when ntiles == 2).
  {
    typedef float f32x32 __attribute__((ext_vector_type(32)));
    typedef int i32x32 __attribute__((ext_vector_type(32)));
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    f32x32 O[4], S[2];
    i32x32 F, QA[2];
    i32x16 P[2], misc;
    i32x8 KA;
    frag_t VA;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        O[2 * b][r] = ot[b][r >> 4][r & 15], O[2 * b + 1][r] = ot[b][2 + (r >> 4)][r & 15];
        S[b][r] = sc[b][r >> 4][r & 15];
        QA[b][r] = qf[b][r >> 2][r & 3];
      }
      F[r] = fr[r >> 2][r & 3];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) P[0][r] = pf[0][r >> 2][r & 3], P[1][r] = pf[1][r >> 2][r & 3];
#pragma unroll
    for (int r = 0; r < 8; ++r) KA[r] = (int)k_ad[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      VA[r] = (int)v_ad[r];
      misc[r] = (int)k_voff[r], misc[4 + r] = (int)v_voff[r];
      // K rows past the last key of a ragged last tile are fetched from the last key (stage_k)
      const int kr = (wave * 4 + r) * 4 + (lane >> 4);
      misc[8 + r] = kr >= k_last_rows ? (int)((k_last_rows - 1) * 256 + (((lane & 15) ^ (kr & 15)) << 4)) : (int)k_voff[r];
    }
    misc[12] = __float_as_int(m_run[0]), misc[13] = __float_as_int(m_run[1]), misc[14] = __float_as_int(l_run[0]), misc[15] = __float_as_int(l_run[1]);
    const uint64_t kb64 = (uint64_t)(uintptr_t)Kb, vb64 = (uint64_t)(uintptr_t)Vb;
    const uint32_t kb_lo = __builtin_amdgcn_readfirstlane((uint32_t)kb64), kb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(kb64 >> 32));
    const uint32_t vb_lo = __builtin_amdgcn_readfirstlane((uint32_t)vb64), vb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(vb64 >> 32));
    const float thr = (float)THR_X16 * 0.0625f;
    asm volatile(FMI_AW4_LOOP_ASM
                 : "+{a[0:31]}"(O[0]), "+{a[32:63]}"(O[1]), "+{a[64:95]}"(O[2]), "+{a[96:127]}"(O[3]), "+{v[0:31]}"(S[0]), "+{v[32:63]}"(S[1]),
                   "+{v[64:79]}"(P[0]), "+{v[80:95]}"(P[1]), "+{v[96:127]}"(F), "+{v[128:135]}"(KA), "+{v[136:139]}"(VA), "+{v[140:155]}"(misc)
                 : "{a[128:159]}"(QA[0]), "{a[160:191]}"(QA[1]), [kb_lo] "s"(kb_lo), [kb_hi] "s"(kb_hi), [vb_lo] "s"(vb_lo), [vb_hi] "s"(vb_hi),
                   [nt] "s"(ntiles), [ntm1] "s"(ntiles - 1), [sl] "s"(scale_log2e), [thr] "s"(thr), [woff] "s"(wave * 4096)
                 : "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "s80",
                   "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "vcc", "scc", "memory");
#pragma unroll
    for (int r = 0; r < 32; ++r) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        ot[b][r >> 4][r & 15] = O[2 * b][r], ot[b][2 + (r >> 4)][r & 15] = O[2 * b + 1][r];
        sc[b][r >> 4][r & 15] = S[b][r];
      }
      fr[r >> 2][r & 3] = F[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pf[0][r >> 2][r & 3] = P[0][r], pf[1][r >> 2][r & 3] = P[1][r];
#pragma unroll
    for (int r = 0; r < 8; ++r) k_ad[r] = (uint32_t)KA[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) v_ad[r] = (uint32_t)VA[r];
    m_run[0] = __int_as_float(misc[12]), m_run[1] = __int_as_float(misc[13]), l_run[0] = __int_as_float(misc[14]), l_run[1] = __int_as_float(misc[15]);
  }
  phase(true, true, 0, 0, Aw4Phase{true, true, 1, 1}, true, 1, ntiles - 2, -1, 0, 2, ntiles - 1, 0, 0);
  phase(true, true, 1, 1, Aw4Phase{true, false, 0, 0}, true, 0, ntiles - 1, 16, 8, 1, ntiles - 1, 3, ntiles - 1);
  phase(true, false, 0, 0, Aw4Phase{true, false, 1, 1}, true, 1, ntiles - 1, -1, 0, 0, 0, 0, 0);  // B(last): PV(0, last) beside softmax(1, last)
  phase(true, false, 1, 1, none, false, 0, 0, -1, 0, 0, 0, 0, 0);                                // post: PV(1, last)
#undef FMI_AW4_MFMA
#undef FMI_AW4_GAP
#undef FMI_AW4_G2
#undef FMI_AW4_G2D
#undef FMI_AW4_G3

  // ---- epilogue: O[q][d] = O^T / l, d = 32 dt + 8 g + 4 hl + {0..3}; staged through LDS so that whole 256-byte rows leave
  // The last MFMAs must drain before O^T is read, and hipcc must not schedule those reads above the drain (it did: an asm
  // output counts as available right behind its statement, so the v_accvgpr_reads of the epilogue sat between the last PV
  // MFMAs and lost their contribution): every accumulator passes through a volatile asm placed behind the nops.
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(ot[b][i]));
  __builtin_amdgcn_s_barrier();  // every wave is done with the rings
  char* stg = smem + wave * TILE;  // 64 rows x 256 B, 16-byte slot c of row r at c ^ (r & 15)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[b]), __float_as_uint(l_run[b]), false, false);
    const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    const int r = 32 * b + l31;
    if (LSE && hl == 0 && q0 + r < Lq) lse[(int64_t)bh * Lq + q0 + r] = m_run[b] + __log2f(__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + g * 8 + 4 * hl;
        const uint2 v = make_uint2(pack_bf16x2(ot[b][dt][4 * g] * inv, ot[b][dt][4 * g + 1] * inv), pack_bf16x2(ot[b][dt][4 * g + 2] * inv, ot[b][dt][4 * g + 3] * inv));
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
