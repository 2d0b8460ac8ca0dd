// gemm_w4q.h — fused bitsandbytes dequant-GEMM for large M (included by gemm_bf16.hip; same translation unit).
//
// Replaces BnbLinear::forward (diffusion_rs_backend/src/bitsandbytes/mod.rs:301-312: dequantise the whole
// weight to a dense tensor, then matmul) for nf4 / fp4 weights (kernels/bitsandbytes/dequant.cu:94-170)
// WITHOUT the dense round trip through HBM: the packed weight tile is read from HBM (32 B per row per K tile
// instead of 128 B), expanded in registers with exactly the arithmetic of the stand-alone dequant kernel —
// bf16(code_value * absmax) — and written into the same XOR-swizzled LDS image the dense kernels' DMA produces:
// the "dequant as an LDS stage" of BASELINE.json's north star; bit-identical to dequantise-then-dense because the MFMA sees the same bf16 operands in the same order.
//
// Structure = gemm_w4_kernel (4 waves, one per SIMD, 128 x 128 of C per wave in 256 accumulator registers, A tile by
// LDS-DMA), with the W side register-staged.  One wave per SIMD hides ~5 single-issue instructions behind every
// 32x32x16 MFMA (MI355X_MICROARCH.md, per-instruction table), so the expansion is cut into micro-steps and one
// micro-step rides in each MFMA gap:
//   * a lane owns one W row of the tile: 64 weights per K tile = 8 "units" of one packed dword;
//   * a 256-entry table in LDS maps a packed BYTE to the f32 values of its two codes (2 KiB); every second MFMA gap
//     one byte is turned into a table address (shift + and) and looked up with ONE ds_read_b64 — with one wave per
//     SIMD the LDS pipe retires narrow reads at a fraction of its rate, so the number of lookups, not their bank
//     conflicts, is what counts (a 16-entry conflict-free table read per nibble ran 0.8x the dense kernel; r02 bench);
//     eight gaps later the two values are multiplied by the row's absmax and rounded to bf16 (v_cvt_pk_bf16_f32), and
//     every eighth gap a 16-byte piece of the row goes to the W ring (ds_write_b128 at slot u ^ ((row >> 1) & 7));
//   * the unit stream is a modulo-8 pipeline that runs across K tiles: the group that looks up unit u consumes unit
//     u-1, the first lookups of tile t+2 ride next to the last writes of tile t+1, so every group is the same.
// LDS: 2 KiB (table) + A ring 2 x 32 KiB + W ring 2 x 32 KiB.  Per K tile t the order is
//   G0..G4 (steps 0, 1, first half of 2): lookups of W(t+1) units 3..7, writes of units 2..6; G0 also issues the packed
//       loads of W(t+2) into a landing register set; G4 waits for them (and thereby for the A(t+1) DMA) and swaps sets;
//   G5: the last unit of W(t+1) is written early in the group, then lgkmcnt / s_barrier: tile t's slots are free,
//       tile t+1 is published;  lookups of W(t+2) unit 0 (no write yet) share the group;
//   G6, G7 (step 3): fragment reads of tile t+1 step 0, the 8 DMA pieces of A(t+2), units 1, 2 of W(t+2).
// Fragments are read ONE k-step ahead into two register buffers (the dense kernel reads two steps ahead into four;
// the 64 registers that frees hold the expansion state).  Past the end of K the loads re-fetch the last tile into
// slots nobody reads again (branch-free body, as in gemm_w4_kernel).
#pragma once

namespace fmi {

constexpr int W4Q_LUT_BYTES = 2048;  // 256 byte values x {value of the high nibble, value of the low nibble} f32

template <int ACT>
__global__ __launch_bounds__(W4_THREADS, 1) void gemm_w4q_kernel(const GemmBatch batch) {
  constexpr int TILE = A_TILE_BYTES;
  constexpr int A_RING = W4Q_LUT_BYTES, W_RING = W4Q_LUT_BYTES + 2 * TILE;
  __shared__ __attribute__((aligned(256))) char smem[W4Q_LUT_BYTES + 4 * TILE];
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
  const int tiles_n = (P.N + 255) / 256;
  int tm, tn;
  tile_coords(t_in, tiles_m, tiles_n, batch.band[pi], tm, tn);
  const int m0 = tm * BM, n0 = tn * 256;
  const int nk = P.K / BK;
  const int klast = nk - 1;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (smem sits at LDS byte 0 — the table lookups rely on it: it is the kernel's only __shared__ object, which the host checks before the first launch, FMI_LDS_GUARD)
  {  // packed byte -> the two code values, weight order (high nibble first); fp4: value * sign of the tree in dequant.cu:12-37
    const float fp4[8] = {0.0f, 5.208333333e-03f, 0.66666667f, 1.0f, 0.33333333f, 0.5f, 0.16666667f, 0.25f};
    auto val = [&](int c) { return P.q_type == 2 ? kNF4[c] : ((c & 8) ? -fp4[c & 7] : fp4[c & 7]); };
    reinterpret_cast<float2*>(smem)[tid] = make_float2(val(tid >> 4), val(tid & 15));  // 256 threads, 256 entries
  }

  // ---- A operand: LDS-DMA pieces of this wave (1-KiB chunks wave*8 + i), as gemm_w4_kernel
  uint32_t a_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (wave * 8 + i) * 8 + (lane >> 3);
    a_off[i] = (uint32_t)((int64_t)(min(m0 + r, P.M - 1) - m0) * P.lda * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4));  // tile-relative
  }
  const char* const a_base = reinterpret_cast<const char*>(P.A) + (int64_t)m0 * P.lda * 2;
  auto dma_a = [&](int kt, int slot, int i) {
    const char* base = a_base + (int64_t)kt * (BK * 2);
    __builtin_amdgcn_global_load_lds((glb_void*)(base + a_off[i]), (lds_void*)(smem + A_RING + slot * TILE + (wave * 8 + i) * 1024), 16, 0, 0);
  };
  // ---- W operand: this lane's row of the tile (32 packed bytes + one scale per K tile)
  const int wrow = wave * 64 + lane;
  const int wn_g = min(n0 + wrow, P.N - 1);
  const uint32_t wq_off = (uint32_t)((int64_t)wn_g * P.K / 2);  // byte offset of the row in Wq
  const int kt_block_shift = __builtin_ctz(P.q_blocksize / BK);  // K tiles per absmax block: a power of two (host-checked)
  const uint32_t am_off = (uint32_t)wn_g * (uint32_t)(P.K / P.q_blocksize) * 4u;
  const uint32_t w_wr = W_RING + wrow * 128 + (((wrow >> 1) & 7) << 4);  // 16-B piece u of the row lives at w_wr ^ (u << 4)
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  i32x4 nxt0, nxt1;  // landing registers of the packed loads
  float nxt_am;
  uint32_t cur[8];   // packed words of the tile being expanded
  float cur_am = 0.f, am_c = 0.f;  // scale of `cur` / of the unit being consumed
  // (inline asm: a load hipcc does not count, so that it never waits vmcnt(0) for it beside the LDS-DMA stream; the
  // s_nop keeps a base pointer fresh from the scalar unit out of the next state's saddr read, guide 5.7 item 2)
  // (the base pointers pass through an empty asm so that they stay in SGPRs: re-reading them from the kernel arguments
  // inside the loop is an s_load + lgkmcnt(0), i.e. a drain of the LDS pipeline once per K tile)
  const char* wq_base = reinterpret_cast<const char*>(P.Wq);
  const char* am_base = reinterpret_cast<const char*>(P.absmax);
  asm volatile("" : "+s"(wq_base), "+s"(am_base));
  auto load_w = [&](int kt, int part) {
    const char* wb = wq_base + (int64_t)kt * 32;
    const char* ab = am_base + (int64_t)(kt >> kt_block_shift) * 4;
    if (part == 0) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(nxt0) : "v"(wq_off), "s"(wb) : "memory");
    if (part == 1) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:16" : "=v"(nxt1) : "v"(wq_off), "s"(wb) : "memory");
    if (part == 2) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(nxt_am) : "v"(am_off), "s"(ab) : "memory");
  };
  auto swap_sets = [&]() {  // after the vmcnt wait that covers the packed loads
#pragma unroll
    for (int e = 0; e < 4; ++e) cur[e] = (uint32_t)nxt0[e], cur[4 + e] = (uint32_t)nxt1[e];
    cur_am = nxt_am;
  };

  // ---- expansion pipeline state
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 L[2][4];            // table values of the unit being looked up / consumed (set = unit & 1): pair jp = weights 2jp, 2jp+1
  uint32_t tw = 0, tw3 = 0;  // the unit's word shifted left by 3: byte k (masked per lookup) is a table address; byte 3 apart (its top bits fall off)
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 pk;                 // the unit's 8 bf16 on their way to LDS
  // byte k of a unit holds weights 2k (high nibble) and 2k+1 (low nibble) (dequant.cu:142-151).
  // (plain int parameters, constant after inlining: a generic lambda cannot name the register arrays in asm operands.)
  auto lookup = [&](int u, int jp) {
    if (jp == 0) {  // (both taken now: group G4 swaps `cur` to the next tile right after its first lookup)
      tw = cur[u] << 3;
      tw3 = (cur[u] >> 21) & 0x7f8u;
    }
    const uint32_t a = jp == 0 ? (tw & 0x7f8u) : jp == 3 ? tw3 : ((tw >> (8 * jp)) & 0x7f8u);
    asm volatile("ds_read_b64 %0, %1" : "=v"(L[u & 1][jp]) : "v"(a));
  };
  // pair jp of unit u: two values * absmax -> one dword of bf16; after the fourth pair the 16-byte piece is stored
  auto consume_pair = [&](int u, int jp, int slot) {
    pk[jp] = pack_bf16x2(L[u & 1][jp][0] * am_c, L[u & 1][jp][1] * am_c);
    if (jp == 3) {
      const uint32_t ad = (w_wr + slot * TILE) ^ (uint32_t)(u << 4);
      asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(pk) : "memory");
    }
  };
  // wait until at most N LDS operations are pending, tied to the two pairs (h = 0: pairs 0, 1; h = 1: pairs 2, 3) it releases
#define FMI_W4Q_LGKM(N, c, h) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(L[c][2 * (h)]), "+v"(L[c][2 * (h) + 1]))

  // ---- prologue: A(0), A(1) by DMA; W(0) and units 0, 1 of W(1) expanded with nothing to overlap
  const int k1 = min(1, klast);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_a(0, 0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_a(k1, 1, i);
  __syncthreads();  // table visible
  auto serial_unit = [&](int u, int slot) {
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) lookup(u, jp);
    if (u & 1) {
      FMI_W4Q_LGKM(0, 1, 0);
      FMI_W4Q_LGKM(0, 1, 1);
    } else {
      FMI_W4Q_LGKM(0, 0, 0);
      FMI_W4Q_LGKM(0, 0, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) consume_pair(u, jp, slot);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto fetch_now = [&](int kt) {
    load_w(kt, 0), load_w(kt, 1), load_w(kt, 2);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt0), "+v"(nxt1), "+v"(nxt_am)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    swap_sets();
    am_c = cur_am;
  };
  fetch_now(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) serial_unit(u, 0);
  fetch_now(k1);
  serial_unit(0, 1), serial_unit(1, 1);
  // unit 2 of W(1): looked up, not yet consumed — the state group G0 of the first K tile expects
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) lookup(2, jp);
  FMI_W4Q_LGKM(0, 0, 0);  // unit 2 -> value set 0
  FMI_W4Q_LGKM(0, 0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();  // A(0), A(1), W(0) published
  __builtin_amdgcn_sched_barrier(0);

  // ---- fragment reads: one VGPR address per (operand, k-step), rebased once per tile; the row block is the immediate
  typedef __attribute__((ext_vector_type(4))) int frag_t;
  frag_t xf[2][4], wf[2][4];
  const int sw = ((lane & 31) >> 1) & 7;
  uint32_t koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = ((s * 2 + (lane >> 5)) ^ sw) << 4;
  const uint32_t a_row = A_RING + (wm * 128 + (lane & 31)) * 128;
  const uint32_t w_row = W_RING + (wn * 128 + (lane & 31)) * 128;
  uint32_t a_ad[4], w_ad[4];
  auto rebase = [&](int slot) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a_ad[s] = a_row + koff[s] + slot * TILE;
      w_ad[s] = w_row + koff[s] + slot * TILE;
    }
  };
#define FMI_W4Q_RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
  auto read_frag = [&](int s, int which) {  // k-step s of the tile a_ad / w_ad point to -> buffer s & 1
    const int b = s & 1;
    switch (which) {
      case 0: FMI_W4Q_RD(xf[b][0], a_ad[s], 0); break;
      case 1: FMI_W4Q_RD(xf[b][1], a_ad[s], 4096); break;
      case 2: FMI_W4Q_RD(xf[b][2], a_ad[s], 8192); break;
      case 3: FMI_W4Q_RD(xf[b][3], a_ad[s], 12288); break;
      case 4: FMI_W4Q_RD(wf[b][0], w_ad[s], 0); break;
      case 5: FMI_W4Q_RD(wf[b][1], w_ad[s], 4096); break;
      case 6: FMI_W4Q_RD(wf[b][2], w_ad[s], 8192); break;
      default: FMI_W4Q_RD(wf[b][3], w_ad[s], 12288); break;
    }
  };
  // LDS operations retire in order: once at most N are pending, everything issued before the newest N has landed.
  // The "+v" ties make the fragments depend on the wait, so no MFMA is scheduled above it.
#define FMI_W4Q_FRAG_WAIT(N, b)                                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                                                   \
               : "+v"(xf[b][0]), "+v"(xf[b][1]), "+v"(xf[b][2]), "+v"(xf[b][3]), "+v"(wf[b][0]), "+v"(wf[b][1]), "+v"(wf[b][2]), "+v"(wf[b][3]))
  rebase(0);
#pragma unroll
  for (int w = 0; w < 8; ++w) read_frag(0, w);
  FMI_W4Q_FRAG_WAIT(0, 0);  // (the in-loop wait of the first step assumes a preceding group of lookups)
  __builtin_amdgcn_sched_barrier(0);

  // One MFMA gap.  G = group 0..7 of the K tile (two per k-step), J = gap 0..7 of the group.
  auto gap = [&](int t, int G, int J) {
    const int step = G >> 1, q = (G & 1) * 8 + J, b = step & 1;
    const int U = (G + 3) & 7;   // unit looked up in this group
    const int UC = (G + 2) & 7;  // unit consumed in this group
    // ---- consume unit UC (W(t+1) through G5, W(t+2) from G6 on) — first in the gap, so that its waits only count
    // operations of earlier gaps.  LDS operations retire in order; a group issues its 4 lookups in gaps 0, 2, 4, 6.
    const int wslot = (G >= 6 ? t : t + 1) & 1;
    if (G == 6 && J == 0) am_c = cur_am;
    if (G == 5) {
      // the last unit of W(t+1) early in the group, so that its store is old when the barrier waits for it
      if (J == 1) FMI_W4Q_LGKM(3, 1, 0);  // (unit 7 -> value set 1) pairs 0, 1: the lookups of G4 gaps 4, 6 and G5 gap 0 are younger
      if (J == 1) consume_pair(UC, 0, wslot);
      if (J == 2) consume_pair(UC, 1, wslot);
      if (J == 3) FMI_W4Q_LGKM(2, 1, 1);  // pairs 2, 3: the lookups of G5 gaps 0, 2 are younger
      if (J == 3) consume_pair(UC, 2, wslot);
      if (J == 4) consume_pair(UC, 3, wslot);
    } else {
      // pairs 0, 1 were looked up in gaps 0, 2 of the previous group (2 younger lookups: its gaps 4, 6); pairs 2, 3 in its
      // gaps 4, 6 (2 younger lookups: this group's gaps 0, 2)
      if (J == 0 && (UC & 1)) FMI_W4Q_LGKM(2, 1, 0);
      if (J == 0 && !(UC & 1)) FMI_W4Q_LGKM(2, 0, 0);
      if (J == 4 && (UC & 1)) FMI_W4Q_LGKM(2, 1, 1);
      if (J == 4 && !(UC & 1)) FMI_W4Q_LGKM(2, 0, 1);
      if (J == 1) consume_pair(UC, 0, wslot);
      if (J == 2) consume_pair(UC, 1, wslot);
      if (J == 5) consume_pair(UC, 2, wslot);
      if (J == 6) consume_pair(UC, 3, wslot);
    }
    if (G % 2 == 0 && J == 0) {
      // this step's fragments were read in the first half of the previous step; the second half issued 4 lookups and a store
      if (b) FMI_W4Q_FRAG_WAIT(4, 1);
      else FMI_W4Q_FRAG_WAIT(4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    acc[q >> 2][q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[b][q & 3]), __builtin_bit_cast(bf16x8_t, xf[b][q >> 2]),
                                                                 acc[q >> 2][q & 3], 0, 0, 0);
    // ---- fragment reads of the next k-step (during step 3: step 0 of tile t + 1; addresses rebased after the barrier)
    if (G % 2 == 0) read_frag((step + 1) & 3, J);
    // ---- look up byte J / 2 of unit U
    if (J % 2 == 0) lookup(U, J >> 1);
    // ---- memory side
    if (G == 0 && J >= 1 && J <= 3) load_w(min(t + 2, klast), J - 1);  // packed W(t+2) -> landing registers
    if (G == 4 && J == 1) {
      // W(t+2) has landed (issued two k-steps ago) and, older than it, the A(t+1) DMA: swap register sets
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt0), "+v"(nxt1), "+v"(nxt_am)::"memory");
      swap_sets();
    }
    if (G >= 6 && (J & 1)) dma_a(min(t + 2, klast), t & 1, (G - 6) * 4 + (J >> 1));  // A slot t & 1: free since the barrier
    if (G == 5 && J == 7) {
      // this wave's reads of tile t (issued by G4) and its stores of W(t+1) (the last one in gap 4, ahead of that gap's
      // lookup) are all older than the newest 2 LDS operations (the lookups of gaps 4 and 6)
      asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rebase((t + 1) & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nk; ++t) {
#pragma unroll
    for (int G = 0; G < 8; ++G)
#pragma unroll
      for (int J = 0; J < 8; ++J) gap(t, G, J);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing dummy DMA / stores must not land in the epilogue's staging
#undef FMI_W4Q_RD
#undef FMI_W4Q_FRAG_WAIT
#undef FMI_W4Q_LGKM
  w4_epilogue<ACT>(P, acc, smem + W4Q_LUT_BYTES, m0, n0, wave, wm, wn, lane);
}

}  // namespace fmi
