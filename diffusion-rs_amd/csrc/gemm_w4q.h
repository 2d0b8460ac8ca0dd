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
//   * the 16 code values sit in LDS as f32 (64 B, one bank each: conflict-free, broadcast); per MFMA gap one
//     nibble is turned into a table address (v_bfe on the pre-shifted word) and looked up (ds_read_b32); eight gaps
//     later the value is multiplied by the row's absmax, pairs are rounded to bf16 (v_cvt_pk_bf16_f32) and every
//     eighth gap a 16-byte piece of the row goes to the W ring (ds_write_b128 at slot u ^ ((row >> 1) & 7));
//   * the unit stream is a modulo-8 pipeline that runs across K tiles: the group that looks up unit u consumes unit
//     u-1, the first lookups of tile t+2 ride next to the last writes of tile t+1, so every group is the same.
// LDS: 256 B (table) + A ring 2 x 32 KiB + W ring 2 x 32 KiB.  Per K tile t the order is
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

constexpr int W4Q_LUT_BYTES = 256;

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
  const int band = t_in / (TILE_BAND * tiles_n);
  const int band_h = min(TILE_BAND, tiles_m - band * TILE_BAND);
  const int tin = t_in - band * TILE_BAND * tiles_n;
  const int tn = tin / band_h, tm = band * TILE_BAND + tin % band_h;
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

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
  if (lds0 != 0) __builtin_trap();  // the table lookups address LDS byte 0 directly (smem is the only __shared__ object)
  if (tid < 16) {                   // code -> value table; fp4: value * sign of the tree in dequant.cu:12-37 (sign is exact)
    const float fp4[8] = {0.0f, 5.208333333e-03f, 0.66666667f, 1.0f, 0.33333333f, 0.5f, 0.16666667f, 0.25f};
    reinterpret_cast<float*>(smem)[tid] = P.q_type == 2 ? kNF4[tid] : ((tid & 8) ? -fp4[tid & 7] : fp4[tid & 7]);
  }

  // ---- A operand: LDS-DMA pieces of this wave (1-KiB chunks wave*8 + i), as gemm_w4_kernel
  uint32_t a_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (wave * 8 + i) * 8 + (lane >> 3);
    a_off[i] = (uint32_t)((int64_t)min(m0 + r, P.M - 1) * P.lda * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
  }
  const char* const a_base = reinterpret_cast<const char*>(P.A);
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
  float L[2][8];            // table values of the unit being looked up / consumed (set = unit & 1)
  uint32_t ty = 0, tz = 0;  // the unit's word pre-shifted and masked so that each byte is a table address
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 pk;                 // the unit's 8 bf16 on their way to LDS
  // weight 2k of a unit is the HIGH nibble of byte k, weight 2k+1 the low one (dequant.cu:142-151)
  // (plain int parameters, constant after inlining: a generic lambda cannot name the register arrays in asm operands)
  auto lookup = [&](int u, int j) {
    if (j == 0) {
      ty = (cur[u] << 2) & 0x3c3c3c3cu;
      tz = (cur[u] >> 2) & 0x3c3c3c3cu;
    }
    const uint32_t a = (((j & 1) ? ty : tz) >> (8 * (j >> 1))) & 0xffu;
    asm volatile("ds_read_b32 %0, %1" : "=v"(L[u & 1][j]) : "v"(a));
  };
#define FMI_W4Q_LGKM(N, a, b) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b))
  // pair jp of unit u: two values * absmax -> one dword of bf16; after the fourth pair the 16-byte piece is stored
  auto consume_pair = [&](int u, int jp, int slot) {
    pk[jp] = pack_bf16x2(L[u & 1][2 * jp] * am_c, L[u & 1][2 * jp + 1] * am_c);
    if (jp == 3) {
      const uint32_t ad = (w_wr + slot * TILE) ^ (uint32_t)(u << 4);
      asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(pk) : "memory");
    }
  };

  // ---- prologue: A(0), A(1) by DMA; W(0) and units 0, 1 of W(1) expanded with nothing to overlap
  const int k1 = min(1, klast);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_a(0, 0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_a(k1, 1, i);
  __syncthreads();  // table visible
  auto serial_unit = [&](int u, int slot) {
#pragma unroll
    for (int j = 0; j < 8; ++j) lookup(u, j);
    const int b = u & 1;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(L[b][0]), "+v"(L[b][1]), "+v"(L[b][2]), "+v"(L[b][3]), "+v"(L[b][4]), "+v"(L[b][5]), "+v"(L[b][6]), "+v"(L[b][7]));
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
  for (int j = 0; j < 8; ++j) lookup(2, j);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(L[0][0]), "+v"(L[0][1]), "+v"(L[0][2]), "+v"(L[0][3]), "+v"(L[0][4]), "+v"(L[0][5]), "+v"(L[0][6]), "+v"(L[0][7])
               :
               : "memory");
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
    if (G % 2 == 0 && J == 0) {
      // this step's fragments were read in the first half of the previous step; the second half issued 8 lookups and a store
      if (b) FMI_W4Q_FRAG_WAIT(8, 1);
      else FMI_W4Q_FRAG_WAIT(8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    acc[q >> 2][q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[b][q & 3]), __builtin_bit_cast(bf16x8_t, xf[b][q >> 2]),
                                                                 acc[q >> 2][q & 3], 0, 0, 0);
    // ---- fragment reads of the next k-step (during step 3: step 0 of tile t + 1; addresses rebased after the barrier)
    if (G % 2 == 0) read_frag((step + 1) & 3, J);
    // ---- consume unit UC: W(t+1) through G5, W(t+2) from G6 on
    const int wslot = (G >= 6 ? t : t + 1) & 1;
    if (G == 6 && J == 0) am_c = cur_am;
    const int c = UC & 1;
    if (G == 5) {
      // the last unit of W(t+1): one pair per gap, so that its store is old when the barrier waits for it.
      // Lookups issued after the younger value of pair J: 6 - 2J in G4 and J in G5.
      if (J == 0) FMI_W4Q_LGKM(6, L[c][0], L[c][1]);
      if (J == 1) FMI_W4Q_LGKM(5, L[c][2], L[c][3]);
      if (J == 2) FMI_W4Q_LGKM(4, L[c][4], L[c][5]);
      if (J == 3) FMI_W4Q_LGKM(3, L[c][6], L[c][7]);
      if (J < 4) consume_pair(UC, J, wslot);
    } else if (J % 2 == 0) {
      FMI_W4Q_LGKM(6, L[c][J], L[c][J + 1]);  // at least 6 lookups were issued after the pair's younger value
    } else {
      consume_pair(UC, J >> 1, wslot);
    }
    // ---- look up nibble J of unit U
    lookup(U, J);
    // ---- memory side
    if (G == 0 && J >= 1 && J <= 3) load_w(min(t + 2, klast), J - 1);  // packed W(t+2) -> landing registers
    if (G == 4 && J == 1) {
      // W(t+2) has landed (issued two k-steps ago) and, older than it, the A(t+1) DMA: swap register sets
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt0), "+v"(nxt1), "+v"(nxt_am)::"memory");
      swap_sets();
    }
    if (G >= 6 && (J & 1)) dma_a(min(t + 2, klast), t & 1, (G - 6) * 4 + (J >> 1));  // A slot t & 1: free since the barrier
    if (G == 5 && J == 7) {
      // this wave's reads of tile t (issued by G4) and its stores of W(t+1) (the last one in gap 3) are all older
      // than the newest 4 LDS operations (the lookups of gaps 4..7)
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
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
  // epilogue: as gemm_w4_kernel — two rounds through the 8-wave epilogue (see there)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  {
    f32x16 hacc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) hacc[i][0] = acc[i][0], hacc[i][1] = acc[i][1];
#pragma clang loop unroll(disable)
    for (int r = 0; r < 2; ++r) {
      gemm_epilogue<2, 4, ACT>(P, hacc, smem + W4Q_LUT_BYTES, m0, n0, wm * 4 + wn * 2 + r, lane_e);
#pragma unroll
      for (int i = 0; i < 4; ++i) hacc[i][0] = acc[i][2], hacc[i][1] = acc[i][3];
      asm volatile("" : "+v"(lane_e));
    }
  }
}

}  // namespace fmi
