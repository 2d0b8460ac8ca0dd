#!/usr/bin/env python
"""Generates diffusion-rs_amd/csrc/attention_w16_loop.inc: the whole KV stream of attention_w16_kernel (attention_w16.h)
as ONE inline-asm statement with hand-assigned registers.

attention_w16_kernel is round 3's joint-attention kernel: the machine mapping of attention_w4_kernel (one wave per SIMD, 64
query rows per wave in two blocks b = 0, 1 that run half a KV tile apart, K / V^T tiles in 4-deep LDS-DMA rings, one barrier
per tile, the softmax of one block sliced into the MFMA gaps of the other) rebuilt on three changes:

  1. `v_mfma_f32_16x16x32_bf16` for both products (this part sustains 14 % more on it than on 32x32x16).  A wave's
     S^T tile of 64 keys x 64 queries is 4 key blocks (a) x 2 query blocks (c) of 16 x 16 per block b; a lane (g = lane / 16,
     n = lane % 16) owns 4 keys x 1 query of each (rows 4 g + i of the key block, column n).  A K / V^T fragment (one
     ds_read_b128 per lane) feeds TWO MFMAs (c = 0, 1), so the LDS fragment traffic per FLOP is unchanged.
  2. scale and the running maximum are folded into the QK^T product: Q is pre-multiplied by scale * log2(e) when its fragments
     are loaded (one rounding to bf16 — a stated parity decision), and the first d-step of a score tile accumulates
     onto a register block holding -m of the lane's query (NM) instead of 0.  The scores leave the matrix pipe as
     s' = s * scale * log2(e) - m: the softmax needs no fma per score any more, p = exp2(s') in place.
  3. the row statistics never cross lanes on the common path: the deferred-rescale decision is "some s' of the wave exceeds the
     threshold" (one running v_max3 over the lane's 32 scores, one compare, one branch); only the rarely taken rescale block
     reduces the per-query maxima over the four lane groups (v_permlane32_swap / v_permlane16_swap).
  4. the row sums come out of the matrix pipe: V^T is extended by one row of ones — a constant A fragment in registers (row 0 =
     1.0, rows 1..15 = 0), one extra MFMA per (k-step, query block) into the accumulators OL[b][c] — so l = sum of the
     bf16-rounded p (exactly the P the second product multiplies) costs 8 MFMAs per KV tile (+6 %) instead of 64 VALU adds.
     Why that trade: tools/gen_issue_model.py (profiles/r03_issue_model.txt) — with one wave per SIMD a 16-clock MFMA hides only
     ~8 clocks of other instructions, v_exp_f32 costs 8, VOP3 4.5, and packed-f32 / dot2 VALU ops stall the matrix pipe outright
     (v_pk_add_f32: +14 clocks each), so the kernel is bound by what is issued BETWEEN the MFMAs, not by the MFMAs.
  => 2 VALU instructions per score (exp, half a max3, half a cvt_pk) instead of 4.

P feeds the second product straight from the S registers: k-step kk (32 keys) of block b, query block c is
pack(S[2kk][c][0..3], S[2kk+1][c][0..3]); the key that (block a, row m) stands for is chosen so that this matches the
k-permutation already baked into V^T (attention.hip: vt_perm) — key = 32 (a >> 1) + 8 (a & 1) + (m & 7) + 16 (m >> 3) — so
V^T and its producers are unchanged; only the K ring's swizzle is this kernel's own (16-byte slot p of row r holds global slot
p ^ f(r), f(r) = (r & 7) | ((r >> 4) & 1) << 3: the 16 rows of a fragment read hit 16 distinct slots).

Stream (n = number of KV tiles >= 2):
    pre   QK(0,0)
    A0    QK(1,0)                 | softmax(0,0) | barrier | DMA V^T(2)          ; K address registers -> slot 1
    loop t = 0 ..:
      B(t)    PV(0,t) + QK(0,t+1) | softmax(1,t)           | DMA K(t+3)          ; t == n-1: QK(0,n) reads a stale ring slot, unused
      if t == n-1: break
      A(t+1)  PV(1,t) + QK(1,t+1) | softmax(0,t+1) | barrier | DMA V^T(t+3)      ; K, V^T address registers -> next slot
    post  PV(1,n-1)
The fragment stream (each fragment read LOOKAHEAD MFMA slots ahead of its first use, into the buffer of a fragment whose last
MFMA has left the front of the matrix pipe — rule 3 of DESIGN 4.4, asserted below) is continuous from pre to the loop's end;
post drains and fetches its own first fragments.

fp8 mode (AW16_MODE=fp8qk -> attention_w16f8_loop.inc; BASELINE configs[4]): Q and K arrive as OCP e4m3 bytes with
static scales (the model's fp8 mode emits them from the fused QKV epilogue).  The score product is then ONE
v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile — the whole head dimension in one instruction, 8 per block and tile instead
of 32, at twice the rate — whose block scales carry 2^-n: the host picks the q scale so that scale * log2(e) / (sq * sk) is
exactly a power of two, and the fold (-m as the accumulator's start value) works unchanged.  A K tile is 8 KiB (rows of 128
bytes, 16-byte slot p of row r holds global slot p ^ f8(r), f8(r) = ((r & 7) >> 1) | ((r >> 4) & 1) << 2), a K fragment is 32
bytes per lane (two ds_read_b128 into two adjacent fragment buffers), Q fragments are 8 registers.  P, V^T and the second
product are the bf16 ones.

Register map (pinned by the operand constraints in attention_w16.h):
  a[0:127]    O^T   O[b][dt][c]  -> a[((8b+dt)*2+c)*4 ..]        a[128:191]  Q fragments QF[b][c][s] -> a[128+((2b+c)*4+s)*4 ..]
  a[192:207]  OL[b][c]: the ones-row product (register 0 of lanes 0..15 = the row sum of query n)
  v[0:63]     S^T   S[b][a][c]   -> v[((4b+a)*2+c)*4 ..]         v[64:95]    P[b][kk][c] -> v[64+((2b+kk)*2+c)*4 ..]
  v[96:127]   fragment buffers FR[0..7]
  v[128:131]  KAD[s]  v[132:133] VAD[kk]  v[134:137] k_voff  v[138:141] v_voff  v[142:145] k_voff clamped (ragged last tile)
  v146        lane key offset 16 (g >> 1) + 4 (g & 1)   v147 DMA offset temporary
  v[148:163]  NM[b][c] (-m of the lane's query, 4 copies)   v[164:167] M[b][c]   v[168:171] the ones fragment (row 0 = bf16 1.0)
  v[184:213]  temporaries (clobbers)   s[80:95] loop state (clobbers)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 16384
VT_RING = 4 * TILE
NBUF = 8          # fragment buffers
LOOKAHEAD = int(os.environ.get("AW16_LOOKAHEAD", "12"))    # a fragment is read this many MFMA slots ahead of its first use (about 6 reads in flight)
MODE = os.environ.get("AW16_MODE", "bf16")  # "bf16" | "fp8qk": Q and K as OCP e4m3 (see the fp8 notes below), P and V^T stay bf16
FP8 = MODE == "fp8qk"
TILE_K = 8192 if FP8 else TILE  # bytes of a K tile in LDS (64 keys x 128 d)
X = os.environ.get("AW16_X", "")  # timing experiments only (wrong results): novalu | noexp | nodma | nobarrier | halfreads | nomfma | nowait

PMAX, TA, TB, T0, T1, AL, DL = (f"v{n}" for n in range(184, 191))
XT = [f"v{n}" for n in range(191, 197)]
PM0, PM1 = "v197", "v198"
PMX = [f"v{n}" for n in range(199, 214)]  # partial maxima of the tree
LKEY, DMAT = "v146", "v147"
S_KP, S_VP, S_MASK = "s[80:81]", "s[82:83]", "s[84:85]"
S_T, S_TILE, S_M0K, S_M0V, S_TMP, S_MKK, S_MKV, S_RAG, S_TMP2, S_FLAG = "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95"
NEG_BIG = "0xf149f2ca"  # -1e30f


def O(b, dt, c):
    lo = ((8 * b + dt) * 2 + c) * 4
    return f"a[{lo}:{lo + 3}]"


def Or(b, c, k):  # k = 0..31: register i of d block dt, k = 4 dt + i
    return f"a{((8 * b + (k >> 2)) * 2 + c) * 4 + (k & 3)}"


def QF(b, c, s):
    lo = 128 + ((2 * b + c) * 4 + s) * 4
    return f"a[{lo}:{lo + 3}]"


def QF8(b, c):
    lo = 128 + (2 * b + c) * 8
    return f"a[{lo}:{lo + 7}]"


SCA, SCB = "v172", "v173"  # fp8 mode: E8M0 block scales of the score product (2^-n on the K side, 1 on the Q side)


def S(b, a, c):
    lo = ((4 * b + a) * 2 + c) * 4
    return f"v[{lo}:{lo + 3}]"


def Sr(b, a, c, i):
    return f"v{((4 * b + a) * 2 + c) * 4 + i}"


def P(b, kk, c):
    lo = 64 + ((2 * b + kk) * 2 + c) * 4
    return f"v[{lo}:{lo + 3}]"


def Pr(b, kk, c, d):
    return f"v{64 + ((2 * b + kk) * 2 + c) * 4 + d}"


NBUFK = 4  # fp8 mode: 32-byte K fragments live in their own pool of four 8-register buffers in the accumulator half, a[208:239]


def FR(n, pool="V", half=None):
    """register range of fragment buffer n of a pool: "V" = v[96:127] (8 x 4 registers: every bf16 fragment);  "K8" = a[208:239]
    (4 x 8 registers; half = 0 / 1 selects the 16 bytes one ds_read_b128 fills)"""
    if pool == "K8":
        lo = 208 + 8 * (n % NBUFK)
        return f"a[{lo}:{lo + 7}]" if half is None else f"a[{lo + 4 * half}:{lo + 4 * half + 3}]"
    lo = 96 + 4 * (n % NBUF)
    return f"v[{lo}:{lo + 3}]"


def KAD(s):
    return f"v{128 + s}"


def VAD(kk):
    return f"v{132 + kk}"


def NM(b, c):
    lo = 148 + (2 * b + c) * 4
    return f"v[{lo}:{lo + 3}]"


def NMr(b, c, i):
    return f"v{148 + (2 * b + c) * 4 + i}"


def M(b, c):
    return f"v{164 + 2 * b + c}"


def OL(b, c):
    lo = 192 + (2 * b + c) * 4
    return f"a[{lo}:{lo + 3}]"


ONESF = "v[168:171]"


# ----------------------------------------------------------------------------------------------------------------------------
# MFMA sequences.  A product's 32 MFMAs use 16 fragments, each for two consecutive MFMAs of that product (c = 0, 1).
def pv_seq(b):
    """(MFMA text, V^T fragment index or None) of PV(b): per k-step 8 d blocks x 2 query blocks, then the ones row (row sums)"""
    out = []
    for kk in range(2):
        for dt in range(8):
            for c in range(2):
                out.append((f"v_mfma_f32_16x16x32_bf16 {O(b, dt, c)}, {{fr}}, {P(b, kk, c)}, {O(b, dt, c)}", 8 * kk + dt))
        for c in range(2):
            out.append((f"v_mfma_f32_16x16x32_bf16 {OL(b, c)}, {ONESF}, {P(b, kk, c)}, {OL(b, c)}", None))
    return out


def qk_seq(b):
    if FP8:  # one MFMA per score tile: key block a = j >> 1, query block c = j & 1; fragment a (32 bytes per lane)
        return [(f"v_mfma_scale_f32_16x16x128_f8f6f4 {S(b, j >> 1, j & 1)}, {{fr}}, {QF8(b, j & 1)}, {NM(b, j & 1)}, {SCA}, {SCB}", j >> 1) for j in range(8)]
    return [(qk_mfma(b, j), j >> 1) for j in range(32)]


def qk_mfma(b, j):
    s, a, c = j >> 3, (j >> 1) & 3, j & 1
    acc = NM(b, c) if s == 0 else S(b, a, c)  # first d-step: start from -m of the lane's query (the fold)
    return f"v_mfma_f32_16x16x32_bf16 {S(b, a, c)}, {{fr}}, {QF(b, c, s)}, {acc}"


def pv_frag(f):  # fragment f = 0..15 of a PV product: k-step kk = f >> 3, d block dt = f & 7
    return [(VAD(f >> 3), (f & 7) * 2048)]


def qk_frag(f):  # fragment f = 0..15 of a QK^T product: d-step s = f >> 2, key block a = f & 3
    if FP8:  # fragment a = f: rows of 128 bytes; two 16-byte reads (KAD[0], KAD[1] = KAD[0] ^ 16) -> 8 registers
        imm = (f >> 1) * 4096 + (f & 1) * 1024
        return [(KAD(0), imm), (KAD(1), imm)]
    a = f & 3
    return [(KAD(f >> 2), (a >> 1) * 8192 + (a & 1) * 2048)]


class Phase:
    """One phase: its MFMA slots (the MFMA text with a {fr} hole, and the index of its fragment in the phase's fragment list)
    and the fragments (address register, immediate) in order of first use."""

    def __init__(self, name, pv_b, qk_b, sm_b, dma, barrier, advance, forced_rescale=False):
        self.name, self.pv_b, self.qk_b, self.sm_b = name, pv_b, qk_b, sm_b
        self.dma, self.barrier, self.advance, self.forced_rescale = dma, barrier, advance, forced_rescale
        self.mfma, self.frags = [], []
        if pv_b is not None and qk_b is not None:
            pv, qk = pv_seq(pv_b), qk_seq(qk_b)
            # alternate PV / QK^T (consecutive MFMAs never share an accumulator); the phase's fragments are numbered in order of first use
            seq = []
            if FP8:  # 36 PV MFMAs and 8 score MFMAs: spread the latter evenly
                npv, nqk, ip, iq = len(pv), len(qk), 0, 0
                while ip < npv or iq < nqk:
                    if iq >= nqk or (ip < npv and (ip + 1) * nqk <= (iq + 1) * npv):
                        t, f = pv[ip]
                        ip += 1
                        seq.append((t, None if f is None else ("V", f)))
                    else:
                        t, f = qk[iq]
                        iq += 1
                        seq.append((t, ("K", f)))
                pv, qk = [], []
            while pv or qk:
                if pv:
                    t, f = pv.pop(0)
                    seq.append((t, None if f is None else ("V", f)))
                if qk:
                    t, f = qk.pop(0)
                    seq.append((t, ("K", f)))
            index = {}
            for t, key in seq:
                if key is not None and key not in index:
                    index[key] = len(self.frags)
                    self.frags.append(pv_frag(key[1]) if key[0] == "V" else qk_frag(key[1]))
                self.mfma.append((t, None if key is None else index[key]))
        else:
            self.mfma = pv_seq(pv_b) if pv_b is not None else qk_seq(qk_b)
            nfr = 1 + max(f for _, f in self.mfma if f is not None)
            self.frags = [pv_frag(f) if pv_b is not None else qk_frag(f) for f in range(nfr)]
        self.n = len(self.mfma)
        self.fu = [min(i for i, (_, ff) in enumerate(self.mfma) if ff == f) for f in range(len(self.frags))]
        self.lu = [max(i for i, (_, ff) in enumerate(self.mfma) if ff == f) for f in range(len(self.frags))]
        assert self.fu == sorted(self.fu), self.fu
        # fragment buffers: per pool, numbered in order of first use; a phase's span in each pool is padded to a multiple of the
        # pool size, so every phase starts at buffer 0 of both pools
        self.pool = ["K8" if len(fr) == 2 else "V" for fr in self.frags]
        self.bpos, cnt = [], {"V": 0, "K8": 0}
        for pl in self.pool:
            self.bpos.append(cnt[pl])
            cnt[pl] += 1
        self.span = {"V": -(-cnt["V"] // NBUF) * NBUF, "K8": -(-cnt["K8"] // NBUFK) * NBUFK}
        self.buf0 = 0


# ----------------------------------------------------------------------------------------------------------------------------
# softmax of block b as instruction streams
def mask_block(b):
    """ragged last tile: scores of keys >= Lk become -1e30 (p = 0).  The key of (a, i) in lane (g, n) is
    32 (a >> 1) + 8 (a & 1) + i + LKEY; S_RAG = keys in the last tile (1..64)."""
    out = [f"v_mov_b32 {T1}, {NEG_BIG}"]
    for a in range(4):
        for i in range(4):
            koff = 32 * (a >> 1) + 8 * (a & 1) + i
            out.append(f"s_sub_i32 {S_TMP2}, {S_RAG}, {koff}")          # key valid <=> LKEY + koff < rag
            out.append(f"v_cmp_le_i32 vcc, {S_TMP2}, {LKEY}")
            for c in range(2):
                out.append(f"v_cndmask_b32 {Sr(b, a, c, i)}, {Sr(b, a, c, i)}, {T1}, vcc")
    return out


def max_chain(dst, regs):
    out = [f"v_max3_f32 {dst}, {regs[0]}, {regs[1]}, {regs[2]}"]
    k = 3
    while k + 1 < len(regs):
        out.append(f"v_max3_f32 {dst}, {dst}, {regs[k]}, {regs[k + 1]}")
        k += 2
    if k < len(regs):
        out.append(f"v_max_f32 {dst}, {dst}, {regs[k]}")
    return out


def max_stream(b):
    """maximum over the lane's 32 scores of block b (both queries): the common path only needs "does any score exceed the
    threshold".  A tree of independent v_max3 (a serial chain on one register costs ~13 clocks per link with one wave per SIMD:
    measured), key blocks in order a = 0..3 (the order the QK^T product finishes them in)."""
    regs = [Sr(b, a, c, i) for a in range(4) for c in range(2) for i in range(4)]
    out, level = [], []
    for k in range(0, 30, 3):
        out.append(f"v_max3_f32 {PMX[k // 3]}, {regs[k]}, {regs[k + 1]}, {regs[k + 2]}")
        level.append(PMX[k // 3])
    out.append(f"v_max_f32 {PMX[10]}, {regs[30]}, {regs[31]}")
    level.append(PMX[10])
    out += [f"v_max3_f32 {PMX[11]}, {level[0]}, {level[1]}, {level[2]}", f"v_max3_f32 {PMX[12]}, {level[3]}, {level[4]}, {level[5]}",
            f"v_max3_f32 {PMX[13]}, {level[6]}, {level[7]}, {level[8]}", f"v_max_f32 {PMX[14]}, {level[9]}, {level[10]}",
            f"v_max3_f32 {PMAX}, {PMX[11]}, {PMX[12]}, {PMX[13]}", f"v_max_f32 {PMAX}, {PMAX}, {PMX[14]}"]
    return out


def rescale_block(b):
    """Taken when some score of the wave exceeds the threshold, and always on a block's first tile (M = -1e30, fold = 0): per
    query c the new maximum m' = max(M, max s' - NM), delta = m' + NM (how far this tile's fold was off), alpha =
    exp2(min(-delta, 0));  then s' -= delta, O^T *= alpha, L *= alpha, M = m', NM = -m'."""
    out = []
    for c, pm in ((0, PM0), (1, PM1)):
        out += max_chain(pm, [Sr(b, a, c, i) for a in range(4) for i in range(4)])
    # reduce over the four lane groups (lanes n, n + 16, n + 32, n + 48 hold the same query): after the first swap the lower half
    # of the wave works on query 0 and the upper half on query 1; the last swap hands every lane both results
    out += ["s_nop 1",
            f"v_permlane32_swap_b32 {PM0}, {PM1}",       # PM0 = [q0.r0, q0.r1, q1.r0, q1.r1]  PM1 = [q0.r2, q0.r3, q1.r2, q1.r3]
            "s_nop 1",
            f"v_max_f32 {TA}, {PM0}, {PM1}",
            f"v_mov_b32 {TB}, {TA}",
            "s_nop 1",
            f"v_permlane16_swap_b32 {TA}, {TB}",         # TA = [r0, r0, r2, r2]  TB = [r1, r1, r3, r3]
            "s_nop 1",
            f"v_max_f32 {TA}, {TA}, {TB}",
            f"v_mov_b32 {TB}, {TA}",
            "s_nop 1",
            f"v_permlane32_swap_b32 {TA}, {TB}",         # TA = query 0's maximum in every lane, TB = query 1's
            "s_nop 1"]
    for c, ps in ((0, TA), (1, TB)):
        out += [f"v_sub_f32 {T0}, {ps}, {NMr(b, c, 0)}",            # the maximum in the unshifted domain: ps' - NM
                f"v_max_f32 {T0}, {M(b, c)}, {T0}",                  # m'
                f"v_add_f32 {DL}, {T0}, {NMr(b, c, 0)}",             # delta = m' - (the m this tile's fold used)
                f"v_max_f32 {AL}, {DL}, 0",
                f"v_sub_f32 {AL}, 0, {AL}",
                f"v_exp_f32 {AL}, {AL}",                             # alpha = exp2(-max(delta, 0))
                f"v_mov_b32 {M(b, c)}, {T0}"]
        out += [f"v_sub_f32 {NMr(b, c, i)}, 0, {T0}" for i in range(4)]
        out += [f"v_sub_f32 {Sr(b, a, c, i)}, {Sr(b, a, c, i)}, {DL}" for a in range(4) for i in range(4)]
        lo = 192 + (2 * b + c) * 4
        out += [f"v_accvgpr_read_b32 {T1}, a{lo}", f"s_nop 0", f"v_mul_f32 {T1}, {T1}, {AL}", f"s_nop 0", f"v_accvgpr_write_b32 a{lo}, {T1}"]
        n = len(XT)
        out.append(f"v_accvgpr_read_b32 {XT[0]}, {Or(b, c, 0)}")
        for r in range(32):  # software pipeline over the 32 accumulator registers of (b, c)
            if r + 1 < 32:
                out.append(f"v_accvgpr_read_b32 {XT[(r + 1) % n]}, {Or(b, c, r + 1)}")
            out.append(f"v_mul_f32 {XT[r % n]}, {XT[r % n]}, {AL}")
            out.append(f"v_accvgpr_write_b32 {Or(b, c, r)}, {XT[r % n]}")
    return out


def exp_stream(b):
    """p = exp2(s') in place, then per unit of 4 scores (a, c): two packs to bf16 (P fragment dwords; the row sums come from the
    ones-row MFMAs).  The packs of unit u follow the exponentials of unit u + 2: a v_exp_f32 result must not be read within the
    next few VALU instructions (gfx950: stale in half of the lanes, DESIGN 4.4 rule 2)."""
    units = [(a, c) for a in range(4) for c in range(2)]
    out = []

    def start(a, c):
        return [f"v_exp_f32 {Sr(b, a, c, i)}, {Sr(b, a, c, i)}" for i in range(4)]

    def finish(a, c):
        kk, h = a >> 1, a & 1  # P[b][kk][c] dwords 2h, 2h + 1
        o = []
        for d in range(2):
            o.append(f"v_cvt_pk_bf16_f32 {Pr(b, kk, c, 2 * h + d)}, {Sr(b, a, c, 2 * d)}, {Sr(b, a, c, 2 * d + 1)}")
        return o

    SKEW = 2
    for u in range(len(units) + SKEW):
        if u < len(units):
            out += start(*units[u])
        if u >= SKEW:
            out += finish(*units[u - SKEW])
    return out


def spread(plan, stream, first, last):
    """stream instructions over gaps first..last (inclusive), as evenly as integer division allows, in order"""
    n = last - first + 1
    for k, ins in enumerate(stream):
        plan[first + k * n // len(stream)].append(ins)


def softmax_plan(ph, uid):
    """instruction lists per gap for the softmax of ph.sm_b: mask branch, running max, decision + rescale branch, exp stream.
    S^T(b) was finished by the previous phase's last QK^T MFMAs (key block 3 by its very last two): nothing reads it before
    gap 4 (>= 4 MFMAs = 64+ clocks behind; an MFMA result needs ~40)."""
    b, n = ph.sm_b, ph.n
    plan = [[] for _ in range(n)]
    g_mask = 4 if n >= 16 else 0   # (fp8 mode's first phases are 8 MFMAs long and entered drained: S^T is complete at slot 0)
    skipm = f".Law16_nomask_{uid}_%="
    plan[g_mask] += [f"s_cmp_eq_u32 {S_FLAG}, 0", f"s_cbranch_scc1 {skipm}"] + mask_block(b) + [f"{skipm}:"]
    span = 8 if n >= 64 else 4 if n >= 16 else 2
    spread(plan, max_stream(b), g_mask + 1, g_mask + span)
    g_dec = g_mask + span + 1
    skip, do = f".Law16_skip_{uid}_%=", f".Law16_resc_{uid}_%="
    dec = [f"v_cmp_lt_f32 vcc, %[thr], {PMAX}"]
    if ph.forced_rescale == "always":      # softmax(0,0): this block's first tile
        dec += []
    elif ph.forced_rescale == "t0":        # softmax(1,t): first tile when t == 0
        dec += [f"s_cbranch_vccnz {do}", f"s_cmp_eq_u32 {S_T}, 0", f"s_cbranch_scc0 {skip}", f"{do}:"]
    elif ph.forced_rescale == "tm1":       # softmax(0,t+1) inside the loop is never a first tile
        dec += [f"s_cbranch_vccz {skip}"]
    plan[g_dec] += dec + rescale_block(b) + [f"{skip}:"]
    spread(plan, exp_stream(b), g_dec + 1, n - 1)
    return plan


# ----------------------------------------------------------------------------------------------------------------------------
def emit_phase(ph, nxt, uid, own_prefetch=False, drain=False):
    """asm lines of one phase.  `nxt` = the phase whose first fragments are fetched behind this phase's last MFMAs (None: none).
    Fragment f of a phase is read in the gap behind MFMA slot fu[f] - LOOKAHEAD (a negative slot: in the previous phase's tail, or
    in front of the phase when own_prefetch)."""
    o = [f"; ==== phase {ph.name}"]
    n = ph.n
    nf = len(ph.frags)
    plan = softmax_plan(ph, uid) if ph.sm_b is not None else [[] for _ in range(n)]
    # ---- the read instructions of every gap, in stream order: (key, register range, reg, imm); key = (0, f) own fragment f, (1, f) the next phase's
    def rd_regs(p_, f, k):
        return FR(p_.bpos[f], p_.pool[f], k if p_.pool[f] == "K8" else None)

    reads = [[] for _ in range(n)]
    early = []                               # read before slot 0 (previous phase's tail or own prefetch), in order
    for f, fr in enumerate(ph.frags):
        i = ph.fu[f] - LOOKAHEAD
        for k, (reg, imm) in enumerate(fr):
            (reads[i] if i >= 0 else early).append(((0, f), rd_regs(ph, f, k), reg, imm))
    own_last_read = max([g for g in range(n) if reads[g]], default=-1)
    if nxt is not None:
        for f, fr in enumerate(nxt.frags):
            i = n + nxt.fu[f] - LOOKAHEAD
            if i < n:
                assert i > own_last_read, (ph.name, "next phase's reads must follow the own ones")
                for k, (reg, imm) in enumerate(fr):
                    reads[i].append(((1, f), rd_regs(nxt, f, k), reg, imm))
    if own_prefetch:
        for (_, b_, reg, imm) in early:
            o.append(f"ds_read_b128 {b_}, {reg} offset:{imm}")
    # position of every read in issue order (early ones first); last[f] = position of fragment f's last read
    order = [key for (key, _, _, _) in early]
    issued_before_slot = [len(order)]
    for g in range(n):
        order += [key for (key, _, _, _) in reads[g]]
        issued_before_slot.append(len(order))   # issued before MFMA slot g + 1
    last = {}
    for k, key in enumerate(order):
        last[key] = k
    # ---- ring-slot advance of the address registers: each register right behind the last own read that uses it (the next
    # phase's reads of that register come later by construction: asserted)
    adv_at = [[] for _ in range(n)]
    if ph.advance:
        regs = ([(KAD(s_), S_MKK) for s_ in range(2 if FP8 else 4)] if "K" in ph.advance else []) + ([(VAD(k_), S_MKV) for k_ in range(2)] if "V" in ph.advance else [])
        for reg, mask in regs:
            own = [g for g in range(n) for (key, _, r_, _) in reads[g] if key[0] == 0 and r_ == reg]
            g_last = max(own, default=0)
            nxt_use = [g for g in range(n) for (key, _, r_, _) in reads[g] if key[0] == 1 and r_ == reg]
            assert all(g > g_last for g in nxt_use), (ph.name, reg, g_last, nxt_use)
            adv_at[g_last].append(f"v_xor_b32 {reg}, {mask}, {reg}")
    q = n // 4
    for i in range(n):
        text, f = ph.mfma[i]
        pre, post = [], []
        # ---- DMA pieces (4 per phase): B stages K(tile) once per quarter, A stages V^T(tile) in the second half (behind the barrier)
        dma = None
        if ph.dma == "K" and i % q == q // 2 - 1 and i // q < (2 if FP8 else 4) and X != "nodma":
            piece = i // q
            pre.append(f"s_add_i32 m0, {S_M0K}, {piece * 1024}")
            pre.append(f"v_cndmask_b32 {DMAT}, v{134 + piece}, v{142 + piece}, {S_MASK}")
            dma = f"global_load_lds_dwordx4 {DMAT}, {S_KP}"
        if ph.dma == "V" and i >= n // 2 and (i - n // 2) % (n // 8) == n // 8 - 1 and X != "nodma":
            piece = (i - n // 2) // (n // 8)
            pre.append(f"s_add_i32 m0, {S_M0V}, {piece * 1024}")
            dma = f"global_load_lds_dwordx4 v{138 + piece}, {S_VP}"
        # ---- counted wait (LDS reads retire in order) every fourth slot, for every fragment first used in slots i .. i + 3
        if i % 4 == 0:
            need = [f2 for f2 in range(nf) if i <= ph.fu[f2] < i + 4]
            if need:
                younger = issued_before_slot[i] - last[(0, max(need))] - 1
                assert 0 <= younger <= 15, (ph.name, i, younger)
                pre.append(f"s_waitcnt lgkmcnt({younger})")
        mf = text.format(fr=FR(ph.bpos[f], ph.pool[f])) if f is not None else text
        rd = [f"ds_read_b128 {b_}, {reg} offset:{imm}" for (_, b_, reg, imm) in reads[i]]
        post += adv_at[i]
        post += plan[i]
        if X == "halfreads":
            rd = [r_ if k % 2 == 0 else "s_nop 0" for k, r_ in enumerate(rd)] if i % 4 < 2 else ["s_nop 0" for _ in rd]
        if X == "nomfma":
            mf = "s_nop 0"
        if X == "halfpv" and f is not None and ph.pool[f] == "V" and (i // 2) % 2:  # every second PV MFMA pair dropped: what would fp8 P V buy?
            mf = "s_nop 0"
        if X == "nopv" and (f is None or ph.pool[f] == "V"):
            mf = "s_nop 0"
        if X == "nowait":
            pre = [p_ for p_ in pre if not p_.startswith("s_waitcnt lgkmcnt")]
        if X.startswith("drop_"):  # drop every instruction whose mnemonic starts with one of the '+'-separated prefixes
            pref = tuple(X[5:].split("+"))
            post = [p_ for p_ in post if not p_.startswith(pref)]
        if X == "novalu":
            post = [p_ for p_ in post if p_.startswith(("s_", ".Law16", "v_xor", "v_cmp"))]
        if X == "noexp":
            post = [p_.replace("v_exp_f32", "v_mov_b32") for p_ in post]
        o.append(f"; slot {i}")
        o += pre + [mf] + rd
        if dma:
            o.append(dma)
        o += post
        if ph.barrier and i == n // 2 and X != "nobarrier":
            # everything but this wave's newest pieces — V^T(t+2) [4] and K(t+3) [4; 2 in fp8 mode] — has landed: K(t+2), V^T(t+1)
            o += [f"s_waitcnt vmcnt({4 + (2 if FP8 else 4)})", "s_barrier"]
    if drain:
        o += ["s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15"]
    return o


def early_reads(ph):
    """the reads of ph's fragments that precede its slot 0 (what the previous phase's tail, or an own prefetch, issues), in order"""
    out = []
    for f, fr in enumerate(ph.frags):
        if ph.fu[f] - LOOKAHEAD < 0:
            for k, (reg, imm) in enumerate(fr):
                out.append(f"ds_read_b128 {FR(ph.bpos[f], ph.pool[f], k if ph.pool[f] == 'K8' else None)}, {reg} offset:{imm}")
    return out


def check_rule3(seq):
    """Linearise a sequence of phases and assert that every read refills a buffer whose previous fragment's LAST MFMA sits
    strictly before the MFMA slot the read is issued behind (so a later MFMA has issued and the old operand has left the front
    of the matrix pipe), and that reads are issued in stream order."""
    base, last_user, prev_rd = 0, {}, None
    for ph in seq:
        for f, fr in enumerate(ph.frags):
            rd = base + ph.fu[f] - LOOKAHEAD
            assert prev_rd is None or rd >= prev_rd, ("stream order", ph.name, f)
            prev_rd = rd
            pb = (ph.pool[f], ph.bpos[f] % (NBUFK if ph.pool[f] == "K8" else NBUF))
            assert last_user.get(pb, -10**9) < rd, ("rule 3", ph.name, f, pb, last_user.get(pb), rd)
            last_user[pb] = base + ph.lu[f]
        base += ph.n


def build():
    pre = Phase("pre: QK(0,0)", None, 0, None, None, False, "")
    a0 = Phase("A0: QK(1,0) | softmax(0,0)", None, 1, 0, "V", True, "K", forced_rescale="always")
    bt = Phase("B(t): PV(0,t) + QK(0,t+1) | softmax(1,t)", 0, 0, 1, "K", False, "", forced_rescale="t0")
    at = Phase("A(t+1): PV(1,t) + QK(1,t+1) | softmax(0,t+1)", 1, 1, 0, "V", True, "KV", forced_rescale="tm1")
    post = Phase("post: PV(1,n-1)", 1, None, None, None, False, "")
    return pre, a0, bt, at, post


def loop():
    pre, a0, bt, at, post = build()
    # every phase's fragment count is a multiple of the pool size (16 / 32 bf16 fragments, 4 fp8 K fragments): all start at buffer 0
    for ph in (pre, a0, bt, at, post):
        assert ph.span["V"] == sum(1 for p_ in ph.pool if p_ == "V") and ph.span["K8"] == sum(1 for p_ in ph.pool if p_ == "K8"), ph.name
    if FP8:  # pre and A0 are only 8 MFMAs long there: each fetches its own first fragments and ends drained (see below)
        check_rule3([pre])
        check_rule3([a0])
        check_rule3([bt, at, bt, at, bt])
    else:
        check_rule3([pre, a0, bt, at, bt, at, bt])
    check_rule3([post])

    def dma_setup():
        """scalar state of one loop iteration: the tile both DMA streams fetch, min(t + 3, n - 1), and its ring slot"""
        kshift = 13 if FP8 else 14   # log2 of a K tile's bytes, in HBM and in LDS (64 keys x 128 d)
        return [f"s_add_i32 {S_TILE}, {S_T}, 3",
                f"s_min_i32 {S_TILE}, {S_TILE}, %[ntm1]",
                f"s_lshl_b32 {S_TMP}, {S_TILE}, {kshift}",
                f"s_add_u32 s80, %[kb_lo], {S_TMP}",
                f"s_addc_u32 s81, %[kb_hi], 0",
                f"s_lshl_b32 {S_TMP}, {S_TILE}, 7",
                f"s_add_u32 s82, %[vb_lo], {S_TMP}",
                f"s_addc_u32 s83, %[vb_hi], 0",
                f"s_and_b32 {S_TMP}, {S_TILE}, 3",
                f"s_lshl_b32 {S_M0K}, {S_TMP}, {kshift}",
                f"s_add_i32 {S_M0K}, {S_M0K}, %[woffk]",
                f"s_lshl_b32 {S_TMP}, {S_TMP}, 14",
                f"s_add_i32 {S_M0V}, {S_TMP}, %[woffv]",
                f"s_cmp_eq_u32 {S_TILE}, %[ntm1]",
                f"s_cselect_b64 {S_MASK}, -1, 0"]

    def rag_flag(tile_reg):
        """S_FLAG = 1 when the softmax of this phase works on the last tile and that tile is ragged"""
        return [f"s_cmp_eq_u32 {tile_reg}, %[ntm1]",
                f"s_cselect_b32 {S_FLAG}, 1, 0",
                f"s_cmp_lt_u32 {S_RAG}, 64",
                f"s_cselect_b32 {S_FLAG}, {S_FLAG}, 0"]

    o = []
    # ---- pre, A0 (= "A(t+1)" with t = -1: it stages V^T(2) and moves the K address registers from slot 0 to slot 1)
    o += [f"s_mov_b32 {S_T}, -1"] + dma_setup()
    o += [f"s_mov_b32 {S_MKK}, {TILE_K}",               # K leaves slot 0: even slot -> xor with one slot size
          f"s_mov_b32 {S_RAG}, %[rag]",
          f"s_mov_b32 {S_FLAG}, 0"]                      # softmax(0,0): tile 0 is never the last (n >= 2)
    if FP8:
        # the first two phases are 8 MFMAs each — shorter than the fragment lookahead and with all eight buffers holding their own
        # four 32-byte K fragments — so each fetches its own first fragments and ends drained, and B(0)'s first fragments are
        # fetched here, in front of the loop label, exactly as A(t+1)'s tail fetches them for B(t+1)
        o += emit_phase(pre, None, "pre", own_prefetch=True, drain=True)
        o += emit_phase(a0, None, "a0", own_prefetch=True, drain=True)
        o += early_reads(bt)
    else:
        o += emit_phase(pre, a0, "pre", own_prefetch=True)
        o += emit_phase(a0, bt, "a0")
    o += [f"s_mov_b32 {S_T}, 0",
          ".Law16_loop_%=:"]
    o += dma_setup()
    # ring-slot masks of A(t+1): K leaves slot t + 1, V^T slot t (slot s -> s + 1: xor 1 << 14 out of an even slot, 3 << 14 out of an odd one)
    o += [f"s_and_b32 {S_TMP}, {S_T}, 1",
          f"s_lshl_b32 {S_TMP2}, {S_TMP}, 15",
          f"s_or_b32 {S_MKV}, {S_TMP2}, {TILE}",               # V^T leaves slot t: t even -> 1 << 14, odd -> 3 << 14
          f"s_xor_b32 {S_TMP}, {S_TMP}, 1",                    # K leaves slot t + 1
          f"s_lshl_b32 {S_TMP}, {S_TMP}, {(13 if FP8 else 14) + 1}",
          f"s_or_b32 {S_MKK}, {S_TMP}, {TILE_K}"]
    o += rag_flag(S_T)                                   # B(t): softmax(1, t)
    o += emit_phase(bt, at, "b")
    o += [f"s_cmp_eq_u32 {S_T}, %[ntm1]",
          "s_cbranch_scc1 .Law16_done_%=",
          f"s_add_i32 {S_TMP}, {S_T}, 1"]
    o += rag_flag(S_TMP)                                 # A(t+1): softmax(0, t+1)
    o += emit_phase(at, bt, "a")
    o += [f"s_add_i32 {S_T}, {S_T}, 1",
          "s_branch .Law16_loop_%=",
          ".Law16_done_%=:",
          # B's tail fetched fragments of an A phase that does not follow: let them land, and let B's last MFMAs leave the front of
          # the matrix pipe before post's own fragments refill their buffers (rule 3)
          "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15"]
    o += emit_phase(post, None, "post", own_prefetch=True, drain=True)
    return o


def main():
    lines = loop()
    stem = "attention_w16f8_loop" if FP8 else "attention_w16_loop"
    path = os.path.join(ROOT, "diffusion-rs_amd", "csrc", stem + ".inc")
    if X:
        os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
        path = os.path.join(ROOT, "build", f"{stem}_{X}.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_attention_w16.py — do not edit.  The whole KV stream of attention_w16_kernel as one asm\n")
        f.write("// statement (pre, A0, loop { B(t); A(t+1) }, post); register map and schedule: see the generator.\n")
        f.write(f"#define FMI_AW16{'F8' if FP8 else ''}_LOOP_ASM \\\n")
        body = ['  "' + ln + '\\n\\t"' for ln in lines if not ln.startswith(";")]
        f.write(" \\\n".join(body))
        f.write("\n")
    if os.environ.get("AW16_DUMP"):
        with open(os.environ["AW16_DUMP"], "w") as f:
            f.write("\n".join(lines) + "\n")
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    n_other = sum(1 for ln in lines if not ln.startswith(";") and not ln.startswith("v_mfma") and not ln.endswith(":"))
    print(f"{path}: {len(lines)} lines, {n_mfma} MFMAs, {n_other} other instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
