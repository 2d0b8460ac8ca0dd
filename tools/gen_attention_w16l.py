#!/usr/bin/env python
"""Generates diffusion-rs_amd/csrc/attention_w16l_loop.inc: the whole KV stream of attention_w16l_kernel (attention_w16l.h) as ONE
inline-asm statement with hand-assigned registers.

attention_w16l is round 4's joint-attention kernel: attention_w16's arithmetic (both products on v_mfma_f32_16x16x32_bf16, scale and the
running maximum folded into the score MFMA, exp2 in place, row sums from a ones-row MFMA, nothing crossing lanes on the common path —
tools/gen_attention_w16.py) on a LOCK-STEP schedule: the wave's four 16-query blocks q = 0..3 go through a KV tile TOGETHER, so a K / V^T
fragment (one ds_read_b128 per lane) feeds FOUR MFMAs instead of two — half the LDS fragment reads and counted waits per FLOP.  Round 3's
measurements (the stream is bound by what is issued BETWEEN the MFMAs; a ds_read_b128 costs ~14 clocks of issue with four waves
reading, a 16-clock MFMA hides ~8) made the LDS reads the largest removable item.

What lock-step costs: the two half-tile-staggered query blocks of attention_w16 gave every MFMA phase an independent softmax to hide; here a
tile's softmax sits between ITS OWN two products.  So the products of neighbouring tiles interleave —

    pre   X(0)                                                            X(t) = S^T(t) = K(t) Q^T   (64 MFMAs, 16 K fragments)
    P1    X(1)   | softmax(0) whole (first tile: forced rescale) | barrier                Y(t) = O^T += V^T(t) P^T(t), l += 1 P^T(t)  (72 MFMAs, 16 V^T fragments)
    loop t = 0 ..:
      Y(t)     | mask / max tree / decision (+ rare: rescale of S, M, NM) / first half of the exponentials of tile t + 1    | DMA K(t+4)
      if t == n-1: break
      X(t+2)   | rare: O^T, l *= alpha | second half of the exponentials and the packs to bf16 (P) of tile t + 1 | barrier | DMA V^T(t+3)

— and S^T is double-buffered (tile t lives in buffer t & 1: X(t+2) overwrites the buffer of the dead tile t while tile t+1 is still being
exponentiated), which makes the loop body two tiles long (register names are static).  P is single-buffered: Y(t) has issued its last read of
P(t) before X(t+2)'s packs write P(t+1).  The deferred rescale is split: the decision (any s' of the wave above the threshold) and everything
that touches S, M and NM run in Y(t)'s gaps, BEFORE X(t+2) folds the new maximum into the next scores; the multiplication of O^T and l by
alpha waits (flag + four saved alphas) until Y(t) — which is still accumulating tile t into them — has finished: it runs at the top of
X(t+2).  With THR = 0 (rescale on every tile) the result is bit-identical to attention_w16 / attention_w32; with the default threshold the
three agree to rounding (this kernel takes its decision over all 64 queries of the wave, attention_w16 per 32).

LDS rings (4 K tiles, 4 V^T tiles, as before) and the one barrier per tile: barrier(t) sits in the middle of X(t+2).  K(t+4) is staged in
Y(t) into the slot of K(t) (every wave passed barrier(t-1), i.e. finished X(t)); V^T(t+3) behind barrier(t) into the slot of V^T(t-1) (every
wave finished Y(t-1)).  At barrier(t) a wave's newest 8 pieces (V^T(t+2), K(t+4)) may be in flight — vmcnt(8) — and what the next tile of
reads needs (K(t+3), V^T(t+1)) is older.  Two tiles of lead for either operand, as in attention_w16.

Register map (pinned by the operand constraints in attention_w16l.h):
  a[0:127]    O^T   O[q][dt]   -> a[((8b+dt)*2+c)*4 ..], q = 2b + c      a[128:191]  Q fragments QF[q][s] -> a[128+(q*4+s)*4 ..]
  a[192:207]  OL[q]: the ones-row product (register 0 of lanes 0..15 = the row sum of query n of block q)
  a[208:239]  fragment buffers FR[0..7] (ds_read_b128 destinations; the MFMA takes its A operand from the accumulator half)
  v[0:63]     S^T buffer 0: S[a][q] -> v[(4a+q)*4 ..]     v[64:127]  S^T buffer 1     v[128:159]  P[kk][q] -> v[128+(4kk+q)*4 ..]
  v[160:163]  KAD[s]  v[164:165] VAD[kk]  v[166:169] k_voff  v[170:173] v_voff  v[174:177] k_voff clamped (ragged last tile)
  v178        lane key offset 16 (g >> 1) + 4 (g & 1)   v179 DMA offset temporary
  v[180:195]  NM[q] (-m of the lane's query, 4 copies)   v[196:199] M[q]   v[200:203] the ones fragment   v[204:207] ALPHA[q] (pending O^T rescale)
  v[208:255]  temporaries (clobbers)   s[80:97] loop state (clobbers)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 16384
VT_RING = 4 * TILE
NBUF = 8
NBUFK = 4         # fp8 mode: the 32-byte K fragments have their own pool of four 8-register buffers, a[160:191]
MODE = os.environ.get("AW16L_MODE", "bf16")  # "bf16" | "fp8qk": Q and K as OCP e4m3 (the model's fp8 mode), P and V^T stay bf16 -> attention_w16lf8_loop.inc
#                                              | "fp8pv" (round 5): P and V^T as e4m3 too, second product on v_mfma_f32_32x32x64_f8f6f4 -> attention_w16lf8pv_loop.inc
PV8 = MODE == "fp8pv"
FP8 = MODE in ("fp8qk", "fp8pv")
TILE_K = 8192 if FP8 else TILE  # bytes of a K tile (64 keys x 128 d) in HBM and in LDS
LOOKAHEAD = int(os.environ.get("AW16L_LOOKAHEAD", "16"))  # a fragment is read this many MFMA slots ahead of its first use (4 fragments in flight)
X = os.environ.get("AW16L_X", "")  # timing experiments only (wrong results): novalu | noexp | nodma | nobarrier | nomfma | nowait
XS = set(x for x in X.split("/") if not x.startswith("drop:"))  # (several experiments at once: separated by "/")
XDROP = next((x[5:] for x in X.split("/") if x.startswith("drop:")), None)
TAG = os.environ.get("AW16L_TAG", X)  # a tag (or an experiment) writes build/attention_w16l_loop_<tag>.inc instead of the committed file
EY = int(os.environ.get("AW16L_EY", "8"))  # how many of a tile's 16 exponential units (4 scores each, key-block major) run in the Y phase; the rest in X
F8_EX = int(os.environ.get("AW16L_F8_EX", "6" if PV8 else "4"))  # fp8 mode: how many of the 16 exponential units (from the last) run in the X phase instead of Y
TREE_END = int(os.environ.get("AW16L_TREE_END", "19"))  # last gap of the max tree in a Y phase (the decision sits two gaps behind it)

PMAX, TA, TB, T0, T1, AL, DL = (f"v{n}" for n in range(208, 215))
XT = [f"v{n}" for n in range(215, 221)]
PM0, PM1 = "v221", "v222"
PMX = [f"v{n}" for n in range(223, 256)]  # partial maxima of the tree (33 registers)
LKEY, DMAT = "v178", "v179"
S_KP, S_VP, S_MASKK = "s[80:81]", "s[82:83]", "s[84:85]"
S_T, S_TILE, S_M0K, S_M0V, S_TMP, S_RESC, S_RAG, S_TMP2, S_FLAG = "s86", "s87", "s88", "s89", "s90", "s91", "s93", "s94", "s95"
S_GODD = "s[96:97]"  # fp8pv: lanes whose 16-lane group is odd (the 32 x 32 accumulators hold query block 2 pair + (group & 1))
NEG_BIG = "0xf149f2ca"  # -1e30f


def O(q, dt):
    b, c = q >> 1, q & 1
    lo = ((8 * b + dt) * 2 + c) * 4
    return f"a[{lo}:{lo + 3}]"


def Or(q, k):  # k = 0..31: register i of d block dt, k = 4 dt + i
    b, c = q >> 1, q & 1
    return f"a{((8 * b + (k >> 2)) * 2 + c) * 4 + (k & 3)}"


def QF(q, s):
    lo = 128 + (q * 4 + s) * 4
    return f"a[{lo}:{lo + 3}]"


def OL(q):
    lo = 192 + q * 4
    return f"a[{lo}:{lo + 3}]"


def S(buf, a, q):
    lo = 64 * buf + (4 * a + q) * 4
    return f"v[{lo}:{lo + 3}]"


def Sr(buf, a, q, i):
    return f"v{64 * buf + (4 * a + q) * 4 + i}"


def P(kk, q):
    lo = 128 + (4 * kk + q) * 4
    return f"v[{lo}:{lo + 3}]"


def Pr(kk, q, d):
    return f"v{128 + (4 * kk + q) * 4 + d}"


# ---- fp8pv register map (differences from the header's): O^T as 32 x 32 accumulators O8[pair][db] -> a[(4 pair + db) * 16 ..] (pair = query blocks
# 2 pair, 2 pair + 1; db = 32 head dimensions), the ones-row products OL8[pair] -> a[192:207], a[240:255]; P as the B operand of the pair,
# v[128 + 8 pair ..] (dword 2 a + c: before the lane swap the four e4m3 scores of key block a, query block 2 pair + c); the ones fragment in
# v[144:151]; four 8-register V^T fragment buffers in a[208:239]
def O8(pair, db):
    lo = (pair * 4 + db) * 16
    return f"a[{lo}:{lo + 15}]"


def O8r(pair, k):  # k = 0..63: register k & 15 of d block k >> 4
    return f"a{(pair * 4 + (k >> 4)) * 16 + (k & 15)}"


def OL8(pair):
    lo = 240 if pair else 192
    return f"a[{lo}:{lo + 15}]"


def P8(pair):
    lo = 128 + 8 * pair
    return f"v[{lo}:{lo + 7}]"


def P8r(pair, a, c):
    return f"v{128 + 8 * pair + 2 * a + c}"


ONES8 = "v[144:151]"


def FR(n, pool="V", half=None):
    """fragment buffer n of a pool: "V" = a[208:239] (8 x 4 registers: every bf16 fragment); "K8" = a[160:191] (4 x 8 registers: the fp8
    mode's 32-byte K fragments; half = 0 / 1 selects the 16 bytes one ds_read_b128 fills)"""
    if pool == "K8":
        lo = 160 + 8 * (n % NBUFK)
        return f"a[{lo}:{lo + 7}]" if half is None else f"a[{lo + 4 * half}:{lo + 4 * half + 3}]"
    if pool == "V8":  # fp8pv: the 32-byte V^T fragments (32 head dimensions x 64 keys of e4m3)
        lo = 208 + 8 * (n % NBUFK)
        return f"a[{lo}:{lo + 7}]" if half is None else f"a[{lo + 4 * half}:{lo + 4 * half + 3}]"
    lo = 208 + 4 * (n % NBUF)
    return f"a[{lo}:{lo + 3}]"


def QF8(q):
    lo = 128 + 8 * q
    return f"a[{lo}:{lo + 7}]"


SCA, SCB = "v162", "v163"  # fp8 mode (KAD[2], KAD[3] are unused there): E8M0 block scales of the score product, 2^-n on the K side, 1 on the Q side


def KAD(s):
    return f"v{160 + s}"


def VAD(kk):
    return f"v{164 + kk}"


def NM(q):
    lo = 180 + 4 * q
    return f"v[{lo}:{lo + 3}]"


def NMr(q, i):
    return f"v{180 + 4 * q + i}"


def M(q):
    return f"v{196 + q}"


def ALPHA(q):
    return f"v{204 + q}"


ONESF = "v[200:203]"


# ----------------------------------------------------------------------------------------------------------------------------
class Phase:
    """One phase: its MFMA slots (text with a {fr} hole + the index of the slot's fragment) and its fragments (address register,
    immediate) in order of first use.  kind "X": S^T(buf) = K Q^T; kind "Y": O^T += V^T P^T and the ones row."""

    def __init__(self, name, kind, buf=None):
        self.name, self.kind, self.buf = name, kind, buf
        self.mfma, self.frags = [], []
        if kind == "X" and FP8:
            # one v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 score tile: the whole head dimension in one instruction, whose E8M0 block
            # scales carry the score factor 2^-n; fragment a = the key block's 32 bytes per lane (two reads: KAD[0], KAD[1] = KAD[0] ^ 16)
            for a in range(4):
                # the key whose score sits in row m of block a: 32 (a >> 1) + 8 (a & 1) + (m & 7) + 16 (m >> 3); fp8pv: 8 a + (m & 7) + 32 (m >> 3), so that a
                # lane's 32 operand bytes of the second product are 32 consecutive keys (plan_x_pv8)
                imm = a * 1024 if PV8 else (a >> 1) * 4096 + (a & 1) * 1024
                self.frags.append([(KAD(0), imm), (KAD(1), imm)])
                for q in range(4):
                    self.mfma.append((f"v_mfma_scale_f32_16x16x128_f8f6f4 {S(buf, a, q)}, {{fr}}, {QF8(q)}, {NM(q)}, {SCA}, {SCB}", a))
        elif kind == "X":
            for s in range(4):          # d-step (32 of the 128 head dimensions)
                for a in range(4):      # key block (16 keys)
                    self.frags.append([(KAD(s), (a >> 1) * 8192 + (a & 1) * 2048)])
                    for q in range(4):
                        acc = NM(q) if s == 0 else S(buf, a, q)  # first d-step: start from -m of the lane's query (the fold)
                        self.mfma.append((f"v_mfma_f32_16x16x32_bf16 {S(buf, a, q)}, {{fr}}, {QF(q, s)}, {acc}", 4 * s + a))
        elif PV8:
            # O^T[32 d x 32 queries] += V^T[32 d x 64 keys] P^T[64 keys x 32 queries]: one v_mfma_f32_32x32x64_f8f6f4 per (d block, query pair), the
            # whole tile's keys in one instruction; fragment db = 32 bytes per lane (row 32 db + (lane & 31), keys 32 (lane >> 5) ..): two reads
            for db in range(4):
                self.frags.append([(VAD(0), db * 2048), (VAD(1), db * 2048)])
                for pair in range(2):
                    self.mfma.append((f"v_mfma_f32_32x32x64_f8f6f4 {O8(pair, db)}, {{fr}}, {P8(pair)}, {O8(pair, db)}", db))
            for pair in range(2):       # V^T extended by a row of ones: the row sums of the e4m3-rounded P
                self.mfma.append((f"v_mfma_f32_32x32x64_f8f6f4 {OL8(pair)}, {ONES8}, {P8(pair)}, {OL8(pair)}", None))
        else:
            for kk in range(2):         # k-step (32 keys)
                for dt in range(8):     # d block (16 head dimensions)
                    self.frags.append([(VAD(kk), dt * 2048)])
                    for q in range(4):
                        self.mfma.append((f"v_mfma_f32_16x16x32_bf16 {O(q, dt)}, {{fr}}, {P(kk, q)}, {O(q, dt)}", 8 * kk + dt))
                for q in range(4):      # V^T extended by a row of ones: the row sums of the bf16-rounded P
                    self.mfma.append((f"v_mfma_f32_16x16x32_bf16 {OL(q)}, {ONESF}, {P(kk, q)}, {OL(q)}", None))
        self.n = len(self.mfma)
        nf = len(self.frags)
        self.fu = [min(i for i, (_, ff) in enumerate(self.mfma) if ff == f) for f in range(nf)]
        self.lu = [max(i for i, (_, ff) in enumerate(self.mfma) if ff == f) for f in range(nf)]
        self.pool = "K8" if (kind == "X" and FP8) else "V8" if PV8 else "V"
        assert self.fu == sorted(self.fu) and nf % (NBUFK if self.pool != "V" else NBUF) == 0
        # where a fragment is read, in MFMA slots relative to the phase's slot 0 (negative: in the previous phase, counted from its end).
        # fp8pv: the slots are not of one length (32 clocks in X, 64 in Y) — a table with >= ~200 clocks of lead; otherwise LOOKAHEAD slots
        self.rd = ([-4, -2, 2, 6] if kind == "X" else [-7, -3, 0, 2]) if PV8 else [self.fu[f] - LOOKAHEAD for f in range(nf)]
        self.wg = 2 if (PV8 and kind == "Y") else 4  # counted waits every wg slots
        # set by build(): valu (list of instruction lists per gap), dma ("K" | "V" | None), barrier, advance ((regs, xor mask) applied behind the last own read)
        self.valu, self.dma, self.barrier, self.advance = [[] for _ in range(self.n)], None, False, None


# ----------------------------------------------------------------------------------------------------------------------------
# softmax of the tile in S^T buffer `buf`, as instruction streams
OOL = int(os.environ.get("AW16L_OOL", "0"))  # 1 = rare paths out of line: the common path falls through a NOT-taken branch (round 5 experiment: a taken branch costs a
#                                              single-wave stream an instruction-buffer refill; three per tile are taken) — measured 0.5-2 %, inside the box noise: not adopted
OOL_BLOCKS = []


def rare(kind, uid, cond_taken_skip, cond_taken_rare, body, pre=()):
    """`body` runs only when the rare condition holds.  In line (OOL = 0): pre + [branch-if-common over it].  Out of line: pre + [branch-if-rare to
    the block behind the stream, which jumps back]."""
    if not OOL:
        lab = f".Law16l_{kind}_{uid}_%="
        return list(pre) + [f"{cond_taken_skip} {lab}"] + body + [f"{lab}:"]
    lab, back = f".Law16l_r{kind}_{uid}_%=", f".Law16l_b{kind}_{uid}_%="
    OOL_BLOCKS.append([f"{lab}:"] + body + [f"s_branch {back}"])
    return list(pre) + [f"{cond_taken_rare} {lab}", f"{back}:"]


def mask_block(buf):
    """ragged last tile: scores of keys >= Lk become -1e30 (p = 0).  The key of (a, i) in lane (g, n) is
    32 (a >> 1) + 8 (a & 1) + i + LKEY; S_RAG = keys in the last tile (1..64)."""
    out = [f"v_mov_b32 {T1}, {NEG_BIG}"]
    for a in range(4):
        for i in range(4):
            koff = 8 * a + i if PV8 else 32 * (a >> 1) + 8 * (a & 1) + i   # (fp8pv: LKEY = 32 (g >> 1) + 4 (g & 1))
            out.append(f"s_sub_i32 {S_TMP2}, {S_RAG}, {koff}")          # key valid <=> LKEY + koff < rag
            out.append(f"v_cmp_le_i32 vcc, {S_TMP2}, {LKEY}")
            for q in range(4):
                out.append(f"v_cndmask_b32 {Sr(buf, a, q, i)}, {Sr(buf, a, q, i)}, {T1}, vcc")
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


def max_tree(buf):
    """maximum over the lane's 64 scores (all four query blocks): the common path only needs "does any score of the wave exceed the
    threshold".  A tree of independent v_max3 (a serial chain on one register costs ~13 clocks per link with one wave per SIMD),
    key blocks in order a = 0..3 (the order the score product finishes them in).  Result in PMAX."""
    level = [Sr(buf, a, q, i) for a in range(4) for q in range(4) for i in range(4)]
    out, nxt = [], 0
    while len(level) > 1:
        new = []
        k = 0
        while k < len(level):
            rest = len(level) - k
            dst = PMAX if (len(level) <= 3) else PMX[nxt % len(PMX)]
            if rest >= 3:
                out.append(f"v_max3_f32 {dst}, {level[k]}, {level[k + 1]}, {level[k + 2]}")
                k += 3
            elif rest == 2:
                out.append(f"v_max_f32 {dst}, {level[k]}, {level[k + 1]}")
                k += 2
            else:
                new.append(level[k])
                k += 1
                continue
            new.append(dst)
            nxt += 1
        level = new
    assert nxt <= len(PMX) + 1, nxt  # no partial is overwritten while live (22 + 8 + 3 + 1 = 34 results, the last in PMAX)
    assert level == [PMAX]
    return out


def rescale_s(buf, set_flag):
    """Taken when some score of the wave exceeds the threshold, and always on the first tile (M = -1e30, fold = 0): per query block q
    the new maximum m' = max(M, max s' - NM), delta = m' + NM (how far this tile's fold was off), alpha = exp2(min(-delta, 0));
    then s' -= delta, M = m', NM = -m' — and alpha is SAVED: O^T and l still receive tile t's second product; they are multiplied at
    the top of the next X phase (rescale_o), flagged by S_RESC."""
    out = []
    for b in range(2):
        for c, pm in ((0, PM0), (1, PM1)):
            out += max_chain(pm, [Sr(buf, a, 2 * b + c, i) for a in range(4) for i in range(4)])
        # reduce over the four lane groups (lanes n, n + 16, n + 32, n + 48 hold the same query): after the first swap the lower half
        # of the wave works on query block 2b and the upper half on 2b + 1; the last swap hands every lane both results
        out += ["s_nop 1",
                f"v_permlane32_swap_b32 {PM0}, {PM1}",
                "s_nop 1",
                f"v_max_f32 {TA}, {PM0}, {PM1}",
                f"v_mov_b32 {TB}, {TA}",
                "s_nop 1",
                f"v_permlane16_swap_b32 {TA}, {TB}",
                "s_nop 1",
                f"v_max_f32 {TA}, {TA}, {TB}",
                f"v_mov_b32 {TB}, {TA}",
                "s_nop 1",
                f"v_permlane32_swap_b32 {TA}, {TB}",         # TA = block 2b's maximum in every lane, TB = block 2b + 1's
                "s_nop 1"]
        for c, ps in ((0, TA), (1, TB)):
            q = 2 * b + c
            out += [f"v_sub_f32 {T0}, {ps}, {NMr(q, 0)}",            # the maximum in the unshifted domain: ps' - NM
                    f"v_max_f32 {T0}, {M(q)}, {T0}",                  # m'
                    f"v_add_f32 {DL}, {T0}, {NMr(q, 0)}",             # delta = m' - (the m this tile's fold used)
                    f"v_max_f32 {AL}, {DL}, 0",
                    f"v_sub_f32 {AL}, 0, {AL}",
                    f"v_exp_f32 {ALPHA(q)}, {AL}",                    # alpha = exp2(-max(delta, 0)), read much later (rescale_o)
                    f"v_mov_b32 {M(q)}, {T0}"]
            out += [f"v_sub_f32 {NMr(q, i)}, 0, {T0}" for i in range(4)]
            out += [f"v_sub_f32 {Sr(buf, a, q, i)}, {Sr(buf, a, q, i)}, {DL}" for a in range(4) for i in range(4)]
    if set_flag:
        out.append(f"s_mov_b32 {S_RESC}, 1")
    return out


def rescale_o():
    """O^T[q] *= alpha[q], l[q] *= alpha[q] for the four query blocks (rare).  Placed >= 4 MFMA slots behind the last MFMA that wrote
    them (the previous Y phase); the X phase around it does not touch these accumulators."""
    out = []
    for q in range(4):
        lo = 192 + q * 4
        out += [f"v_accvgpr_read_b32 {T1}, a{lo}", "s_nop 0", f"v_mul_f32 {T1}, {T1}, {ALPHA(q)}", "s_nop 0", f"v_accvgpr_write_b32 a{lo}, {T1}"]
        n = len(XT)
        out.append(f"v_accvgpr_read_b32 {XT[0]}, {Or(q, 0)}")
        for r in range(32):  # software pipeline over the 32 accumulator registers of block q
            if r + 1 < 32:
                out.append(f"v_accvgpr_read_b32 {XT[(r + 1) % n]}, {Or(q, r + 1)}")
            out.append(f"v_mul_f32 {XT[r % n]}, {XT[r % n]}, {ALPHA(q)}")
            out.append(f"v_accvgpr_write_b32 {Or(q, r)}, {XT[r % n]}")
    out.append(f"s_mov_b32 {S_RESC}, 0")
    return out


def exps(buf, a, q):
    return [f"v_exp_f32 {Sr(buf, a, q, i)}, {Sr(buf, a, q, i)}" for i in range(4)]


def packs(buf, a, q):
    kk, h = a >> 1, a & 1  # P[kk][q] dwords 2h, 2h + 1
    return [f"v_cvt_pk_bf16_f32 {Pr(kk, q, 2 * h + d)}, {Sr(buf, a, q, 2 * d)}, {Sr(buf, a, q, 2 * d + 1)}" for d in range(2)]


UNITS = [(a, q) for a in range(4) for q in range(4)]  # a tile's 16 units of 4 scores, in the order the score product finishes them


def exp_pack_stream(buf, units, SKEW=2):
    """p = exp2(s') in place, then per unit of 4 scores (a, q) two packs to bf16.  The packs of unit u follow the exponentials of unit
    u + SKEW: a v_exp_f32 result must not be read within the next few VALU instructions (gfx950: stale in half of the lanes, DESIGN
    4.4 rule 2)."""
    out = []
    for u in range(len(units) + SKEW):
        if u < len(units):
            out += exps(buf, *units[u])
        if u >= SKEW:
            out += packs(buf, *units[u - SKEW])
    return out


def spread(plan, stream, first, last):
    """stream instructions over gaps first..last (inclusive), as evenly as integer division allows, in order"""
    n = last - first + 1
    for k, ins in enumerate(stream):
        plan[first + k * n // len(stream)].append(ins)


def plan_y(ph, buf, uid):
    """Y(t)'s gaps: mask (ragged last tile), max tree, decision (+ rare rescale of S / M / NM), first half of the exponentials — of tile
    t + 1 in S^T buffer `buf`, which the X phase in front of this one completed (its last MFMAs are >= 4 slots behind gap 4)."""
    n, plan = ph.n, ph.valu
    plan[4] += rare("nomask", uid, "s_cbranch_scc1", "s_cbranch_scc0", mask_block(buf), [f"s_cmp_eq_u32 {S_FLAG}, 0"])
    if not ("halftree" in XS and uid == "yo"):  # (timing experiment: the maximum taken on every other tile only)
        spread(plan, max_tree(buf), 5, TREE_END)
        plan[TREE_END + 2] += rare("skip", uid, "s_cbranch_vccz", "s_cbranch_vccnz", rescale_s(buf, True), [f"v_cmp_lt_f32 vcc, %[thr], {PMAX}"])
    spread(plan, [i for (a, q) in UNITS[:EY] for i in exps(buf, a, q)], TREE_END + 3, n - 1)


def plan_x(ph, buf, uid):
    """X(t+2)'s gaps: the pending O^T / l rescale (rare), then for tile t + 1 (buffer `buf`): packs of key blocks 0, 1 (exponentiated in
    the Y phase before), exponentials + packs of key blocks 2, 3.  P[kk = 0] is complete by mid-phase, P[kk = 1] by the last gap (Y(t+1)
    reads it from its slot 36 on)."""
    n, plan = ph.n, ph.valu
    plan[4] += rare("noresc", uid, "s_cbranch_scc1", "s_cbranch_scc0", rescale_o(), [f"s_cmp_eq_u32 {S_RESC}, 0"])
    early = [i for (a, q) in UNITS[:EY] for i in packs(buf, a, q)]
    late = exp_pack_stream(buf, UNITS[EY:])
    # the 16 early packs alternate with the first 16 instructions of the late stream: P[kk = 0] is complete after a third of the
    # phase, and the exponentials start at once; the last pack sits a few gaps in front of the next phase
    stream = []
    for k, ins in enumerate(late):
        stream.append(ins)
        if k < len(early):
            stream.append(early[k])
    spread(plan, stream, 5, n - 3)


def plan_y_f8(ph, buf, uid):
    """fp8 mode: an X phase is 16 double-length MFMAs — a quarter of the gaps of the bf16 one — so the softmax lives in the Y phases.
    Y(t): [packs of key blocks 2, 3 of tile t (the OTHER buffer) -> P[kk = 1], which this phase reads from slot 36 on], then tile t + 1
    (buffer `buf`): mask, max tree, decision (+ rare rescale of S / M / NM), ALL 64 exponentials.  X(t+2) keeps the packs of key blocks 0, 1."""
    n, plan = ph.n, ph.valu
    spread(plan, [i for (a, q) in UNITS[8:] for i in packs(buf ^ 1, a, q)], 0, 28)
    plan[4] += rare("nomask", uid, "s_cbranch_scc1", "s_cbranch_scc0", mask_block(buf), [f"s_cmp_eq_u32 {S_FLAG}, 0"])
    spread(plan, max_tree(buf), 5, TREE_END)
    plan[TREE_END + 2] += rare("skip", uid, "s_cbranch_vccz", "s_cbranch_vccnz", rescale_s(buf, True), [f"v_cmp_lt_f32 vcc, %[thr], {PMAX}"])
    spread(plan, [i for (a, q) in UNITS[:16 - F8_EX] for i in exps(buf, a, q)], TREE_END + 3, n - 1)


def plan_x_f8(ph, buf, uid):
    """fp8 mode, X(t+2)'s 16 gaps: the pending O^T / l rescale (rare; gap 2 is 64+ clocks behind Y(t)'s last MFMA), then the packs of key
    blocks 0, 1 of tile t + 1 (buffer `buf`, exponentiated in Y(t)) -> P[kk = 0], complete three gaps in front of Y(t+1)."""
    n, plan = ph.n, ph.valu
    plan[2] += rare("noresc", uid, "s_cbranch_scc1", "s_cbranch_scc0", rescale_o(), [f"s_cmp_eq_u32 {S_RESC}, 0"])
    late_exps = [i for (a, q) in UNITS[16 - F8_EX:] for i in exps(buf, a, q)]  # (their packs are Y(t+1)'s first job: far enough behind)
    pk = [i for (a, q) in UNITS[:8] for i in packs(buf, a, q)]
    stream = []
    for k in range(max(len(late_exps), len(pk))):
        stream += late_exps[k:k + 1] + pk[k:k + 1]
    spread(plan, stream, 3, n - 3)


def plan_p1_f8(ph, uid):
    """fp8 mode, P1 = X(1) | all exponentials of tile 0 (buffer 0) and the packs of its key blocks 0, 1; the packs of key blocks 2, 3
    are Y(0)'s first job, as in the steady state (a one-time crowd: 80 instructions in 14 gaps)."""
    n, plan = ph.n, ph.valu
    spread(plan, [i for (a, q) in UNITS for i in exps(0, a, q)] + [i for (a, q) in UNITS[:8] for i in packs(0, a, q)], 1, n - 3)


# ---- fp8pv: P as e4m3, the B operand of v_mfma_f32_32x32x64_f8f6f4 --------------------------------------------------------------
# The score product leaves lane (g, n) with, per key block a and query block q, the four scores of key rows 4 g + i of query n.  The second
# product wants lane l = (g, n) to hold 32 consecutive key bytes (those of half g >> 1) of ONE query of the pair: query block 2 pair + (g & 1).
# So: four scores -> one dword of e4m3 (two v_cvt_pk_fp8_f32), the dwords of the pair's two query blocks in neighbouring registers, and one
# v_permlane16_swap_b32 per (pair, a) hands the odd groups' block-0 dword to the even groups and the even groups' block-1 dword to the odd
# ones.  Afterwards register 2 a (+1) of lane (g, n) = key rows 8 (g >> 1) + 0..3 (4..7) of block a, i.e. with the key of (a, row m) chosen as
# 8 a + (m & 7) + 32 (m >> 3) (Phase: the K fragment immediates) byte j of the lane's operand is key 32 (g >> 1) + j: V^T is read in plain order.
def cvts8(buf, a, q):
    d = P8r(q >> 1, a, q & 1)
    return [f"v_cvt_pk_fp8_f32 {d}, {Sr(buf, a, q, 0)}, {Sr(buf, a, q, 1)}",
            f"v_cvt_pk_fp8_f32 {d}, {Sr(buf, a, q, 2)}, {Sr(buf, a, q, 3)} op_sel:[0,0,1]"]


def swap8(pair, a):
    # (one stream element: the interleaving of two streams must not pull the wait states away from the swap; main() splits at " ;; ")
    return [f"s_nop 1 ;; v_permlane16_swap_b32 {P8r(pair, a, 0)}, {P8r(pair, a, 1)} ;; s_nop 1"]


CVT_GRP = int(os.environ.get("AW16L_CVT_GRP", "4"))  # the two conversions into one dword are dependent (the second keeps the first's half): units go
#                                                       in groups of CVT_GRP — all first halves, then all second halves (measured: ~10 clocks per back-to-back pair)


def cvt_swap_stream(buf, units, done=()):
    """cvts of `units` in order; the swap of (pair, a) behind the last of its four cvts (units in `done` were converted earlier)"""
    out, have = [], set(done)
    for g0 in range(0, len(units), CVT_GRP):
        grp = units[g0:g0 + CVT_GRP]
        pairs = [cvts8(buf, a, q) for (a, q) in grp]
        out += [c[0] for c in pairs] + [c[1] for c in pairs]
        for (a, q) in grp:
            have.add((a, q))
            if (a, q ^ 1) in have:
                out += swap8(q >> 1, a)
    return out


def rescale_o_pv8():
    """fp8pv form of rescale_o: the 32 x 32 accumulators of a pair hold query block 2 pair in the even 16-lane groups and 2 pair + 1 in the
    odd ones — alpha is selected per lane (S_GODD)."""
    out = []
    n = len(XT)
    for pair in range(2):
        lo = 240 if pair else 192
        out += [f"v_cndmask_b32 {AL}, {ALPHA(2 * pair)}, {ALPHA(2 * pair + 1)}, {S_GODD}",
                f"v_accvgpr_read_b32 {T1}, a{lo}", "s_nop 0", f"v_mul_f32 {T1}, {T1}, {AL}", "s_nop 0", f"v_accvgpr_write_b32 a{lo}, {T1}"]
        out.append(f"v_accvgpr_read_b32 {XT[0]}, {O8r(pair, 0)}")
        for r in range(64):
            if r + 1 < 64:
                out.append(f"v_accvgpr_read_b32 {XT[(r + 1) % n]}, {O8r(pair, r + 1)}")
            out.append(f"v_mul_f32 {XT[r % n]}, {XT[r % n]}, {AL}")
            out.append(f"v_accvgpr_write_b32 {O8r(pair, r)}, {XT[r % n]}")
    out.append(f"s_mov_b32 {S_RESC}, 0")
    return out


PV8_TREE_END = int(os.environ.get("AW16L_PV8_TREE_END", "1"))


def plan_y_pv8(ph, buf, uid):
    """fp8pv, Y(t)'s 10 gaps (64-clock MFMAs), tile t + 1 in S^T buffer `buf`: mask (ragged last tile), max tree, decision (+ rare rescale of S / M /
    NM), then the exponentials of the first 16 - F8_EX units.  The VALU work of a tile (64 quarter-rate exponentials ~ 1000 clocks) is about as long
    as its MFMAs (512 + 640 clocks): it is spread in proportion to the slot lengths (plan_x_pv8 takes the rest)."""
    n, plan = ph.n, ph.valu
    # (rare, and in front of everything: the score product's last MFMAs may still be in the pipe — wait them out inside the branch)
    plan[0] += rare("nomask", uid, "s_cbranch_scc1", "s_cbranch_scc0", ["s_nop 15", "s_nop 15", "s_nop 15"] + mask_block(buf), [f"s_cmp_eq_u32 {S_FLAG}, 0"])
    spread(plan, max_tree(buf), 0, PV8_TREE_END)   # (key blocks in order: block 3's scores, the last to finish, are read at the end of the first level)
    plan[PV8_TREE_END + 1] += rare("skip", uid, "s_cbranch_vccz", "s_cbranch_vccnz", rescale_s(buf, True), [f"v_cmp_lt_f32 vcc, %[thr], {PMAX}"])
    spread(plan, [i for (a, q) in UNITS[:16 - F8_EX] for i in exps(buf, a, q)], PV8_TREE_END + 2, n - 1)


def plan_x_pv8(ph, buf, uid):
    """fp8pv, X(t+2)'s 16 gaps: the pending O^T / l rescale (rare), the last F8_EX units of exponentials of tile t + 1, every conversion to e4m3
    and the eight lane swaps -> P(t+1), complete two gaps in front of Y(t+1) (whose every MFMA reads it)."""
    n, plan = ph.n, ph.valu
    plan[2] += rare("noresc", uid, "s_cbranch_scc1", "s_cbranch_scc0", rescale_o_pv8(), [f"s_cmp_eq_u32 {S_RESC}, 0"])
    ny = 16 - F8_EX
    late_exps = [i for (a, q) in UNITS[ny:] for i in exps(buf, a, q)]
    early = cvt_swap_stream(buf, UNITS[:ny])
    stream = []
    for k in range(max(len(late_exps), len(early))):
        stream += late_exps[k:k + 1] + early[k:k + 1]
    # (the late units' conversions follow all of the late exponentials: a v_exp_f32 result is not read by the next few VALU instructions)
    stream += cvt_swap_stream(buf, UNITS[ny:], UNITS[:ny])
    spread(plan, stream, 3, n - 3)


def plan_p1_pv8(ph, uid):
    """fp8pv, P1 = X(1) | tile 0 (buffer 0): all exponentials, conversions and swaps (a one-time crowd)"""
    n, plan = ph.n, ph.valu
    ex = [i for (a, q) in UNITS for i in exps(0, a, q)]
    spread(plan, ex + ["s_nop 7"] + cvt_swap_stream(0, UNITS), 1, n - 3)



def plan_p1(ph, uid):
    """P1 = X(1) | all exponentials and packs of tile 0 (buffer 0).  Tile 0's maxima (the rescale block, unconditional on a first tile:
    M = -1e30, NM = 0) were taken BETWEEN pre and P1 — X(1) folds -m into its scores, so NM must be final before its first MFMA;
    everywhere else in the stream the rescale of tile t + 1 runs inside Y(t), where no score product is in flight."""
    n, plan = ph.n, ph.valu
    spread(plan, exp_pack_stream(0, UNITS), 1, n - 3)


# ----------------------------------------------------------------------------------------------------------------------------
def emit_phase(ph, nxt, own_prefetch=False, drain=False):
    """asm lines of one phase.  `nxt` = the phase whose first fragments are fetched behind this phase's last MFMAs (None: none).
    Fragment f of a phase is read in the gap behind MFMA slot fu[f] - LOOKAHEAD (a negative slot: in the previous phase's tail, or in
    front of the phase when own_prefetch)."""
    o = [f"; ==== phase {ph.name}"]
    n, nf = ph.n, len(ph.frags)
    reads = [[] for _ in range(n)]
    early = []
    def dest(p_, f, k):
        return FR(f, p_.pool, k if p_.pool != "V" else None)

    for f, fr in enumerate(ph.frags):
        i = ph.rd[f]
        for k, (reg, imm) in enumerate(fr):
            (reads[i] if i >= 0 else early).append(((0, f), dest(ph, f, k), reg, imm))
    own_last_read = max([g for g in range(n) if reads[g]], default=-1)
    if nxt is not None:
        for f, fr in enumerate(nxt.frags):
            i = n + nxt.rd[f]
            if i < n:
                assert i > own_last_read, (ph.name, "next phase's reads must follow the own ones")
                for k, (reg, imm) in enumerate(fr):
                    reads[i].append(((1, f), dest(nxt, f, k), reg, imm))
    if own_prefetch:
        for (_, b_, reg, imm) in early:
            o.append(f"ds_read_b128 {b_}, {reg} offset:{imm}")
    order = [key for (key, _, _, _) in early]
    issued_before_slot = [len(order)]
    for g in range(n):
        order += [key for (key, _, _, _) in reads[g]]
        issued_before_slot.append(len(order))
    last = {}
    for k, key in enumerate(order):
        last[key] = k
    # ring-slot advance of the address registers: each right behind the last own read that uses it
    adv_at = [[] for _ in range(n)]
    if ph.advance:
        regs, mask = ph.advance
        for reg in regs:
            own = [g for g in range(n) for (key, _, r_, _) in reads[g] if key[0] == 0 and r_ == reg]
            g_last = max(own, default=0)
            nxt_use = [g for g in range(n) for (key, _, r_, _) in reads[g] if key[0] == 1 and r_ == reg]
            assert all(g > g_last for g in nxt_use), (ph.name, reg, g_last, nxt_use)
            adv_at[g_last].append(f"v_xor_b32 {reg}, 0x{mask:x}, {reg}")
    for i in range(n):
        text, f = ph.mfma[i]
        pre, post = [], []
        dma = None
        if ph.dma == "K" and i % (n // 4) == n // 8 - 1 and i // (n // 4) < (2 if FP8 else 4) and "nodma" not in XS:  # 4 pieces (fp8: 2), one per quarter
            piece = i // (n // 4)
            pre.append(f"s_add_i32 m0, {S_M0K}, {piece * 1024}")
            pre.append(f"v_cndmask_b32 {DMAT}, v{166 + piece}, v{174 + piece}, {S_MASKK}")
            dma = f"global_load_lds_dwordx4 {DMAT}, {S_KP}"
        if ph.dma == "V" and i >= n // 2 and (i - n // 2) % (n // 8) == n // 8 - 1 and "nodma" not in XS and (
                not PV8 or (i - n // 2) // (n // 8) < 2):   # 4 pieces in the second half, behind the barrier (fp8pv: a V^T tile is 8 KiB, 2 pieces)
            piece = (i - n // 2) // (n // 8)
            pre.append(f"s_add_i32 m0, {S_M0V}, {piece * 1024}")
            dma = f"global_load_lds_dwordx4 v{170 + piece}, {S_VP}"
        if i % ph.wg == 0:  # counted wait (LDS reads retire in order) for every fragment first used in slots i .. i + 3
            need = [f2 for f2 in range(nf) if i <= ph.fu[f2] < i + ph.wg]
            if need:
                younger = issued_before_slot[i] - last[(0, max(need))] - 1
                assert 0 <= younger <= 15, (ph.name, i, younger)
                pre.append(f"s_waitcnt lgkmcnt({younger})")
        mf = text.format(fr=FR(f, ph.pool)) if f is not None else text
        rd = [f"ds_read_b128 {b_}, {reg} offset:{imm}" for (_, b_, reg, imm) in reads[i]]
        post += adv_at[i]
        post += ph.valu[i]
        if "nomfma" in XS:
            mf = "s_nop 0"
        if "nowait" in XS:
            pre = [p_ for p_ in pre if not p_.startswith("s_waitcnt lgkmcnt")]
        if "novalu" in XS:
            post = [p_ for p_ in post if p_.startswith(("s_", ".Law16l", "v_xor", "v_cmp"))]
        if "noexp" in XS:
            post = [p_.replace("v_exp_f32", "v_mov_b32") for p_ in post]
        if XDROP:  # drop:<prefix>+<prefix>: the gap instructions that start with one of the prefixes are left out ("~" = a space)
            post = [p_ for p_ in post if not p_.startswith(tuple(XDROP.replace("~", " ").split("+")))]
        if "nolds" in XS:
            rd = []
        o.append(f"; slot {i}")
        if ph.barrier and FP8 and i == 0 and "nobarrier" not in XS:
            # fp8 mode: an X phase is 16 slots long and the NEXT Y phase's first V^T fragments are read from its slot 0 on (look-ahead 16) —
            # the barrier that publishes V^T(t+1) must stand in front of them (in the bf16 stream those reads start at slot 48, behind
            # the mid-phase barrier).  Same accounting: the newest V^T(t+2) [4] + K(t+4) [2] pieces may fly.
            o += [f"s_waitcnt vmcnt({4 if PV8 else 6})", "s_barrier"]
        o += pre + [mf] + rd
        if dma:
            o.append(dma)
        o += post
        if ph.barrier and not FP8 and i == n // 2 and "nobarrier" not in XS:
            # everything but this wave's newest pieces — V^T(t+2) [4] and K(t+4) [4; 2 in fp8 mode] — has landed: K(t+3), V^T(t+1)
            o += ["s_waitcnt vmcnt(8)", "s_barrier"]
    if drain:
        o += ["s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15"]
    return o


def early_reads(ph):
    """the reads of ph's fragments that precede its slot 0 (what the previous phase's tail, or an own prefetch, issues), in order"""
    out = []
    for f, fr in enumerate(ph.frags):
        if ph.rd[f] < 0:
            for k, (reg, imm) in enumerate(fr):
                out.append(f"ds_read_b128 {FR(f, ph.pool, k if ph.pool != 'V' else None)}, {reg} offset:{imm}")
    return out


def check_rule3(seq):
    """Linearise a sequence of phases and assert that every read refills a buffer whose previous fragment's LAST MFMA sits strictly
    before the MFMA slot the read is issued behind (so a later MFMA has issued and the old operand has left the front of the matrix
    pipe — DESIGN 4.4 rule 3), and that reads are issued in stream order."""
    base, last_user, prev_rd = 0, {}, None
    for ph in seq:
        for f in range(len(ph.frags)):
            rd = base + ph.rd[f]
            assert prev_rd is None or rd >= prev_rd or PV8, ("stream order", ph.name, f)  # (fp8pv: two pools, each in its own order)
            prev_rd = rd
            pb = (ph.pool, f % (NBUFK if ph.pool != "V" else NBUF))
            assert last_user.get(pb, -10**9) < rd, ("rule 3", ph.name, f, pb, last_user.get(pb), rd)
            last_user[pb] = base + ph.lu[f]
        base += ph.n


def build():
    K_REGS, V_REGS = [KAD(s) for s in range(2 if FP8 else 4)], [VAD(k) for k in range(2)]
    EVEN, ODD = TILE, 3 * TILE  # ring slot s -> s + 1: xor one slot size out of an even slot, three out of an odd one
    KEVEN, KODD = TILE_K, 3 * TILE_K
    py, px, pp1 = (plan_y_pv8, plan_x_pv8, plan_p1_pv8) if PV8 else (plan_y_f8, plan_x_f8, plan_p1_f8) if FP8 else (plan_y, plan_x, plan_p1)
    pre = Phase("pre: X(0) -> S0", "X", 0)
    pre.advance = (K_REGS, KEVEN)                    # K leaves slot 0
    p1 = Phase("P1: X(1) -> S1 | softmax(0)", "X", 1)
    p1.advance, p1.barrier = (K_REGS, KODD), True    # K leaves slot 1
    pp1(p1, "p1")
    ye = Phase("Y(t), t even | softmax 1st half of tile t+1 (S1)", "Y")
    ye.advance, ye.dma = (V_REGS, EVEN), "K"
    py(ye, 1, "ye")
    xe = Phase("X(t+2) -> S0, t even | softmax 2nd half of tile t+1 (S1)", "X", 0)
    xe.advance, xe.dma, xe.barrier = (K_REGS, KEVEN), "V", True   # t + 2 even
    px(xe, 1, "xe")
    yo = Phase("Y(t), t odd | softmax 1st half of tile t+1 (S0)", "Y")
    yo.advance, yo.dma = (V_REGS, ODD), "K"
    py(yo, 0, "yo")
    xo = Phase("X(t+2) -> S1, t odd | softmax 2nd half of tile t+1 (S0)", "X", 1)
    xo.advance, xo.dma, xo.barrier = (K_REGS, KODD), "V", True
    px(xo, 0, "xo")
    return pre, p1, ye, xe, yo, xo


def stream():
    pre, p1, ye, xe, yo, xo = build()
    if FP8:  # pre and P1 are 16 MFMAs each, as long as the look-ahead and back to back on the same four K buffers: each fetches its own
        check_rule3([pre])  # fragments and ends drained, and Y(0)'s first fragments are fetched in front of the loop label
        check_rule3([p1])
        check_rule3([ye, xe, yo, xo, ye, xe, yo, xo, ye])
    else:
        check_rule3([pre, p1, ye, xe, yo, xo, ye, xe, yo, xo, ye])

    def dma_setup():
        """scalar state of one tile: K(min(t + 4, n - 1)) and V^T(min(t + 3, n - 1)): HBM base and ring slot of either"""
        return [f"s_add_i32 {S_TILE}, {S_T}, 4",
                f"s_min_i32 {S_TILE}, {S_TILE}, %[ntm1]",
                f"s_lshl_b32 {S_TMP}, {S_TILE}, {13 if FP8 else 14}",   # log2 of a K tile's bytes, in HBM and in LDS
                "s_add_u32 s80, %[kb_lo], " + S_TMP,
                "s_addc_u32 s81, %[kb_hi], 0",
                f"s_and_b32 {S_TMP}, {S_TILE}, 3",
                f"s_lshl_b32 {S_M0K}, {S_TMP}, {13 if FP8 else 14}",
                f"s_add_i32 {S_M0K}, {S_M0K}, %[woffk]",
                f"s_cmp_eq_u32 {S_TILE}, %[ntm1]",
                f"s_cselect_b64 {S_MASKK}, -1, 0",
                f"s_add_i32 {S_TILE}, {S_T}, 3",
                f"s_min_i32 {S_TILE}, {S_TILE}, %[ntm1]",
                f"s_lshl_b32 {S_TMP}, {S_TILE}, {6 if PV8 else 7}",   # a tile's 64 keys in a V^T row: 128 bytes of bf16, 64 of e4m3
                "s_add_u32 s82, %[vb_lo], " + S_TMP,
                "s_addc_u32 s83, %[vb_hi], 0",
                f"s_and_b32 {S_TMP}, {S_TILE}, 3",
                f"s_lshl_b32 {S_TMP}, {S_TMP}, 14",
                f"s_add_i32 {S_M0V}, {S_TMP}, %[woffv]"]

    def rag_flag():
        """S_FLAG = 1 when the softmax of this Y phase (tile t + 1) works on the last tile and that tile is ragged"""
        return [f"s_add_i32 {S_TMP}, {S_T}, 1",
                f"s_cmp_eq_u32 {S_TMP}, %[ntm1]",
                f"s_cselect_b32 {S_FLAG}, 1, 0",
                f"s_cmp_lt_u32 {S_RAG}, 64",
                f"s_cselect_b32 {S_FLAG}, {S_FLAG}, 0"]

    o = [f"s_mov_b32 {S_RAG}, %[rag]", f"s_mov_b32 {S_FLAG}, 0", f"s_mov_b32 {S_RESC}, 0", f"s_mov_b32 {S_T}, 0"]
    if PV8:  # the lane-group mask of the 32 x 32 accumulators and the 8-register ones fragment (row 0 of the A operand = e4m3 1.0, handed over in v200)
        o += ["s_mov_b32 s96, 0xffff0000", "s_mov_b32 s97, 0xffff0000"] + [f"v_mov_b32 v{144 + r}, v200" for r in range(8)]
    o += emit_phase(pre, None if FP8 else p1, own_prefetch=True, drain=FP8)
    # the one un-hidden softmax piece of a workgroup: tile 0's maxima.  X(0)'s last MFMAs must have written S0 (an MFMA result needs
    # ~40 clocks); tile 0 is never the ragged last tile (n >= 2); O^T = l = 0: no rescale_o, no flag
    o += ["s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15"] + rescale_s(0, False)
    if FP8:
        o += emit_phase(p1, None, own_prefetch=True, drain=True) + early_reads(ye)
    else:
        o += emit_phase(p1, ye)
    o += [".Law16l_loop_%=:"]
    for y, x in ((ye, xe), (yo, xo)):
        o += [] if "nosetup" in XS else dma_setup() + rag_flag()
        o += emit_phase(y, x)
        o += [f"s_cmp_eq_u32 {S_T}, %[ntm1]",
              "s_cbranch_scc1 .Law16l_done_%="]
        o += emit_phase(x, yo if y is ye else ye)
        o += [f"s_add_i32 {S_T}, {S_T}, 1"]
    o += ["s_branch .Law16l_loop_%=",
          ".Law16l_done_%=:",
          # Y's tail fetched fragments of an X phase that does not follow: let them land; the statement ends drained
          "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15"]
    if OOL_BLOCKS:  # the rare paths, behind the stream: each ends in a jump back to where it was called from
        o += ["s_branch .Law16l_end_%="] + [ln for blk in OOL_BLOCKS for ln in blk] + [".Law16l_end_%=:"]
    return o


def main():
    lines = [x for ln in stream() for x in ln.split(" ;; ")]
    stem = "attention_w16lf8pv_loop" if PV8 else "attention_w16lf8_loop" if FP8 else "attention_w16l_loop"
    path = os.path.join(ROOT, "diffusion-rs_amd", "csrc", stem + ".inc")
    if TAG:
        os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
        path = os.path.join(ROOT, "build", f"{stem}_{TAG}.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_attention_w16l.py — do not edit.  The whole KV stream of attention_w16l_kernel as one asm\n")
        f.write("// statement (pre, P1, loop { Y(t); X(t+2) } unrolled over two tiles); register map and schedule: see the generator.\n")
        f.write(f"#define FMI_AW16L{'F8PV' if PV8 else 'F8' if FP8 else ''}_LOOP_ASM \\\n")
        body = ['  "' + ln + '\\n\\t"' for ln in lines if not ln.startswith(";")]
        f.write(" \\\n".join(body))
        f.write("\n")
    if os.environ.get("AW16L_DUMP"):
        with open(os.environ["AW16L_DUMP"], "w") as f:
            f.write("\n".join(lines) + "\n")
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    n_other = sum(1 for ln in lines if not ln.startswith(";") and not ln.startswith("v_mfma") and not ln.endswith(":"))
    print(f"{path}: {len(lines)} lines, {n_mfma} MFMAs, {n_other} other instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
