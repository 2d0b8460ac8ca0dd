#!/usr/bin/env python
"""Generates diffusion-rs_amd/csrc/attention_w32_loop.inc: the whole KV stream of attention_w32_kernel (attention_w32.h)
as ONE inline-asm statement with hand-assigned registers.

The same design as tools/gen_attention_w32.py (read its header first: folded scale / running maximum, no cross-lane traffic on
the common softmax path, exp2 in place, row sums from a ones-row MFMA, whole stream generated, rule-3 fragment buffers) on
`v_mfma_f32_32x32x16_bf16` instead of 16x16x32.  Why both exist: with one wave per SIMD the kernel is bound by what is issued
between the MFMAs (tools/gen_issue_model.py, profiles/r03_issue_model.txt): a 32-clock 32x32x16 MFMA hides ~19 clocks of other
instructions, a 16-clock 16x16x32 MFMA ~8, i.e. 19 vs 16 per 32 K FLOP — and half as many MFMAs and counted waits are issued.

Layout (attention_w4_kernel's): block b = 32 queries; S^T[b][u] (u = key half, 32 keys) is one 32 x 32 accumulator: lane
(hl = lane / 32, q = lane % 32) owns query q and the keys 8 (r >> 2) + 4 hl + (r & 3), r = 0..15, of the half.  QK^T: 2 halves x
8 d-steps of 16; PV: 4 d blocks of 32 x 4 k-steps of 16 keys, P[b][c] = pack of S registers (the V^T k-permutation is the one
this layout was designed around); the ones row adds one MFMA per k-step into OL[b] (row 0 = lanes 0..31, register 0).

Register map (pinned by the operand constraints in attention_w32.h):
  a[0:127]    O^T  O[b][dt] -> a[(4b+dt)*16 ..]     a[128:191] Q fragments QF[b][s] -> a[128+(8b+s)*4 ..]     a[192:223] OL[b] -> a[192+16b ..]
  v[0:63]     S^T  S[b][u]  -> v[(2b+u)*16 ..]      v[64:95]   P[b][c] -> v[64+(4b+c)*4 ..]                    v[96:127]  fragment buffers FR[0..7]
  v[128:135]  KAD[s]  v[136:139] VAD[c]  v[140:143] k_voff  v[144:147] v_voff  v[148:151] k_voff clamped (ragged last tile)
  v152 lane key offset 4 hl   v153 DMA offset temporary   v[154:155] M[b]   v[156:159] the ones fragment (row 0 = bf16 1.0)
  v[160:191]  NM[b] (-m of the lane's query, 16 copies)   v[192:219] temporaries (clobbers)   s[80:95] loop state (clobbers)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 16384
VT_RING = 4 * TILE
NBUF = 8          # fragment buffers
LOOKAHEAD = int(os.environ.get("AW32_LOOKAHEAD", "6"))    # a fragment is read this many MFMA slots ahead of its first use (about 6 reads in flight)
X = os.environ.get("AW32_X", "")  # timing experiments only (wrong results): novalu | noexp | nodma | nobarrier | halfreads | nomfma | nowait

PMAX, TA, TB, T0, T1, AL, DL = (f"v{n}" for n in range(192, 199))
XT = [f"v{n}" for n in range(199, 205)]
PMX = [f"v{n}" for n in range(205, 220)]  # partial maxima of the tree
LKEY, DMAT = "v152", "v153"
S_KP, S_VP, S_MASK = "s[80:81]", "s[82:83]", "s[84:85]"
S_T, S_TILE, S_M0K, S_M0V, S_TMP, S_MKK, S_MKV, S_RAG, S_TMP2, S_FLAG = "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95"
NEG_BIG = "0xf149f2ca"  # -1e30f
ONESF = "v[156:159]"


def O(b, dt):
    lo = (4 * b + dt) * 16
    return f"a[{lo}:{lo + 15}]"


def Or(b, k):  # k = 0..63
    return f"a{64 * b + k}"


def QF(b, s):
    lo = 128 + (8 * b + s) * 4
    return f"a[{lo}:{lo + 3}]"


def OL(b):
    lo = 192 + 16 * b
    return f"a[{lo}:{lo + 15}]"


def S(b, u):
    lo = (2 * b + u) * 16
    return f"v[{lo}:{lo + 15}]"


def Sr(b, u, r):
    return f"v{(2 * b + u) * 16 + r}"


def P(b, c):
    lo = 64 + (4 * b + c) * 4
    return f"v[{lo}:{lo + 3}]"


def Pr(b, c, d):
    return f"v{64 + (4 * b + c) * 4 + d}"


def FR(n):
    lo = 96 + 4 * (n % NBUF)
    return f"v[{lo}:{lo + 3}]"


def KAD(s):
    return f"v{128 + s}"


def VAD(c):
    return f"v{136 + c}"


def NM(b):
    lo = 160 + 16 * b
    return f"v[{lo}:{lo + 15}]"


def NMr(b, i):
    return f"v{160 + 16 * b + i}"


def M(b):
    return f"v{154 + b}"


# ----------------------------------------------------------------------------------------------------------------------------
# MFMA sequences.  A product's 32 MFMAs use 16 fragments, each for two consecutive MFMAs of that product (c = 0, 1).
def pv_seq(b):
    """(MFMA text, V^T fragment index or None) of PV(b): per k-step c (16 keys) 4 d blocks, then the ones row (row sums)"""
    out = []
    for c in range(4):
        for dt in range(4):
            out.append((f"v_mfma_f32_32x32x16_bf16 {O(b, dt)}, {{fr}}, {P(b, c)}, {O(b, dt)}", 4 * c + dt))
        out.append((f"v_mfma_f32_32x32x16_bf16 {OL(b)}, {ONESF}, {P(b, c)}, {OL(b)}", None))
    return out


def qk_seq(b):
    """QK^T(b): d-step s = j >> 1, key half u = j & 1; the first d-step accumulates onto -m of the lane's query (the fold)"""
    out = []
    for j in range(16):
        s_, u = j >> 1, j & 1
        acc = NM(b) if s_ == 0 else S(b, u)
        out.append((f"v_mfma_f32_32x32x16_bf16 {S(b, u)}, {{fr}}, {QF(b, s_)}, {acc}", j))
    return out


def pv_frag(f):  # fragment f = 0..15 of a PV product: k-step c = f >> 2, d block dt = f & 3
    return VAD(f >> 2), (f & 3) * 4096


def qk_frag(f):  # fragment f = 0..15 of a QK^T product: d-step s = f >> 1, key half u = f & 1
    return KAD(f >> 1), (f & 1) * 8192


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
            npv, nqk, ip, iq = len(pv), len(qk), 0, 0
            while ip < npv or iq < nqk:   # proportional merge: the 20 PV MFMAs (16 + 4 ones rows) spread evenly among the 16 QK^T ones
                if iq >= nqk or (ip < npv and ip * nqk <= iq * npv):
                    t, f = pv[ip]
                    ip += 1
                    seq.append((t, None if f is None else ("V", f)))
                else:
                    t, f = qk[iq]
                    iq += 1
                    seq.append((t, ("K", f)))
            index = {}
            for t, key in seq:
                if key is not None and key not in index:
                    index[key] = len(self.frags)
                    self.frags.append(pv_frag(key[1]) if key[0] == "V" else qk_frag(key[1]))
                self.mfma.append((t, None if key is None else index[key]))
        else:
            self.mfma = pv_seq(pv_b) if pv_b is not None else qk_seq(qk_b)
            self.frags = [pv_frag(f) if pv_b is not None else qk_frag(f) for f in range(16)]
        self.n = len(self.mfma)
        self.fu = [min(i for i, (_, ff) in enumerate(self.mfma) if ff == f) for f in range(len(self.frags))]
        self.lu = [max(i for i, (_, ff) in enumerate(self.mfma) if ff == f) for f in range(len(self.frags))]
        assert self.fu == sorted(self.fu), self.fu
        self.buf0 = 0


# ----------------------------------------------------------------------------------------------------------------------------
# softmax of block b as instruction streams
def mask_block(b):
    """ragged last tile: scores of keys >= Lk become -1e30 (p = 0).  The key of (u, r) in lane (hl, q) is
    32 u + 8 (r >> 2) + (r & 3) + LKEY (= 4 hl); S_RAG = keys in the last tile (1..64)."""
    out = [f"v_mov_b32 {T1}, {NEG_BIG}"]
    for u in range(2):
        for r in range(16):
            koff = 32 * u + 8 * (r >> 2) + (r & 3)
            out.append(f"s_sub_i32 {S_TMP2}, {S_RAG}, {koff}")          # key valid <=> LKEY + koff < rag
            out.append(f"v_cmp_le_i32 vcc, {S_TMP2}, {LKEY}")
            out.append(f"v_cndmask_b32 {Sr(b, u, r)}, {Sr(b, u, r)}, {T1}, vcc")
    return out


def max_stream(b):
    """maximum over the lane's 32 scores of block b (one query): a tree of independent v_max3, key half 0 first"""
    regs = [Sr(b, u, r) for u in range(2) for r in range(16)]
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
    """Taken when some score of the wave exceeds the threshold, and always on a block's first tile (M = -1e30, fold = 0): the
    lane's query has its maximum in PMAX (this lane's 32 keys) and in lane ^ 32 (the other 32): m' = max(M, max s' - NM),
    delta = m' + NM (how far this tile's fold was off), alpha = exp2(-max(delta, 0));  then s' -= delta, O^T *= alpha,
    OL *= alpha, M = m', NM = -m'."""
    out = [f"v_mov_b32 {TA}, {PMAX}",
           "s_nop 1",
           f"v_permlane32_swap_b32 {PMAX}, {TA}",
           "s_nop 1",
           f"v_max_f32 {TA}, {PMAX}, {TA}",
           f"v_sub_f32 {T0}, {TA}, {NMr(b, 0)}",            # the maximum in the unshifted domain: ps' - NM
           f"v_max_f32 {T0}, {M(b)}, {T0}",                  # m'
           f"v_add_f32 {DL}, {T0}, {NMr(b, 0)}",             # delta = m' - (the m this tile's fold used)
           f"v_max_f32 {AL}, {DL}, 0",
           f"v_sub_f32 {AL}, 0, {AL}",
           f"v_exp_f32 {AL}, {AL}",                          # alpha = exp2(-max(delta, 0))
           f"v_mov_b32 {M(b)}, {T0}"]
    out += [f"v_sub_f32 {NMr(b, i)}, 0, {T0}" for i in range(16)]
    out += [f"v_sub_f32 {Sr(b, u, r)}, {Sr(b, u, r)}, {DL}" for u in range(2) for r in range(16)]
    lo = 192 + 16 * b
    out += [f"v_accvgpr_read_b32 {T1}, a{lo}", "s_nop 0", f"v_mul_f32 {T1}, {T1}, {AL}", "s_nop 0", f"v_accvgpr_write_b32 a{lo}, {T1}"]
    n = len(XT)
    out.append(f"v_accvgpr_read_b32 {XT[0]}, {Or(b, 0)}")
    for r in range(64):  # software pipeline over the 64 accumulator registers of block b
        if r + 1 < 64:
            out.append(f"v_accvgpr_read_b32 {XT[(r + 1) % n]}, {Or(b, r + 1)}")
        out.append(f"v_mul_f32 {XT[r % n]}, {XT[r % n]}, {AL}")
        out.append(f"v_accvgpr_write_b32 {Or(b, r)}, {XT[r % n]}")
    return out


def exp_stream(b):
    """p = exp2(s') in place, then per unit of 4 scores two packs to bf16 (P fragment dwords; the row sums come from the ones-row
    MFMAs).  Unit k = registers 4 (k & 3) .. + 3 of half u = k >> 2  ->  P[b][2 u + ((k & 3) >> 1)] dwords 2 (k & 1), + 1.  The
    packs of unit k follow the exponentials of unit k + 2: a v_exp_f32 result must not be read within the next few VALU
    instructions (gfx950: stale in half of the lanes, DESIGN 4.4 rule 2)."""
    out = []

    def start(k):
        u, r0 = k >> 2, 4 * (k & 3)
        return [f"v_exp_f32 {Sr(b, u, r0 + i)}, {Sr(b, u, r0 + i)}" for i in range(4)]

    def finish(k):
        u, r0 = k >> 2, 4 * (k & 3)
        c, d0 = 2 * u + ((k & 3) >> 1), 2 * (k & 1)
        return [f"v_cvt_pk_bf16_f32 {Pr(b, c, d0 + d)}, {Sr(b, u, r0 + 2 * d)}, {Sr(b, u, r0 + 2 * d + 1)}" for d in range(2)]

    SKEW = 2
    for k in range(8 + SKEW):
        if k < 8:
            out += start(k)
        if k >= SKEW:
            out += finish(k - SKEW)
    return out


def spread(plan, stream, first, last):
    """stream instructions over gaps first..last (inclusive), as evenly as integer division allows, in order"""
    n = last - first + 1
    for k, ins in enumerate(stream):
        plan[first + k * n // len(stream)].append(ins)


def softmax_plan(ph, uid):
    """instruction lists per gap for the softmax of ph.sm_b: mask branch, max tree, decision + rescale branch, exp stream.
    S^T(b) was finished by the previous phase's last QK^T MFMAs: nothing reads it before gap 2 (>= 2 MFMAs = 64+ clocks behind;
    an MFMA result needs ~50)."""
    b, n = ph.sm_b, ph.n
    plan = [[] for _ in range(n)]
    g_mask = 2
    skipm = f".Law32_nomask_{uid}_%="
    plan[g_mask] += [f"s_cmp_eq_u32 {S_FLAG}, 0", f"s_cbranch_scc1 {skipm}"] + mask_block(b) + [f"{skipm}:"]
    span = 4 if n >= 30 else 3
    spread(plan, max_stream(b), g_mask + 1, g_mask + span)
    g_dec = g_mask + span + 1
    skip, do = f".Law32_skip_{uid}_%=", f".Law32_resc_{uid}_%="
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
    # ---- the reads of every gap, in stream order
    reads = [[] for _ in range(n)]           # (stream index relative to ph.buf0, reg, imm)
    early = []                               # read before slot 0 (previous phase's tail or own prefetch), in order
    for f, (reg, imm) in enumerate(ph.frags):
        i = ph.fu[f] - LOOKAHEAD
        (reads[i] if i >= 0 else early).append((f, reg, imm))
    own_last_read = max([g for g in range(n) if reads[g]], default=-1)
    nxt_first_read = n
    if nxt is not None:
        for f, (reg, imm) in enumerate(nxt.frags):
            i = n + nxt.fu[f] - LOOKAHEAD
            if i < n:
                assert i > own_last_read, (ph.name, "next phase's reads must follow the own ones")
                reads[i].append((nf + f, reg, imm))
                nxt_first_read = min(nxt_first_read, i)
    if own_prefetch:
        for (f, reg, imm) in early:
            o.append(f"ds_read_b128 {FR(ph.buf0 + f)}, {reg} offset:{imm}")
    # position of every read in issue order (early ones first): younger(f, i) = reads issued after f's and before MFMA slot i
    order = [f for (f, _, _) in early]
    issued_before_slot = [len(order)]
    for g in range(n):
        order += [f for (f, _, _) in reads[g]]
        issued_before_slot.append(len(order))   # issued before MFMA slot g + 1
    pos = {f: k for k, f in enumerate(order)}
    # ---- ring-slot advance of the address registers: each register right behind the last own read that uses it (the next
    # phase's reads of that register come later by construction: asserted)
    adv_at = [[] for _ in range(n)]
    if ph.advance:
        regs = ([(KAD(s_), S_MKK) for s_ in range(8)] if "K" in ph.advance else []) + ([(VAD(k_), S_MKV) for k_ in range(4)] if "V" in ph.advance else [])
        for reg, mask in regs:
            own = [g for g in range(n) for (fs, r_, _) in reads[g] if fs < nf and r_ == reg]
            g_last = max(own, default=0)
            nxt_use = [g for g in range(n) for (fs, r_, _) in reads[g] if fs >= nf and r_ == reg]
            assert all(g > g_last for g in nxt_use), (ph.name, reg, g_last, nxt_use)
            adv_at[g_last].append(f"v_xor_b32 {reg}, {mask}, {reg}")
    # DMA pieces (4 per phase): B stages K(tile) spread over the phase, A stages V^T(tile) in the second half (behind the barrier)
    k_slots = [(2 * k + 1) * n // 8 for k in range(4)]
    v_slots = [n // 2 + 1 + k * (n - n // 2 - 2) // 3 for k in range(4)]
    assert len(set(k_slots)) == 4 and len(set(v_slots)) == 4 and max(v_slots) < n and min(v_slots) > n // 2
    for i in range(n):
        text, f = ph.mfma[i]
        pre, post = [], []
        dma = None
        if ph.dma == "K" and i in k_slots and X != "nodma":
            piece = k_slots.index(i)
            pre.append(f"s_add_i32 m0, {S_M0K}, {piece * 1024}")
            pre.append(f"v_cndmask_b32 {DMAT}, v{140 + piece}, v{148 + piece}, {S_MASK}")
            dma = f"global_load_lds_dwordx4 {DMAT}, {S_KP}"
        if ph.dma == "V" and i in v_slots and X != "nodma":
            piece = v_slots.index(i)
            pre.append(f"s_add_i32 m0, {S_M0V}, {piece * 1024}")
            dma = f"global_load_lds_dwordx4 v{144 + piece}, {S_VP}"
        # ---- counted wait (LDS reads retire in order) every second slot, for every fragment first used in slots i, i + 1
        if i % 2 == 0:
            need = [f2 for f2 in range(nf) if i <= ph.fu[f2] < i + 2]
            if need:
                younger = issued_before_slot[i] - pos[max(need)] - 1
                assert 0 <= younger <= 15, (ph.name, i, younger)
                pre.append(f"s_waitcnt lgkmcnt({younger})")
        mf = text.format(fr=FR(ph.buf0 + f)) if f is not None else text
        rd = [f"ds_read_b128 {FR(ph.buf0 + fs)}, {reg} offset:{imm}" for (fs, reg, imm) in reads[i]]
        post += adv_at[i]
        post += plan[i]
        if X == "halfreads":
            rd = [r_ if k % 2 == 0 else "s_nop 0" for k, r_ in enumerate(rd)] if i % 4 < 2 else ["s_nop 0" for _ in rd]
        if X == "nomfma":
            mf = "s_nop 0"
        if X == "nowait":
            pre = [p_ for p_ in pre if not p_.startswith("s_waitcnt lgkmcnt")]
        if X.startswith("drop_"):  # drop every instruction whose mnemonic starts with one of the '+'-separated prefixes
            pref = tuple(X[5:].split("+"))
            post = [p_ for p_ in post if not p_.startswith(pref)]
        if X == "novalu":
            post = [p_ for p_ in post if p_.startswith(("s_", ".Law32", "v_xor", "v_cmp"))]
        if X == "noexp":
            post = [p_.replace("v_exp_f32", "v_mov_b32") for p_ in post]
        o.append(f"; slot {i}")
        o += pre + [mf] + rd
        if dma:
            o.append(dma)
        o += post
        if ph.barrier and i == n // 2 and X != "nobarrier":
            o += ["s_waitcnt vmcnt(8)", "s_barrier"]
    if drain:
        o += ["s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15"]
    return o


def check_rule3(seq):
    """Linearise a sequence of phases and assert that every read refills a buffer whose previous fragment's LAST MFMA sits
    strictly before the MFMA slot the read is issued behind (so a later MFMA has issued and the old operand has left the front
    of the matrix pipe), and that reads are issued in stream order."""
    base, stream = 0, []
    for ph in seq:
        for f in range(len(ph.frags)):
            stream.append((base + ph.fu[f] - LOOKAHEAD, base + ph.fu[f], base + ph.lu[f], ph.name))
        base += ph.n
    for k, (rd, fu, lu, name) in enumerate(stream):
        if k >= 1:
            assert rd >= stream[k - 1][0], ("stream order", name, k)
        if k >= NBUF:
            assert stream[k - NBUF][2] < rd, ("rule 3", name, k, stream[k - NBUF], rd)
        if rd >= 0 and k + 1 < len(stream):
            pass


def build():
    pre = Phase("pre: QK(0,0)", None, 0, None, None, False, "")
    a0 = Phase("A0: QK(1,0) | softmax(0,0)", None, 1, 0, "V", True, "K", forced_rescale="always")
    bt = Phase("B(t): PV(0,t) + QK(0,t+1) | softmax(1,t)", 0, 0, 1, "K", False, "", forced_rescale="t0")
    at = Phase("A(t+1): PV(1,t) + QK(1,t+1) | softmax(0,t+1)", 1, 1, 0, "V", True, "KV", forced_rescale="tm1")
    post = Phase("post: PV(1,n-1)", 1, None, None, None, False, "")
    return pre, a0, bt, at, post


def loop():
    pre, a0, bt, at, post = build()
    # buffer numbering: pre (16 fragments), A0 (16), then the loop body B (32), A (32): every phase starts at a multiple of NBUF
    pos = 0
    for ph in (pre, a0, bt, at):
        ph.buf0 = pos
        pos += len(ph.frags)
        assert ph.buf0 % NBUF == 0
    post.buf0 = 0
    check_rule3([pre, a0, bt, at, bt, at, bt])
    check_rule3([post])

    def dma_setup():
        """scalar state of one loop iteration: the tile both DMA streams fetch, min(t + 3, n - 1), and its ring slot"""
        return [f"s_add_i32 {S_TILE}, {S_T}, 3",
                f"s_min_i32 {S_TILE}, {S_TILE}, %[ntm1]",
                f"s_lshl_b32 {S_TMP}, {S_TILE}, 14",
                f"s_add_u32 s80, %[kb_lo], {S_TMP}",
                f"s_addc_u32 s81, %[kb_hi], 0",
                f"s_lshl_b32 {S_TMP}, {S_TILE}, 7",
                f"s_add_u32 s82, %[vb_lo], {S_TMP}",
                f"s_addc_u32 s83, %[vb_hi], 0",
                f"s_and_b32 {S_TMP}, {S_TILE}, 3",
                f"s_lshl_b32 {S_TMP}, {S_TMP}, 14",
                f"s_add_i32 {S_M0K}, {S_TMP}, %[woff]",
                f"s_add_i32 {S_M0V}, {S_M0K}, {VT_RING}",
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
    o += [f"s_mov_b32 {S_MKK}, {TILE}",                 # K leaves slot 0: even slot -> xor 1 << 14
          f"s_mov_b32 {S_RAG}, %[rag]",
          f"s_mov_b32 {S_FLAG}, 0"]                      # softmax(0,0): tile 0 is never the last (n >= 2)
    o += emit_phase(pre, a0, "pre", own_prefetch=True)
    o += emit_phase(a0, bt, "a0")
    o += [f"s_mov_b32 {S_T}, 0",
          ".Law32_loop_%=:"]
    o += dma_setup()
    # ring-slot masks of A(t+1): K leaves slot t + 1, V^T slot t (slot s -> s + 1: xor 1 << 14 out of an even slot, 3 << 14 out of an odd one)
    o += [f"s_and_b32 {S_TMP}, {S_T}, 1",
          f"s_lshl_b32 {S_TMP}, {S_TMP}, 15",
          f"s_or_b32 {S_MKV}, {S_TMP}, {TILE}",
          f"s_xor_b32 {S_MKK}, {S_MKV}, {2 * TILE}"]
    o += rag_flag(S_T)                                   # B(t): softmax(1, t)
    o += emit_phase(bt, at, "b")
    o += [f"s_cmp_eq_u32 {S_T}, %[ntm1]",
          "s_cbranch_scc1 .Law32_done_%=",
          f"s_add_i32 {S_TMP}, {S_T}, 1"]
    o += rag_flag(S_TMP)                                 # A(t+1): softmax(0, t+1)
    o += emit_phase(at, bt, "a")
    o += [f"s_add_i32 {S_T}, {S_T}, 1",
          "s_branch .Law32_loop_%=",
          ".Law32_done_%=:",
          # B's tail fetched fragments of an A phase that does not follow: let them land, and let B's last MFMAs leave the front of
          # the matrix pipe before post's own fragments refill their buffers (rule 3)
          "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15", "s_nop 15"]
    o += emit_phase(post, None, "post", own_prefetch=True, drain=True)
    return o


def main():
    lines = loop()
    path = os.path.join(ROOT, "diffusion-rs_amd", "csrc", "attention_w32_loop.inc")
    if X:
        os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
        path = os.path.join(ROOT, "build", f"attention_w32_loop_{X}.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_attention_w32.py — do not edit.  The whole KV stream of attention_w32_kernel as one asm\n")
        f.write("// statement (pre, A0, loop { B(t); A(t+1) }, post); register map and schedule: see the generator.\n")
        f.write("#define FMI_AW32_LOOP_ASM \\\n")
        body = ['  "' + ln + '\\n\\t"' for ln in lines if not ln.startswith(";")]
        f.write(" \\\n".join(body))
        f.write("\n")
    if os.environ.get("AW32_DUMP"):
        with open(os.environ["AW32_DUMP"], "w") as f:
            f.write("\n".join(lines) + "\n")
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    n_other = sum(1 for ln in lines if not ln.startswith(";") and not ln.startswith("v_mfma") and not ln.endswith(":"))
    print(f"{path}: {len(lines)} lines, {n_mfma} MFMAs, {n_other} other instructions", file=sys.stderr)


if __name__ == "__main__":
    main()
