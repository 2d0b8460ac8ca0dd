#!/usr/bin/env python
"""Generates diffusion-rs_amd/csrc/attention_w4_loop.inc: the steady-state KV loop of attention_w4_kernel
(attention_w4.h) as ONE inline-asm statement with hand-assigned registers.

Why: the C++ form of the loop is a chain of small asm statements (one MFMA + fragment read, one softmax slice
per MFMA gap).  hipcc pads an `s_nop` between two asm statements whenever the second reads a VGPR the first one
wrote (it assumes a destination-select forwarding hazard for every asm def) — 52 issue slots per KV tile pair in
a loop whose speed is set by the number of instructions issued between the MFMAs — and places its own address /
loop-control instructions wherever they fall.  Inside one statement nothing is padded and every instruction
sits in the gap this script puts it in.

The schedule is the one documented in attention_w4.h (phases B(t), A(t+1); MFMA order, fragment stream,
softmax slices, DMA pieces, barrier); the arithmetic instructions and their order are those of the C++
phases, so the result stays bit-identical to attention_pp_kernel.

Register map (per lane; pinned by the operand constraints in attention_w4.h):
  a[0:127]    O^T accumulators  ot[b][i]  -> a[(4b+i)*16 ...]
  a[128:191]  Q fragments       qf[b][s]  -> a[128 + (8b+s)*4 ...]
  v[0:63]     S^T               sc[b][u]  -> v[(2b+u)*16 ...]
  v[64:95]    P fragments       pf[b][c]  -> v[64 + (4b+c)*4 ...]
  v[96:127]   streamed K / V^T fragments fr[0..7]
  v[128:135]  k_ad   v[136:139] v_ad
  v[140:143]  k_voff v[144:147] v_voff  v[148:151] k_voff clamped to the last key (ragged last tile)
  v[152:153]  m_run  v[154:155] l_run
  v[156:171]  temporaries (clobbers)
  s[80:95]    loop state (clobbers)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 16384
PF = 8

E0, E1, LS, PMAX, T0, T1, SUM, TA, TB, TD = (f"v{n}" for n in range(156, 166))
XT = [f"v{n}" for n in range(166, 172)]
S_KP, S_VP, S_MASK = "s[80:81]", "s[82:83]", "s[84:85]"
S_T, S_TILE, S_M0K, S_M0V, S_TMP, S_MKK, S_MKV = "s86", "s87", "s88", "s89", "s90", "s91", "s92"


def OT(b, i):
    lo = (4 * b + i) * 16
    return f"a[{lo}:{lo + 15}]"


def OTr(b, n):  # n = 0..63
    return f"a{64 * b + n}"


def QF(b, s):
    lo = 128 + (8 * b + s) * 4
    return f"a[{lo}:{lo + 3}]"


def SC(b, u):
    lo = (2 * b + u) * 16
    return f"v[{lo}:{lo + 15}]"


def SCr(b, u, r):
    return f"v{(2 * b + u) * 16 + r}"


def PFt(b, c):
    lo = 64 + (4 * b + c) * 4
    return f"v[{lo}:{lo + 3}]"


def PFd(b, c, d):
    return f"v{64 + (4 * b + c) * 4 + d}"


def FR(i):
    lo = 96 + 4 * (i % PF)
    return f"v[{lo}:{lo + 3}]"


def KAD(s):
    return f"v{128 + s}"


def VAD(c):
    return f"v{136 + c}"


def M(b):
    return f"v{152 + b}"


def L(b):
    return f"v{154 + b}"


def frag_src(k):
    """address register and immediate of the fragment of MFMA k of a two-product phase"""
    j = k >> 1
    if k % 2 == 0:  # PV step j: d block j & 3, k-step j >> 2
        return VAD(j >> 2), (j & 3) * 4096
    return KAD(j >> 1), (j & 1) * 8192  # QK^T step j: key half j & 1, d-step j >> 1


def exp_step(b, k):
    """step k = 0..17 of the exp stream of block b: start pair k (two scores: fma, fma ... exp, exp) and finish pair k - 2 (row sum,
    bf16 pack).  Two pairs are in flight, in alternating register sets: a transcendental co-issues with the VALU instructions
    behind it, so its result must not be read, nor its source overwritten, within the next few instructions — with a skew of
    one pair and the stream cut at instruction granularity (exp, exp, fma, fma, add in one gap) half of the lanes read a stale
    exponential (measured).  Same operations in the same order per element as the C++ phases: p = exp2(fma(s, scale, -m));
    lsum += p0 + p1 in pair order; pack(p0, p1)."""
    out = []
    ea = (E0, E1) if k % 2 == 0 else (XT[2], XT[3])            # pair k
    ec = (E0, E1) if (k - 2) % 2 == 0 else (XT[2], XT[3])      # pair k - 2 (same set, two steps later)
    t0, t1 = (T0, T1) if k & 1 else (XT[0], XT[1])
    if k < 16:
        u, r = k >> 3, 2 * (k & 7)
        out += [f"v_fma_f32 {t0}, {SCr(b, u, r)}, %[sl], -{M(b)}", f"v_fma_f32 {t1}, {SCr(b, u, r + 1)}, %[sl], -{M(b)}"]
    if k >= 2:
        kp = k - 2
        pk = PFd(b, 2 * (kp >> 3) + ((kp & 7) >> 2), kp & 3)
        out += [f"v_add_f32 {SUM}, {ec[0]}, {ec[1]}", f"v_cvt_pk_bf16_f32 {pk}, {ec[0]}, {ec[1]}", f"v_add_f32 {LS}, {LS}, {SUM}"]
    if k < 16:
        out += [f"v_exp_f32 {ea[0]}, {t0}", f"v_exp_f32 {ea[1]}, {t1}"]
    if k == 17:
        out.append(f"v_add_f32 {L(b)}, {L(b)}, {LS}")
    return out


def max_step(b, g):
    u, r = g >> 2, 4 * (g & 3)
    s = [SCr(b, u, r + e) for e in range(4)]
    if g == 0:
        return [f"v_max_f32 {PMAX}, {s[0]}, {s[1]}", f"v_max3_f32 {PMAX}, {PMAX}, {s[2]}, {s[3]}"]
    return [f"v_max3_f32 {PMAX}, {PMAX}, {s[0]}, {s[1]}", f"v_max3_f32 {PMAX}, {PMAX}, {s[2]}, {s[3]}"]


def rescale_block(b, label):
    """taken when some row's maximum moved more than the threshold: m_run, l_run and O^T(b) rescaled (sm_decide)"""
    out = [f"v_max_f32 {TB}, {M(b)}, {TA}",        # mn = max(m_run, ps)
           f"v_sub_f32 {T0}, {M(b)}, {TB}",
           f"v_exp_f32 {T0}, {T0}",                 # alpha
           f"v_mov_b32 {M(b)}, {TB}",
           f"v_accvgpr_read_b32 {XT[0]}, {OTr(b, 0)}",
           f"v_mul_f32 {L(b)}, {L(b)}, {T0}"]
    n = len(XT)
    # software pipeline over the 64 accumulator registers: read r + 1 is issued before r is scaled and written back
    for r in range(64):
        if r + 1 < 64:
            out.append(f"v_accvgpr_read_b32 {XT[(r + 1) % n]}, {OTr(b, r + 1)}")
        out.append(f"v_mul_f32 {XT[r % n]}, {XT[r % n]}, {T0}")
        out.append(f"v_accvgpr_write_b32 {OTr(b, r)}, {XT[r % n]}")
    return out


def fixed_load(i, is_a):
    """instructions between MFMA i and the next MFMA besides the softmax slice: the fragment read and gap i's trailing
    work, plus what the next gap issues in front of its MFMA"""
    n = 1                                              # the fragment read
    if not is_a and (i & 7) == 3:
        n += 1                                         # K piece
    if is_a and i >= 16 and (i & 3) == 3:
        n += 1                                         # V^T piece
    if is_a and i == 16:
        n += 2                                         # vmcnt wait, barrier
    if is_a and i >= 24:
        n += 3 if i == 31 else 1                       # ring-slot xors behind the read
    j = (i + 1) % 32                                   # the next gap's head (for i = 31: gap 0 of the other phase)
    nxt_a = is_a if i < 31 else not is_a
    if (j & 3) == 0:
        n += 1                                         # counted lgkm wait
    if not nxt_a and (j & 7) == 3:
        n += 2                                         # m0, clamp select
    if nxt_a and j >= 16 and (j & 3) == 3:
        n += 1                                         # m0
    if nxt_a and j == 24:
        n += 2                                         # the two ring-slot xors the next reads need
    return n


def softmax_plan(b, is_a, uid):
    """The softmax of block b cut into per-gap slices.  The wave issues in order and an MFMA occupies the matrix pipe for 32
    clocks: a gap whose other instructions issue in less is free, one that needs more stalls the pipe — so the slices are
    levelled: every gap carries about the same number of instructions (MFMA + read + fixed work + softmax), instead of whole
    7-instruction exp steps in some gaps and nothing in others.
      gaps 0..4   the running maximum, 8 steps of 4 scores (1, 2, 2, 2, 1 per gap) — S^T(b) was finished by the previous
                  phase's MFMAs 29 (first half) / 31 (second half): three MFMAs before gap 0 reads the first scores and
                  three before gap 2 reads the second half
      gap 5       row maximum across the two halves of the wave, deferred-rescale decision
      gap 6       branch over the (rarely taken) rescale block
      gaps 6..31  the exp stream: 16 pairs of (fma, fma | exp, exp | add, cvt_pk, add), skewed by one pair so that a
                  transcendental's result is not used by the next instruction, cut at instruction granularity."""
    plan = [[] for _ in range(32)]
    for g in range(8):
        plan[1 + g // 2] += max_step(b, g)
    plan[4].append(f"v_mov_b32 {TA}, {PMAX}")
    plan[5] += [f"v_permlane32_swap_b32 {PMAX}, {TA}", f"v_mov_b32 {LS}, 0", f"v_max_f32 {TA}, {PMAX}, {TA}", f"v_mul_f32 {TA}, %[sl], {TA}",
                f"v_sub_f32 {TB}, {TA}, {M(b)}", f"v_cmp_lt_f32 vcc, %[thr], {TB}"]
    skip = f".Law4_skip_{uid}_%="
    plan[6] += [f"s_cbranch_vccz {skip}"] + rescale_block(b, uid) + [f"{skip}:"]
    stream = []
    for k in range(18):
        stream += exp_step(b, k)
    first, last = 6, 31
    # level: instructions between two MFMAs as equal as the fixed work allows
    fixed = [fixed_load(i, is_a) + (1 if i == first else 0) for i in range(32)]
    total = sum(fixed[first:last + 1]) + len(stream)
    take = [0] * 32
    left = len(stream)
    level = -(-total // (last - first + 1))
    while True:
        take = [max(0, level - fixed[i]) if first <= i <= last else 0 for i in range(32)]
        if sum(take) >= len(stream):
            break
        level += 1
    # trim the surplus from the last gaps backwards (the stream must still end in gap 31: its tail feeds the next phase late)
    surplus = sum(take) - len(stream)
    i = first
    while surplus > 0:
        if take[i] > 0:
            take[i] -= 1
            surplus -= 1
        i = i + 1 if i < last else first
    pos = 0
    for i in range(first, last + 1):
        plan[i] += stream[pos:pos + take[i]]
        pos += take[i]
    assert pos == len(stream)
    return plan


def phase(name, b_pv, b_qk, b_sm, is_a, uid):
    """one two-product phase of 32 MFMAs; returns asm lines.  The phase that follows is also a two-product phase (its first
    PF fragments are fetched behind this phase's last MFMAs)."""
    o = [f"; ---- phase {name}: PV(b={b_pv}) + QK^T(b={b_qk}) beside softmax(b={b_sm})"]
    sm_plan = softmax_plan(b_sm, is_a, uid)
    lazy_xor = []
    for i in range(32):
        pre = []    # before the MFMA
        post = []   # after the fragment read
        # ---- DMA pieces: B stages K(tile) at gaps 3, 11, 19, 27; A stages V^T(tile) at gaps 19, 23, 27, 31
        dma = None
        if not is_a and (i & 7) == 3:
            piece = i >> 3
            pre.append(f"s_add_i32 m0, {S_M0K}, {piece * 1024}")
            pre.append(f"v_cndmask_b32 {TD}, v{140 + piece}, v{148 + piece}, {S_MASK}")
            dma = f"global_load_lds_dwordx4 {TD}, {S_KP}"
        if is_a and i >= 16 and (i & 3) == 3:
            piece = (i - 16) >> 2
            pre.append(f"s_add_i32 m0, {S_M0V}, {piece * 1024}")
            dma = f"global_load_lds_dwordx4 v{144 + piece}, {S_VP}"
        # ---- address registers move to the next ring slot behind this phase's last own read (A only).  The fragments of the
        # next phase fetched in gaps 25..31 need v_ad[0] (gap 25), k_ad[0] (gap 26) and k_ad[1] (gap 30); the other nine
        # follow one per gap (the own reads end with gap 24's: fragment 31, k_ad[7]).
        if is_a and i == 25:
            pre += [f"v_xor_b32 {VAD(0)}, {S_MKV}, {VAD(0)}", f"v_xor_b32 {KAD(0)}, {S_MKK}, {KAD(0)}"]
            lazy_xor = [f"v_xor_b32 {KAD(1)}, {S_MKK}, {KAD(1)}"] + [f"v_xor_b32 {VAD(c)}, {S_MKV}, {VAD(c)}" for c in (1, 2, 3)] + \
                       [f"v_xor_b32 {KAD(s)}, {S_MKK}, {KAD(s)}" for s in range(2, 8)]
        # ---- the MFMA of gap i, behind a counted wait every fourth gap
        if (i & 3) == 0:
            pre.append("s_waitcnt lgkmcnt(2)" if os.environ.get("AW4_X") == "halfreads" else f"s_waitcnt lgkmcnt({PF - 5})")
        j = i >> 1
        if i % 2 == 0:
            mf = f"v_mfma_f32_32x32x16_bf16 {OT(b_pv, j & 3)}, {FR(i)}, {PFt(b_pv, j >> 2)}, {OT(b_pv, j & 3)}"
        else:
            u, s = j & 1, j >> 1
            mf = f"v_mfma_f32_32x32x16_bf16 {SC(b_qk, u)}, {FR(i)}, {QF(b_qk, s)}, {'0' if s == 0 else SC(b_qk, u)}"
        # ---- the read behind MFMA i refills the buffer that MFMA i - 1 consumed (fragment i + PF - 1 of this phase, or
        # i + PF - 1 - 32 of the next).  Never the buffer of MFMA i itself: an MFMA can still be waiting for the matrix pipe
        # when the LDS data returns, and nothing orders the LDS write to its A operand behind the operand read — with other
        # work on the CU the MFMA then multiplied the fragment meant for MFMA i + 8 (tools/attn_race.hip).  MFMA i has
        # issued, so MFMA i - 1 has left the pipe's front: its operands are free by construction, not by timing.
        reg, imm = frag_src((i + PF - 1) % 32)
        rd = f"ds_read_b128 {FR(i + PF - 1)}, {reg} offset:{imm}"
        # ---- softmax slice of gap i (see `softmax_plan`)
        post += sm_plan[i]
        if lazy_xor and i >= 25:
            post.append(lazy_xor.pop(0))
        if is_a and i == 31:
            post += lazy_xor
            lazy_xor = []
        o.append(f"; gap {i}")
        # timing experiments only (wrong results): AW4_X=halfreads | novalu | noexp | nobarrier | nodma
        X = os.environ.get("AW4_X", "")
        if X == "halfreads" and (i & 1):
            rd = "s_nop 0"
        if X == "novalu":
            post = [p_ for p_ in post if p_.startswith(("s_cbranch", ".Law4", "v_xor", "v_cmp"))]
        if X == "noexp":
            post = [p_.replace("v_exp_f32", "v_mov_b32") for p_ in post]
        if X == "nodma":
            dma = None
        o += pre + [mf, rd] + post
        if dma:
            o.append(dma)
        if is_a and i == 16 and os.environ.get("AW4_X") != "nobarrier":
            o += ["s_waitcnt vmcnt(8)", "s_barrier"]
    return o


def loop():
    o = []
    # the first PF - 1 fragments of B(0) (the statement is entered with nothing in flight), then zero iterations when ntiles == 2
    for k in range(PF - 1):
        reg, imm = frag_src(k)
        o.append(f"ds_read_b128 {FR(k)}, {reg} offset:{imm}")
    o += [f"s_mov_b32 {S_T}, 0",
          "s_cmp_lt_i32 %[nt], 3",
          "s_cbranch_scc1 .Law4_done_%=",
          ".Law4_loop_%=:",
          # the tile both DMA streams fetch in this iteration: min(t + 3, ntiles - 1), into ring slot (tile & 3)
          f"s_add_i32 {S_TILE}, {S_T}, 3",
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
          f"s_add_i32 {S_M0V}, {S_M0K}, {4 * TILE}",
          f"s_cmp_eq_u32 {S_TILE}, %[ntm1]",
          f"s_cselect_b64 {S_MASK}, -1, 0",
          # ring-slot masks of A(t + 1): K leaves slot t + 1, V^T slot t
          # (slot s -> s + 1 is an xor with 1 << 14 out of an even slot, 3 << 14 out of an odd one)
          f"s_and_b32 {S_TMP}, {S_T}, 1",
          f"s_lshl_b32 {S_TMP}, {S_TMP}, 15",
          f"s_or_b32 {S_MKV}, {S_TMP}, {TILE}",
          f"s_xor_b32 {S_MKK}, {S_MKV}, {2 * TILE}"]
    o += phase("B(t)", 0, 0, 1, False, "b")
    o += phase("A(t+1)", 1, 1, 0, True, "a")
    o += [f"s_add_i32 {S_T}, {S_T}, 1",
          f"s_add_i32 {S_TMP}, {S_T}, 2",
          f"s_cmp_lt_i32 {S_TMP}, %[nt]",
          "s_cbranch_scc1 .Law4_loop_%=",
          ".Law4_done_%=:",
          # the statement ends drained: hipcc knows nothing of the reads / MFMAs in flight and may copy the operands behind it
          "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 15"]
    return o


def main():
    lines = loop()
    path = os.path.join(ROOT, "diffusion-rs_amd", "csrc", "attention_w4_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_attention_w4_loop.py — do not edit.  The steady-state KV loop of attention_w4_kernel\n")
        f.write("// (phases B(t), A(t+1) for t = 0 .. ntiles-3) as one asm statement; register map and schedule: see the generator.\n")
        f.write("#define FMI_AW4_LOOP_ASM \\\n")
        body = []
        for ln in lines:
            if ln.startswith(";"):
                continue
            body.append('  "' + ln + '\\n\\t"')
        f.write(" \\\n".join(body))
        f.write("\n")
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    n_other = sum(1 for ln in lines if not ln.startswith(";") and not ln.startswith("v_mfma") and not ln.endswith(":"))
    print(f"{path}: {len(lines)} lines, {n_mfma} MFMAs, {n_other} other instructions (incl. 2 x 199 of the rarely taken rescale blocks)")


if __name__ == "__main__":
    main()
