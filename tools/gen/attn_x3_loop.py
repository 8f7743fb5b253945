#!/usr/bin/env python
"""Generator of the hand-scheduled key loop of attn_frag_x3q2_kernel (csrc/attn2.hip): writes csrc/attn_x3_loop.inc.

    python tools/gen/attn_x3_loop.py            (the .inc is committed; re-run after editing the schedule)

The loop is ONE inline-assembly statement: hipcc allocates the operands that are used as whole vectors (Q fragments, the
reference-maximum splats, the output accumulators), everything that is touched element by element lives in physical
registers v116 .. v255 that the statement lists as clobbers.  What the schedule is and why: DESIGN.md section 5
("hand-scheduled key loop"); the kernel side of the contract (operand order, LDS layout, ring of four 64-key tiles) is in
csrc/attn2.hip next to the statement.

One wave = two 32-query blocks A and B, one 32-key block per step.  The two blocks run half a step apart:
    phase X(c):  VALU  softmax of A on key block c        | MFMA  scores of B for block c,     P.V of B for block c - 1
    phase Y(c):  VALU  softmax of B on key block c        | MFMA  scores of A for block c + 1, P.V of A for block c
so every phase pairs 12 big MFMAs of one query block (32 cycles each on the SIMD's matrix pipe) with the 56 VALU
instructions of the other one (16 exponentials, the 24-instruction hi + lo split, 16 row-sum adds) and 4 fragment reads:
five single-issue instructions per MFMA gap, placed by hand (MI355X_MICROARCH.md: <= 5 fillers fit a gap).

Round 5: a second statement, ATTN_X3Q2P_ASM -- the "P16" arithmetic (profiles/r05_flip_frontier.txt): the
probabilities enter P.V as their fp16 hi parts only (P_hi . V_hi + P_hi . V_lo: four MFMAs instead of six, no lo split), and
the row sums are taken from the SAME rounded values on v_mfma_f32_4x4x4_16b_f16 with an all-ones A operand (numerator and
denominator of the softmax see identical probabilities, fp16 subnormals included, so the rounding largely cancels in O / l).
Per step and query block: 10 big + 4 small MFMAs and 24 VALU instructions (16 exponentials, 8 conversions) against 12 and 56.
"""
import os

KBX = 2                      # 32-key blocks per LDS tile
BLK = 4096                   # bytes of one [hi 2 KB | lo 2 KB] block
TILE = KBX * BLK             # K (or V) part of a ring buffer
BUF = 2 * TILE               # one ring buffer: [K tile | V tile]
NBUF = 4

# ---- operands (order = the asm statement's operand list in attn2.hip) -------------------------------------------------
OPS = ["accA", "accB", "lA", "lB", "t", "soff",               # "+v" x4, "+s" x2 (tile index; global byte offset of tile t + 3)
       "qA0", "qA1", "qA0l", "qA1l", "qB0", "qB1", "qB0l", "qB1l", "negmA", "negmB",   # "v"
       "klane", "vlane", "dmaoff",                            # "v": LDS byte address of this lane's K / V fragment in buffer 0; tid * 16
       "rk", "rv", "m0base", "m1", "nfull"]                   # "s": descriptors, LDS address of smem + wave * 1024, -1.0f, loop end
IDX = {n: i for i, n in enumerate(OPS)}


def o(name):
    return "%" + str(IDX[name])


# ---- physical registers -----------------------------------------------------------------------------------------------
def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


S = {"A": 240, "B": 224}          # scores / probabilities, 16 each
H = {"A": 216, "B": 200}          # packed hi halves of the probabilities, 8 each
L = {"A": 208, "B": 192}          # packed lo halves
KB_ = [176, 160]                  # K fragment buffers (k0, k1, k0l, k1l: 4 registers each)
VB_ = [144, 128]                  # V fragment buffers (v0, v1, v0l, v1l)
T = 116                           # temporaries: T+0..4 row-sum tree, T+5 kcur, T+6 knext, T+7 vcur, T+8 vnext
CLOBBER_V = list(range(116, 125)) + list(range(128, 256))   # (v125 .. v127 are not touched by the three-term statement)
ST = 88                           # SGPR temporaries s88 .. s95
CLOBBER_S = list(range(88, 96))

K_OFF = [0, 512, 2048, 2560]      # k0, k1, k0l, k1l inside a block (lane part in the address register)
V_OFF = [0, 1024, 2048, 3072]     # v0, v1, v0l, v1l


def mfma(d, a, b, c):
    return f"v_mfma_f32_32x32x16_f16 {d}, {a}, {b}, {c}"


def score_mfmas(q, kbuf):
    """S_q = negm_q + K . Q_q^T: small terms first (k0l q0, k1l q1, k0 q0l, k1 q1l, k0 q0, k1 q1)"""
    s = vr(S[q], 16)
    k = [vr(KB_[kbuf] + 4 * i, 4) for i in range(4)]
    qq = [o(f"q{q}0"), o(f"q{q}1"), o(f"q{q}0l"), o(f"q{q}1l")]
    seq = [(k[2], qq[0]), (k[3], qq[1]), (k[0], qq[2]), (k[1], qq[3]), (k[0], qq[0]), (k[1], qq[1])]
    out = []
    for i, (a, b) in enumerate(seq):
        out.append(mfma(s, a, b, o(f"negm{q}") if i == 0 else s))
    return out


def pv_mfmas(q, vbuf):
    """O_q^T += V^T . P_q^T: (v0l h0, v1l h1, v0 l0, v1 l1, v0 h0, v1 h1)"""
    acc = o(f"acc{q}")
    v = [vr(VB_[vbuf] + 4 * i, 4) for i in range(4)]
    h0, h1, l0, l1 = vr(H[q], 4), vr(H[q] + 4, 4), vr(L[q], 4), vr(L[q] + 4, 4)
    seq = [(v[2], h0), (v[3], h1), (v[0], l0), (v[1], l1), (v[0], h0), (v[1], h1)]
    return [mfma(acc, a, b, acc) for a, b in seq]


def softmax_fillers(q):
    """the 56 VALU instructions of one query block's softmax step, grouped per MFMA gap (12 groups)"""
    s = lambda r: vr(S[q] + r)      # noqa: E731
    h = lambda j: vr(H[q] + j)      # noqa: E731
    lo = lambda j: vr(L[q] + j)     # noqa: E731
    t = lambda i: vr(T + i)         # noqa: E731
    E = [f"v_exp_f32_e32 {s(r)}, {s(r)}" for r in range(16)]
    C = [f"v_cvt_pk_f16_f32 {h(j)}, {s(2 * j)}, {s(2 * j + 1)}" for j in range(8)]
    ML = [f"v_fma_mixlo_f16 {lo(j)}, {h(j)}, {o('m1')}, {s(2 * j)} op_sel:[0,0,0] op_sel_hi:[1,0,0]" for j in range(8)]
    MH = [f"v_fma_mixhi_f16 {lo(j)}, {h(j)}, {o('m1')}, {s(2 * j + 1)} op_sel:[1,0,0] op_sel_hi:[1,0,0]" for j in range(8)]
    add = lambda d, a, b: f"v_add_f32_e32 {d}, {a}, {b}"   # noqa: E731
    a, b, c, d, e = t(0), t(1), t(2), t(3), t(4)
    lq = o(f"l{q}")
    return [
        E[0:4], E[4:8], E[8:12], E[12:16],
        [C[0], C[1], add(a, s(0), s(1)), add(b, s(2), s(3)), ML[0]],
        [ML[1], MH[0], MH[1], C[2], C[3]],
        [add(a, a, b), add(c, s(4), s(5)), add(d, s(6), s(7)), ML[2], ML[3]],
        [MH[2], MH[3], add(c, c, d), add(a, a, c), C[4]],
        [C[5], add(e, s(8), s(9)), add(b, s(10), s(11)), ML[4], ML[5]],
        [MH[4], MH[5], add(e, e, b), C[6], C[7]],
        [add(c, s(12), s(13)), add(d, s(14), s(15)), ML[6], ML[7], MH[6]],
        [MH[7], add(c, c, d), add(e, e, c), add(a, a, e), add(lq, lq, a)],
    ]


ONES = 126                        # v[126:127]: packed fp16 (1, 1, 1, 1) -- the A operand of the row-sum MFMAs (P16)


def pv_mfmas_p(q, vbuf):
    """P16: O_q^T += V^T . P_hi^T: (v0l h0, v1l h1, v0 h0, v1 h1)"""
    acc = o(f"acc{q}")
    v = [vr(VB_[vbuf] + 4 * i, 4) for i in range(4)]
    h0, h1 = vr(H[q], 4), vr(H[q] + 4, 4)
    return [mfma(acc, a, b, acc) for a, b in [(v[2], h0), (v[3], h1), (v[0], h0), (v[1], h1)]]


def rowsum_mfmas(q):
    """P16: l_q (four registers, all equal) += the lane's 16 rounded probabilities, four at a time: D = ones(4x4) . B"""
    lq = o(f"l{q}")
    return [f"v_mfma_f32_4x4x4_16b_f16 {lq}, {vr(ONES, 2)}, {vr(H[q] + 2 * i, 2)}, {lq}" for i in range(4)]


def phase_p(sm, mm, pv_vbuf, sc_kbuf, reads, head=(), dma=None):
    """P16 phase: MFMAs of block `mm` -- P.V (4) and row sums (4 small) of its previous key block, scores (6) of its next one --
    beside the 24 VALU instructions of block `sm`'s softmax step.  Order: pv0 | R0 | sc0 sc1 | pv1 | R1 | sc2 sc3 | pv2 | R2 |
    sc4 sc5 | pv3 | R3.  Same-accumulator big MFMAs are either issued back to back (the score pairs: exact-overlap forwarding,
    as in the prologue) or have another big MFMA in between (>= 16 quad cycles); a small MFMA always follows a filler group, so
    that the wave issues its VALU work in the preceding big MFMA's shadow before it queues for the pipe; the last score MFMA
    is followed by pv3, R3 and the next phase's pv0 before the first exponential reads the scores (>= 18 quad cycles by pipe
    occupancy alone; 12 are required)."""
    pv, sc, rs = pv_mfmas_p(mm, pv_vbuf), score_mfmas(mm, sc_kbuf), rowsum_mfmas(mm)
    s = lambda r: vr(S[sm] + r)      # noqa: E731
    E = [f"v_exp_f32_e32 {s(r)}, {s(r)}" for r in range(16)]
    C = [f"v_cvt_pk_f16_f32 {vr(H[sm] + j)}, {s(2 * j)}, {s(2 * j + 1)}" for j in range(8)]
    g = [E[0:4] + [reads[0]],
         E[4:8] + [reads[1]] + E[8:12] + [reads[2]],
         E[12:16] + [reads[3]],
         C[0:4],
         C[4:6],
         C[6:8],
         []]
    if dma:   # refill of the ring (X(1) only): address arithmetic first, then one LDS-DMA instruction per filler group
        g[2] = g[2] + dma[0]
        for i in range(4):
            g[3 + i] = g[3 + i] + dma[1 + i]
    out = list(head)
    out += [pv[0]] + g[0] + [rs[0]]
    out += [sc[0], sc[1]] + g[1]
    out += [pv[1]] + g[2] + [rs[1]]
    out += [sc[2], sc[3]] + g[3]
    out += [pv[2]] + g[4] + [rs[2]]
    out += [sc[4], sc[5]] + g[5]
    out += [pv[3]] + g[6] + [rs[3]]
    return out


def build_p():
    """the P16 statement: same ring, same phases, same fragment buffers as build()"""
    kcur, knext = vr(T + 5), vr(T + 6)
    vcur, vnext = vr(T + 7), vr(T + 8)
    s0, s1 = f"s{ST}", f"s{ST + 1}"
    A = []
    A += ["s_nop 4",
          f"v_mov_b32_e32 {vr(ONES)}, 0x3c003c00", f"v_mov_b32_e32 {vr(ONES + 1)}, 0x3c003c00",
          f"s_and_b32 {s0}, {o('t')}, {NBUF - 1}", f"s_lshl_b32 {s0}, {s0}, {BUF.bit_length() - 1}",
          f"v_add_u32_e32 {kcur}, {s0}, {o('klane')}"]
    A += frag_reads("K", 0, kcur, 0)
    A += [f"v_add_u32_e32 {knext}, {s0}, {o('vlane')}"]
    A += frag_reads("V", 0, knext, 0) + frag_reads("V", 1, knext, 0)
    for j in range(8):
        A += [f"v_mov_b32_e32 {vr(H['B'] + j)}, 0"]
    A += ["s_waitcnt lgkmcnt(0)"]
    A += score_mfmas("A", 0)
    A += ["s_nop 7", "s_nop 7"]
    A += ["Lloop%=:"]
    A += [f"s_and_b32 {s0}, {o('t')}, {NBUF - 1}", f"s_lshl_b32 {s0}, {s0}, {BUF.bit_length() - 1}",
          f"s_add_u32 {s1}, {o('t')}, 1", f"s_and_b32 {s1}, {s1}, {NBUF - 1}", f"s_lshl_b32 {s1}, {s1}, {BUF.bit_length() - 1}",
          f"v_add_u32_e32 {kcur}, {s0}, {o('klane')}", f"v_add_u32_e32 {vcur}, {s0}, {o('vlane')}",
          f"v_add_u32_e32 {knext}, {s1}, {o('klane')}", f"v_add_u32_e32 {vnext}, {s1}, {o('vlane')}"]
    A += phase_p("A", "B", pv_vbuf=1, sc_kbuf=0, reads=frag_reads("K", 1, kcur, BLK))
    A += phase_p("B", "A", pv_vbuf=0, sc_kbuf=1, reads=frag_reads("V", 1, vcur, BLK), head=["s_waitcnt lgkmcnt(0)"])
    s_dst = f"s{ST + 3}"
    refill_prep = [f"s_add_u32 {s_dst}, {o('t')}, 3", f"s_and_b32 {s_dst}, {s_dst}, {NBUF - 1}",
                   f"s_lshl_b32 {s_dst}, {s_dst}, {BUF.bit_length() - 1}", f"s_add_u32 {s_dst}, {s_dst}, {o('m0base')}"]
    dma = [refill_prep] + [dma_group(i, s_dst, o("soff")) for i in range(4)]
    A += phase_p("A", "B", pv_vbuf=0, sc_kbuf=1, reads=frag_reads("K", 0, knext, 0),
                 head=["s_waitcnt vmcnt(4) lgkmcnt(0)", "s_barrier"], dma=dma)
    A += phase_p("B", "A", pv_vbuf=1, sc_kbuf=0, reads=frag_reads("V", 0, vnext, 0), head=["s_waitcnt lgkmcnt(0)"])
    A += [f"s_add_u32 {o('soff')}, {o('soff')}, {BUF // 2}",
          f"s_add_u32 {o('t')}, {o('t')}, 1",
          f"s_cmp_lt_i32 {o('t')}, {o('nfull')}",
          "s_cbranch_scc1 Lloop%="]
    # drain: P.V and row sums of B for the last block (V buffer 1); dependent MFMAs kept apart by explicit wait states
    pv, rs = pv_mfmas_p("B", 1), rowsum_mfmas("B")
    for i in range(4):
        A += [pv[i], rs[i], "s_nop 7", "s_nop 3"]
    A += ["s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7"]
    return A


def frag_reads(kind, buf, addr, blk_off):
    """four ds_read_b128 of one block's K or V fragments (hi and lo) into fragment buffer `buf`"""
    base = (KB_ if kind == "K" else VB_)[buf]
    offs = K_OFF if kind == "K" else V_OFF
    extra = 0 if kind == "K" else TILE
    return [f"ds_read_b128 {vr(base + 4 * i, 4)}, {addr} offset:{extra + blk_off + offs[i]}" for i in range(4)]


def phase(sm, mm, pv_vbuf, sc_kbuf, reads, head=(), tail_groups=None):
    """one phase: MFMAs of block `mm` (P.V from V buffer pv_vbuf, then scores from K buffer sc_kbuf, interleaved),
    softmax of block `sm` in the gaps, `reads` = four fragment reads placed in the first four gaps"""
    # The two chains of the phase alternate (score, P.V, score, P.V ...): two MFMAs on the SAME accumulator with other
    # instructions issued in between lose the back-to-back accumulator forwarding (+43 cycles each, MI355X_MICROARCH.md);
    # alternated, the next MFMA of a chain issues 64 cycles after its predecessor, whose result is long written.  Hazards:
    # the scores are read by the VALU (exponentials) in the NEXT phase's first gap -- the last score MFMA is the eleventh,
    # 12 instructions ahead of them (an MFMA result may not be read by anything but an accumulating MFMA for 12 wait
    # states); the first P.V MFMA reads probability words whose last half was written six instructions earlier.
    pv, sc = pv_mfmas(mm, pv_vbuf), score_mfmas(mm, sc_kbuf)
    mf = [x for pair in zip(sc, pv) for x in pair]
    fill = softmax_fillers(sm)
    out = list(head)
    if ABL & 4:
        out = [h for h in out if "barrier" not in h]
    for i in range(12):
        if not ((ABL & 16) and i % 2 == 0) and not ((ABL & 8) and i % 2 == 1):
            out.append(mf[i])
        if not (ABL & 1):
            out += fill[i]
        if i < 4 and not (ABL & 2):
            out.append(reads[i])
        if tail_groups and i in tail_groups and not (ABL & 4):
            out += tail_groups[i]
    return out


def dma_group(piece, s_dst, s_off):
    """one LDS-DMA instruction of the refill (1 KB per wave): piece 0, 1 = K pieces, 2, 3 = V pieces of the tile"""
    rs = o("rk") if piece < 2 else o("rv")
    lds_add = (piece & 1) * 4096 + (TILE if piece >= 2 else 0)
    glb_add = (piece & 1) * 4096
    return [f"s_add_u32 m0, {s_dst}, {lds_add}",
            f"s_add_u32 s{ST + 2}, {s_off}, {glb_add}",
            f"buffer_load_dwordx4 {o('dmaoff')}, {rs}, s{ST + 2} offen lds"]


ABL = 0   # development ablations (results are garbage): 1 no softmax VALU, 2 no fragment reads in the loop, 4 no refill / barrier,
          # 8 no P.V MFMAs, 16 no score MFMAs


def build():
    kcur, knext = vr(T + 5), vr(T + 6)
    s0, s1 = f"s{ST}", f"s{ST + 1}"
    A = []
    # ---- fill: addresses of tile t, K(0) -> K buffer 0, V(0) -> both V buffers (the first P.V of B multiplies zeros), zero
    # B's probability words, scores of A for block 0
    A += ["s_nop 4",
          f"s_and_b32 {s0}, {o('t')}, {NBUF - 1}", f"s_lshl_b32 {s0}, {s0}, {BUF.bit_length() - 1}",
          f"v_add_u32_e32 {kcur}, {s0}, {o('klane')}"]
    A += frag_reads("K", 0, kcur, 0)
    # (V reads take the K lane address minus the K lane part plus the V lane part: separate address register)
    A += [f"v_add_u32_e32 {knext}, {s0}, {o('vlane')}"]
    A += frag_reads("V", 0, knext, 0) + frag_reads("V", 1, knext, 0)
    for j in range(8):
        A += [f"v_mov_b32_e32 {vr(H['B'] + j)}, 0", f"v_mov_b32_e32 {vr(L['B'] + j)}, 0"]
    A += ["s_waitcnt lgkmcnt(0)"]
    A += score_mfmas("A", 0)
    A += ["s_nop 7", "s_nop 7"]   # (MFMA result -> VALU read: 12 wait states; once per workgroup)
    A += [f"Lloop%=:"]
    # ---- per tile: addresses (K reads: kcur / knext, V reads: vcur / vnext kept in T+7 / T+8) -------------------------------
    vcur, vnext = vr(T + 7), vr(T + 8)
    A += [f"s_and_b32 {s0}, {o('t')}, {NBUF - 1}", f"s_lshl_b32 {s0}, {s0}, {BUF.bit_length() - 1}",
          f"s_add_u32 {s1}, {o('t')}, 1", f"s_and_b32 {s1}, {s1}, {NBUF - 1}", f"s_lshl_b32 {s1}, {s1}, {BUF.bit_length() - 1}",
          f"v_add_u32_e32 {kcur}, {s0}, {o('klane')}", f"v_add_u32_e32 {vcur}, {s0}, {o('vlane')}",
          f"v_add_u32_e32 {knext}, {s1}, {o('klane')}", f"v_add_u32_e32 {vnext}, {s1}, {o('vlane')}"]
    # block c = 0 of the tile (global parity even): K(c) in K buffer 0, V(c) in V buffer 0, V(c - 1) in V buffer 1
    A += phase("A", "B", pv_vbuf=1, sc_kbuf=0, reads=frag_reads("K", 1, kcur, BLK))                       # X(0): reads K(1)
    A += phase("B", "A", pv_vbuf=0, sc_kbuf=1, reads=frag_reads("V", 1, vcur, BLK), head=["s_waitcnt lgkmcnt(0)"])  # Y(0): reads V(1)
    # block c = 1: the next tile must have landed before its K(0) is read; the refill of the ring goes into this phase's later gaps
    s_dst = f"s{ST + 3}"
    refill_prep = [f"s_add_u32 {s_dst}, {o('t')}, 3", f"s_and_b32 {s_dst}, {s_dst}, {NBUF - 1}",
                   f"s_lshl_b32 {s_dst}, {s_dst}, {BUF.bit_length() - 1}", f"s_add_u32 {s_dst}, {s_dst}, {o('m0base')}"]
    tg = {4: refill_prep, 5: dma_group(0, s_dst, o("soff")), 6: dma_group(1, s_dst, o("soff")),
          7: dma_group(2, s_dst, o("soff")), 8: dma_group(3, s_dst, o("soff"))}
    A += phase("A", "B", pv_vbuf=0, sc_kbuf=1, reads=frag_reads("K", 0, knext, 0),
               head=["s_waitcnt vmcnt(4) lgkmcnt(0)", "s_barrier"], tail_groups=tg)                                    # X(1): reads K(0) of tile t + 1
    A += phase("B", "A", pv_vbuf=1, sc_kbuf=0, reads=frag_reads("V", 0, vnext, 0), head=["s_waitcnt lgkmcnt(0)"])  # Y(1): reads V(0) of tile t + 1
    A += [f"s_add_u32 {o('soff')}, {o('soff')}, {BUF // 2}",     # (a tile of K is TILE bytes in global memory; so is V)
          f"s_add_u32 {o('t')}, {o('t')}, 1",
          f"s_cmp_lt_i32 {o('t')}, {o('nfull')}",
          f"s_cbranch_scc1 Lloop%="]
    # ---- drain: P.V of B for the last block (V buffer 1), then the MFMA results may be read by compiler code
    A += pv_mfmas("B", 1)
    A += ["s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7"]   # (the fragment reads of the next tile land in registers hipcc may reuse)
    return A


def main():
    global ABL
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "..", "beat_this_amd", "csrc", "attn_x3_loop.inc")
    lines = build()
    n_mfma = sum("v_mfma" in x for x in lines)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/attn_x3_loop.py -- do not edit; the schedule is described there and in DESIGN.md.\n")
        f.write(f"// {len(lines)} instructions, {n_mfma} MFMAs; operands: " + ", ".join(f"%{i} {n}" for i, n in enumerate(OPS)) + "\n")
        f.write("// (-DBT_X3Q2_ABL=n, development builds: ablated forms of the loop whose results are garbage -- what the loop spends where)\n")
        f.write("#ifndef BT_X3Q2_ABL\n#define BT_X3Q2_ABL 0\n#endif\n")
        for abl in (0, 1, 2, 3, 4, 7, 8, 16, 9, 17):
            ABL = abl
            f.write(f"#if BT_X3Q2_ABL == {abl}\n#define ATTN_X3Q2_ASM \\\n")
            for x in build():
                f.write(f'  "{x}\\n\\t" \\\n')
            f.write('  ""\n#endif\n')
        ABL = 0
        lp = build_p()
        f.write(f"// P16 statement: {len(lp)} instructions, {sum('v_mfma_f32_32x32' in x for x in lp)} big + "
                f"{sum('v_mfma_f32_4x4x4' in x for x in lp)} small MFMAs; operands as above with lA / lB four registers each\n")
        f.write("#define ATTN_X3Q2P_ASM \\\n")
        for x in lp:
            f.write(f'  "{x}\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define ATTN_X3Q2_CLOBBERS " + ", ".join(f'"v{i}"' for i in CLOBBER_V) + ", " +
                ", ".join(f'"s{i}"' for i in CLOBBER_S) + ', "scc", "memory"\n')
        # (P16 leaves the lo words, the row-sum tree's temporaries and v125 to the compiler: its row sums are four registers)
        used_p = sorted(set(range(S["A"], S["A"] + 16)) | set(range(S["B"], S["B"] + 16)) | set(range(H["A"], H["A"] + 8)) |
                        set(range(H["B"], H["B"] + 8)) | set(range(KB_[1], KB_[0] + 16)) | set(range(VB_[1], VB_[0] + 16)) |
                        set(range(T + 5, T + 9)) | {ONES, ONES + 1})
        f.write("#define ATTN_X3Q2P_CLOBBERS " + ", ".join(f'"v{i}"' for i in used_p) + ", " +
                ", ".join(f'"s{i}"' for i in CLOBBER_S) + ', "scc", "memory"\n')
        f.write(f"#define ATTN_X3Q2_KBX {KBX}\n#define ATTN_X3Q2_NBUF {NBUF}\n")
    print(f"wrote {os.path.normpath(out)}: {len(lines)} instructions, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
