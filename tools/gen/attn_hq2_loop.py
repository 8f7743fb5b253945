#!/usr/bin/env python
"""Generator of the hand-scheduled key loop of attn_frag_hq2_kernel (csrc/attn2.hip), the fp16 path's attention on two query
blocks per wave (round 6): writes csrc/attn_hq2_loop.inc.

    python tools/gen/attn_hq2_loop.py           (the .inc is committed; re-run after editing the schedule)

Same design as tools/gen/attn_x3_loop.py's P16 statement with the lo terms gone -- ONE inline-assembly statement, two 32-query
blocks A and B per wave half a step apart, a ring of four [K tile | V tile] buffers of 64 keys:
    phase X(c):  VALU  softmax of A on key block c        | MFMA  P.V + row sums of B for block c - 1, scores of B for block c
    phase Y(c):  VALU  softmax of B on key block c        | MFMA  P.V + row sums of A for block c,     scores of A for block c + 1
Per phase: 4 big MFMAs (v_mfma_f32_32x32x16_f16: two score, two P.V) + 4 small ones (v_mfma_f32_4x4x4_16b_f16 with an all-ones
A operand: the row sums of the packed probabilities) = 160 matrix-pipe cycles beside 24 VALU instructions (16 v_exp_f32, 8
v_cvt_pk_f16_f32: ~126 cycles at the measured issue costs) and two fragment reads.  The compiler-scheduled attn_frag_kernel
spends 337 cycles per (query block, key block) pair on the same work.

ARITHMETIC IS THAT OF attn_frag_kernel, BIT FOR BIT (the forward picks the kernel by launch size): scores = (-m - P_SHIFT) +
k0 . q0 + k1 . q1 in that order on the accumulator input, exp2 in place, round-to-nearest packing, row sums H[0:2], H[2:4],
H[4:6], H[6:8] in that order, P.V = v0 . H[0:4] then v1 . H[4:8].
"""
import os

KBX = 2                      # 32-key blocks per LDS tile
BLK = 2048                   # bytes of one fragment-major block
TILE = KBX * BLK             # K (or V) part of a ring buffer
BUF = 2 * TILE               # one ring buffer: [K tile | V tile]
NBUF = 4

OPS = ["accA", "accB", "lA", "lB", "t", "soff",               # "+v" x4, "+s" x2 (tile index; global byte offset of tile t + 3)
       "qA0", "qA1", "qB0", "qB1", "negmA", "negmB",          # "v"
       "klane", "vlane", "dmaoff",                            # "v": LDS byte address of this lane's K / V fragment in buffer 0; tid * 16
       "rk", "rv", "m0base", "nfull"]                         # "s": descriptors, LDS address of smem + wave * 1024, loop end
IDX = {n: i for i, n in enumerate(OPS)}


def o(name):
    return "%" + str(IDX[name])


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


S = {"A": 240, "B": 224}          # scores / probabilities, 16 each
H = {"A": 216, "B": 208}          # packed probabilities, 8 each
KB_ = [200, 192]                  # K fragment buffers (k0, k1: 4 registers each)
VB_ = [184, 176]                  # V fragment buffers (v0, v1)
ONES = 174                        # v[174:175]: packed fp16 (1, 1, 1, 1)
T = 170                           # kcur, knext, vcur, vnext
CLOBBER_V = list(range(170, 256))
ST = 88                           # SGPR temporaries s88 .. s95
CLOBBER_S = list(range(88, 96))
K_OFF = [0, 512]                  # k0, k1 inside a block (lane part in the address register)
V_OFF = [0, 1024]                 # v0, v1


def mfma(d, a, b, c):
    return f"v_mfma_f32_32x32x16_f16 {d}, {a}, {b}, {c}"


def score_mfmas(q, kbuf):
    s = vr(S[q], 16)
    k = [vr(KB_[kbuf] + 4 * i, 4) for i in range(2)]
    return [mfma(s, k[0], o(f"q{q}0"), o(f"negm{q}")), mfma(s, k[1], o(f"q{q}1"), s)]


def pv_mfmas(q, vbuf):
    acc = o(f"acc{q}")
    v = [vr(VB_[vbuf] + 4 * i, 4) for i in range(2)]
    return [mfma(acc, v[0], vr(H[q], 4), acc), mfma(acc, v[1], vr(H[q] + 4, 4), acc)]


def rowsum_mfmas(q):
    lq = o(f"l{q}")
    return [f"v_mfma_f32_4x4x4_16b_f16 {lq}, {vr(ONES, 2)}, {vr(H[q] + 2 * i, 2)}, {lq}" for i in range(4)]


def frag_reads(kind, buf, addr, blk_off):
    base = (KB_ if kind == "K" else VB_)[buf]
    offs = K_OFF if kind == "K" else V_OFF
    extra = 0 if kind == "K" else TILE
    return [f"ds_read_b128 {vr(base + 4 * i, 4)}, {addr} offset:{extra + blk_off + offs[i]}" for i in range(2)]


def dma_group(piece, s_dst, s_off):
    """one LDS-DMA instruction of the refill (1 KB per wave): piece 0 = the K tile, 1 = the V tile"""
    rs = o("rk") if piece == 0 else o("rv")
    # (m0 written by a SALU instruction may not be used by the very next LDS-DMA: one instruction in between)
    return [f"s_add_u32 m0, {s_dst}, {TILE if piece else 0}", "s_nop 0",
            f"buffer_load_dwordx4 {o('dmaoff')}, {rs}, {s_off} offen lds"]


def phase(sm, mm, pv_vbuf, sc_kbuf, reads, head=(), dma=None):
    """MFMAs of block `mm` -- P.V (2) and row sums (4 small) of its previous key block, scores (2) of its next one -- beside the
    24 VALU instructions of block `sm`'s softmax step:
        pv0 | E0-3 rd0 | rs0 | sc0 | E4-7 C0 C1 | rs1 | pv1 | E8-11 rd1 C2 C3 | rs2 | sc1 | E12-15 C4 C5 | rs3 | C6 C7
    NO two matrix instructions are adjacent: a wave issues in order, so an MFMA that has to wait for the pipe holds back every
    VALU instruction behind it (the first form of this loop had the score pair back to back and ran 5 % SLOWER than the
    compiler-scheduled kernel).  Every big MFMA is followed by ~31 cycles of VALU work (4 exponentials at 6.5, two conversions at
    2.8 or a fragment read) for its 32 cycles on the pipe; the two MFMAs of a chain (sc0 -> sc1, pv0 -> pv1, the four row sums)
    have another big MFMA between them, so the dependent one finds its input written.  Hazards: the scores are read by the VALU
    in the NEXT phase (sc1 .. first exponential: 11 instructions and a big MFMA); the packed words C0-3 are read by the next
    phase's pv0 / rs0 / rs1 at least a gap later, C6 C7 by its pv1 / rs3."""
    pv, sc, rs = pv_mfmas(mm, pv_vbuf), score_mfmas(mm, sc_kbuf), rowsum_mfmas(mm)
    s = lambda r: vr(S[sm] + r)      # noqa: E731
    E = [f"v_exp_f32_e32 {s(r)}, {s(r)}" for r in range(16)]
    C = [f"v_cvt_pk_f16_f32 {vr(H[sm] + j)}, {s(2 * j)}, {s(2 * j + 1)}" for j in range(8)]
    d = dma or [[], [], []]
    out = list(head)
    out += [pv[0]] + E[0:4] + [reads[0]] + [rs[0]] + d[0]
    out += [sc[0]] + E[4:8] + C[0:2] + [rs[1]] + d[1]
    out += [pv[1]] + E[8:12] + [reads[1]] + C[2:4] + [rs[2]] + d[2]
    out += [sc[1]] + E[12:16] + C[4:6] + [rs[3]]
    out += C[6:8]
    return out


def build():
    kcur, knext, vcur, vnext = vr(T), vr(T + 1), vr(T + 2), vr(T + 3)
    s0, s1 = f"s{ST}", f"s{ST + 1}"
    A = []
    # ---- fill: K(0) -> K buffer 0, V(0) -> both V buffers (the first P.V / row sums of B multiply zeros), scores of A for block 0
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
    A += ["s_nop 7", "s_nop 7"]   # (MFMA result -> VALU read; once per workgroup)
    A += ["Lloop%=:"]
    A += [f"s_and_b32 {s0}, {o('t')}, {NBUF - 1}", f"s_lshl_b32 {s0}, {s0}, {BUF.bit_length() - 1}",
          f"s_add_u32 {s1}, {o('t')}, 1", f"s_and_b32 {s1}, {s1}, {NBUF - 1}", f"s_lshl_b32 {s1}, {s1}, {BUF.bit_length() - 1}",
          f"v_add_u32_e32 {kcur}, {s0}, {o('klane')}", f"v_add_u32_e32 {vcur}, {s0}, {o('vlane')}",
          f"v_add_u32_e32 {knext}, {s1}, {o('klane')}", f"v_add_u32_e32 {vnext}, {s1}, {o('vlane')}"]
    # block c = 0 of the tile: K(c) in K buffer 0, V(c) in V buffer 0, V(c - 1) in V buffer 1
    A += phase("A", "B", pv_vbuf=1, sc_kbuf=0, reads=frag_reads("K", 1, kcur, BLK))                                   # X(0): reads K(1)
    A += phase("B", "A", pv_vbuf=0, sc_kbuf=1, reads=frag_reads("V", 1, vcur, BLK), head=["s_waitcnt lgkmcnt(0)"])  # Y(0): reads V(1)
    # block c = 1: the next tile must have landed before its K(0) is read; the refill of the ring rides in this phase
    s_dst = f"s{ST + 3}"
    refill_prep = [f"s_add_u32 {s_dst}, {o('t')}, 3", f"s_and_b32 {s_dst}, {s_dst}, {NBUF - 1}",
                   f"s_lshl_b32 {s_dst}, {s_dst}, {BUF.bit_length() - 1}", f"s_add_u32 {s_dst}, {s_dst}, {o('m0base')}"]
    dma = [refill_prep, dma_group(0, s_dst, o("soff")), dma_group(1, s_dst, o("soff"))]
    A += phase("A", "B", pv_vbuf=0, sc_kbuf=1, reads=frag_reads("K", 0, knext, 0),
               head=["s_waitcnt vmcnt(2) lgkmcnt(0)", "s_barrier"], dma=dma)                                          # X(1): reads K(0) of tile t + 1
    A += phase("B", "A", pv_vbuf=1, sc_kbuf=0, reads=frag_reads("V", 0, vnext, 0), head=["s_waitcnt lgkmcnt(0)"])   # Y(1): reads V(0) of tile t + 1
    A += [f"s_add_u32 {o('soff')}, {o('soff')}, {TILE}",      # (a tile of K is TILE bytes in global memory; so is V)
          f"s_add_u32 {o('t')}, {o('t')}, 1",
          f"s_cmp_lt_i32 {o('t')}, {o('nfull')}",
          "s_cbranch_scc1 Lloop%="]
    # ---- drain: P.V and row sums of B for the last block (V buffer 1); dependent MFMAs kept apart by explicit wait states
    pv, rs = pv_mfmas("B", 1), rowsum_mfmas("B")
    A += [pv[0], rs[0], "s_nop 7", "s_nop 3", rs[1], "s_nop 7", "s_nop 3", pv[1], rs[2], "s_nop 7", "s_nop 3", rs[3]]
    A += ["s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7"]   # (the fragment reads of the next tile land in registers hipcc may reuse)
    return A


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "..", "beat_this_amd", "csrc", "attn_hq2_loop.inc")
    lines = build()
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen/attn_hq2_loop.py -- do not edit; the schedule is described there.\n")
        f.write(f"// {len(lines)} instructions, {sum('v_mfma_f32_32x32' in x for x in lines)} big + {sum('v_mfma_f32_4x4x4' in x for x in lines)} "
                "small MFMAs; operands: " + ", ".join(f"%{i} {n}" for i, n in enumerate(OPS)) + "\n")
        f.write("#define ATTN_HQ2_ASM \\\n")
        for x in lines:
            f.write(f'  "{x}\\n\\t" \\\n')
        f.write('  ""\n')
        f.write("#define ATTN_HQ2_CLOBBERS " + ", ".join(f'"v{i}"' for i in CLOBBER_V) + ", " +
                ", ".join(f'"s{i}"' for i in CLOBBER_S) + ', "scc", "memory"\n')
        f.write(f"#define ATTN_HQ2_KBX {KBX}\n#define ATTN_HQ2_NBUF {NBUF}\n")
    print(f"wrote {os.path.normpath(out)}: {len(lines)} instructions")


if __name__ == "__main__":
    main()
