#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of GEMM tile configuration 8 (csrc/gemm_bf16_cfg8.hip).

Configuration 8 is a 256x256x64 tile run by FOUR waves (one per SIMD), each owning a 128x128 output sub-tile: all 256
accumulator registers of a wave live in AGPRs a[0:255], fragments in v[0:127], and the whole K loop is ONE inline-asm
statement whose instruction order is written here - every ds_read_b128, every LDS-DMA instruction and every wait is
assigned to a gap between two MFMAs by this script (the compiler schedules nothing inside it).

    python tools/gen_gemm_a4.py            # rewrites construction-clip_amd/csrc/gemm_a4_kloop.inc

Loop structure (BK = 64, two 64 KiB LDS stages, ROTATED like configuration 7):
    iteration kt:  s_waitcnt vmcnt(0); s_barrier           tile kt visible, everybody done with stage (kt-1)&1
                   phase R: 64 MFMAs on fragment set 1 (tile kt-1, k-step 1)
                            + the 16 DMA instructions of tile kt+1 + the 16 reads of (kt, k-step 0) -> set 0
                   phase Q: 64 MFMAs on set 0 + the 16 reads of (kt, k-step 1) -> set 1
Register map (literal, listed as clobbers):
    a[4*(8*mt+nt) .. +3]   accumulator of m-tile mt (16 rows) x n-tile nt (16 columns) of the wave's 128x128
    v[0:31] / v[32:63]     set 0: activation fragments (8 m-tiles) / weight fragments (8 n-tiles)
    v[64:95] / v[96:127]   set 1
    v[128:135] / v[136:143]  per-lane source offsets of the wave's 8 + 8 DMA pieces (A tile / B tile)
    v144..v147             LDS read addresses of the current stage: A k-step 0, A k-step 1, B k-step 0, B k-step 1
    s[60:61], s[62:63]     A / B source base of the next tile to stage (advance 128 B per K-tile)
    s64                    loop counter;  s65  M0 base of the stage the next DMA group writes
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "construction-clip_amd", "csrc", "gemm_a4_kloop.inc")

MFMA = '" CCLIP_MFMA_ASM "'          # spliced C macro: "v_mfma_f32_16x16x32_bf16" or "..._f16"


def acc(mt, nt):
    b = 4 * (8 * mt + nt)
    return f"a[{b}:{b + 3}]"


def afrag(s, mt):
    b = 64 * s + 4 * mt
    return f"v[{b}:{b + 3}]"


def bfrag(s, nt):
    b = 64 * s + 32 + 4 * nt
    return f"v[{b}:{b + 3}]"


def mfma_list(s, order="mt_outer"):
    """The 64 MFMAs of one phase on fragment set s.  Swapped operands: srcA = weight fragment (rows of D = n),
    srcB = activation fragment (columns of D = m)."""
    out = []
    if order == "mt_outer":
        for mt in range(8):
            for nt in range(8):
                out.append(f"{MFMA} {acc(mt, nt)}, {bfrag(s, nt)}, {afrag(s, mt)}, {acc(mt, nt)}")
    else:
        for nt in range(8):
            for mt in range(8):
                out.append(f"{MFMA} {acc(mt, nt)}, {bfrag(s, nt)}, {afrag(s, mt)}, {acc(mt, nt)}")
    return out


def reads(s, ks):
    """16 ds_read_b128 of k-step ks into set s: weight fragments first (the first MFMAs of an mt-outer phase need all 8)."""
    va, vb = (144, 146) if ks == 0 else (145, 147)
    out = []
    for nt in range(8):
        out.append(f"ds_read_b128 {bfrag(s, nt)}, v{vb} offset:{2048 * nt}")
    for mt in range(8):
        out.append(f"ds_read_b128 {afrag(s, mt)}, v{va} offset:{2048 * mt}")
    return out


def dma_pairs():
    """(M0 setup, DMA) pairs of one tile: 8 pieces of the A tile, 8 of the B tile (each wave: pieces w, w+4, ...)."""
    out = []
    for i in range(8):
        out.append((f"s_add_u32 m0, s65, {4096 * i}", f"global_load_lds_dwordx4 v{128 + i}, s[60:61]"))
    for i in range(8):
        out.append((f"s_add_u32 m0, s65, {32768 + 4096 * i}", f"global_load_lds_dwordx4 v{136 + i}, s[62:63]"))
    return out


ADVANCE = ["s_add_u32 s60, s60, 128", "s_addc_u32 s61, s61, 0", "s_add_u32 s62, s62, 128", "s_addc_u32 s63, s63, 0",
           "s_xor_b32 s65, s65, 0x10000"]
TOGGLE = [f"v_xor_b32 v{r}, 0x10000, v{r}" for r in (144, 145, 146, 147)]
# L2 prefetch stream (variant {"pf": 2}; measured in round 3 and NOT instantiated: -3 % on the K = 3072 projection, +2 % on the K = 768 ones -
# the activation rows a projection reads were written by the kernel before it and sit in the Infinity Cache): one dword per 128-byte line of the A tile's K-slab `pf` K-tiles ahead of the DMA stream - lane l of
# wave w touches row 64 w + l (offset v149, destination v148 never read).  s[72:73] = its source, advanced while s74 > 0.
PF_STEP = ["s_cmp_lg_u32 s74, 0", "s_cselect_b32 s75, 128, 0", "s_cselect_b32 s76, 1, 0", "s_add_u32 s72, s72, s75", "s_addc_u32 s73, s73, 0",
           "s_sub_u32 s74, s74, s76"]
PF_LOAD = "global_load_dword v148, v149, s[72:73]"


def phase(mf, fillers):
    """Interleave: fillers[g] = instructions issued right after MFMA g."""
    out = []
    for g, m in enumerate(mf):
        out.append(m)
        out.extend(fillers.get(g, []))
    return out


def phase_R(dma, variant):
    """64 MFMAs on set 1 + (optionally) the DMA group of the next tile + the reads of (kt, k-step 0) into set 0."""
    fill = {}
    rd = reads(0, 0)
    if dma:
        pairs = dma_pairs()
        if variant["dma_spread"] == 4:      # one DMA per 4 gaps over the whole phase; reads in the two free gaps of the first 8 groups
            for j, (m0, ld) in enumerate(pairs):
                fill.setdefault(4 * j, []).append(m0)
                fill.setdefault(4 * j + 1, []).append(ld)
            for r, ins in enumerate(rd):
                fill.setdefault(4 * (r // 2) + 2 + (r % 2), []).append(ins)
            fill.setdefault(62, []).extend(ADVANCE[:2])
            fill.setdefault(63, []).extend(ADVANCE[2:])
            if variant.get("pf"):           # L2 prefetch of the A rows of K-tile kt + 1 + pf: youngest operation of the iteration
                fill.setdefault(61, []).append(PF_LOAD)        # (after the last DMA of the group, which sits in this gap too)
                fill.setdefault(62, []).extend(PF_STEP[:3])
                fill.setdefault(63, []).extend(PF_STEP[3:])
        else:                               # dma_spread == 2: DMA group in the first half, reads in the second
            for j, (m0, ld) in enumerate(pairs):
                fill.setdefault(2 * j, []).append(m0)
                fill.setdefault(2 * j + 1, []).append(ld)
            for r, ins in enumerate(rd):
                fill.setdefault(32 + r, []).append(ins)
            fill.setdefault(33, []).extend(ADVANCE[:2])
            fill.setdefault(34, []).extend(ADVANCE[2:])
    else:
        for r, ins in enumerate(rd):
            fill.setdefault(2 * r + 1, []).append(ins)
    return phase(mfma_list(1, variant["order"]), fill)


def phase_Q(variant):
    fill = {}
    for r, ins in enumerate(reads(1, 1)):
        fill.setdefault(2 * r + 1, []).append(ins)
    for i, t in enumerate(TOGGLE):
        fill.setdefault(40 + 2 * i, []).append(t)
    return phase(mfma_list(0, variant["order"]), fill)


def kloop(variant):
    L = []
    pf = variant.get("pf", 0)
    wait = f"s_waitcnt vmcnt({1 if pf else 0})"
    L += ["s_mov_b64 s[60:61], %[abase]", "s_mov_b64 s[62:63], %[bbase]", "s_mov_b32 s64, %[niter]", "s_mov_b32 s65, %[m0base]"]
    if pf:
        # the prefetch stream starts at K-tile 2 (tiles 0 and 1 are staged right away) and runs up to the last K-tile: nkt - 3 advances
        L += ["s_mov_b64 s[72:73], %[abase]", "s_add_u32 s72, s72, 256", "s_addc_u32 s73, s73, 0", "s_sub_u32 s74, s64, 1", "s_max_i32 s74, s74, 0"]
    L.append("s_nop 4")
    # prologue: tile 0 -> stage 0
    for m0, ld in dma_pairs():
        L += [m0, "s_nop 0", ld]
    L += ADVANCE
    for i in range(256):
        L.append(f"v_accvgpr_write_b32 a{i}, 0")
    # kt = 0: no phase R
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    for m0, ld in dma_pairs():              # tile 1 -> stage 1 (nkt >= 2 is a launch condition)
        L += [m0, "s_nop 0", ld]
    L += ADVANCE
    if pf:
        for _ in range(pf):                 # K-tiles 2 .. 1 + pf: nothing else will touch their lines before their DMA
            L += [PF_LOAD] + PF_STEP
    L += reads(0, 0)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += phase_Q(variant)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += ["s_cmp_eq_u32 s64, 0", "s_cbranch_scc1 LAST_%="]
    L += ["LOOP_%=:"]
    L += [wait, "s_barrier"]
    L += phase_R(True, variant)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += phase_Q(variant)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += ["s_sub_u32 s64, s64, 1", "s_cmp_lg_u32 s64, 0", "s_cbranch_scc1 LOOP_%="]
    L += ["LAST_%=:"]
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    L += phase_R(False, variant)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += phase_Q(variant)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += mfma_list(1, variant["order"])
    L += ["s_nop 15", "s_nop 15"]
    return L


def clobbers():
    c = ['"memory"', '"vcc"', '"scc"']
    c += [f'"s{i}"' for i in range(60, 66)]
    c += [f'"v{i}"' for i in range(0, 128)]
    return c


def emit(variant, name):
    pf = variant.get("pf", 0)
    """The asm statement as a macro.  Operands pinned to physical registers: the 64 accumulators ACC[8 mt + nt] (f32x4 each; wider pinned tuples crash this compiler's copy lowering)
    (outputs: the compiler owns them afterwards and reads them in the epilogue), the DMA offsets v[128:143] (inputs) and the
    LDS read addresses v[144:147] (in/out: the loop toggles their stage bit)."""
    lines = kloop(variant)
    s = [f"// GENERATED by tools/gen_gemm_a4.py (variant {variant}) - do not edit; the K loop of gemm_a4_kernel as one asm statement.",
         f"#define {name}(ACC, ABASE, BBASE, NITER, M0BASE, OA03, OA47, OB03, OB47, LADDR{', PFOFF' if pf else ''}) \\", "  asm volatile( \\"]
    for ln in lines:
        s.append(f'    "{ln}\\n\\t" \\')
    outs = [f'"={{a[{4 * i}:{4 * i + 3}]}}"(ACC[{i}])' for i in range(64)] + ['"+{v[144:147]}"(LADDR)']
    s.append("    : " + ", ".join(outs) + " \\")
    ops = ['[abase] "s"(ABASE)', '[bbase] "s"(BBASE)', '[niter] "s"(NITER)', '[m0base] "s"(M0BASE)',
           '"{v[128:131]}"(OA03)', '"{v[132:135]}"(OA47)', '"{v[136:139]}"(OB03)', '"{v[140:143]}"(OB47)']
    cl = clobbers()
    if pf:
        ops.append('"{v149}"(PFOFF)')
        cl += ['"v148"'] + [f'"s{i}"' for i in range(72, 77)]
    s.append("    : " + ", ".join(ops) + " \\")
    s.append("    : " + ", ".join(cl) + ")")
    return "\n".join(s) + "\n"


# ======================================================================================================================
# Persistent form (tile configuration 10): the same BK = 64 rotated loop, but a work-group walks MANY tiles and its DMA stream
# does not stop at a tile boundary: the last iteration of a tile stages K-tiles 0 and 1 of the work-group's NEXT tile (into the
# two stages as they fall free), so the next tile finds its first operands in LDS, and the epilogue's global stores - younger than
# those DMAs in the in-order vmcnt queue - are NOT waited for at the next tile's start: COUNTED waits (all but the W youngest,
# W = the stores the epilogue is known to have issued) let them drain under the next tile's first two K iterations.
#   extra registers: v[148:163] the next tile's DMA offsets; s[66:67], s[68:69] its A / B source; s70 / s71 the wait selectors of
#   the tile's first two barriers (0: vmcnt(0), 1: vmcnt(32), 2: vmcnt(48), 3: vmcnt(63)).
def dma_pairs_next():
    out = []
    for i in range(8):
        out.append((f"s_add_u32 m0, s65, {4096 * i}", f"global_load_lds_dwordx4 v{148 + i}, s[66:67]"))
    for i in range(8):
        out.append((f"s_add_u32 m0, s65, {32768 + 4096 * i}", f"global_load_lds_dwordx4 v{156 + i}, s[68:69]"))
    return out


ADVANCE_NEXT = ["s_add_u32 s66, s66, 128", "s_addc_u32 s67, s67, 0", "s_add_u32 s68, s68, 128", "s_addc_u32 s69, s69, 0",
                "s_xor_b32 s65, s65, 0x10000"]


def mfma_zero(s, order):
    out = []
    rng = [(mt, nt) for mt in range(8) for nt in range(8)] if order == "mt_outer" else [(mt, nt) for nt in range(8) for mt in range(8)]
    for mt, nt in rng:
        out.append(f"{MFMA} {acc(mt, nt)}, {bfrag(s, nt)}, {afrag(s, mt)}, 0")
    return out


def p_phase_R(dma, variant, rd=True, set_=1):
    """64 MFMAs on set 1 + a DMA group (None / 'cur' / 'next') + (rd) the reads of (kt, k-step 0) into set 0."""
    fill = {}
    if dma:
        pairs = dma_pairs() if dma == "cur" else dma_pairs_next()
        adv = ADVANCE if dma == "cur" else ADVANCE_NEXT
        for j, (m0, ld) in enumerate(pairs):
            fill.setdefault(4 * j, []).append(m0)
            fill.setdefault(4 * j + 1, []).append(ld)
        fill.setdefault(62, []).extend(adv[:2])
        fill.setdefault(63, []).extend(adv[2:])
    if rd:
        for r, ins in enumerate(reads(0, 0)):
            fill.setdefault(4 * (r // 2) + 2 + (r % 2), []).append(ins)
    return phase(mfma_list(set_, variant["order"]), fill)


def p_phase_Q(variant, zero=False):
    fill = {}
    for r, ins in enumerate(reads(1, 1)):
        fill.setdefault(2 * r + 1, []).append(ins)
    for i, t in enumerate(TOGGLE):
        fill.setdefault(40 + 2 * i, []).append(t)
    mf = mfma_zero(0, variant["order"]) if zero else mfma_list(0, variant["order"])
    return phase(mf, fill)


def wait_sel(sreg, tag):
    return [f"s_cmp_eq_u32 {sreg}, 3", f"s_cbranch_scc1 W63{tag}_%=", f"s_cmp_eq_u32 {sreg}, 2", f"s_cbranch_scc1 W48{tag}_%=",
            f"s_cmp_eq_u32 {sreg}, 1", f"s_cbranch_scc1 W32{tag}_%=", "s_waitcnt vmcnt(0)", f"s_branch WD{tag}_%=",
            f"W63{tag}_%=:", "s_waitcnt vmcnt(63)", f"s_branch WD{tag}_%=", f"W48{tag}_%=:", "s_waitcnt vmcnt(48)", f"s_branch WD{tag}_%=",
            f"W32{tag}_%=:", "s_waitcnt vmcnt(32)", f"WD{tag}_%=:"]


def pers_first():
    """Work-group prologue: K-tiles 0 and 1 of its first tile."""
    L = ["s_mov_b64 s[60:61], %[abase]", "s_mov_b64 s[62:63], %[bbase]", "s_mov_b32 s65, %[m0st]", "s_nop 4"]
    for _ in range(2):
        for m0, ld in dma_pairs():
            L += [m0, "s_nop 0", ld]
        L += ADVANCE
    return L


def pers_tile(variant):
    """One tile; K-tiles 0 and 1 are staged (or on their way) on entry.  s64 = nkt - 2 >= 1 loop iterations (kt = 1 .. nkt-2)."""
    L = ["s_mov_b64 s[60:61], %[abase]", "s_mov_b64 s[62:63], %[bbase]", "s_mov_b64 s[66:67], %[nabase]", "s_mov_b64 s[68:69], %[nbbase]",
         "s_mov_b32 s64, %[niter]", "s_mov_b32 s65, %[m0st]", "s_mov_b32 s70, %[w0]", "s_mov_b32 s71, %[w1]"]
    L += wait_sel("s70", "A") + ["s_barrier"]
    L += reads(0, 0) + ["s_waitcnt lgkmcnt(0)"]
    L += p_phase_Q(variant, zero=True) + ["s_waitcnt lgkmcnt(0)"]
    # kt = 1: K-tile 1 was staged before the epilogue's stores -> counted wait
    L += wait_sel("s71", "B") + ["s_barrier"]
    L += p_phase_R("cur", variant) + ["s_waitcnt lgkmcnt(0)"]
    L += p_phase_Q(variant) + ["s_waitcnt lgkmcnt(0)"]
    L += ["s_sub_u32 s64, s64, 1", "s_cmp_eq_u32 s64, 0", "s_cbranch_scc1 PLAST_%="]
    L += ["PLOOP_%=:", "s_waitcnt vmcnt(0)", "s_barrier"]
    L += p_phase_R("cur", variant) + ["s_waitcnt lgkmcnt(0)"]
    L += p_phase_Q(variant) + ["s_waitcnt lgkmcnt(0)"]
    L += ["s_sub_u32 s64, s64, 1", "s_cmp_lg_u32 s64, 0", "s_cbranch_scc1 PLOOP_%="]
    L += ["PLAST_%=:", "s_waitcnt vmcnt(0)", "s_barrier"]
    L += p_phase_R("next", variant) + ["s_waitcnt lgkmcnt(0)"]          # K-tile 0 of the next tile -> the stage of kt - 1
    L += p_phase_Q(variant) + ["s_waitcnt lgkmcnt(0)", "s_barrier"]     # everybody is done reading the last K-tile's stage
    L += p_phase_R("next", variant, rd=False)                           # k-step 1 of the last K-tile + K-tile 1 of the next tile
    L += ["s_mov_b32 %[m0out], s65", "s_nop 15", "s_nop 15"]
    return L


def pers_clobbers():
    c = ['"memory"', '"vcc"', '"scc"']
    c += [f'"s{i}"' for i in range(60, 72)]
    c += [f'"v{i}"' for i in range(0, 128)]
    return c


def emit_pers(variant, name):
    s = [f"// GENERATED by tools/gen_gemm_a4.py (persistent variant {variant}) - do not edit.",
         f"#define {name}_FIRST(ABASE, BBASE, M0ST, OA03, OA47, OB03, OB47) \\", "  asm volatile( \\"]
    for ln in pers_first():
        s.append(f'    "{ln}\\n\\t" \\')
    s.append("    : \\")
    s.append('    : [abase] "s"(ABASE), [bbase] "s"(BBASE), [m0st] "s"(M0ST), "{v[128:131]}"(OA03), "{v[132:135]}"(OA47), "{v[136:139]}"(OB03), "{v[140:143]}"(OB47) \\')
    s.append('    : "memory", "scc", "s60", "s61", "s62", "s63", "s65")')
    s.append("")
    s += [f"#define {name}_TILE(ACC, ABASE, BBASE, NABASE, NBBASE, NITER, M0ST, W0, W1, OA03, OA47, OB03, OB47, NOA03, NOA47, NOB03, NOB47, LADDR) \\",
          "  asm volatile( \\"]
    for ln in pers_tile(variant):
        s.append(f'    "{ln}\\n\\t" \\')
    outs = [f'"={{a[{4 * i}:{4 * i + 3}]}}"(ACC[{i}])' for i in range(64)] + ['"+{v[144:147]}"(LADDR)', '[m0out] "=s"(M0ST)']
    s.append("    : " + ", ".join(outs) + " \\")
    ops = ['[abase] "s"(ABASE)', '[bbase] "s"(BBASE)', '[nabase] "s"(NABASE)', '[nbbase] "s"(NBBASE)', '[niter] "s"(NITER)', '[m0st] "s"(M0ST)',
           '[w0] "s"(W0)', '[w1] "s"(W1)',
           '"{v[128:131]}"(OA03)', '"{v[132:135]}"(OA47)', '"{v[136:139]}"(OB03)', '"{v[140:143]}"(OB47)',
           '"{v[148:151]}"(NOA03)', '"{v[152:155]}"(NOA47)', '"{v[156:159]}"(NOB03)', '"{v[160:163]}"(NOB47)']
    s.append("    : " + ", ".join(ops) + " \\")
    s.append("    : " + ", ".join(pers_clobbers()) + ")")
    return "\n".join(s) + "\n"


PERS_VARIANTS = {"GEMM_A4P": {"order": "mt_outer"}}

# ======================================================================================================================
# Weight-gradient form (tile configuration 11): both operands K-STRIDED (dW = dY^T X, the contraction runs over tokens): the same
# four 128x128 waves and rotated BK = 64 loop, fragments by ds_read_b64_tr_b16 (hardware transpose) from the [64 k][128 col]
# sub-tile image of the other configurations.
#   LDS: region(X, stage) at X * 32768 + stage * 16384, X = A sub-tile 0 / 1, B sub-tile 0 / 1 - the stage is an IMMEDIATE
#   offset, the loop is unrolled by two (one body per stage), no address toggling.
#   registers: v[144:207] LDS read addresses: v[144 + 32 op + 4 t + 2 ks + h] = transposing read (half h) of tile t, k-step ks;
#   v[208:239] the bias gradient's accumulators (row sums of dY^T by 8 extra MFMAs per k-step against an all-ones fragment,
#   v[240:243]; only in tiles of the first column block: s68); s66 / s67 A / B source stride per K-tile (64 rows).
def wg_reads(stage, set_, ks):
    """32 transposing reads of k-step ks of the tile in `stage` into fragment set set_ (weight fragments first)."""
    out = []
    for op in (1, 0):
        for t in range(8):
            for h in range(2):
                va = 144 + 32 * op + 4 * t + 2 * ks + h
                b = 64 * set_ + (32 if op else 0) + 4 * t + 2 * h
                # (the operand's region and the wave's sub-tile are folded into the address register: the offset field is 16 bits)
                out.append(f"ds_read_b64_tr_b16 v[{b}:{b + 1}], v{va} offset:{stage * 16384}")
    return out


def wg_dma(stage):
    out = []
    for op, (base, sreg) in enumerate(((128, "s[60:61]"), (136, "s[62:63]"))):
        for i in range(8):
            dst = (2 * op + (i >> 2)) * 32768 + stage * 16384 + (i & 3) * 4096
            out.append((f"s_add_u32 m0, s65, {dst}", f"global_load_lds_dwordx4 v{base + i}, {sreg}"))
    return out


WG_ADVANCE = ["s_add_u32 s60, s60, s66", "s_addc_u32 s61, s61, 0", "s_add_u32 s62, s62, s67", "s_addc_u32 s63, s63, 0"]


def wg_mfma(set_, cs):
    out = []
    for mt in range(8):
        for nt in range(8):
            out.append(f"{MFMA} {acc(mt, nt)}, {bfrag(set_, nt)}, {afrag(set_, mt)}, {acc(mt, nt)}")
        if cs:
            b = 208 + 4 * mt
            out.append(f"{MFMA} v[{b}:{b + 3}], v[240:243], {afrag(set_, mt)}, v[{b}:{b + 3}]")
    return out


def wg_phase(set_, cs, rd, dma):
    """MFMAs on fragment set set_ with the 32 reads `rd` and (optionally) the 16 DMA pairs `dma` dealt into the gaps."""
    mf = wg_mfma(set_, cs)
    n = len(mf)
    fill = {}
    if dma:
        for j, (m0, ld) in enumerate(dma):
            fill.setdefault(4 * j, []).append(m0)
            fill.setdefault(4 * j + 1, []).append(ld)
        fill.setdefault(n - 2, []).extend(WG_ADVANCE[:2])
        fill.setdefault(n - 1, []).extend(WG_ADVANCE[2:])
    for r, ins in enumerate(rd or []):
        fill.setdefault(min(2 + (r * (n - 8)) // 32, n - 3), []).append(ins)
    return phase(mf, fill)


def wg_iter(stage, cs, dma):
    """One K iteration on the tile in `stage`: k-step 1 of the previous tile (+ DMA of the next tile into the other stage + reads of
    k-step 0), then k-step 0 (+ reads of k-step 1)."""
    L = ["s_waitcnt vmcnt(0)", "s_barrier"]
    L += wg_phase(1, cs, wg_reads(stage, 0, 0), wg_dma(stage ^ 1) if dma else None) + ["s_waitcnt lgkmcnt(0)"]
    L += wg_phase(0, cs, wg_reads(stage, 1, 1), None) + ["s_waitcnt lgkmcnt(0)"]
    return L


def wg_body(cs, tag):
    L = []
    # kt = 0 (stage 0): no previous tile
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    for m0, ld in wg_dma(1):
        L += [m0, "s_nop 0", ld]
    L += WG_ADVANCE
    L += wg_reads(0, 0, 0) + ["s_waitcnt lgkmcnt(0)"]
    L += wg_phase(0, cs, wg_reads(0, 1, 1), None) + ["s_waitcnt lgkmcnt(0)"]
    L += [f"WLOOP{tag}_%=:", "s_cmp_eq_u32 s64, 0", f"s_cbranch_scc1 WLAST1{tag}_%="]
    L += wg_iter(1, cs, True) + ["s_sub_u32 s64, s64, 1", "s_cmp_eq_u32 s64, 0", f"s_cbranch_scc1 WLAST0{tag}_%="]
    L += wg_iter(0, cs, True) + ["s_sub_u32 s64, s64, 1", f"s_branch WLOOP{tag}_%="]
    L += [f"WLAST1{tag}_%=:"] + wg_iter(1, cs, False) + [f"s_branch WTAIL{tag}_%="]
    L += [f"WLAST0{tag}_%=:"] + wg_iter(0, cs, False)
    L += [f"WTAIL{tag}_%=:"] + wg_mfma(1, cs)
    return L


def wg_kloop():
    L = ["s_mov_b64 s[60:61], %[abase]", "s_mov_b64 s[62:63], %[bbase]", "s_mov_b32 s64, %[niter]", "s_mov_b32 s65, %[m0base]",
         "s_mov_b32 s66, %[astride]", "s_mov_b32 s67, %[bstride]", "s_mov_b32 s68, %[cs]", "s_nop 4"]
    for m0, ld in wg_dma(0):
        L += [m0, "s_nop 0", ld]
    L += WG_ADVANCE
    for i in range(256):
        L.append(f"v_accvgpr_write_b32 a{i}, 0")
    for i in range(208, 240):
        L.append(f"v_mov_b32 v{i}, 0")
    L += ["s_cmp_eq_u32 s68, 0", "s_cbranch_scc1 WPLAIN_%="]
    L += wg_body(True, "C") + ["s_branch WEND_%="]
    L += ["WPLAIN_%=:"] + wg_body(False, "P")
    L += ["WEND_%=:", "s_nop 15", "s_nop 15"]
    return L


def emit_wg(name):
    s = ["// GENERATED by tools/gen_gemm_a4.py (weight-gradient form) - do not edit.",
         f"#define {name}(ACC, ACCB, ABASE, BBASE, NITER, M0BASE, ASTRIDE, BSTRIDE, CS, OA03, OA47, OB03, OB47, LA, ONES) \\", "  asm volatile( \\"]
    for ln in wg_kloop():
        s.append(f'    "{ln}\\n\\t" \\')
    outs = [f'"={{a[{4 * i}:{4 * i + 3}]}}"(ACC[{i}])' for i in range(64)] + [f'"={{v[{208 + 4 * i}:{211 + 4 * i}]}}"(ACCB[{i}])' for i in range(8)]
    s.append("    : " + ", ".join(outs) + " \\")
    ops = ['[abase] "s"(ABASE)', '[bbase] "s"(BBASE)', '[niter] "s"(NITER)', '[m0base] "s"(M0BASE)', '[astride] "s"(ASTRIDE)',
           '[bstride] "s"(BSTRIDE)', '[cs] "s"(CS)',
           '"{v[128:131]}"(OA03)', '"{v[132:135]}"(OA47)', '"{v[136:139]}"(OB03)', '"{v[140:143]}"(OB47)']
    ops += [f'"{{v[{144 + 4 * i}:{147 + 4 * i}]}}"(LA[{i}])' for i in range(16)] + ['"{v[240:243]}"(ONES)']
    s.append("    : " + ", ".join(ops) + " \\")
    cl = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(60, 69)] + [f'"v{i}"' for i in range(0, 128)]
    s.append("    : " + ", ".join(cl) + ")")
    return "\n".join(s) + "\n"


VARIANTS = {
    "GEMM_A4R": {"order": "mt_outer", "read_gap0": 2, "read_stride": 2, "dma_gap0": 4, "dma_stride": 7},
}

VARIANTS = {
    "GEMM_A4_KLOOP": {"order": "mt_outer", "dma_spread": 4},
}

if __name__ == "__main__":
    txt = "".join(emit(v, n) + "\n" for n, v in VARIANTS.items()) + "".join(emit_pers(v, n) + "\n" for n, v in PERS_VARIANTS.items())
    open(OUT, "w").write(txt)
    print(f"wrote {OUT}: {len(txt.splitlines())} lines")
    wout = OUT.replace("gemm_a4_kloop.inc", "gemm_a4w_kloop.inc")
    wtxt = emit_wg("GEMM_A4W_KLOOP")
    open(wout, "w").write(wtxt)
    print(f"wrote {wout}: {len(wtxt.splitlines())} lines")
