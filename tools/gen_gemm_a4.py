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
    L += ["s_mov_b64 s[60:61], %[abase]", "s_mov_b64 s[62:63], %[bbase]", "s_mov_b32 s64, %[niter]", "s_mov_b32 s65, %[m0base]"]
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
    L += reads(0, 0)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += phase_Q(variant)
    L += ["s_waitcnt lgkmcnt(0)"]
    L += ["s_cmp_eq_u32 s64, 0", "s_cbranch_scc1 LAST_%="]
    L += ["LOOP_%=:"]
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
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
    """The asm statement as a macro.  Operands pinned to physical registers: the 64 accumulators ACC[8 mt + nt] (f32x4 each; wider pinned tuples crash this compiler's copy lowering)
    (outputs: the compiler owns them afterwards and reads them in the epilogue), the DMA offsets v[128:143] (inputs) and the
    LDS read addresses v[144:147] (in/out: the loop toggles their stage bit)."""
    lines = kloop(variant)
    s = [f"// GENERATED by tools/gen_gemm_a4.py (variant {variant}) - do not edit; the K loop of gemm_a4_kernel as one asm statement.",
         f"#define {name}(ACC, ABASE, BBASE, NITER, M0BASE, OA03, OA47, OB03, OB47, LADDR) \\", "  asm volatile( \\"]
    for ln in lines:
        s.append(f'    "{ln}\\n\\t" \\')
    outs = [f'"={{a[{4 * i}:{4 * i + 3}]}}"(ACC[{i}])' for i in range(64)] + ['"+{v[144:147]}"(LADDR)']
    s.append("    : " + ", ".join(outs) + " \\")
    ops = ['[abase] "s"(ABASE)', '[bbase] "s"(BBASE)', '[niter] "s"(NITER)', '[m0base] "s"(M0BASE)',
           '"{v[128:131]}"(OA03)', '"{v[132:135]}"(OA47)', '"{v[136:139]}"(OB03)', '"{v[140:143]}"(OB47)']
    s.append("    : " + ", ".join(ops) + " \\")
    s.append("    : " + ", ".join(clobbers()) + ")")
    return "\n".join(s) + "\n"


# ======================================================================================================================
# Persistent ring kernel (tile configuration 10): BK = 32 steps through a ring of four 32 KiB LDS slots, the DMA stream
# runs FOUR steps ahead of the MFMAs and does not stop at tile boundaries (the last four steps of a tile stage the first four
# steps of the work-group's next tile), so neither the fill latency nor the next tile's first operands are exposed.
#   LDS slot (32 KiB) = [A: 256 rows x 64 B][B: 256 rows x 64 B]; 16-byte chunk g (k = 8g..8g+7) of row r sits at
#   r*64 + ((g ^ ((-(r >> 2)) & 3)) << 4): conflict-free for the ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS).
#   step j: s_waitcnt vmcnt(16); s_barrier  -> step j+1's slot is visible, everybody is done reading slot j
#           64 MFMAs on fragment set j&1  +  16 reads of slot j+1 -> set (j+1)&1  +  8 DMA pieces of step j+4 -> slot j
# Register map: v[0:63] / v[64:127] fragment sets; v[128:131] / v[132:135] A / B DMA offsets of the current tile,
# v[136:143] the same for the next tile; v144/v145 A read base (slots 0-1 / slots 2-3), v146/v147 the B ones;
# s[60:61], s[62:63] A / B source of the current tile's next step to stage, s[66:67], s[68:69] the next tile's; s64 loop
# counter; s65 M0 base (LDS base + wave * 1024).
SLOT = 32768


def r_afrag(s, mt):
    return afrag(s, mt)


def r_reads(slot, s):
    """16 ds_read_b128 of ring slot `slot` into fragment set s."""
    va, vb = (144, 146) if slot < 2 else (145, 147)
    so = (slot & 1) * SLOT
    out = []
    for nt in range(8):
        out.append(f"ds_read_b128 {bfrag(s, nt)}, v{vb} offset:{so + 16384 + 1024 * nt}")
    for mt in range(8):
        out.append(f"ds_read_b128 {afrag(s, mt)}, v{va} offset:{so + 1024 * mt}")
    return out


def r_dma(slot, nxt):
    """8 (M0 setup, DMA) pairs of one step into ring slot `slot`: 4 pieces of the A half-tile, 4 of the B half-tile."""
    ob = 136 if nxt else 128
    sa, sb = ("s[66:67]", "s[68:69]") if nxt else ("s[60:61]", "s[62:63]")
    out = []
    for i in range(4):
        out.append((f"s_add_u32 m0, s65, {slot * SLOT + 4096 * i}", f"global_load_lds_dwordx4 v{ob + i}, {sa}"))
    for i in range(4):
        out.append((f"s_add_u32 m0, s65, {slot * SLOT + 16384 + 4096 * i}", f"global_load_lds_dwordx4 v{ob + 4 + i}, {sb}"))
    return out


def r_advance(nxt):
    if nxt:
        return ["s_add_u32 s66, s66, 64", "s_addc_u32 s67, s67, 0", "s_add_u32 s68, s68, 64", "s_addc_u32 s69, s69, 0"]
    return ["s_add_u32 s60, s60, 64", "s_addc_u32 s61, s61, 0", "s_add_u32 s62, s62, 64", "s_addc_u32 s63, s63, 0"]


def r_mfma(s, zero, order):
    out = []
    rng = [(mt, nt) for mt in range(8) for nt in range(8)] if order == "mt_outer" else [(mt, nt) for nt in range(8) for mt in range(8)]
    for mt, nt in rng:
        c = "0" if zero else acc(mt, nt)
        out.append(f"{MFMA} {acc(mt, nt)}, {bfrag(s, nt)}, {afrag(s, mt)}, {c}")
    return out


def r_step(slot, zero, nxt, variant, read_next=True, first=False):
    """One ring step on slot `slot`: wait + barrier, 64 MFMAs on set slot&1 with the reads of the following slot and the DMA
    group of the step four ahead dealt into the gaps."""
    L = []
    if not first:
        L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    fill = {}
    if read_next:
        rd = r_reads((slot + 1) & 3, (slot + 1) & 1)
        for r, ins in enumerate(rd):
            fill.setdefault(variant["read_gap0"] + variant["read_stride"] * r, []).append(ins)
    pairs = r_dma(slot, nxt)
    for j, (m0, ld) in enumerate(pairs):
        g0 = variant["dma_gap0"] + variant["dma_stride"] * j
        fill.setdefault(g0, []).append(m0)
        fill.setdefault(g0 + 1, []).append(ld)
    adv = r_advance(nxt)
    fill.setdefault(62, []).extend(adv[:2])
    fill.setdefault(63, []).extend(adv[2:])
    L += phase(r_mfma(slot & 1, zero, variant["order"]), fill)
    L += ["s_waitcnt lgkmcnt(0)"]
    return L


def ring_first():
    """Work-group prologue: the DMA groups of steps 0..3 of its first tile (from the 'current tile' registers)."""
    L = ["s_mov_b64 s[60:61], %[abase]", "s_mov_b64 s[62:63], %[bbase]", "s_mov_b32 s65, %[m0base]", "s_nop 4"]
    for slot in range(4):
        for m0, ld in r_dma(slot, False):
            L += [m0, "s_nop 0", ld]
        L += r_advance(False)
    return L


def ring_tile(variant):
    """One tile: K/32 = 4 (s64 + 2) steps.  Steps 0..3 of this tile are in flight / landed on entry."""
    L = ["s_mov_b64 s[60:61], %[abase]", "s_mov_b64 s[62:63], %[bbase]", "s_mov_b64 s[66:67], %[nabase]", "s_mov_b64 s[68:69], %[nbbase]",
         "s_mov_b32 s64, %[niter]", "s_mov_b32 s65, %[m0base]"]
    L += ["s_waitcnt vmcnt(0)", "s_barrier"]
    L += r_reads(0, 0)
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    # first group of four steps (accumulators start from zero), DMA from the current tile
    L += r_step(0, True, False, variant, first=True)
    for u in (1, 2, 3):
        L += r_step(u, False, False, variant)
    L += ["s_cmp_eq_u32 s64, 0", "s_cbranch_scc1 RLAST_%="]
    L += ["RLOOP_%=:"]
    for u in range(4):
        L += r_step(u, False, False, variant)
    L += ["s_sub_u32 s64, s64, 1", "s_cmp_lg_u32 s64, 0", "s_cbranch_scc1 RLOOP_%="]
    L += ["RLAST_%=:"]
    for u in range(4):           # last group: the DMA stream is already in the next tile
        L += r_step(u, False, True, variant, read_next=(u != 3))
    L += ["s_nop 15", "s_nop 15"]
    return L


def ring_clobbers():
    c = ['"memory"', '"vcc"', '"scc"']
    c += [f'"s{i}"' for i in range(60, 70)]
    c += [f'"v{i}"' for i in range(0, 128)]
    return c


def emit_ring(variant, name):
    s = [f"// GENERATED by tools/gen_gemm_a4.py (ring variant {variant}) - do not edit.",
         f"#define {name}_FIRST(ABASE, BBASE, M0BASE, OA, OB) \\", "  asm volatile( \\"]
    for ln in ring_first():
        s.append(f'    "{ln}\\n\\t" \\')
    s.append("    : \\")
    s.append('    : [abase] "s"(ABASE), [bbase] "s"(BBASE), [m0base] "s"(M0BASE), "{v[128:131]}"(OA), "{v[132:135]}"(OB) \\')
    s.append('    : "memory", "scc", "s60", "s61", "s62", "s63", "s65")')
    s.append("")
    s += [f"#define {name}_TILE(ACC, ABASE, BBASE, NABASE, NBBASE, NITER, M0BASE, OA, OB, NOA, NOB, LADDR) \\", "  asm volatile( \\"]
    for ln in ring_tile(variant):
        s.append(f'    "{ln}\\n\\t" \\')
    outs = [f'"={{a[{4 * i}:{4 * i + 3}]}}"(ACC[{i}])' for i in range(64)]
    s.append("    : " + ", ".join(outs) + " \\")
    ops = ['[abase] "s"(ABASE)', '[bbase] "s"(BBASE)', '[nabase] "s"(NABASE)', '[nbbase] "s"(NBBASE)', '[niter] "s"(NITER)', '[m0base] "s"(M0BASE)',
           '"{v[128:131]}"(OA)', '"{v[132:135]}"(OB)', '"{v[136:139]}"(NOA)', '"{v[140:143]}"(NOB)', '"{v[144:147]}"(LADDR)']
    s.append("    : " + ", ".join(ops) + " \\")
    s.append("    : " + ", ".join(ring_clobbers()) + ")")
    return "\n".join(s) + "\n"


RING_VARIANTS = {
    "GEMM_A4R": {"order": "mt_outer", "read_gap0": 2, "read_stride": 2, "dma_gap0": 4, "dma_stride": 7},
}

VARIANTS = {
    "GEMM_A4_KLOOP": {"order": "mt_outer", "dma_spread": 4},
    "GEMM_A4_KLOOP_V2": {"order": "mt_outer", "dma_spread": 2},
}

if __name__ == "__main__":
    txt = "".join(emit(v, n) + "\n" for n, v in VARIANTS.items()) + "".join(emit_ring(v, n) + "\n" for n, v in RING_VARIANTS.items())
    open(OUT, "w").write(txt)
    print(f"wrote {OUT}: {len(txt.splitlines())} lines")
