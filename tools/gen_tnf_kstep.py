#!/usr/bin/env python3
"""Generates nys_koop_lqr_amd/csrc/nk_tnf_kstep.inc: the whole k loop of the fp32 TN engine's Gram launches (nk_gemm_tn_f32.hip)
as ONE inline-assembly block.  Per k-step (32 contraction rows) and wave: 64 v_mfma_f32_32x32x2_f32, and in the gaps between
them the ds_read_b32 operand fetches of the next k-pair, the 8 LDS-DMA copies of the next step with their scalar address
updates, the step's barrier -- and the FLUSH of the fp32 accumulators into the fp64 shadow accumulators (v_cvt_f64_f32 +
v_add_f64 per element, 128 vector instructions per step, two per gap).

A step runs in two halves: the upper two 32 x 32 blocks of the wave's 64 x 64 sub-tile over all 16 k-pairs (two accumulator
chains alternating), then the lower two.  While one half accumulates -- starting from srcC = 0 -- the vector ALU empties the
accumulators of the other half into the shadows, so a flush has 32 matrix instructions of cover and never delays one.
(Tried first: accumulators double buffered in the accumulation registers a[0:127] and emptied over a whole step.  The loop
alone reaches 0.91 of the fp32 matrix peak, but a v_accvgpr_read_b32 beside running matrix instructions costs ~15 cycles of
the matrix pipe each: 0.75 with the 64 reads of a step alone, 0.68 with conversions and additions.)

Registers: accumulators v[0:63] (block t = 2 i + j at v[16 t ...]), fragments and temporaries v[100:115] (clobbers), shadows
v[128:255] (operands with fixed registers: inline assembly cannot name one register of a tuple operand).

    python3 tools/gen_tnf_kstep.py > nys_koop_lqr_amd/csrc/nk_tnf_kstep.inc
"""
ROW_B = 128 * 4            # bytes per LDS row (128 floats)
KK_B = 2 * ROW_B           # one k-pair = two rows
PANEL_B = 32 * ROW_B       # B panel behind the A panel inside a stage
STAGE_B = 2 * PANEL_B
QSTEP_B = 8 * ROW_B        # LDS distance between the row pairs a wave moves (rp = wave + 4 q)

SETS = [(100, 101, 102), (103, 104, 105)]   # fragment sets: (a, b0, b1)
TMP = [108, 110, 112, 114]                  # four fp64 temporaries (register pairs)
SH0 = 128                                   # shadows: block t at v[128 + 32 t ...]


def acc(t):
    return f"v[{16 * t}:{16 * t + 15}]"


def mfma(i, j, s, first):
    a, b0, b1 = SETS[s]
    t = 2 * i + j
    return f"v_mfma_f32_32x32x2_f32 {acc(t)}, v{a}, v{(b0, b1)[j]}, {'0' if first else acc(t)}"


def reads(stage, kk, i, s):
    a, b0, b1 = SETS[s]
    off = stage * STAGE_B + kk * KK_B
    return [f"ds_read_b32 v{a}, %[ard] offset:{off + 128 * i}", f"ds_read_b32 v{b0}, %[brd] offset:{off}",
            f"ds_read_b32 v{b1}, %[brd] offset:{off + 128}"]


def dma(stage):
    """the 8 LDS-DMA copies of the next step into `stage`; s[92:93] / s[94:95] run over this wave's row pairs"""
    seq = ["s_mov_b32 m0, %[dst0]" if stage == 0 else f"s_add_u32 m0, %[dst0], {STAGE_B}"]
    for q in range(4):
        seq += ["s_nop 0", "global_load_lds_dwordx4 %[voa], s[92:93]", f"s_add_u32 m0, m0, {PANEL_B}",
                "s_add_u32 s92, s92, %[stra]", "s_addc_u32 s93, s93, 0", "global_load_lds_dwordx4 %[vob], s[94:95]"]
        if q < 3:
            seq += [f"s_sub_u32 m0, m0, {PANEL_B - QSTEP_B}"]
        seq += ["s_add_u32 s94, s94, %[strb]", "s_addc_u32 s95, s95, 0"]
    return seq


def flush(t):
    """shadow += (double) accumulator for the 16 values of block t, four elements in flight: 32 instructions"""
    import os
    mode = os.environ.get("FLUSHMODE", "full")  # timing experiments only: "none", "cvt" (no additions)
    out = []
    for g in range(0, 16, 4):
        for u in range(4):
            if mode != "none":
                out.append(f"v_cvt_f64_f32 v[{TMP[u]}:{TMP[u] + 1}], v{16 * t + g + u}")
        for u in range(4):
            r = SH0 + 32 * t + 2 * (g + u)
            if mode == "full":
                out.append(f"v_add_f64 v[{r}:{r + 1}], v[{r}:{r + 1}], v[{TMP[u]}:{TMP[u] + 1}]")
    return out


def main():
    lines = []

    def half(st, i, last_step):
        """16 k-pairs of block row i on LDS stage st (32 matrix instructions); empties the two blocks of the OTHER row.  The
        fragments of its k-pair 0 are in set 0 on entry; on exit set 0 holds k-pair 0 of what follows: row 1 of this stage
        (after i = 0), row 0 of the next stage (after i = 1; not after the last step)."""
        other = 1 - i
        fl = flush(2 * other) + flush(2 * other + 1)      # 64 instructions over 32 gaps
        d = dma(1 - st) if (i == 0 and not last_step) else []
        gaps = [fl[2 * g:2 * g + 2] for g in range(32)]
        if d:
            per = (len(d) + 11) // 12
            for n in range(12):       # k-pairs 2 .. 7 of the upper half
                gaps[4 + n] = gaps[4 + n] + d[n * per:(n + 1) * per]
        for kk in range(16):
            s = kk & 1
            lines.append("s_waitcnt lgkmcnt(0)")
            if kk < 15:
                nxt = reads(st, kk + 1, i, 1 - s)
            elif i == 0:
                nxt = reads(st, 0, 1, 0)
            else:
                nxt = [] if last_step else reads(1 - st, 0, 0, 0)
            lines.append(mfma(i, 0, s, kk == 0))
            if kk == 15 and i == 1 and not last_step:
                # the step's barrier behind its 63rd matrix instruction: this wave has every fragment of the stage in registers,
                # its DMA of the next stage was issued 50 instructions ago; then the next stage's first fragments
                lines.extend(["s_waitcnt vmcnt(0)", "s_barrier"])
            lines.extend(nxt)
            lines.extend(gaps[2 * kk])
            lines.append(mfma(i, 1, s, kk == 0))
            lines.extend(gaps[2 * kk + 1])

    def step(st, last_step):
        half(st, 0, last_step)
        half(st, 1, last_step)

    def emit(name, body):
        nonlocal lines
        lines = []
        body()
        print(f"#define {name} \\")
        for ln in lines:
            print(f'  "{ln}\\n\\t" \\')
        print('  ""')

    def whole():
        lines.extend(["s_mov_b64 s[92:93], %[rowa]", "s_mov_b64 s[94:95], %[rowb]"])
        for r in range(32, 64):           # the first half step empties the lower blocks: make them zeros
            lines.append(f"v_mov_b32 v{r}, 0")
        lines.extend(reads(0, 0, 0, 0))   # k-pair 0 of block row 0, stage 0 (filled and fenced by the caller)
        lines.extend(["s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 nk_tnf_after_%=", "nk_tnf_loop_%=:"])
        step(0, False)
        step(1, False)
        lines.extend(["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 nk_tnf_loop_%=",
                      "nk_tnf_after_%=:", "s_bitcmp1_b32 %[flags], 0", "s_cbranch_scc0 nk_tnf_even_%="])
        step(0, False)
        step(1, True)
        lines.extend(["s_branch nk_tnf_end_%=", "nk_tnf_even_%=:"])
        step(0, True)
        lines.append("nk_tnf_end_%=:")
        lines.extend(["s_nop 7", "s_nop 7", "s_nop 7"])   # the last matrix instructions have written their blocks
        lines.extend(flush(2) + flush(3))              # (the upper blocks were emptied during the last lower half)

    print("// GENERATED by tools/gen_tnf_kstep.py -- do not edit by hand.")
    print("// The k loop of the fp32 Gram launches: cnt (+s) trips of two steady k-steps (LDS stage 0, then 1), one more steady step on")
    print("// stage 0 if bit 0 of flags (s) is set, then the final step of the K range (no DMA, no barrier) and the flush of its lower")
    print("// blocks.  Operands: sh00 sh01 sh10 sh11 (+{v[128:159]} .. +{v[224:255]}: the fp64 shadows); ard / brd (v: LDS byte")
    print("// address of this lane's fragment base in stage 0, A / B panel); voa / vob (v: per-lane byte offsets of the DMA);")
    print("// rowa / rowb (s, 64 bit: this wave's first row pair of the step after the first); stra / strb (s: 8 rows in bytes);")
    print("// dst0 (s: LDS byte address of this wave's first DMA row pair in stage 0).")
    print("// Clobbers v[0:63], v[100:115], s[92:95], m0, scc, memory.")
    emit("NK_TNF_KLOOP_ASM", whole)
    cl = ['"memory"', '"m0"', '"scc"', '"s92"', '"s93"', '"s94"', '"s95"'] + [f'"v{r}"' for r in range(64)] + \
         [f'"v{r}"' for r in range(100, 116)]
    print("#define NK_TNF_CLOBBERS " + ", ".join(cl))


if __name__ == "__main__":
    main()
