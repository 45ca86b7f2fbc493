#!/usr/bin/env python3
"""Generates nys_koop_lqr_amd/csrc/nk_tnf_kstep.inc: the whole k loop of the fp32 TN engine's Gram launches (nk_gemm_tn_f32.hip)
as ONE inline-assembly block.  Per k-step (32 contraction rows) and wave: 64 v_mfma_f32_32x32x2_f32 -- four accumulator chains
(the four 32 x 32 blocks of the wave's 64 x 64 sub-tile) alternating over 16 k-pairs -- and between them the 64 ds_read_b32
operand fetches of the next k-pair, the 8 LDS-DMA copies of the next step with their scalar address updates, the step's
barrier, and the FLUSH of the fp32 accumulators into the fp64 shadow accumulators (v_cvt_f64_f32 + v_add_f64 per element).
A block is emptied where its chain pauses: the upper two behind the last two matrix instructions of a step (their chains
restart from srcC = 0 with the first two of the next step), the lower two behind those first two.

What was measured on the way (C5, Gram launches, fraction of the fp32 matrix peak; the compiler-scheduled loop: 0.69):
  * the loop without any flush: 0.91 -- the flush is what there is to lose;
  * accumulators double buffered in the accumulation registers a[0:127], emptied over the whole next step: 0.68 -- a
    v_accvgpr_read_b32 beside running matrix instructions costs ~15 cycles of the matrix pipe (the 64 reads of a step
    alone: 0.75);
  * accumulators in v[0:63], a step in two halves (upper blocks over all k-pairs, then the lower ones, each half emptied
    while the other accumulates): 0.76 (0.87 without flush: two chains instead of four, 96 fragment reads instead of 64);
  * this schedule: 0.76.  Wherever they stand, the 128 vector instructions of a flush cost ~3.5 cycles of the matrix pipe
    each (64 conversions: -6 points, 64 additions: -5): instruction issue, not latency.  FLUSHMODE = none | cvt regenerates
    the block without them (timing experiments only: wrong results).

Registers: accumulators v[0:63] (block t = 2 i + j at v[16 t ...]), fragments and temporaries v[100:115] (clobbers), shadows
v[128:255] (operands with fixed registers: inline assembly cannot name one register of a tuple operand).

    python3 tools/gen_tnf_kstep.py > nys_koop_lqr_amd/csrc/nk_tnf_kstep.inc
"""
ROW_B = 128 * 4            # bytes per LDS row (128 floats)
KK_B = 2 * ROW_B           # one k-pair = two rows
PANEL_B = 32 * ROW_B       # B panel behind the A panel inside a stage
STAGE_B = 2 * PANEL_B
QSTEP_B = 8 * ROW_B        # LDS distance between the row pairs a wave moves (rp = wave + 4 q)

SETS = [((100, 101), (102, 103)), ((104, 105), (106, 107))]   # fragment sets: ((a0, a1), (b0, b1))
TMP = [108, 110, 112, 114]                                     # four fp64 temporaries (register pairs)
SH0 = 128                                                      # shadows: block t at v[128 + 32 t ...]


def acc(t):
    return f"v[{16 * t}:{16 * t + 15}]"


def mfma(i, j, s, first):  # first: start the chain from srcC = 0
    a, b = SETS[s]
    t = 2 * i + j
    return f"v_mfma_f32_32x32x2_f32 {acc(t)}, v{a[i]}, v{b[j]}, {'0' if first else acc(t)}"


def reads(stage, kk, s):
    a, b = SETS[s]
    off = stage * STAGE_B + kk * KK_B
    return [f"ds_read_b32 v{a[0]}, %[ard] offset:{off}", f"ds_read_b32 v{a[1]}, %[ard] offset:{off + 128}",
            f"ds_read_b32 v{b[0]}, %[brd] offset:{off}", f"ds_read_b32 v{b[1]}, %[brd] offset:{off + 128}"]


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

    def step(st, last_step, plain=False):
        """one k-step on LDS stage st: four accumulator chains (the four 32 x 32 blocks of the wave's sub-tile) alternating
        over the 16 k-pairs.  The blocks are emptied into the shadows where their chain pauses: the lower two (of the
        PREVIOUS step) behind the first two matrix instructions -- which restart the upper two from srcC = 0 --, the upper
        two behind the last two."""
        d = [] if last_step else dma(1 - st)
        gaps = [[] for _ in range(64)]
        if not plain:
            gaps[0] = flush(2)
            gaps[1] = flush(3)
            gaps[62] = flush(0)
            gaps[63] = flush(1)
        if d:
            per = (len(d) + 23) // 24
            for n in range(24):       # k-pairs 2 .. 7
                gaps[8 + n] = gaps[8 + n] + d[n * per:(n + 1) * per]
        for kk in range(16):
            s = kk & 1
            lines.append("s_waitcnt lgkmcnt(0)")
            nxt = reads(st, kk + 1, 1 - s) if kk < 15 else ([] if last_step else reads(1 - st, 0, 0))
            for t, (i, j) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
                lines.append(mfma(i, j, s, kk == 0 and not plain))
                if kk == 15 and not last_step:
                    # the step's barrier behind its 61st matrix instruction (this wave has every fragment of the stage in
                    # registers, its DMA of the next stage was issued 40 instructions ago), then the next stage's first fragments
                    if t == 0:
                        lines.extend(["s_waitcnt vmcnt(0)", "s_barrier"])
                        lines.extend(nxt[0:2])
                    elif t == 1:
                        lines.extend(nxt[2:4])
                elif t < len(nxt):
                    lines.append(nxt[t])
                lines.extend(gaps[4 * kk + t])

    def emit(name, body):
        nonlocal lines
        lines = []
        body()
        print(f"#define {name} \\")
        for ln in lines:
            print(f'  "{ln}\\n\\t" \\')
        print('  ""')

    def whole(plain=False):
        lines.extend(["s_mov_b64 s[92:93], %[rowa]", "s_mov_b64 s[94:95], %[rowb]"])
        if not plain:
            for r in range(32, 64):       # the first step empties the lower blocks first: make them zeros
                lines.append(f"v_mov_b32 v{r}, 0")
        lines.extend(reads(0, 0, 0))      # k-pair 0 of stage 0 (filled and fenced by the caller)
        lines.extend(["s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 nk_tnf_after_%=", "nk_tnf_loop_%=:"])
        step(0, False, plain)
        step(1, False, plain)
        lines.extend(["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 nk_tnf_loop_%=",
                      "nk_tnf_after_%=:", "s_bitcmp1_b32 %[flags], 0", "s_cbranch_scc0 nk_tnf_even_%="])
        step(0, False, plain)
        step(1, True, plain)
        lines.extend(["s_branch nk_tnf_end_%=", "nk_tnf_even_%=:"])
        step(0, True, plain)
        lines.append("nk_tnf_end_%=:")
        lines.extend(["s_nop 7", "s_nop 7", "s_nop 7"])   # the last matrix instructions have written their blocks
        if not plain:
            lines.extend(flush(2) + flush(3))          # (the upper blocks were emptied behind the last two matrix instructions)

    print("// GENERATED by tools/gen_tnf_kstep.py -- do not edit by hand.")
    print("// The k loop of the fp32 Gram launches: cnt (+s) trips of two steady k-steps (LDS stage 0, then 1), one more steady step on")
    print("// stage 0 if bit 0 of flags (s) is set, then the final step of the K range (no DMA, no barrier) and the flush of its lower")
    print("// blocks.  Operands: sh00 sh01 sh10 sh11 (+{v[128:159]} .. +{v[224:255]}: the fp64 shadows); ard / brd (v: LDS byte")
    print("// address of this lane's fragment base in stage 0, A / B panel); voa / vob (v: per-lane byte offsets of the DMA);")
    print("// rowa / rowb (s, 64 bit: this wave's first row pair of the step after the first); stra / strb (s: 8 rows in bytes);")
    print("// dst0 (s: LDS byte address of this wave's first DMA row pair in stage 0).")
    print("// Clobbers v[0:63], v[100:115], s[92:95], m0, scc, memory.")
    emit("NK_TNF_KLOOP_ASM", whole)
    base = ['"memory"', '"m0"', '"scc"', '"s92"', '"s93"', '"s94"', '"s95"']
    print("#define NK_TNF_CLOBBERS " + ", ".join(base + [f'"v{r}"' for r in range(64)] + [f'"v{r}"' for r in range(100, 116)]))
    print("// The same loop without shadows and flush, for the kernel-matrix launches: the four accumulator blocks are operands")
    print("// (c00 c01 c10 c11: +{v[0:15]} .. +{v[48:63]}, zero on entry) and are carried through all steps.")
    emit("NK_TNF_KLOOP_PLAIN_ASM", lambda: whole(True))
    print("#define NK_TNF_PLAIN_CLOBBERS " + ", ".join(base + [f'"v{r}"' for r in range(100, 108)]))


if __name__ == "__main__":
    main()
