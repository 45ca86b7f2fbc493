#!/usr/bin/env python3
"""Generates nys_koop_lqr_amd/csrc/nk_tn_kstep.inc: the hand-scheduled k-step of the fp64 TN engine (nk_gemm_tn.hip) as ONE
inline-assembly block -- 64 v_mfma_f64_16x16x4_f64 with every other instruction of the step (32 ds_read_b64 operand
fetches one sub-step ahead, the 8 LDS-DMA row copies of the NEXT step with their scalar address updates, the waits and
the step's single barrier) placed in the gaps BETWEEN matrix instructions.

Why by hand: two waves share a SIMD's matrix pipe and the arbiter alternates between them, which keeps two waves running
the same loop in phase; whatever one wave does outside its MFMA stream (address arithmetic, DMA issue, LDS waits) the
other does at the same moment, and the pipe idles for all of it (measured: 9-13 % of the launch with the compiler's
schedule, whichever way the C++ was arranged -- profiles/r03_gram_experiments.md).  Spread over the gaps it costs nothing.

    python3 tools/gen_tn_kstep.py > nys_koop_lqr_amd/csrc/nk_tn_kstep.inc
"""
import sys

TSTRIDE_B = 144 * 8          # bytes per LDS row
KS_B = 4 * TSTRIDE_B         # bytes per k-sub-step (4 contraction rows)
BPANEL_B = 16 * TSTRIDE_B    # B panel after the A panel inside a stage


def mfma(i, j, s):
    return f"v_mfma_f64_16x16x4_f64 %[c{i}{j}], %[a{s}{i}], %[b{s}{j}], %[c{i}{j}]"


def reads(ks, s, stage):
    """the 8 operand fetches of sub-step ks of LDS stage `stage` into register set s"""
    ra, rb = f"%[ard{stage}]", f"%[brd{stage}]"
    out = []
    for i in range(4):
        out.append(f"ds_read_b64 %[a{s}{i}], {ra} offset:{ks * KS_B + i * 128}")
    for j in range(4):
        out.append(f"ds_read_b64 %[b{s}{j}], {rb} offset:{ks * KS_B + j * 128}")
    return out


def dma(stage):
    """8 LDS-DMA row copies (4 rows of the A panel, 4 of the B panel) of the next step into LDS stage `stage`.  s[92:93] /
    s[94:95] are the running row pointers of this wave (rows wave, wave + 4, ... of the next step); after the four
    copies they have advanced by 16 rows, i.e. they point at this wave's first row of the step after."""
    seq = [f"s_mov_b32 m0, %[dst{stage}]"]
    for q in range(4):
        seq += ["s_nop 0",
                "global_load_lds_dwordx4 %[voa], s[92:93]",
                f"s_add_u32 m0, m0, {BPANEL_B}",
                "s_add_u32 s92, s92, %[stra]",
                "s_addc_u32 s93, s93, 0",
                "global_load_lds_dwordx4 %[vob], s[94:95]"]
        if q < 3:
            seq += [f"s_sub_u32 m0, m0, {BPANEL_B - KS_B}"]
        seq += ["s_add_u32 s94, s94, %[strb]",
                "s_addc_u32 s95, s95, 0"]
    return seq


def main():
    lines = []
    order = [(i, j) for i in range(4) for j in range(4)]

    def block(s, gaps, pre=(), mid=None):
        """16 MFMAs on register set s; gaps[n] = instructions issued after MFMA n; mid = (n, [instr]) issued BEFORE MFMA n"""
        lines.extend(pre)
        for n, (i, j) in enumerate(order):
            if mid and mid[0] == n:
                lines.extend(mid[1])
            lines.append(mfma(i, j, s))
            lines.extend(gaps.get(n, []))

    def step(st):
        """one k-step on LDS stage st; the DMA of the next step goes to stage 1 - st"""
        # sub-step 0 (set 0): fetch sub-step 1 into set 1 (gaps 0-7), DMA of the next step (gaps 8-15)
        r = reads(1, 1, st)
        d = dma(1 - st)
        per = (len(d) + 7) // 8
        g = {n: [r[n]] for n in range(8)}
        for n in range(8):
            g[8 + n] = d[n * per:(n + 1) * per]
        block(0, g, pre=["s_waitcnt lgkmcnt(0)"])
        # sub-step 1 (set 1): fetch sub-step 2 into set 0
        r = reads(2, 0, st)
        block(1, {n: [r[n]] for n in range(8)}, pre=["s_waitcnt lgkmcnt(0)"])
        # sub-step 2 (set 0): fetch sub-step 3 into set 1
        r = reads(3, 1, st)
        block(0, {n: [r[n]] for n in range(8)}, pre=["s_waitcnt lgkmcnt(0)"])
        # sub-step 3 (set 1): after 8 MFMAs the barrier (this wave has read all of the stage; its DMAs of the next stage,
        # issued 40+ MFMAs ago, have landed), then the first fragments of the next stage into set 0, four MFMAs before the end
        r = reads(0, 0, 1 - st)
        g = {8: r[0:2], 9: r[2:4], 10: r[4:6], 11: r[6:8]}
        block(1, g, pre=["s_waitcnt lgkmcnt(0)"], mid=(8, ["s_waitcnt vmcnt(0)", "s_barrier"]))

    def last(st):
        """the final step of a K range whose last step is a full one: no DMA, no barrier, nothing fetched for a next step"""
        r = reads(1, 1, st)
        block(0, {n: [r[n]] for n in range(8)}, pre=["s_waitcnt lgkmcnt(0)"])
        r = reads(2, 0, st)
        block(1, {n: [r[n]] for n in range(8)}, pre=["s_waitcnt lgkmcnt(0)"])
        r = reads(3, 1, st)
        block(0, {n: [r[n]] for n in range(8)}, pre=["s_waitcnt lgkmcnt(0)"])
        block(1, {}, pre=["s_waitcnt lgkmcnt(0)"])

    def emit(name, body, counts):
        nonlocal lines
        lines = []
        body()
        print(f"#define {name} \\")
        for ln in lines:
            print(f'  "{ln}\\n\\t" \\')
        print('  ""')
        nm = sum(1 for ln in lines if ln.startswith("v_mfma"))
        nr = sum(1 for ln in lines if ln.startswith("ds_read"))
        nd = sum(1 for ln in lines if ln.startswith("global_load_lds"))
        assert (nm, nr, nd) == counts, (name, nm, nr, nd)

    def whole():
        """every full step of a K range in ONE block (one register allocation for the compiler to respect): cnt trips of two
        steady steps, then -- by the bits of `flags` -- one more steady step (bit 0) and the final step (bit 1)"""
        lines.extend(["s_mov_b64 s[92:93], %[rowa]", "s_mov_b64 s[94:95], %[rowb]",
                      "s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 nk_tn_after_%=", "nk_tn_loop_%=:"])
        step(0)
        step(1)
        lines.extend(["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 nk_tn_loop_%=",
                      "nk_tn_after_%=:", "s_bitcmp1_b32 %[flags], 0", "s_cbranch_scc0 nk_tn_even_%="])
        step(0)
        lines.extend(["s_bitcmp1_b32 %[flags], 1", "s_cbranch_scc0 nk_tn_end_%="])
        last(1)
        lines.extend(["s_branch nk_tn_end_%=", "nk_tn_even_%=:", "s_bitcmp1_b32 %[flags], 1", "s_cbranch_scc0 nk_tn_end_%="])
        last(0)
        lines.append("nk_tn_end_%=:")

    print("// GENERATED by tools/gen_tn_kstep.py -- do not edit by hand.")
    print("// k-steps (16 contraction rows each) of the 128 x 128 fp64 tile, every non-matrix instruction in a gap between two MFMAs.")
    print("// Operands: c00..c33 (+v, 8 VGPRs each: the accumulators); a00..a03 b00..b03 (+v: register set 0, holds sub-step 0 of")
    print("// the current step on entry and of the next unprocessed step on exit); a10..b13 (=&v: scratch set); ard0/brd0, ard1/brd1")
    print("// (v: LDS read addresses of the fragments in stage 0 / 1); voa/vob (v: per-lane byte offsets of the DMA); rowa/rowb")
    print("// (s, 64 bit: this wave's first row of the step AFTER the current one); stra/strb (s: 4 rows in bytes); dst0/dst1 (s:")
    print("// LDS byte address of this wave's first DMA row in stage 0 / 1).  Clobbers s[92:95], m0, scc.")
    print("// NK_TN_KSTEPS_ASM: cnt (+s) trips of TWO steady steps (LDS stage 0, then 1); then, if bit 0 of flags (s) is set, ONE more")
    print("// steady step (stage 0); then, if bit 1 is set, the final step of the K range (no DMA, no barrier) on the stage that follows.")
    emit("NK_TN_KSTEPS_ASM", whole, (64 * 5, 32 * 3 + 24 * 2, 8 * 3))


if __name__ == "__main__":
    main()

