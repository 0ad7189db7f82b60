#!/usr/bin/env python3
"""Generates raytracer_amd/csrc/filter_asm.inc: the sphere-filter loops as gfx950 assembly.

Why assembly: on gfx950 a three-source VALU instruction (v_fma_f32, v_bitop3_b32) issues in
~2.1 cycles per wave only when its distinct source registers sit in different VGPR banks
(bank = register number mod 4); two sources in one bank cost ~4.2 cycles, as does any SGPR
operand (tools/valu_issue_bench.hip).  hipcc's register allocation ignores banks, so the
filter loop - >90 % of the frame's instructions - is emitted here with a fixed register plan
that the generator checks instruction by instruction.

What a loop computes, per lane, for K rays that share an origin and for each sphere of the
staged array {px, py, pz, w} (p = centre relative to the scene's filter centre):
    trace loops:    -c'' = fma(Bx, px, fma(By, py, fma(Bz, pz, w))) - C0      (4 instructions)
    primary loop:   -c'' = w                                                 (precomputed)
    per ray k:      x' = fma(dz, pz, fma(dy, py, fma(dx, px, e_k)))          (3)
                    v  = fma(x', |x'|, -c'')                                 (1)
    verdict:        AND of the v's: sign bit clear = some ray could hit     (1 per 2-3 rays)
The meaning of B, C0, e_k, w and the proof that no hit is lost: rt_kernels.hip, "Sphere
filter".

Register plan (VGPR numbers; b0..b3 = bank):
  v[4k : 4k+3]      ray k < 5:   (e_k b0, dx b1, dy b2, dz b3)
  v[20:23]          lane:        (Bx b0, By b1, Bz b2, C0 b3)       trace loops only
  v24               LDS address of the trip being fetched (SEL loops)
  v25 (b1)          the trip's reject mask, then the queued word
  v[26+4u : 29+4u]  sphere set 0, u < 4:  (px b2, py b3, pz b0, w b1);  -c'' replaces w
  v[42+4u : 45+4u]  sphere set 1
  quads from v58, quad i = (v58+4i b2, v59+4i b3, v60+4i b0, v61+4i b1):
      i < 4:  SA(i) b2, SB(i) b3   temporaries of sphere i's -c'' chain; later SB holds v of ray
                                   2 and SA the sphere's combined verdict word
      i < 6:  TA(i) b0, TB(i) b1   temporaries of ray chain i
      4..6 :  VT(2(i-4)) b2, VT(2(i-4)+1) b3   v1 of ray chain j (GROUP loops)
  s[98:99]          saved exec mask during the append, s88..s97 scratch

A trip handles 8 consecutive entries of the staged array in two batches of 4, the two register
sets in ping-pong (the LDS reads of one batch fly while the other is computed).  The sign bits
of a trip's 8 verdict words are shifted into one reject mask per lane (entry m of the trip =
bit 7-m, 1 = cannot hit).  Two kinds of loop are generated:

  GROUP loops  walk the array of bounding balls (one ball per 8 spheres) from entry i to cnt,
      trip after trip, and store every trip's mask to the lane's queue slot of that trip
      (slot s of thread t at qaddr0 + 1024*s + 4*t, so a wave's write is conflict-free).
      The verdict is "the ray's half-line meets the ball":
          v1 = fma(x', x', -c'');  reject = (sign(v1) | sign(x')) & sign(c'')
      Behind the last trip there must be 4 more entries (the loop reads one batch ahead).
  GSEL loops   run the trips of the ball array that a wave-uniform bitmap names (same bitmap
      format and prefetching as the SEL loops) and store each trip's mask to the queue slot of
      that trip like a GROUP loop; the caller fills the slots of the other trips.
  SEL loops    run the sphere trips named by a wave-uniform bitmap of two 64-bit halves (bit b
      of the low half = trip trip0 + (b ^ 7); the high half follows 64 trips later), lowest
      bit first; the first batch of the next selected trip is
      fetched while the second batch of the current one is computed.  Lanes with a clear
      mask bit append the word (first sphere index << 8 | mask) to their queue, tested with
      ONE compare and branch per trip, and the loop goes on; the C++ caller runs the exact
      evaluation over each lane's queue afterwards.  The loop leaves early when some lane's
      queue is full (its qaddr has reached qlim), with the trips not yet run left in the
      bitmap; the caller drains the queues and re-enters.
"""
import os

NB = 4   # spheres per batch
DN = lambda k: 4 * k
LANE = 20
VADDR = 24
MASK = 25
# Per-ray reject masks of the split GROUP loops (ray k's verdict register sits in bank 1, 0, 3, 1
# for k = 0..3 and must not share a bank with the mask it is shifted into).
RAYMASK = (86, 87, 88, 90)
QTMP = 89   # queue address of the trip a GSEL loop stores
SET = (lambda u: 26 + 4 * u, lambda u: 42 + 4 * u)
QUAD = lambda i: 58 + 4 * i
SA = lambda u: QUAD(u)
SB = lambda u: QUAD(u) + 1
TA = lambda j: QUAD(j) + 2
TB = lambda j: QUAD(j) + 3
VT = lambda j: QUAD(4 + j // 2) + (j % 2)
QSTRIDE = 1024   # bytes between a thread's consecutive queue slots (256 threads x 4 B)
VARIANT = os.environ.get("FA_VARIANT", "")   # experiments only (tools/filter_loop_bench.hip)

assert MASK % 4 == 1


def bank(r):
    return r % 4


def check3(*regs):
    """Distinct source registers of a three-source instruction must be in distinct banks."""
    regs = sorted(set(regs))
    banks = [bank(r) for r in regs]
    assert len(set(banks)) == len(banks), "bank conflict among " + ", ".join(f"v{r}" for r in regs)


def fma(d, a, b, c):
    check3(a, b, c)
    return f"v_fma_f32 v{d}, v{a}, v{b}, v{c}"


def and3(d, a, b, c):
    check3(a, b, c)
    return f"v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x80"


def group_size(K):
    """Spheres whose ray chains are interleaved (group_size * K independent chains)."""
    g = os.environ.get("FA_GROUP")
    if g:
        return int(g)
    return {1: 4, 2: 2, 3: 2, 4: 1, 5: 1}[K]


def phase(L, K, trace, cur, ph, ball=False, split=False):
    """Shifts the reject bits of the NB entries in register set `cur` into MASK, or with
    `split` every ray's own reject bit into RAYMASK[k]."""
    s = SET[cur]
    px, py, pz, w = (lambda u: s(u)), (lambda u: s(u) + 1), (lambda u: s(u) + 2), (lambda u: s(u) + 3)
    if trace:
        bx, by, bz, c0 = LANE, LANE + 1, LANE + 2, LANE + 3
        for u in range(NB):
            L.append(fma(SA(u), bz, pz(u), w(u)))
        for u in range(NB):
            L.append(fma(SB(u), by, py(u), SA(u)))
        for u in range(NB):
            L.append(fma(SA(u), bx, px(u), SB(u)))
        for u in range(NB):
            L.append(f"v_sub_f32 v{w(u)}, v{SA(u)}, v{c0}")
    G = group_size(K)
    for g0 in range(0, NB, G):
        us = range(g0, g0 + G)
        chains = [(u, k) for u in us for k in range(K)]
        tmp = {ch: j for j, ch in enumerate(chains)}
        assert len(chains) <= 6

        def vreg(u, k):
            """Where ray k's v of sphere u goes: banks b1, b0, b3, b1, b0 for k = 0..4."""
            if K == 1:
                return SA(u)
            j = tmp[u, k]
            return (TB(j), TA(j), SB(u), TB(j), TA(j))[k]

        for (u, k) in chains:
            L.append(fma(TA(tmp[u, k]), DN(k) + 1, px(u), DN(k)))
        for (u, k) in chains:
            L.append(fma(TB(tmp[u, k]), DN(k) + 2, py(u), TA(tmp[u, k])))
        for (u, k) in chains:
            L.append(fma(TA(tmp[u, k]), DN(k) + 3, pz(u), TB(tmp[u, k])))
        if ball:
            for (u, k) in chains:
                x = TA(tmp[u, k])
                check3(x, w(u))
                L.append(f"v_fma_f32 v{VT(tmp[u, k])}, v{x}, v{x}, v{w(u)}")
            for (u, k) in chains:
                j = tmp[u, k]
                check3(VT(j), TA(j), w(u))
                # reject = (sign(v1) | sign(x')) & [c'' > 0]; the register holds -c'', whose sign
                # bit is set exactly when c'' > 0:  (a | b) & c = 0xa8
                L.append(f"v_bitop3_b32 v{vreg(u, k)}, v{VT(j)}, v{TA(j)}, v{w(u)} bitop3:0xa8")
        else:
            for (u, k) in chains:
                x = TA(tmp[u, k])
                check3(x, w(u))
                L.append(f"v_fma_f32 v{vreg(u, k)}, v{x}, |v{x}|, v{w(u)}")
        if split:
            for u in us:
                for k in range(K):
                    m, v = RAYMASK[k], vreg(u, k)
                    if ph == 0 and u == 0:
                        L.append(f"v_lshrrev_b32 v{m}, 31, v{v}")
                    else:
                        check3(m, v)
                        L.append(f"v_alignbit_b32 v{m}, v{m}, v{v}, 31")
            continue
        for u in us:
            acc = SA(u)
            if K == 2:
                L.append(f"v_and_b32 v{acc}, v{vreg(u, 0)}, v{vreg(u, 1)}")
            elif K >= 3:
                L.append(and3(acc, vreg(u, 0), vreg(u, 1), vreg(u, 2)))
                if K == 4:
                    L.append(f"v_and_b32 v{acc}, v{acc}, v{vreg(u, 3)}")
                elif K == 5:
                    L.append(and3(acc, acc, vreg(u, 3), vreg(u, 4)))
            if ph == 0 and u == 0:
                L.append(f"v_lshrrev_b32 v{MASK}, 31, v{acc}")
            else:
                check3(MASK, acc)
                L.append(f"v_alignbit_b32 v{MASK}, v{MASK}, v{acc}, 31")


def loads(L, which, byte_offset, first=False):
    if "nolds" in VARIANT and not first:
        return
    for u in range(NB):
        r = SET[which](u)
        L.append(f"ds_read_b128 v[{r}:{r + 3}], %[addr] offset:{byte_offset + 16 * u}")


def group_loop(K, trace, split=False):
    """GROUP loop: every trip of the ball array from entry i to cnt; masks stored per trip.
    With `split` (2 <= K <= 4) the stored word holds one reject mask per ray, ray k's in bits
    8k .. 8k+7."""
    L = []
    loads(L, 0, 0, first=True)
    L.append(".LFA_LOOP%=:")
    L.append("s_waitcnt lgkmcnt(0)")
    loads(L, 1, 16 * NB)
    phase(L, K, trace, 0, 0, ball=True, split=split)
    L.append("s_waitcnt lgkmcnt(0)")
    loads(L, 0, 32 * NB)
    phase(L, K, trace, 1, 1, ball=True, split=split)
    if split:
        for k in range(1, K):
            src = RAYMASK[0] if k == 1 else MASK
            check3(RAYMASK[k], src)
            L.append(f"v_lshl_or_b32 v{MASK}, v{RAYMASK[k]}, {8 * k}, v{src}")
    L.append(f"ds_write_b32 %[qaddr], v{MASK}")
    L.append(f"v_add_u32 %[qaddr], {QSTRIDE}, %[qaddr]")
    L.append(f"s_add_u32 %[i], %[i], {2 * NB}")
    L.append(f"v_add_u32 %[addr], {32 * NB}, %[addr]")
    L.append("s_cmp_lt_i32 %[i], %[cnt]")
    L.append("s_cbranch_scc1 .LFA_LOOP%=")
    L.append("s_waitcnt lgkmcnt(0)")          # nothing of ours may still be in flight outside
    return L


def rotate(L, tag):
    """The bitmap is two 64-bit halves: when the low half is used up the high half takes its
    place (trip0 moves on by 64)."""
    L.append("s_cmp_eq_u64 %[wlo], 0")
    L.append(f"s_cbranch_scc0 .LFA_{tag}%=")
    L.append("s_mov_b64 %[wlo], %[whi]")
    L.append("s_mov_b64 %[whi], 0")
    L.append("s_add_u32 %[trip0], %[trip0], 64")
    L.append(f".LFA_{tag}%=:")


def next_trip(L, dst, tag):
    """Takes the lowest set bit of the bitmap (which must not be empty): dst = its trip,
    v24 = the trip's LDS address."""
    rotate(L, tag)
    L.append("s_ff1_i32_b64 s96, %[wlo]")
    L.append("s_bitset0_b64 %[wlo], s96")
    L.append(f"s_xor_b32 {dst}, s96, 7")
    L.append(f"s_add_u32 {dst}, {dst}, %[trip0]")
    L.append(f"s_lshl_b32 s94, {dst}, {(2 * NB * 16).bit_length() - 1}")
    L.append("s_add_u32 s94, s94, %[addr0]")
    L.append(f"v_mov_b32 v{VADDR}, s94")


def sel_loads(L, which, byte_offset):
    for u in range(NB):
        r = SET[which](u)
        L.append(f"ds_read_b128 v[{r}:{r + 3}], v{VADDR} offset:{byte_offset + 16 * u}")


def sel_loop(K, trace):
    """SEL loop: the sphere trips named by the bitmap; s95 = current trip, s93 = next or -1."""
    L = []
    L.append("s_waitcnt lgkmcnt(0)")          # the in-order counting below starts from zero
    next_trip(L, "s95", "R0")
    sel_loads(L, 0, 0)
    L.append(".LFA_LOOP%=:")
    sel_loads(L, 1, 16 * NB)
    L.append(f"s_waitcnt lgkmcnt({NB})")      # LDS returns in order: the first batch has arrived
    phase(L, K, trace, 0, 0)
    L.append("s_mov_b32 s93, -1")
    L.append("s_or_b64 s[88:89], %[wlo], %[whi]")
    L.append("s_cmp_eq_u64 s[88:89], 0")
    L.append("s_cbranch_scc1 .LFA_LAST%=")
    next_trip(L, "s93", "R1")
    L.append("s_waitcnt lgkmcnt(0)")
    sel_loads(L, 0, 0)                         # the next trip's first batch flies during phase 1
    L.append("s_branch .LFA_PH1%=")
    L.append(".LFA_LAST%=:")
    L.append("s_waitcnt lgkmcnt(0)")
    L.append(".LFA_PH1%=:")
    phase(L, K, trace, 1, 1)
    # lanes with a clear bit in MASK queue (first sphere index << 8 | MASK)
    L.append(f"v_cmp_ne_u32 vcc, 0xff, v{MASK}")
    L.append("s_cbranch_vccz .LFA_NOAPP%=")
    L.append("s_mov_b64 s[98:99], exec")
    L.append("s_lshl_b32 s97, s95, 11")
    L.append("s_and_b64 exec, exec, vcc")
    L.append(f"v_or_b32 v{MASK}, s97, v{MASK}")
    L.append(f"ds_write_b32 %[qaddr], v{MASK}")
    L.append(f"v_add_u32 %[qaddr], {QSTRIDE}, %[qaddr]")
    L.append("s_mov_b64 exec, s[98:99]")
    L.append("v_cmp_le_u32 vcc, %[qlim], %[qaddr]")
    L.append("s_cbranch_vccnz .LFA_FULL%=")
    L.append(".LFA_NOAPP%=:")
    L.append("s_mov_b32 s95, s93")
    L.append("s_cmp_lt_i32 s93, 0")
    L.append("s_cbranch_scc0 .LFA_LOOP%=")
    L.append("s_branch .LFA_DONE%=")
    # A queue is full: the trip already being fetched goes back into the bitmap.
    L.append(".LFA_FULL%=:")
    L.append("s_cmp_lt_i32 s93, 0")
    L.append("s_cbranch_scc1 .LFA_DONE%=")
    L.append("s_sub_u32 s96, s93, %[trip0]")
    L.append("s_xor_b32 s96, s96, 7")
    L.append("s_bitset1_b64 %[wlo], s96")
    L.append(".LFA_DONE%=:")
    L.append("s_waitcnt lgkmcnt(0)")          # nothing of ours may still be in flight outside
    return L


def gsel_loop(K, trace, split=False):
    """GSEL loop: the trips of the ball array named by the bitmap (as in the SEL loops), each
    trip's mask(s) stored to the lane's queue slot of that trip (as in the GROUP loops);
    s95 = current trip, s93 = next or -1."""
    L = []
    L.append("s_waitcnt lgkmcnt(0)")          # the in-order counting below starts from zero
    next_trip(L, "s95", "R0")
    sel_loads(L, 0, 0)
    L.append(".LFA_LOOP%=:")
    sel_loads(L, 1, 16 * NB)
    L.append(f"s_waitcnt lgkmcnt({NB})")      # in order: the first batch (and the last store) are done
    phase(L, K, trace, 0, 0, ball=True, split=split)
    L.append("s_mov_b32 s93, -1")
    L.append("s_or_b64 s[88:89], %[wlo], %[whi]")
    L.append("s_cmp_eq_u64 s[88:89], 0")
    L.append("s_cbranch_scc1 .LFA_LAST%=")
    next_trip(L, "s93", "R1")
    L.append("s_waitcnt lgkmcnt(0)")
    sel_loads(L, 0, 0)                         # the next trip's first batch flies during phase 1
    L.append("s_branch .LFA_PH1%=")
    L.append(".LFA_LAST%=:")
    L.append("s_waitcnt lgkmcnt(0)")
    L.append(".LFA_PH1%=:")
    phase(L, K, trace, 1, 1, ball=True, split=split)
    if split:
        for k in range(1, K):
            src = RAYMASK[0] if k == 1 else MASK
            check3(RAYMASK[k], src)
            L.append(f"v_lshl_or_b32 v{MASK}, v{RAYMASK[k]}, {8 * k}, v{src}")
    L.append(f"s_lshl_b32 s97, s95, {QSTRIDE.bit_length() - 1}")
    L.append(f"v_add_u32 v{QTMP}, s97, %[qaddr]")
    L.append(f"ds_write_b32 v{QTMP}, v{MASK}")
    L.append("s_mov_b32 s95, s93")
    L.append("s_cmp_lt_i32 s93, 0")
    L.append("s_cbranch_scc0 .LFA_LOOP%=")
    L.append("s_waitcnt lgkmcnt(0)")          # nothing of ours may still be in flight outside
    return L


def written_registers(lines):
    """Every VGPR an instruction of the loop writes (its first operand, or a ds_read's tuple)."""
    regs = set()
    for l in lines:
        op, _, rest = l.partition(" ")
        if not (op.startswith("v_") or op.startswith("ds_read")) or op.startswith("v_cmp"):
            continue
        dst = rest.split(",")[0].strip()
        if dst.startswith("v["):
            lo, hi = dst[2:-1].split(":")
            regs.update(range(int(lo), int(hi) + 1))
        elif dst.startswith("v"):
            regs.add(int(dst[1:]))
    return sorted(regs)


def emit(name, K, trace, kind, split=False):
    lines = {"group": group_loop, "gsel": gsel_loop}[kind](K, trace, split) if kind != "sel" else sel_loop(K, trace)
    probe = []
    phase(probe, K, trace, 0, 0, ball=(kind != "sel"), split=split)
    phase(probe, K, trace, 1, 1, ball=(kind != "sel"), split=split)
    valu = sum(1 for l in probe if l.startswith("v_")) + (2 if kind == "group" else 3) + (K - 1 if split else 0)
    what = "sphere" if kind == "sel" else "ball"
    out = [f"// {name}: {kind} loop, {'shared origin' if trace else 'precomputed -c'}, {K} ray(s) per lane, "
           f"~{valu} VALU instructions per trip of {2 * NB} {what}s when nothing is queued"]
    out.append(f"#define {name}_VALU {valu}")
    out.append(f"#define {name}_ASM \\")
    for l in lines:
        out.append(f'    "{l}\\n" \\')
    out[-1] = out[-1][:-2]
    inputs = set(range(DN(0), DN(K - 1) + 4)) | (set(range(LANE, LANE + 4)) if trace else set())
    regs = written_registers(lines)
    assert not (set(regs) & inputs), "the loop must not write its pinned inputs"
    cl = ", ".join(f'"v{r}"' for r in regs)
    sregs = ", ".join(f'"s{r}"' for r in range(88, 100))
    out.append(f"#define {name}_CLOBBERS {cl}, {sregs}, \"vcc\", \"scc\", \"memory\"")
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("FA_OUT") or os.path.join(here, "..", "raytracer_amd", "csrc", "filter_asm.inc")
    out = ["// GENERATED by tools/gen_filter_asm.py - do not edit; edit the generator and re-run it.",
           "// Sphere-filter loops for gfx950 with a VGPR-bank-conflict-free register plan.",
           "// Register plan and calling convention: see the generator's module docstring.", ""]
    out.append("// Pinned operand registers (constraint strings).")
    for k in range(4):
        out.append(f'#define FA_DN{k} "{{v[{DN(k)}:{DN(k) + 3}]}}"   // ray {k}: (e, dx, dy, dz)')
    out.append(f'#define FA_LANE "{{v[{LANE}:{LANE + 3}]}}"   // (Bx, By, Bz, C0), trace loops')
    out.append(f"#define FA_QUEUE_STRIDE {QSTRIDE}   // bytes between consecutive queue slots of one thread")
    out.append(f"#define FA_BATCH {NB}")
    out.append(f"#define FA_TRIP {2 * NB}   // entries per trip = spheres per bounding ball")
    out.append(f"#define FA_LDS_SLACK {NB}   // entries a GROUP loop reads behind the last trip")
    out.append("")
    out += emit("FA_PGROUP4", 4, False, "group")
    out.append("")
    out += emit("FA_PSEL4", 4, False, "sel")
    out.append("")
    out += emit("FA_PGSEL4", 4, False, "gsel")
    out.append("")
    # Trace loops, two families.  Combined (scenes with few lights, K <= 3): the K rays of a
    # record share one reject mask per ball and per sphere.  Split (K <= 4): the balls are
    # tested for all K rays at once (they share the per-origin term) with one reject mask per
    # ray, and the sphere trips are then run ray by ray (FA_SEL1), each ray over the groups
    # its own mask selects.
    for K in range(1, 4):
        out += emit(f"FA_GROUP{K}", K, True, "group")
        out.append("")
        out += emit(f"FA_SEL{K}", K, True, "sel")
        out.append("")
        out += emit(f"FA_GSEL{K}", K, True, "gsel")
        out.append("")
    for K in range(2, 5):
        out += emit(f"FA_GROUPS{K}", K, True, "group", split=True)
        out.append("")
        out += emit(f"FA_GSELS{K}", K, True, "gsel", split=True)
        out.append("")
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
