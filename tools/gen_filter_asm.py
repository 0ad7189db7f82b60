#!/usr/bin/env python3
"""Generates raytracer_amd/csrc/filter_asm.inc: the sphere-filter loops as gfx950 assembly.

Why assembly: on gfx950 a three-source VALU instruction (v_fma_f32, v_bitop3_b32) issues in
~2.1 cycles per wave only when its three sources sit in three different VGPR banks
(bank = register number mod 4); two sources in one bank cost ~4.2 cycles, as does any SGPR
operand (tools/valu_issue_bench.hip).  hipcc's register allocation ignores banks, so the
filter loop - >90 % of the frame's instructions - is emitted here with a fixed register plan
in which every FMA reads banks {1,0,2}, {2,1,0}, {3,2,1} or {0,0,3}.

Register plan (VGPR numbers; b0..b3 = bank):
  v[4k : 4k+3]      ray k:      (pad, dx, dy, dz)        k < 5      dx b1, dy b2, dz b3
  v[20:23]          origin:     (ox, oy, oz, pad)        trace loops only
  v25 (b1)  k1 = 1 - 2^-18,  v26 (b2)  fm = front margin
  v[28+4u : 31+4u]  sphere set 0, u < 4:  (x b0, y b1, z b2, w b3)
  v[44+4u : 47+4u]  sphere set 1
  v[60+4u : 63+4u]  trace: (opx b0, opy b1, opz b2, -c'' b3) of sphere u
  v76 v78 v80 v82   trace: |o-p|^2 accumulators (fmac: the accumulator's bank is free)
  v[84+4j] (b0), v[85+4j] (b1), j < 10   chain temporaries ta, tb
  v86+4m, m < 8 (b2)     verdict word of sphere m of the trip;  v87 their AND
  v24, v77 v79 v81 v83   batch index and the four sphere indices for the queue append
  s[98:99]               saved exec mask during the append

A loop walks the staged spheres from index i in batches of 4, two batches per trip with the
two register sets in ping-pong (the LDS reads of one batch fly while the other is computed).
When some lane's verdict for a batch is "could hit", the lanes concerned append the sphere's
index to their own candidate queue in LDS (slot-major: slot s of thread t at qaddr0 + 1024*s
+ 4*t, so a wave's write is conflict-free) and the loop goes on; the C++ caller runs the exact
evaluation over each lane's queue afterwards, in index order.  The loop leaves early only when
some lane's queue could overflow in the next trip (its qaddr has reached qlim); the caller
then drains the queues and re-enters at the returned index.  The verdicts of a whole trip
(8 spheres) are ANDed into one word and tested with ONE compare and branch per trip.

The sphere array in LDS holds {x, y, z, w} per entry: for the primary loop x,y,z = o - p and
w = -c'' precomputed per frame; for the trace loops x,y,z = p and w = r*r.  Behind the last
batch there must be 4 more entries (any finite or infinite values): the loop always reads one
batch ahead.  The loop is entered at multiples of 8 spheres and cnt is a multiple of 8.
"""
import os

DN = lambda k: 4 * k
O = 20
K1, FM = 25, 26
SET = (lambda u: 28 + 4 * u, lambda u: 44 + 4 * u)
OP = lambda u: 60 + 4 * u
Q = lambda u: 76 + 2 * u
TA = lambda j: 84 + 4 * j
TB = lambda j: 85 + 4 * j
ACC = lambda u, ph=0: 86 + 4 * (NB_ * ph + u)
NB_ = 4
ALL = 87
NB = 4  # spheres per batch


def bank(r):
    return r % 4


def check3(a, b, c):
    """All distinct registers among the sources must be in distinct banks."""
    regs = sorted(set((a, b, c)))
    banks = [bank(r) for r in regs]
    assert len(set(banks)) == len(banks), f"bank conflict among v{a}, v{b}, v{c}"


def fma(d, a, b, c, na=False, nc=False):
    check3(a, b, c)
    return f"v_fma_f32 v{d}, {'-' if na else ''}v{a}, v{b}, {'-' if nc else ''}v{c}"


def filter_set(lines, K, trace, cur, ph=None):
    if ph is None:
        ph = cur
    """Verdicts of the NB spheres in register set `cur` for K rays -> ACC(u), ALL."""
    src = OP if trace else SET[cur]
    if trace:
        s = SET[cur]
        for u in range(NB):
            for c in range(3):
                lines.append(f"v_sub_f32 v{OP(u) + c}, v{O + c}, v{s(u) + c}")
        for u in range(NB):
            lines.append(f"v_mul_f32 v{Q(u)}, v{OP(u)}, v{OP(u)}")
        for c in (1, 2):
            for u in range(NB):
                lines.append(f"v_fmac_f32 v{Q(u)}, v{OP(u) + c}, v{OP(u) + c}")
        for u in range(NB):
            # -c'' = fma(q, -k1, r*r)
            check3(Q(u), K1, s(u) + 3)
            lines.append(f"v_fma_f32 v{OP(u) + 3}, v{Q(u)}, -v{K1}, v{s(u) + 3}")
    for half in range(NB // 2):
        us = (2 * half, 2 * half + 1)
        chains = [(u, k) for u in us for k in range(K)]
        tmp = {ch: j for j, ch in enumerate(chains)}
        for (u, k) in chains:   # x' = fm - dx*opx
            lines.append(fma(TA(tmp[u, k]), DN(k) + 1, src(u) + 0, FM, na=True))
        for (u, k) in chains:   #      - dy*opy
            lines.append(fma(TB(tmp[u, k]), DN(k) + 2, src(u) + 1, TA(tmp[u, k]), na=True))
        for (u, k) in chains:   #      - dz*opz
            lines.append(fma(TA(tmp[u, k]), DN(k) + 3, src(u) + 2, TB(tmp[u, k]), na=True))
        for (u, k) in chains:   # v = x'*x' - c''     (w holds -c'')
            lines.append(fma(TB(tmp[u, k]), TA(tmp[u, k]), TA(tmp[u, k]), src(u) + 3))
        for u in us:
            j = tmp[u, 0]
            lines.append(f"v_or_b32 v{ACC(u, ph)}, v{TB(j)}, v{TA(j)}")
            for k in range(1, K):
                j = tmp[u, k]
                check3(TB(j), ACC(u, ph), TA(j))
                # (v | x') & acc
                lines.append(f"v_bitop3_b32 v{ACC(u, ph)}, v{TB(j)}, v{ACC(u, ph)}, v{TA(j)} bitop3:0xc8")
    if ph == 0:
        lines.append(f"v_and_b32 v{ALL}, v{ACC(0, ph)}, v{ACC(1, ph)}")
    else:
        lines.append(f"v_and_b32 v{ALL}, v{ALL}, v{ACC(0, ph)}")
        lines.append(f"v_and_b32 v{ALL}, v{ALL}, v{ACC(1, ph)}")
    lines.append(f"v_and_b32 v{ALL}, v{ALL}, v{ACC(2, ph)}")
    lines.append(f"v_and_b32 v{ALL}, v{ALL}, v{ACC(3, ph)}")


VARIANT = os.environ.get("FA_VARIANT", "")   # experiments only (tools/filter_loop_bench.hip)


def loads(lines, which, byte_offset, first=False):
    if "nolds" in VARIANT and not first:
        return
    for u in range(NB):
        r = SET[which](u)
        lines.append(f"ds_read_b128 v[{r}:{r + 3}], %[addr] offset:{byte_offset + 16 * u}")


IDX0 = 24
IDX = lambda u: 77 + 2 * u
QSTRIDE = 1024   # bytes between a thread's consecutive queue slots (256 threads x 4 B)


def append_block(L):
    """After both phases of a trip: lanes whose verdict for a sphere is 'could hit' queue its
    index (8 spheres per trip, in index order)."""
    L.append(f"v_cmp_lt_i32 vcc, -1, v{ALL}")
    L.append("s_cbranch_vccz .LFA_NOAPP%=")
    L.append("s_mov_b64 s[98:99], exec")
    L.append(f"v_mov_b32 v{IDX0}, %[i]")
    for ph in range(2):
        for u in range(NB):
            L.append("s_mov_b64 exec, s[98:99]")   # the compare must see every lane again
            L.append(f"v_add_u32 v{IDX(0)}, {ph * NB + u}, v{IDX0}")
            L.append(f"v_cmp_lt_i32 vcc, -1, v{ACC(u, ph)}")
            L.append("s_and_b64 exec, s[98:99], vcc")
            L.append(f"ds_write_b32 %[qaddr], v{IDX(0)}")
            L.append(f"v_add_u32 %[qaddr], {QSTRIDE}, %[qaddr]")
    L.append("s_mov_b64 exec, s[98:99]")
    L.append("v_cmp_le_u32 vcc, %[qlim], %[qaddr]")
    L.append("s_cbranch_vccnz .LFA_FULL%=")
    L.append(".LFA_NOAPP%=:")


def gen_loop(K, trace):
    L = []
    loads(L, 0, 0, first=True)
    if "nolds" in VARIANT:
        loads(L, 1, 16 * NB, first=True)
    L.append(".LFA_LOOP%=:")
    L.append("s_waitcnt lgkmcnt(0)")
    loads(L, 1, 16 * NB)
    filter_set(L, K, trace, 0, 0)
    L.append("s_waitcnt lgkmcnt(0)")
    loads(L, 0, 32 * NB)
    filter_set(L, K, trace, 1, 1)
    append_block(L)
    L.append(f"s_add_u32 %[i], %[i], {2 * NB}")
    L.append(f"v_add_u32 %[addr], {32 * NB}, %[addr]")
    L.append("s_cmp_lt_i32 %[i], %[cnt]")
    L.append("s_cbranch_scc1 .LFA_LOOP%=")
    L.append("s_branch .LFA_DONE%=")
    # A queue is (nearly) full: return the index of the next trip.
    L.append(".LFA_FULL%=:")
    L.append(f"s_add_u32 %[i], %[i], {2 * NB}")
    L.append(f"v_add_u32 %[addr], {32 * NB}, %[addr]")
    L.append(".LFA_DONE%=:")
    L.append("s_waitcnt lgkmcnt(0)")          # nothing of ours may still be in flight outside
    return L


def clobbers(K, trace):
    regs = set()
    for u in range(NB):
        regs.update(range(SET[0](u), SET[0](u) + 4))
        regs.update(range(SET[1](u), SET[1](u) + 4))
        regs.update((ACC(u, 0), ACC(u, 1), IDX(u)))
    regs.add(IDX0)
    for j in range(2 * K):
        regs.update((TA(j), TB(j)))
    regs.add(ALL)
    if trace:
        for u in range(NB):
            regs.update(range(OP(u), OP(u) + 4))
            regs.add(Q(u))
    return sorted(regs)


def emit(name, K, trace):
    lines = gen_loop(K, trace)
    probe = []
    filter_set(probe, K, trace, 0, 0)
    filter_set(probe, K, trace, 1, 1)
    valu = sum(1 for l in probe if l.startswith("v_")) + 2
    out = [f"// {name}: {'trace' if trace else 'primary'} loop, {K} ray(s) per lane, {valu} VALU instructions per "
           f"trip of {2 * NB} spheres when nothing is queued"]
    out.append(f"#define {name}_VALU {valu}")
    out.append(f"#define {name}_ASM \\")
    for l in lines:
        out.append(f'    "{l}\\n" \\')
    out[-1] = out[-1][:-2]
    cl = ", ".join(f'"v{r}"' for r in clobbers(K, trace))
    out.append(f"#define {name}_CLOBBERS {cl}, \"s98\", \"s99\", \"vcc\", \"scc\", \"memory\"")
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("FA_OUT") or os.path.join(here, "..", "raytracer_amd", "csrc", "filter_asm.inc")
    out = ["// GENERATED by tools/gen_filter_asm.py - do not edit; edit the generator and re-run it.",
           "// Sphere-filter loops for gfx950 with a VGPR-bank-conflict-free register plan.",
           "// Register plan and calling convention: see the generator's module docstring.", ""]
    out.append("// Pinned operand registers (constraint strings).")
    for k in range(5):
        out.append(f'#define FA_DN{k} "{{v[{DN(k)}:{DN(k) + 3}]}}"')
    out.append(f'#define FA_O "{{v[{O}:{O + 3}]}}"')
    out.append(f'#define FA_K1 "{{v{K1}}}"')
    out.append(f'#define FA_FM "{{v{FM}}}"')
    out.append(f"#define FA_QUEUE_STRIDE {QSTRIDE}   // bytes between consecutive queue slots of one thread")
    out.append(f"#define FA_BATCH {NB}")
    out.append(f"#define FA_TRIP {2 * NB}   // spheres per loop trip; the loop is entered at multiples of it")
    out.append(f"#define FA_LDS_SLACK {NB}   // never-pass entries required behind the last batch")
    out.append("")
    out += emit("FA_PRIMARY4", 4, False)
    out.append("")
    for K in range(1, 6):
        out += emit(f"FA_TRACE{K}", K, True)
        out.append("")
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
