#!/usr/bin/env python3
"""Static instruction mix of one kernel in the device assembly of rt_kernels.hip.

    hipcc <the library's flags> -x hip --offload-device-only -S raytracer_amd/csrc/rt_kernels.hip -o /tmp/rtk.s
    python tools/isa_mix.py /tmp/rtk.s k_plainILb0

Prints instructions by class for the whole kernel and for its basic blocks that branch backwards (loops), so that
the light loop of k_plain can be read off: how much of it is FP32 arithmetic the reference asks for, how much is
IEEE divide / square-root expansion, double precision, compares and selects, address arithmetic, memory, scalar.
"""
import re
import sys
from collections import Counter


def klass(op: str) -> str:
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp", "v_rsq", "v_sqrt", "v_frexp", "v_ldexp")):
        return "div/sqrt expansion"
    if re.search(r"_f64|_f64_", op):
        return "f64"
    if op.startswith(("v_cmp", "v_cmpx", "v_cndmask", "v_min", "v_max", "v_med3")):
        return "compare/select/minmax"
    if op.startswith(("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac", "v_fmac", "v_pk_", "v_mad_f32", "v_subrev_f32")):
        return "f32 arithmetic"
    if op.startswith(("v_add_co", "v_addc", "v_lshl", "v_lshr", "v_ashr", "v_add_u32", "v_sub_u32", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i",
                      "v_add3", "v_lshl_add", "v_add_lshl", "v_and", "v_or", "v_xor", "v_bfe", "v_bfi", "v_alignbit", "v_not", "v_subrev_u32",
                      "v_subb", "v_mad_co", "v_perm", "v_sub_co", "v_ashrrev", "v_lshlrev", "v_lshrrev")):
        return "integer/address"
    if op.startswith(("v_cvt", "v_trunc", "v_floor", "v_fract", "v_rndne", "v_ceil")):
        return "convert/round"
    if op.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap", "v_nop")):
        return "move/lane"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vector memory"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_load", "s_buffer_load")):
        return "scalar memory"
    if op.startswith(("s_waitcnt", "s_barrier", "s_sleep", "s_nop")):
        return "wait/barrier"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op.startswith("s_"):
        return "scalar alu"
    if op.startswith("v_"):
        return "other valu"
    return "other"


def main(path: str, key: str) -> None:
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", ln))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    print(f"# {lines[start].split(':')[0]}: lines {start + 1}..{end + 1}")
    labels = {m.group(1): i for i, ln in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", ln))}
    ops = []
    for i, ln in enumerate(body):
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\b(.*)", ln)
        if m and not ln.strip().startswith((".", ";")):
            ops.append((i, m.group(1), m.group(2)))
    total = Counter(klass(op) for _, op, _ in ops)
    n = sum(total.values())
    print(f"whole kernel: {n} instructions")
    for k, v in total.most_common():
        print(f"  {k:24s} {v:6d}  {100.0 * v / n:5.1f} %")
    # loops: backward branches
    loops = []
    for i, op, rest in ops:
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = rest.strip().split()[0] if rest.strip() else ""
            if tgt in labels and labels[tgt] < i:
                loops.append((labels[tgt], i))
    for a, b in sorted(set(loops), key=lambda ab: ab[0] - ab[1])[:6]:
        inner = Counter(klass(op) for i, op, _ in ops if a <= i <= b)
        m = sum(inner.values())
        print(f"loop lines {start + a + 1}..{start + b + 1}: {m} instructions: " +
              ", ".join(f"{k} {v}" for k, v in inner.most_common(8)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
