#!/usr/bin/env python3
"""Static instruction-mix / issue-cost estimate of a generated kernel (no GPU needed).

    python tools/isa_cost.py c2            # one of the benchmark cinematics
    python tools/isa_cost.py path/to/file.game

Disassembles the gfx950 code object and prices each VALU instruction with the issue costs measured
by tools/valu_microbench.hip on MI355X (cycles per wave64 instruction per SIMD): 2 for
add/sub/mul/fma/bitwise with VGPR or literal operands, 4 for min/max/fract/floor/rndne/cvt/integer
multiply/packed/f64, 8 for transcendental (rcp, sqrt, rsq, sin, exp, log). Straight-line kernels
only (loops are counted once)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as T  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_sin", "v_cos", "v_exp", "v_log")
HALF = ("v_max", "v_min", "v_fract", "v_floor", "v_rndne", "v_trunc", "v_ceil", "v_cvt", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i",
        "v_pk_", "v_med3", "v_ldexp", "v_frexp", "v_div_scale", "v_div_fmas", "v_div_fixup", "v_cmp", "v_cndmask", "v_bfe", "v_mul_u32",
        "v_mul_i32", "v_readfirstlane", "v_lshl_add_u64", "v_mad_u64")


def cost(op, operands):
    if op.startswith(TRANS):
        return 8
    if "_f64" in op:
        return 4
    if op.startswith(HALF):
        return 4
    if re.search(r"\bs\d+|\bs\[", operands) and op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma", "v_mac", "v_fmac")):
        return 4     # SGPR operand halves the rate of the simple FP32 ops
    return 2


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "c2"
    srcs = {"c1": T.C1_HELLO, "c2": T.C2_FBM, "c2b": T.C2B_NOISE, "c3": T.C3_SHOWCASE, "c4": T.C4_TRAILS, "c5": T.C5_GRAVITY}
    src = srcs[key] if key in srcs else open(key).read()
    kernel = sys.argv[2] if len(sys.argv) > 2 else "frame"
    game = T.load_product()
    cin = game.compile(src).cinematics[0]
    prefix = "/tmp/isa_cost_" + re.sub(r"\W", "_", key)
    cin.dump_code(prefix)
    dis = subprocess.run([OBJDUMP, "-d", f"{prefix}_{kernel}.co"], capture_output=True, text=True).stdout
    open(prefix + f"_{kernel}.s", "w").write(dis)
    hist, cyc = collections.Counter(), collections.Counter()
    total_valu = total_cycles = salu = other = 0
    for line in dis.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m:
            continue
        op, operands = m.group(1), m.group(2)
        if op.startswith("v_"):
            c = cost(op, operands)
            base = re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", op)
            hist[base] += 1
            cyc[base] += c
            total_valu += 1
            total_cycles += c
        elif op.startswith("s_"):
            if op not in ("s_nop", "s_endpgm", "s_waitcnt", "s_branch", "s_cbranch_execz", "s_cbranch_scc0", "s_cbranch_scc1"):
                salu += 1
        else:
            other += 1
    print(f"{key}/{kernel}: {total_valu} VALU instructions, {total_cycles} issue cycles per wave ({total_cycles / max(total_valu, 1):.2f}/instr), {salu} SALU, {other} memory/other")
    print(f"  disassembly: {prefix}_{kernel}.s   source: {prefix}_{kernel}.hip")
    for op, n in cyc.most_common(28):
        print(f"  {op:28s} x{hist[op]:5d}  {n:6d} cycles")
    # at 100 % issue: pixels/s = SIMDs * clock / cycles_per_wave * 64
    print(f"  issue-bound ceiling at 2.4 GHz: {1024 * 2.4e9 / max(total_cycles, 1) * 64 / 1e9:.1f} Gpixel/s")


if __name__ == "__main__":
    main()
