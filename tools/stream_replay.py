#!/usr/bin/env python3
"""Replays the VALU stream of generated kernels as inline asm, with optional edits, to find what
the issue rate of a real kernel depends on (operand kinds, instruction classes, order).

    python tools/stream_replay.py out.hip name1=dis1.s[:edit,edit...] name2=...

Edits: nolit (literals -> VGPR), nosgpr (SGPR sources -> VGPR), noaux (fract/floor/min/max/cvt/cmp/
cndmask -> v_add_f32), nodiv (div_scale/fmas/fixup/rcp/sqrt -> v_fma_f32), e64to32 (no effect
marker), reverse (stream order reversed), sortlit (literal-carrying instructions moved apart)."""
import re, sys

AUX = ("v_fract", "v_floor", "v_max", "v_min", "v_cvt", "v_cmp", "v_cndmask", "v_med3", "v_rndne", "v_trunc")
DIV = ("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp", "v_sqrt", "v_rsq")

def load(path):
    ins = []
    for line in open(path):
        m = re.match(r"\s+(v_[a-z_0-9]+)\s*(.*?)\s*//", line)
        if not m: continue
        op, operands = m.group(1), m.group(2)
        if op.startswith(("v_readfirstlane", "v_cmpx")): continue
        ins.append([op, operands])
    return ins

def vdst(operands):
    m = re.match(r"v(\d+)", operands)
    return m.group(0) if m else "v1"

def edit(ins, e):
    out = []
    if e == "reverse": return ins[::-1]
    if e.startswith("rot"):
        n = int(e[3:]); return ins[n:] + ins[:n]
    if e.startswith("pre:") or e.startswith("post:") or e.startswith("at"):
        kind, _, text = e.partition(":")
        new = [[t.strip().split(" ", 1)[0], t.strip().split(" ", 1)[1].replace(";", ",")] for t in text.split("|")]
        if kind == "pre": return new + ins
        if kind == "post": return ins + new
        n = int(kind[2:]); return ins[:n] + new + ins[n:]
    if e.startswith("sub:"):
        pre = tuple(e[4:].split("+"))
        return [[("v_add_f32_e32" if op.startswith(pre) else op), (vdst(ops) + ", v63, v59" if op.startswith(pre) else ops)] for op, ops in ins]
    if e.startswith("drop:"):
        pre = tuple(e[5:].split("+"))
        return [[op, ops] for op, ops in ins if not op.startswith(pre)]
    for op, ops in ins:
        parts = [p.strip() for p in ops.split(",")]
        if e == "nolit":
            parts = [("v60" if re.fullmatch(r"-?0x[0-9a-f]+", p) else p) for p in parts]
            if op.startswith(("v_fmaak_f32", "v_fmamk_f32")): op = "v_fma_f32"
        elif e == "nosgpr":
            parts = [parts[0]] + [("v61" if re.fullmatch(r"-?s\d+", p) else p) for p in parts[1:]]
        elif e == "noinline" and "u64" not in op:
            parts = [parts[0]] + [("v62" if re.fullmatch(r"-?\d+(\.\d+)?", p) else p) for p in parts[1:]]
        elif e == "noaux" and op.startswith(AUX):
            d = vdst(ops); op, parts = "v_add_f32_e32", [d if d.startswith("v") and "[" not in ops.split(",")[0] else "v1", "v63", "v59"]
        elif e == "nodiv" and op.startswith(DIV):
            d = vdst(ops); op, parts = "v_add_f32_e32", [d, "v63", "v59"]
        elif e == "vop2only" and not op.endswith("_e32"):
            d = vdst(ops); op, parts = "v_add_f32_e32", [d, "v63", "v59"]
        elif e == "nofma" and op.startswith("v_fma_f32"):
            d = vdst(ops); op, parts = "v_add_f32_e32", [d, "v63", "v59"]
        out.append([op, ", ".join(parts)])
    return out

PROLOGUES = {
    "": "",
    "br": "if ((blockIdx.x >> 3) & 1) asm volatile(\"s_nop 0\\n s_nop 0\\n s_nop 0\\n s_nop 0\");",
    "n0": "if ((blockIdx.x >> 3) & 1) asm volatile(\"s_nop 0\");",
    "n1": "if ((blockIdx.x >> 3) & 1) asm volatile(\"s_nop 1\");",
    "n2": "if ((blockIdx.x >> 3) & 1) asm volatile(\"s_nop 2\");",
    "n3": "if ((blockIdx.x >> 3) & 1) asm volatile(\"s_nop 3\");",
    "n5": "if ((blockIdx.x >> 3) & 1) asm volatile(\"s_nop 5\");",
    "sl": "if ((blockIdx.x >> 3) & 1) asm volatile(\"s_sleep 1\");",
    "all0": "asm volatile(\"s_nop 0\");",
    "all1": "asm volatile(\"s_nop 1\");",
    "b0": "if (blockIdx.x & 1) asm volatile(\"s_nop 1\");",
    "b4": "if ((blockIdx.x >> 4) & 1) asm volatile(\"s_nop 1\");",
    "b8": "if ((blockIdx.x >> 8) & 1) asm volatile(\"s_nop 1\");",
    "w0": "if ((threadIdx.x >> 6) & 1) asm volatile(\"s_nop 1\");",
    "w1": "if ((threadIdx.x >> 7) & 1) asm volatile(\"s_nop 1\");",
    "never": "if (blockIdx.x == 0x7fffffff) asm volatile(\"s_nop 1\");",
    "always": "if (blockIdx.x != 0x7fffffff) asm volatile(\"s_nop 1\");",
    "rnd": "{ unsigned h = blockIdx.x * 2654435761u; if ((h >> 13) & 1) asm volatile(\"s_nop 0\"); if ((h >> 17) & 1) asm volatile(\"s_nop 1\"); }",
}

def emit(name, ins, prologue=""):
    body = "\\n".join(f"{op} {ops}" for op, ops in ins)
    clob = ",".join(f'"v{i}"' for i in range(0, 80)) + "," + ",".join(f'"s{i}"' for i in range(8, 40)) + ',"vcc","memory"'
    init = "\\n".join(f"v_cvt_f32_u32_e32 v{i}, v0" for i in range(1, 80))
    return f'''extern "C" __global__ __launch_bounds__(256) void k_{name}(float* out)
{{
    {PROLOGUES[prologue]}
    asm volatile("{init}\\n s_mov_b32 s8, 0x3f000000\\n s_mov_b32 s9, s8\\n s_mov_b32 s10, s8\\n s_mov_b32 s11, s8\\n s_mov_b32 s12, s8\\n s_mov_b32 s13, s8\\n s_mov_b32 s14, s8\\n s_mov_b32 s15, s8\\n{body}" ::: {clob});
    if (out) out[0] = 1.0f;
}}
'''

def main():
    outp = sys.argv[1]
    specs = sys.argv[2:]
    src = ['#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>\n#include <algorithm>\n']
    names = []
    for s in specs:
        name, rest = s.split("=", 1)
        name, _, prologue = name.partition("@")
        path, _, edits = rest.partition(":")
        ins = load(path)
        # SGPR operands of the original refer to kernel arguments: renumber them into s8..s15
        for i in ins:
            i[1] = re.sub(r"\bs(\d+)\b", lambda m: f"s{8 + int(m.group(1)) % 8}", i[1])
            i[1] = re.sub(r"s\[(\d+):(\d+)\]", lambda m: f"s[{16 + 2 * (int(m.group(1)) % 8)}:{17 + 2 * (int(m.group(1)) % 8)}]", i[1])
        for e in filter(None, edits.split(",")):
            ins = edit(ins, e)
        src.append(emit(name, ins, prologue)); names.append((name, len(ins)))
    src.append('''
template <class K> static void run(const char* name, int n, K k) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int r = 0; r < 7; r++) {
        hipEventRecord(a); hipLaunchKernelGGL(k, dim3(324000), dim3(256), 0, 0, (float*)nullptr); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    double waves = 324000.0 * 4, simds = 1024;
    printf("%-28s %4d VALU  median %.3f ms  -> %.2f cycles/instr/SIMD at 2.4 GHz\\n", name, n, t[3], t[3] * 1e-3 * 2.4e9 / (waves / simds * n));
}
int main() {
''')
    for name, n in names:
        src.append(f'    run("{name}", {n}, k_{name});\n')
    src.append('    return 0;\n}\n')
    open(outp, "w").write("".join(src))

if __name__ == "__main__":
    main()
