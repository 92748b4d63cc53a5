#!/usr/bin/env python3
"""Dependency-distance statistics of a generated kernel's VALU stream (no GPU needed).

    GAME_HIP_FLAGS="..." python tools/isa_deps.py c2

For every VALU instruction: the distance (in VALU instructions) back to the producer of each VGPR
source. Prints the histogram of the nearest-producer distance, the number of VGPR source reads, and
how many of them read the result of the instruction just before (a read the forwarding path can
serve without a register-file access)."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as T
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
key = sys.argv[1] if len(sys.argv) > 1 else "c2"
srcs = {"c1": T.C1_HELLO, "c2": T.C2_FBM, "c2b": T.C2B_NOISE, "c3": T.C3_SHOWCASE, "c4": T.C4_TRAILS}
game = T.load_product()
cin = game.compile(srcs[key]).cinematics[0]
prefix = "/tmp/isa_deps_" + key
cin.dump_code(prefix)
dis = subprocess.run([OBJDUMP, "-d", prefix + "_frame.co"], capture_output=True, text=True).stdout
if len(sys.argv) > 2: open(sys.argv[2], "w").write(dis)
last_write = {}
n = 0
hist = collections.Counter()
reads = prev_reads = lit = sg = 0
sizes = collections.Counter()
for line in dis.splitlines():
    m = re.match(r"\s+(v_[a-z_0-9]+)\s*(.*?)\s*//\s*[0-9A-F]+:\s*((?:[0-9A-F]{8}\s*)+)", line)
    if not m: continue
    op, operands, enc = m.group(1), m.group(2), m.group(3).split()
    sizes[len(enc) * 4] += 1
    ops = [o.strip() for o in operands.split(",")]
    dst, srcs_ = ops[0], ops[1:]
    n += 1
    near = None
    for s in srcs_:
        mm = re.fullmatch(r"[-|]*v(\d+)\|?", s)
        if mm:
            r = int(mm.group(1)); reads += 1
            if r in last_write:
                d = n - last_write[r]
                if d == 1: prev_reads += 1
                near = d if near is None else min(near, d)
        elif re.match(r"0x", s): lit += 1
        elif re.match(r"s\d+|s\[", s): sg += 1
    hist[min(near, 9) if near else 0] += 1
    mm = re.fullmatch(r"v(\d+)", dst)
    if mm: last_write[int(mm.group(1))] = n
print(f"{key} [{os.environ.get('GAME_HIP_FLAGS','')}]: {n} VALU, {reads} VGPR reads ({reads/n:.2f}/instr), {prev_reads} read the previous result, {lit} literals, {sg} SGPR operands, sizes {dict(sizes)}")
print("  nearest producer distance (0 = none, 9 = 9+):", " ".join(f"{k}:{hist[k]}" for k in sorted(hist)))
