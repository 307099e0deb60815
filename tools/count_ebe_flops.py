#!/usr/bin/env python3
"""tools/count_ebe_flops.py -- fp64 work of the element-by-element kernels as compiled: compiles elem_kernels.hip to gfx950
assembly and counts, per k_ebe_apply instantiation, the double-precision vector instructions one lane (= one element)
executes: straight-line code once, the Gauss-pair loop of the general hex8 kernel four times.  flops = 2 per fma, 1 per
add / mul (v_rcp and conversions not counted).  bench.py's roofline uses these figures (FLOPS_* constants there)."""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "elem_kernels_count.s")
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(root, "fem_amd/csrc"),
                       "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only", "-w", "-o", out,
                       os.path.join(root, "fem_amd/csrc/elem_kernels.hip")], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z11k_ebe_applyILi(\d)ELi(\d)ELi(\d)ELi(\d)E\w+):", lines[i])
    if not m:
        i += 1
        continue
    name = "k_ebe_apply<%s,%s,%s,%s>" % m.group(2, 3, 4, 5)
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    body = lines[i:j]
    # loop: from the label marked "Loop Header" to the backward s_cbranch_scc1
    weight = [1] * len(body)
    for a, l in enumerate(body):
        if "Loop Header" in l:
            lab = l.split(":")[0].strip()
            for b in range(a, len(body)):
                if "s_cbranch_scc" in body[b] and lab in body[b]:
                    trips = 4 if m.group(5) in "12" else 1
                    for c in range(a, b + 1):
                        weight[c] = trips
                    break
    cnt = {"fma": 0, "add": 0, "mul": 0, "valu": 0}
    for l, w in zip(body, weight):
        op = l.strip().split(" ")[0].split("\t")[0]
        if op.startswith("v_"):
            cnt["valu"] += w
        if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
            cnt["fma"] += w
        elif op.startswith("v_add_f64"):
            cnt["add"] += w
        elif op.startswith("v_mul_f64"):
            cnt["mul"] += w
    f64 = cnt["fma"] + cnt["add"] + cnt["mul"]
    print(f"{name}: fma {cnt['fma']} add {cnt['add']} mul {cnt['mul']} -> fp64 instr {f64}, flops {2 * cnt['fma'] + cnt['add'] + cnt['mul']}, "
          f"all vector instr {cnt['valu']}")
    i = j
