#!/usr/bin/env python3
"""tools/count_ebe_flops.py -- fp64 work of the element-by-element kernels as compiled: compiles elem_kernels.hip and
ebe_tile.hip to gfx950 assembly and counts the double-precision vector instructions one lane executes per element:
  k_ebe_apply<...>: the whole kernel once (one element per lane), the Gauss-pair loop of the general hex8 path four times;
  k_ebe_tile<...>:  the body of the step loop (one element visit per lane and step), its Gauss-pair loop four times.
flops = 2 per fma, 1 per add / mul (v_rcp, conversions, moves not counted).  bench.py's roofline uses these figures
(EBE_WORK there); re-run after changing a kernel."""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_of(src):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".count.s")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(root, "fem_amd/csrc"),
                           "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only", "-w", "-o", out,
                           os.path.join(root, src)], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def loops_of(body):
    """[(first, last)] of the natural loops: header label line .. the s_cbranch that jumps back to it"""
    res = []
    for a, l in enumerate(body):
        if "Loop Header" in l:
            lab = l.split(":")[0].strip()
            last = None
            for b in range(a, len(body)):
                if re.search(r"s_cbranch_\w+\s+" + re.escape(lab) + r"\b", body[b]):
                    last = b
            if last is not None:
                res.append((a, last, "Inner Loop" in l or "Depth=2" in l))
    return res


def count(body, weight):
    cnt = {"fma": 0, "add": 0, "mul": 0, "valu": 0}
    for l, w in zip(body, weight):
        op = l.strip().split(" ")[0].split("\t")[0]
        if not w:
            continue
        if op.startswith("v_"):
            cnt["valu"] += w
        if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
            cnt["fma"] += w
        elif op.startswith("v_add_f64"):
            cnt["add"] += w
        elif op.startswith("v_mul_f64"):
            cnt["mul"] += w
    return cnt


def report(name, cnt):
    f64 = cnt["fma"] + cnt["add"] + cnt["mul"]
    print(f"{name}: fma {cnt['fma']} add {cnt['add']} mul {cnt['mul']} -> fp64 instr {f64}, "
          f"flops {2 * cnt['fma'] + cnt['add'] + cnt['mul']}, all vector instr {cnt['valu']}")


lines = asm_of("fem_amd/csrc/elem_kernels.hip")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z11k_ebe_applyILi(\d)ELi(\d)ELi(\d)ELi(\d)E\w+):", lines[i])
    if not m:
        i += 1
        continue
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    body = lines[i:j]
    weight = [1] * len(body)
    for a, b, _ in loops_of(body):
        for c in range(a, b + 1):
            weight[c] = 4
    report("k_ebe_apply<%s,%s,%s,%s>" % m.group(2, 3, 4, 5), count(body, weight))
    i = j

lines = asm_of("fem_amd/csrc/ebe_tile.hip")
# k_ebe_tile_x4 (four steps unrolled, no loop): everything between the stage-in barrier and the barrier behind the last step, a quarter of it
i = 0
while i < len(lines):
    m4 = re.match(r"^(_ZN3fem13k_ebe_tile_x4ILi([12])E\w+):", lines[i])
    if not m4:
        i += 1
        continue
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    body = lines[i:j]
    bars = [k for k, l in enumerate(body) if re.match(r"^\s*s_barrier", l)]
    cnt = count(body[bars[0]:bars[-1]], [1] * (bars[-1] - bars[0]))
    report("k_ebe_tile_x4<%s> per element visit" % {"1": "affine", "2": "cartesian"}[m4.group(2)], {k: (v + 2) // 4 for k, v in cnt.items()})
    i = j
i = 0
while i < len(lines):
    m = re.match(r"^(_ZN3fem(?:10k_ebe_tile|12k_ebe_tile_x)ILi([012])E(?:Li(\d)E)?\w+):", lines[i])
    mq = re.match(r"^(_ZN3fem14k_ebe_tile_q4xILb([01])E\w+):", lines[i])     # quad4, x and coordinates in LDS (config 2)
    if mq:
        m = mq
    if not m or (not mq and m.group(3) and m.group(3) != "0"):   # (the stress passes <0,1>, <0,2> are not products)
        i += 1
        continue
    xk = "k_ebe_tile_x" in m.group(1) or bool(mq)
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    body = lines[i:j]
    # the step loop: the loop that holds the LDS adds -- every block the compiler marks "in Loop: Header=<that one>"; loops inside it (the Gauss pairs of the general
    # path) four times
    first_add = next(k for k, l in enumerate(body) if "ds_add_f64" in l)
    depth = 1
    hdr = None
    for k in range(first_add, -1, -1):
        mm = re.search(r"Header=(BB\d+_\d+) Depth=%d" % depth, body[k])
        if mm:
            hdr = mm.group(1)
            break
        mm = re.match(r"^\.L(BB\d+_\d+):.*Loop Header: Depth=%d" % depth, body[k])
        if mm:
            hdr = mm.group(1)
            break
    weight = [0] * len(body)
    cur = 0
    for k, l in enumerate(body):
        if re.match(r"^(\.LBB|; %bb)", l):
            if ("Header=" + hdr + " Depth=%d" % depth) in l or re.match(r"^\.L" + hdr + ":", l):
                cur = 1
            elif ("Depth=%d" % (depth + 1)) in l and ("Parent Loop " + hdr) in l:
                cur = 4
            elif not xk and ("Depth=2" in l or ("Parent Loop " + hdr) in l):
                cur = 4
            else:
                cur = 0
        weight[k] = cur
    if mq:
        report("k_ebe_tile_q4x<%s> per element visit" % ("fused CG opening" if m.group(2) == "1" else "plain"), count(body, weight))
    else:
        uni = bool(re.match(r"^_ZN3fem12k_ebe_tile_xILi[12]ELb[01]ELb1E", m.group(1)))   # one geometry record per tile
        report("%s<%s%s> per element visit" % ("k_ebe_tile_x" if xk else "k_ebe_tile", {"0": "general", "1": "affine", "2": "cartesian"}[m.group(2)],
                                               ", one record" if uni else ""), count(body, weight))
    i = j
