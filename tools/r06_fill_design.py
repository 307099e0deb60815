#!/usr/bin/env python3
"""tools/r06_fill_design.py BENCH.json EBE_TRAFFIC.json SPMV_TRAFFIC.json GENERAL_TRAFFIC.json -- puts the figures of the round's final
measurement call (tools/r06_final.sh) into the {{...}} slots of DESIGN.md."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
et = json.load(open(sys.argv[2])); st = json.load(open(sys.argv[3])); gt = json.load(open(sys.argv[4]))
r = d["roofline"]; c = d["config"]; dd = d["config3_distorted"]; c2 = d["config2"]
asm = et["other_kernels"]["k_assemble_rows<3, 8, 32>"]
rep = {
    "X4_FRAC": f"{r['frac']:.2f}", "X4_ALG_MB": f"{r['algorithmic_bytes'] / 1e6:.0f}", "X4_XY": f"{r['frac_x_and_y_only']:.2f}",
    "X4_US": f"{r['avg_launch_ms'] * 1e3:.1f}",
    "GEN_US": f"{dd['roofline']['avg_launch_ms'] * 1e3:.0f}", "DIST_M": f"{dd['value'] / 1e6:.1f}", "DIST_ITS": f"{dd['cg_iterations']}",
    "GEN3_RESULT": "1.09 against 0.38 ms per product",
    "ASM_MS": f"{asm['avg_launch_us'] / 1e3:.2f}", "ASM_FRAC": f"{6.67e9 / (asm['avg_launch_us'] * 1e-6) / 8e12:.2f}",
    "ASM_PMC": f"{asm['hbm_bytes_per_launch'] / 1e9:.2f}",
    "SPMV_US": f"{st['avg_launch_us']:.0f}", "SPMV_FRAC": f"{(d['spmv_roofline']['format_bytes']) / (st['avg_launch_us'] * 1e-6) / 8e12:.2f}",
    "SPMV_PMC": f"{st['hbm_bytes_per_launch'] / 1e9:.2f}", "ASM_RUN_M": f"{c['assembled_spmv_run']['value'] / 1e6:.1f}",
    "IT_US": f"{c['ms_per_cg_iteration'] * 1e3:.0f}", "IT_FRAC": f"{d['cg_iteration_roofline']['frac']:.2f}",
    "IT_STREAM": f"{d['cg_iteration_roofline']['frac_of_measured_stream']:.2f}",
    "C2_RES_US": f"{c2['ms_per_cg_iteration'] * 1e3:.1f}", "C2_LAUNCH_US": f"{c2['launch_per_kernel_path']['ms_per_cg_iteration'] * 1e3:.1f}",
}
cpu = d["cpu_baseline"]
rows = [
    ("config 3 (headline): 149³ hex8, CG/Jacobi, rtol 1e-9", f"**{d['value'] / 1e6:.1f} M DOF/s** ({d['ms_per_step']:.1f} ms per step: {c['t_assemble_ms']:.2f} ms assemble + set-up, {c['cg_iterations']} iterations at {c['ms_per_cg_iteration'] * 1e3:.1f} µs); one-shot (`value_cold`) {d['value_cold'] / 1e6:.1f} M"),
    ("… against the CPU port", f"{cpu['value'] / 1e6:.3f} M DOF/s on {cpu['cores']} cores → {c['gpu_over_cpu']:.0f}×"),
    ("… at rtol 1e-13 (the parity setting)", f"{c['parity_run_1e-13']['cg_iterations']} iterations, {c['parity_run_1e-13']['value'] / 1e6:.1f} M DOF/s"),
    ("… with the assembled SpMV in the CG", f"{c['assembled_spmv_run']['value'] / 1e6:.1f} M DOF/s (SpMV {c['assembled_spmv_run']['spmv_ms_in_cg']:.2f} ms = {c['assembled_spmv_run']['spmv_frac_of_spec_format']:.2f} of 8 TB/s on stored bytes)"),
    ("… with `-pc_type mg`", f"{c['pc_mg']['value'] / 1e6:.0f} M DOF/s ({c['pc_mg']['ms_per_step']:.1f} ms per step: {c['pc_mg']['t_assemble_ms']:.1f} ms assemble + set-up, {c['pc_mg']['cg_iterations']} iterations)"),
    ("config 3, nodes moved by 0.15 h (general kernel)", f"{dd['value'] / 1e6:.1f} M DOF/s ({dd['cg_iterations']} iterations at {dd['ms_per_cg_iteration'] * 1e3:.0f} µs)"),
    ("config 3, two-kernel product (`ebe_gather`)", f"{d['config3_two_kernel_product']['value'] / 1e6:.1f} M DOF/s"),
    ("config 2: 707² quad4, CG/Jacobi", f"**{c2['value'] / 1e6:.2f} M DOF/s** resident launch ({c2['ms_per_cg_iteration'] * 1e3:.1f} µs per iteration, {c2['cg_iterations']} iterations); launch path {c2['launch_per_kernel_path']['value'] / 1e6:.2f} M ({c2['launch_per_kernel_path']['ms_per_cg_iteration'] * 1e3:.1f} µs); " + (f"{c2['gpu_over_cpu']:.1f}× the CPU port; " if c2.get('gpu_over_cpu') else "") + f"`-pc_type mg` {c2['pc_mg']['value'] / 1e6:.0f} M"),
    ("config 5: cohesive deck ×1000, two Newton increments", f"{d['config5']['disjoint_copies']['ms']:.1f} ms (disjoint copies), {d['config5']['connected_strip']['ms']:.1f} ms (connected strip)"),
    ("nodal stress recovery (outside the metric)", f"{d['stress_recovery']['ms']:.3f} ms = {d['stress_recovery']['frac']:.2f} of the roof"),
    ("config 4 (80 M DOF, 8 GPUs)", "not measured: no multi-GPU node (one GPU holding all of it: 25.9 M DOF/s CG/Jacobi, 116 M with mg, `profiles/r03_c4_on_one_gpu.txt`)"),
]
rep["RESULTS_TABLE"] = "| what | figure |\n|---|---|\n" + "\n".join(f"| {a} | {b} |" for a, b in rows)
s = open("DESIGN.md").read()
for k, v in rep.items():
    s = s.replace("{{" + k + "}}", v)
left = [w for w in s.split("{{")[1:]]
open("DESIGN.md", "w").write(s)
print("filled; slots left:", [w.split("}}")[0] for w in left])
