#!/usr/bin/env python3
"""Instruction mix of the hot loops of K1 and K3 from the gfx950 disassembly (no GPU needed), priced with the two issue
classes measured in profiles/r01_valu_issue_rates.md.

    python scripts/valu_mix.py > profiles/valu_mix.json

For each kernel: the device assembly (hipcc -S --cuda-device-only, poly_amd/build.py's flags), the LARGEST loop body
that contains the kernel's signature instruction (the murmur3 chain's v_mad_u64_u32 / the packed maximum3), and its VALU
instructions split into
    full  v_add_u32 / v_sub_u32 / v_subrev_u32 / v_and_b32 / v_or_b32 / v_xor_b32 / v_lshrrev_b32 in plain VOP1/VOP2
          form: 2 cycles per wave64 instruction per SIMD
    half  everything else (shifts left, rotates, 3-operand forms, SDWA / DPP, min / max, packed ops, compares, every
          multiply): 4 cycles
The average cycles per VALU instruction of that loop is what bench.py prices the kernel's VALU-issue ceiling with
(applied to the counter-measured instructions per unit; instructions outside the loop are assumed to have the same mix)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poly_amd import build  # noqa: E402

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32"}
KERNELS = [
    ("mash_sketch.hip", r"sketch_slab_kernelILi21E", "v_mad_u64_u32", "K1 polyhip::k1::sketch_slab_kernel<21>"),
    ("sw_packed.hip", r"sw_pk1_kernelILi152ELb0E", "v_pk_maximum3_f16", "K3 polyhip::k3p::sw_pk1_kernel<152,false>"),
    ("sw_packed.hip", r"sw_pk1x2_kernelILi76ELb0E", "v_pk_maximum3_f16", "K3 polyhip::k3p::sw_pk1x2_kernel<76,false>"),
]


def asm_of(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x.s")
        cmd = [build._hipcc()] + build.CXXFLAGS + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        return open(out).read()


def function_body(asm, name_re):
    m = re.search(r"^(_Z\w*" + name_re + r"\w*):", asm, flags=re.M)
    if not m:
        raise SystemExit(f"kernel {name_re} not found")
    end = asm.index(".Lfunc_end", m.end())
    return m.group(1), asm[m.end():end].splitlines()


def hot_loop(lines, signature):
    """largest [label .. backward branch to label] span whose body holds the signature instruction"""
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, ln in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            if any(signature in b for b in body) and (best is None or len(body) > len(best)):
                best = body
    return best


def classify(body):
    full = half = salu = lds = vmem = other = 0
    hist = {}
    for ln in body:
        m = re.match(r"\s+([a-z_0-9]+)\s", ln + " ")
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            base = op
            modified = "sdwa" in ln or "dpp" in ln or op.endswith("_e64") or "row_" in ln
            base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
            if base in FULL and not ("sdwa" in op or "dpp" in op or "sdwa" in ln or "row_" in ln or "quad_perm" in ln):
                full += 1
            else:
                half += 1
            hist[base] = hist.get(base, 0) + 1
        elif op.startswith("s_"):
            salu += 1
        elif op.startswith("ds_"):
            lds += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            vmem += 1
        else:
            other += 1
    return full, half, salu, lds, vmem, hist


def main():
    out = {"source": "scripts/valu_mix.py: hot-loop disassembly histogram (hipcc -S, gfx950) x the issue classes of "
                     "profiles/r01_valu_issue_rates.md (full rate 2 cycles: add/sub/and/or/xor/lshr in plain form; all else 4)",
           "kernels": {}}
    for src, name_re, sig, title in KERNELS:
        sym, lines = function_body(asm_of(src), name_re)
        body = hot_loop(lines, sig)
        if body is None:
            raise SystemExit(f"no loop with {sig} in {sym}")
        full, half, salu, lds, vmem, hist = classify(body)
        n = full + half
        out["kernels"][title] = {
            "symbol": sym, "loop_instructions": len([b for b in body if re.match(r"\s+[a-z]", b)]), "valu": n, "valu_full_rate": full,
            "valu_half_rate": half, "salu": salu, "lds": lds, "vmem": vmem,
            "cycles_per_valu_instruction": (2.0 * full + 4.0 * half) / n,
            "top_valu_opcodes": dict(sorted(hist.items(), key=lambda kv: -kv[1])[:12])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
