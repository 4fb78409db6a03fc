#!/usr/bin/env python3
"""Audit of the attn4 code object: the hand-placed stream owns v[64:255] and every accumulator register; hipcc must stay inside v[0:63].
Scans the -save-temps assembly of csrc/attn4.hip: every instruction OUTSIDE ';;#ASMSTART' / ';;#ASMEND' that names v64+ or an a-register is
an error (a compiler copy / spill into the stream's registers is silent corruption), as are scratch traffic and spills.

    python tools/audit_attn4.py [path/to/attn4-hip-amdgcn-amd-amdhsa-gfx950.s]      (without a path: compiles csrc/attn4.hip and csrc/attn4_bwd.hip first)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_s(name="attn4"):
    d = tempfile.mkdtemp(prefix="attn4_audit_")
    src = os.path.join(REPO, "metamorph_amd", "csrc", name + ".hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
                           "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-c", src,
                           "-o", os.path.join(d, name + ".o"), "-save-temps=obj"], cwd=os.path.dirname(src))
    return os.path.join(d, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def audit(path):
    bad = []
    in_asm = False
    kernel = None
    stats = {}
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(_ZN\d+attn4b?\d+\w+|_ZN\d+_GLOBAL__N_\w*gemm_st_kernel\w+):", t)
        if m:
            kernel = m.group(1) if "nstat" not in m.group(1) else None      # (plain helper kernels own no stream)
            if kernel:
                stats[kernel] = {"mfma": 0, "asm_lines": 0, "compiler_lines": 0}
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or kernel is None:
            if ".vgpr_spill_count" in t or ".private_segment_fixed_size" in t:
                if int(t.split(":")[1]) != 0:
                    bad.append((ln, t))
            continue
        code = t.split(";")[0]
        if in_asm:
            stats[kernel]["asm_lines"] += 1
            stats[kernel]["mfma"] += code.startswith("v_mfma")
            continue
        stats[kernel]["compiler_lines"] += 1
        if "scratch_" in code:
            bad.append((ln, t))
        lim = 96 if ("dkdv" in kernel or "gemm_st" in kernel) else (80 if "dq_kernel" in kernel else 64)      # the kernel's amdgpu_num_vgpr
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
            if int(b) >= lim:
                bad.append((ln, t))
        for a in re.findall(r"\bv(\d+)\b", code):
            if int(a) >= lim:
                bad.append((ln, t))
        if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", code):
            bad.append((ln, t))
    return bad, stats


if __name__ == "__main__":
    paths = sys.argv[1:] if len(sys.argv) > 1 else [compile_s("attn4"), compile_s("attn4_bwd"), compile_s("gemm_st")]
    failed = False
    for path in paths:
        bad, stats = audit(path)
        for k, v in stats.items():
            print(k, v)
        if bad:
            failed = True
            print("AUDIT FAILED (%s): %d compiler instructions touch the stream's registers / scratch" % (path, len(bad)))
            for ln, t in bad[:40]:
                print("  line %d: %s" % (ln, t))
    if failed:
        sys.exit(1)
    print("audit ok: hipcc stays inside its registers (v[0:63] / v[0:95] / v[0:79] by kernel), no scratch, no VGPR spills")
