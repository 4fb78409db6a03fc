#!/usr/bin/env python3
"""Audit of the attn4 code object: the hand-placed stream owns v[64:255] and every accumulator register; hipcc must stay inside v[0:63].
Scans the -save-temps assembly of csrc/attn4.hip: every instruction OUTSIDE ';;#ASMSTART' / ';;#ASMEND' that names v64+ or an a-register is
an error (a compiler copy / spill into the stream's registers is silent corruption), as are scratch traffic and spills.

    python tools/audit_attn4.py [path/to/attn4-hip-amdgcn-amd-amdhsa-gfx950.s]      (without a path: compiles csrc/attn4.hip first)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_s():
    d = tempfile.mkdtemp(prefix="attn4_audit_")
    src = os.path.join(REPO, "metamorph_amd", "csrc", "attn4.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-c", src,
                           "-o", os.path.join(d, "attn4.o"), "-save-temps=obj"], cwd=os.path.dirname(src))
    return os.path.join(d, "attn4-hip-amdgcn-amd-amdhsa-gfx950.s")


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
        m = re.match(r"^(_ZN5attn4\w+):", t)
        if m:
            kernel = m.group(1)
            stats[kernel] = {"mfma": 0, "asm_lines": 0, "compiler_lines": 0}
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or kernel is None:
            if ".vgpr_spill_count" in t or ".sgpr_spill_count" in t or ".private_segment_fixed_size" in t:
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
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
            if int(b) >= 64:
                bad.append((ln, t))
        for a in re.findall(r"\bv(\d+)\b", code):
            if int(a) >= 64:
                bad.append((ln, t))
        if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", code):
            bad.append((ln, t))
    return bad, stats


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_s()
    bad, stats = audit(path)
    for k, v in stats.items():
        print(k, v)
    if bad:
        print("AUDIT FAILED: %d compiler instructions touch the stream's registers / scratch" % len(bad))
        for ln, t in bad[:40]:
            print("  line %d: %s" % (ln, t))
        sys.exit(1)
    print("audit ok: hipcc stays inside v[0:63], no scratch, no spills")
