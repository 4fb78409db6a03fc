"""Builds metamorph_amd/lib/libmm355.so from csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmm355.so")
# MM355_LEGACY_VARIANTS=1 (tools only): also compile the kernel generations no product path selects -- the round-2 d == 128 attention
# (attn3.hip, attention variant 3), the one-wave-per-SIMD GEMM stream (gemm_st.hip, GEMM variants 13 / 14) and GEMM variants 3-6, 8, 10, 12 --
# for A/B timing (tools/bench_attn4*.py, tools/bench_gemm_*.py) and their tests (MM355_TEST_LEGACY=1).  The product library has none of them.
LEGACY = os.environ.get("MM355_LEGACY_VARIANTS") == "1"
SOURCES = ["gemm_bf16.hip", "rowwise.hip", "elementwise.hip", "attn.hip", "attn2.hip", "attn4.hip", "attn4_bwd.hip", "decode.hip", "losses.hip", "linear_ce.hip"]
if LEGACY:
    SOURCES += ["attn3.hip", "gemm_st.hip"]
# (attn3_kernels.h stays in the digest of the default build: attn4.hip / attn4_bwd.hip include it for its tile-DMA and lane helpers; the
# round-2 kernel templates in it are instantiated only by attn3.hip, i.e. only under MM355_LEGACY_VARIANTS)
HEADERS = ["mm355_common.h", "gemm_common.h", "attn2.h", "rowsum.h", "attn3_kernels.h"] + sorted(
    os.path.join(d, f) for d in ("attn4_gen", "attn4_bwd_gen") + (("gemm_st_gen",) if LEGACY else ())
    for f in os.listdir(os.path.join(CSRC, d)) if f.endswith(".inc"))   # tools/gen_attn4*.py output
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + (["-DMM355_LEGACY_VARIANTS"] if LEGACY else [])
# MFMA results stay in arch VGPRs (<= 256 registers, two waves per SIMD): no accumulator <-> VGPR moves around the VALU phases
FLAGS = BASE_FLAGS + ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# the hand-placed streams own the accumulator file: hipcc must not park its own spills there (tools/audit_attn4.py checks the result)
FLAGS_OF = {f: BASE_FLAGS + ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"] for f in ("attn4.hip", "attn4_bwd.hip", "gemm_st.hip")}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(PKG), "include", "mm355.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FLAGS_OF.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libmm355.stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS_OF.get(src, FLAGS), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[mm355 build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    for stale in ("attn3.o", "gemm_st.o"):                       # objects of another build flavour must not be linked (or shipped)
        if stale.replace(".o", ".hip") not in SOURCES and os.path.exists(os.path.join(LIBDIR, stale)):
            os.remove(os.path.join(LIBDIR, stale))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print("[mm355 build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
