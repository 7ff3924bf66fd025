"""Build the gfx950 HIP library in-tree: enspara_amd/libenspara_hip.so.

`python -m enspara_amd.build` (or __graft_entry__.build()).  hipcc
cross-compiles for gfx950 without a GPU.  The library is linked without an
rpath so that, inside a process that has already imported torch, the dynamic
loader binds it to the HIP runtime torch loaded (both have the soname
libamdhip64.so.7); outside such a process it resolves through the normal
search path (/opt/rocm/lib).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
OUT = os.path.join(HERE, "libenspara_hip.so")
SOURCES = ["ek_prepare.hip", "ek_kcenters.hip", "ek_spec.hip", "ek_pass16.hip", "ek_chain.hip", "ek_round.hip", "ek_mshard.hip", "ek_assign.hip", "ek_pam.hip", "ek_pam_sparse.hip",
           "ek_msm.hip", "ek_krylov.hip", "ek_features.hip", "ek_api.hip", "ek_api_pam.hip",
           "ek_api_ms.hip"]
HEADERS = ["ek_common.h", "ek_ctx.h", "ek_qcp.h", "ek_reduce.h", "ek_chain_dev.h", "ek_top_dev.h", "ek_pam_sparse.h", "ek_lanes.h", os.path.join("..", "..", "include",
                                                   "enspara_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: FMAs are explicit in the sources (numerical contract,
# csrc/ek_qcp.h); hipcc's default would fuse a*b+c on its own.
# -fno-slp-vectorize: otherwise the FMA chains become v_pk_fma_f32 plus one
# v_mov per operand pair, which measured slower than plain v_fmac_f32 here.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall",
         "-Wextra",
         "-Wno-unused-parameter"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None, tag=""):
    """``out``/``tag``/``extra_flags`` build an experimental variant next to
    the default library (tools/ only)."""
    global OBJ, OUT
    if out:
        OUT = out
        OBJ = os.path.join(HERE, "_build" + tag)
        force = True
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if not force and _newer(op, [sp] + hdrs):
            continue
        cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if force or not _newer(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
