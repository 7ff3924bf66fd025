"""Build the CPU oracle (test infrastructure) into oracle/_build/liboracle.so.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.py and bench.py's
cpu_baseline leg may call this.  There is no oracle/_ref build: the reference's
RMSD arithmetic lives in mdtraj, whose sources are not under /root/reference
(see oracle/README.md), so there is nothing of the reference to compile here.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "qcp_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")

# -ffp-contract=off: every fused multiply-add in the contract is written
# explicitly (fmaf/fma); nothing else may be fused.  -mfma -mavx2 make those
# calls single instructions (x86-64-v3); results are identical without them,
# only slower, because fma()/fmaf() are correctly rounded either way.
CFLAGS = ["-O3", "-std=c11", "-fPIC", "-shared", "-fopenmp",
          "-ffp-contract=off", "-fno-fast-math", "-mavx2", "-mfma",
          "-Wall", "-Wextra"]


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= os.path.getmtime(SRC)):
        return OUT
    cmd = ["gcc"] + CFLAGS + ["-o", OUT, SRC, "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
