#!/usr/bin/env python3
"""Build libopadpo_hip.so for gfx950 (MI355X) in-tree: hipcc cross-compiles without a GPU.

    python opa-dpo_amd/build.py [--force]

Outputs opa-dpo_amd/lib/libopadpo_hip.so (git-ignored; travels to the GPU box with the snapshot).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libopadpo_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "elementwise.hip", "head_optim.hip", "decode.hip", "capi.hip", "ctx.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    # everything a translation unit can #include or is generated from: headers, the generated asm K-loops (*.inc) and their generators (*_gen.py)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc", "_gen.py"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "opadpo_hip.h"))
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    stamp = os.path.join(OBJDIR, "stamp")
    dig = _digest(srcs + headers)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB

    def cc(src):
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        ostamp = obj + ".stamp"                     # per-object stamp: only the sources that changed (or any header) are recompiled
        odig = _digest([src] + headers)
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == odig:
            return obj
        guard = os.path.basename(src) == "gemm.hip"
        cmd = [HIPCC] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if guard else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if guard:
            # the 256x256 kernels run their K-loop as one asm block with physical registers: it knows nothing about a scratch descriptor, so an
            # instantiation that SPILLS faults at run time (round 5: the bias / activation epilogue in the streaming kernel did) - refuse to build one
            bad, cur = [], None
            for ln in r.stderr.splitlines():
                if "remark: Function Name:" in ln:
                    cur = ln.split("Function Name:")[1].split("[")[0].strip()
                elif cur and ("gemm_nt_w4" in cur or "gemm_tn_w4" in cur) and "ScratchSize [bytes/lane]:" in ln:
                    if int(ln.split("ScratchSize [bytes/lane]:")[1].split("[")[0]) != 0:
                        bad.append(cur)
            if bad:
                os.remove(obj)
                raise RuntimeError("these asm-K-loop kernels use scratch (register spills) and would fault at run time: " + ", ".join(bad))
            r_stderr = "\n".join(ln for ln in r.stderr.splitlines() if "kernel-resource-usage" not in ln)
        else:
            r_stderr = r.stderr
        if verbose and r_stderr.strip():
            print(r_stderr, file=sys.stderr)
        with open(ostamp, "w") as f:
            f.write(odig)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
