"""Build libt2v_hip.so (gfx950) in-tree with hipcc.  No CMake, no JIT cache: the .so lands next
to the package so it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = ["gemm.hip", "gemm_exp.hip", "gemm_fuse.hip", "conv_halo.hip", "linear_pr.hip", "norm.hip", "attention.hip", "elementwise.hip", "backward.hip", "backward_unet.hip", "train.hip", "attention_bwd.hip", "wgrad_tn.hip", "full_grad.hip", "replay.hip"]
# T2V_EXPERIMENTAL=1: the library WITH the measured negative results (the second t2v_gemm kernel family, the one-launch feed-forward, the
# alternative flash-attention forms, the direct small-Cout conv: -DT2V_EXPERIMENTAL) as libt2v_hip_exp.so next to the product library —
# tools and tests load it through T2V_HIP_LIB; the product library and include/t2v_hip.h's default view do not carry them.
EXPERIMENTAL = os.environ.get("T2V_EXPERIMENTAL") == "1"
if EXPERIMENTAL:
    SOURCES = SOURCES + ["gemm2.hip", "ffn.hip"]
# T2V_HIP_LIB_OUT: build a variant (e.g. with ablation switches) next to the product library instead of over it
LIB = os.path.abspath(os.environ["T2V_HIP_LIB_OUT"]) if os.environ.get("T2V_HIP_LIB_OUT") else os.path.join(PKG, "libt2v_hip_exp.so" if EXPERIMENTAL else "libt2v_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         "-I", os.path.join(ROOT, "include"), "-I", HERE] + (["-DT2V_EXPERIMENTAL"] if EXPERIMENTAL else []) + os.environ.get("T2V_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, "common.h"), os.path.join(HERE, "gn_bwd_common.h"), os.path.join(HERE, "tile80.h"), os.path.join(HERE, "gemm2.h"), os.path.join(HERE, "gelu_poly.h"),
                                                        os.path.join(ROOT, "include", "t2v_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


# what each translation unit includes besides common.h and the C-ABI header (an object is rebuilt when it is older than any of them)
EXTRA_DEPS = {"gemm.hip": ["gemm2.h", "gelu_poly.h"], "gemm_exp.hip": ["gemm.hip", "gemm2.h", "gelu_poly.h"], "gemm_fuse.hip": ["gemm.hip", "gemm2.h", "gelu_poly.h"],
              "linear_pr.hip": ["gelu_poly.h"],
              "conv_halo.hip": ["tile80.h"], "gemm2.hip": ["tile80.h", "gemm2.h"], "backward.hip": ["gn_bwd_common.h"],
              "backward_unet.hip": ["gn_bwd_common.h"]}


def _stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(HERE, src), os.path.join(HERE, "common.h"), os.path.join(ROOT, "include", "t2v_hip.h"), os.path.abspath(__file__)]
    deps += [os.path.join(HERE, d) for d in EXTRA_DEPS.get(src, [])]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        # a build with its own output path OR its own flags is a variant: its objects never share a name with the product's (a
        # flagged object left behind as foo.o would look fresh to the next plain build and be linked into libt2v_hip.so)
        variant = bool(os.environ.get("T2V_HIP_LIB_OUT") or os.environ.get("T2V_EXTRA_HIPCC_FLAGS") or EXPERIMENTAL)
        o = os.path.join(HERE, s.replace(".hip", ".variant.o" if variant else ".o"))
        objs.append(o)
        # (variant builds are always recompiled; the product build only recompiles what changed)
        if not force and not variant and not _stale(s, o):
            continue
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(HERE, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
