"""Build libt2v_hostsim.so: the library's SIMT-only kernel sources (no MFMA / asm / LDS-DMA) compiled with g++ against the host
SIMT simulator (tests/hostsim/hip/hip_runtime.h).  TEST INFRASTRUCTURE ONLY — nothing in the product imports it.

The sources are compiled as they lie in t2v-turbo_amd/csrc; the only textual change is `extern __shared__ T name[];`
(dynamic LDS) -> a pointer to the simulator's buffer."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "t2v-turbo_amd", "csrc")
SOURCES = ["backward.hip", "backward_unet.hip", "train.hip"]
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libt2v_hostsim.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "gn_bwd_common.h"), os.path.join(HERE, "common.h"),
                                                      os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__),
                                                      os.path.join(ROOT, "include", "t2v_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for src in SOURCES + ["gn_bwd_common.h"]:
        text = open(os.path.join(CSRC, src)).read()
        text = re.sub(r"extern\s+__shared__\s+(\w+)\s+(\w+)\[\];", r"\1* \2 = (\1*)hostsim::dyn_shared();", text)
        open(os.path.join(OUT, src.replace(".hip", ".cpp")), "w").write(text)
    for src in SOURCES:
        cpp = os.path.join(OUT, src.replace(".hip", ".cpp"))
        obj = cpp.replace(".cpp", ".o")
        # include order: the simulator's common.h / hip_runtime.h shadow the device ones; gn_bwd_common.h is the (patched) copy
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-w", "-I", HERE, "-I", OUT, "-I", os.path.join(ROOT, "include"), "-c", cpp, "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call(["g++", "-shared", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
