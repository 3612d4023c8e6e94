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
SOURCES = ["backward.hip", "backward_unet.hip", "train.hip", "wgrad_tn.hip", "full_grad.hip"]
GEMM_SOURCES = ["gemm.hip", "gemm_exp.hip", "gemm_fuse.hip", "conv_halo.hip", "gemm2.hip", "linear_pr.hip"]   # MFMA + LDS-DMA: simulated as wave collectives (separate, slower-to-build library)
GEMM_HEADERS = ["tile80.h", "gemm2.h", "gelu_poly.h"]
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libt2v_hostsim.so")
GEMM_LIB = os.path.join(OUT, "libt2v_hostsim_gemm.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "gn_bwd_common.h"), os.path.join(HERE, "common.h"),
                                                      os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__),
                                                      os.path.join(ROOT, "include", "t2v_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def patch(text):
    """The textual differences between device source and what g++ sees: dynamic LDS, address-space casts, AMDGPU inline asm."""
    text = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];",
                  r"\1* \2 = (\1*)hostsim::dyn_shared();", text)
    text = re.sub(r"__attribute__\(\(address_space\(\d\)\)\)", "", text)
    # A wave runs in lockstep on the hardware: lanes may exchange data through LDS with nothing but "my LDS operations have
    # landed" (s_waitcnt lgkmcnt(0)) or a compiler fence between the writes and the reads.  Fibers do not: those two markers
    # become wave rendezvous points, which restores exactly that ordering.
    text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)"[^;]*;', "hostsim::wave_rendezvous();", text)
    text = re.sub(r'asm volatile\(""\s*:::\s*"memory"\);', "hostsim::wave_rendezvous();", text)
    text = re.sub(r"__attribute__\(\(ext_vector_type\((\d+)\)\)\)", lambda m: f"__attribute__((vector_size({4 * int(m.group(1))})))", text)
    text = re.sub(r'asm volatile\("s_[^"]*"[^;]*;', ";", text)                 # s_waitcnt vmcnt / s_nop: no asynchrony on the host
    text = re.sub(r'asm volatile\(""\s*:\s*"\+[vs]"\([^;]*;', ";", text)       # register-pinning barriers
    text = text.replace('#include "gemm.hip"', '#include "gemm.cpp"')
    return text


def build_gemm(force=False):
    """t2v_gemm (+ the experimental tile ids; -DT2V_EXPERIMENTAL: the simulator keeps testing the kernels the product library leaves out) on the simulator.  Minutes of g++ time: built on demand by its own test module."""
    deps = [os.path.join(CSRC, s) for s in GEMM_SOURCES + GEMM_HEADERS] + [os.path.join(HERE, "common.h"), os.path.join(HERE, "hip", "hip_runtime.h"),
                                                           os.path.abspath(__file__), os.path.join(ROOT, "include", "t2v_hip.h")]
    if not force and os.path.exists(GEMM_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(GEMM_LIB) for d in deps):
        return GEMM_LIB
    os.makedirs(OUT, exist_ok=True)
    procs, objs = [], []
    for hdr in GEMM_HEADERS:   # (device headers the sources include: patched like them)
        open(os.path.join(OUT, hdr), "w").write(patch(open(os.path.join(CSRC, hdr)).read()))
    for src in GEMM_SOURCES:
        cpp = os.path.join(OUT, src.replace(".hip", ".cpp"))
        open(cpp, "w").write(patch(open(os.path.join(CSRC, src)).read()))
    for src in GEMM_SOURCES:
        cpp = os.path.join(OUT, src.replace(".hip", ".cpp"))
        obj = cpp.replace(".cpp", ".o")
        cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-w", "-fno-strict-aliasing", "-DT2V_EXPERIMENTAL", "-I", HERE, "-I", OUT, "-I", os.path.join(ROOT, "include"), "-c", cpp, "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("g++ failed: " + " ".join(cmd))
    subprocess.check_call(["g++", "-shared", "-o", GEMM_LIB] + objs)
    return GEMM_LIB


FULL_SOURCES = ["gemm.hip", "gemm_exp.hip", "gemm_fuse.hip", "conv_halo.hip", "gemm2.hip", "linear_pr.hip", "ffn.hip", "norm.hip", "attention.hip", "elementwise.hip", "backward.hip", "backward_unet.hip", "train.hip", "attention_bwd.hip", "wgrad_tn.hip", "full_grad.hip", "replay.hip"]
FULL_LIB = os.path.join(OUT, "libt2v_hostsim_full.so")


def build_full(force=False):
    """EVERY source of libt2v_hip.so on the simulator: the engines can then run on the real C-ABI end to end, on the CPU."""
    deps = [os.path.join(CSRC, s) for s in FULL_SOURCES + GEMM_HEADERS] + [os.path.join(CSRC, "gn_bwd_common.h"), os.path.join(HERE, "common.h"),
                                                           os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__),
                                                           os.path.join(ROOT, "include", "t2v_hip.h")]
    if not force and os.path.exists(FULL_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(FULL_LIB) for d in deps):
        return FULL_LIB
    full = os.path.join(OUT, "full")
    os.makedirs(full, exist_ok=True)
    for src in FULL_SOURCES + ["gn_bwd_common.h"] + GEMM_HEADERS:
        open(os.path.join(full, src.replace(".hip", ".cpp")), "w").write(patch(open(os.path.join(CSRC, src)).read()))
    procs, objs = [], []
    for src in FULL_SOURCES:
        cpp = os.path.join(full, src.replace(".hip", ".cpp"))
        obj = cpp.replace(".cpp", ".o")
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-w", "-fno-strict-aliasing", "-DHOSTSIM_FULL", "-DT2V_EXPERIMENTAL", "-I", HERE, "-I", full, "-I", os.path.join(ROOT, "include"),
               "-c", cpp, "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("g++ failed: " + " ".join(cmd))
    subprocess.check_call(["g++", "-shared", "-o", FULL_LIB] + objs)
    return FULL_LIB


def build(force=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for src in SOURCES + ["gn_bwd_common.h"]:
        open(os.path.join(OUT, src.replace(".hip", ".cpp")), "w").write(patch(open(os.path.join(CSRC, src)).read()))
    for src in SOURCES:
        cpp = os.path.join(OUT, src.replace(".hip", ".cpp"))
        obj = cpp.replace(".cpp", ".o")
        # include order: the simulator's common.h / hip_runtime.h shadow the device ones; gn_bwd_common.h is the (patched) copy
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-w", "-fno-strict-aliasing", "-I", HERE, "-I", OUT, "-I", os.path.join(ROOT, "include"), "-c", cpp, "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call(["g++", "-shared", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--gemm" in sys.argv:
        print(build_gemm(force="--force" in sys.argv))
    if "--full" in sys.argv:
        print(build_full(force="--force" in sys.argv))
