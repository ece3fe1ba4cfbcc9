#!/usr/bin/env python3
"""Build tools/_build/emu/libpnerf_emu.so: the kernels of pointnerf_amd/csrc compiled for the HOST on top of the emulation
shim (tools/emu/hip/hip_runtime.h).  Test infrastructure: same C ABI as libpnerf_hip.so, host pointers instead of device
pointers.  The sources are not edited; three HIP-only constructs are rewritten on the fly:
  extern __shared__ ... T name[];   ->  T *name = (T *)emu::lds();
  asm volatile("" : "+v"(x) ...);   ->  ;          (register pins, no semantics)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "pointnerf_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FILES = ["scan", "grid", "query", "aggregate", "render", "backward", "optim", "pointinit", "embed2d", "prof"]
EXACT = {"grid", "query", "pointinit", "embed2d"}
OPT = {}


def preprocess(text):
    text = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];", r"\1 *\2 = (\1 *)emu::lds();", text)
    text = re.sub(r'asm volatile\(""[^;]*\);', ";", text)
    text = text.replace("(__attribute__((address_space(3))) void *)", "(void *)")
    text = text.replace("(__attribute__((address_space(3))) pn_s4 *)", "(const pn_s4 *)")
    return text


def newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(verbose=False):
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libpnerf_emu.so")
    deps = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith((".h", ".hip"))]
    deps += [os.path.join(ROOT, "include", "pnerf.h"), os.path.join(ROOT, "tools", "emu", "hip", "hip_runtime.h"),
             os.path.join(ROOT, "tools", "emu", "emu_runtime.cpp"), os.path.abspath(__file__)]
    if os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    objs, procs = [], []
    base = [CLANG, "-std=c++17", "-g", "-fPIC", "-march=native", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-ignored-attributes",
            "-I" + os.path.join(ROOT, "tools", "emu"), "-I" + os.path.join(ROOT, "include"), "-I" + SRC]
    for h in os.listdir(SRC):
        if h.endswith(".h"):
            with open(os.path.join(SRC, h)) as f, open(os.path.join(OUT, h), "w") as g:
                g.write(preprocess(f.read()))
    for name in FILES:
        cpp = os.path.join(OUT, name + ".cpp")
        with open(os.path.join(SRC, name + ".hip")) as f, open(cpp, "w") as g:
            g.write(preprocess(f.read()))
        obj = os.path.join(OUT, name + ".o")
        cmd = base + [OPT.get(name, "-O2")] + (["-ffp-contract=off"] if name in EXACT else []) + ["-I" + OUT, "-c", cpp, "-o", obj]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    rt = os.path.join(OUT, "emu_runtime.o")
    procs.append(("emu_runtime", subprocess.Popen(base + ["-O2", "-c", os.path.join(ROOT, "tools", "emu", "emu_runtime.cpp"), "-o", rt],
                                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs.append(rt)
    bad = False
    for name, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0 or (verbose and out.strip()):
            sys.stderr.write("---- %s\n%s\n" % (name, out))
        bad = bad or p.returncode != 0
    if bad:
        raise RuntimeError("emu build failed")
    subprocess.check_call([CLANG, "-shared", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(verbose=True))
