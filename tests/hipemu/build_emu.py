"""Builds tests/hipemu/librainbow_emu.so: the kernel sources of rainbow_amd/csrc compiled for
x86 against the hipemu host interpreter.  TEST INFRASTRUCTURE ONLY (see hipemu.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "rainbow_amd", "csrc")
OUT = os.path.join(HERE, "librainbow_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, f) for f in ("hipemu.h", "hipemu.cpp")]
    deps.append(os.path.join(ROOT, "include", "rainbow_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    if not force and not needs_build():
        return OUT
    cc = CLANG if os.path.exists(CLANG) else "clang++"
    cmd = [cc, "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-DRB_HOST_INTERP",
           "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-I", HERE, "-I", CSRC]
    for s in sources():
        cmd += ["-x", "c++", s]
    tmp = "%s.%d.tmp" % (OUT, os.getpid())           # atomic: a concurrent loader never sees a half-written library
    cmd += ["-x", "c++", os.path.join(HERE, "hipemu.cpp"), "-o", tmp]
    subprocess.check_call(cmd)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
