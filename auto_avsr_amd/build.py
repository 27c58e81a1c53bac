"""Build driver for libavsr_hip.so (gfx950) -- and, for the CPU unit tests only,
the host emulator build of the same sources (tests/emu/libavsr_emu.so).

Usage:  python -m auto_avsr_amd.build [--emu] [--force]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libavsr_hip.so")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libavsr_emu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOST_CXX = os.environ.get("AVSR_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha1()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + os.path.basename(cmd[-1]))
    return r


def _file_flags(src):
    """Per-file compiler flags: a source may carry a line `// AVSR_CXXFLAGS: <flags>` (e.g. augment.hip turns FMA
    contraction off for bit-exact f32 arithmetic; the flag is part of the source digest, so a change rebuilds)."""
    with open(src) as f:
        for line in f:
            if "AVSR_CXXFLAGS:" in line:
                return line.split("AVSR_CXXFLAGS:", 1)[1].split()
    return []


def _build(objdir, out, compile_cmd, link_cmd, extra_inputs, force):
    os.makedirs(objdir, exist_ok=True)
    srcs = _sources()
    headers = sorted(
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + list(extra_inputs)
    )
    hdig = _digest(headers)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        stamp = o + ".sha1"
        dig = _digest([s]) + hdig + " ".join(compile_cmd)
        objs.append(o)
        if not force and os.path.exists(o) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((s, o, stamp, dig))

    def one(j):
        s, o, stamp, dig = j
        _run(compile_cmd + _file_flags(s) + ["-c", s, "-o", o])
        with open(stamp, "w") as f:
            f.write(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(one, jobs))
    if jobs or not os.path.exists(out):
        _run(link_cmd + objs + ["-o", out])
    return out


def build_hip(force=False):
    """hipcc --offload-arch=gfx950 build of every kernel into one C-ABI shared object."""
    cc = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
          "-Wno-unused-result", "-ffp-contract=fast"]
    ld = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"]
    return _build(os.path.join(HERE, "build", "hip"), LIB, cc, ld, [], force)


def build_emu(force=False):
    """Host emulator build (tests only)."""
    cc = [HOST_CXX, "-x", "c++", "-DAVSR_EMU", "-O2", "-std=c++17", "-fPIC", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-I", EMU_DIR,
          "-Wno-unused-result", "-pthread"]
    emu_obj = os.path.join(HERE, "build", "emu", "hip_emu.o")
    os.makedirs(os.path.dirname(emu_obj), exist_ok=True)
    _run([HOST_CXX, "-O2", "-std=c++17", "-fPIC", "-pthread", "-I", EMU_DIR, "-c",
          os.path.join(EMU_DIR, "hip_emu.cpp"), "-o", emu_obj])
    ld = [HOST_CXX, "-shared", "-fPIC", "-pthread", emu_obj]
    return _build(os.path.join(HERE, "build", "emu"), EMU_LIB, cc, ld,
                  [os.path.join(EMU_DIR, "hip_emu.h"), os.path.join(EMU_DIR, "hip_emu.cpp")], force)


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--emu" in sys.argv:
        print(build_emu(force))
    else:
        print(build_hip(force))
