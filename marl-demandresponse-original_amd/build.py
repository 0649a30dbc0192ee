"""Build csrc/libmdr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ("mdr_kernels.hip", "mdr_multi.hip", "mdr_persist.hip", "mdr_control.hip", "mdr_api.hip", "mdr_policy.hip")
HEADERS = ("mdr_device.h", "mdr_kernels.h", "mdr_step_common.h", os.path.join("..", "..", "include", "mdr.h"), os.path.join("..", "..", "include", "mdr_policy.h"))
OUTPUT = os.path.join(CSRC, "libmdr_hip.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmdr_hip.so)")


def is_stale() -> bool:
    if not os.path.isfile(OUTPUT):
        return True
    built = os.path.getmtime(OUTPUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps)


def build_variant(name: str, defines=(), verbose: bool = False) -> str:
    """Experiment build csrc/libmdr_hip_<name>.so with extra -D switches (e.g. MDR_NT_STORES=0 for the cache-control sweep of
    tools/bench_size_sweep.py); select it with the MDR_HIP_LIB environment variable.  Never the product library."""
    out = os.path.join(CSRC, "libmdr_hip_%s.so" % name)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    if os.path.isfile(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17", "-fPIC", "-shared"]
    cmd += ["-D" + d for d in defines] + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


FLAGS = ("--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17", "-fPIC")


def build_native(force: bool = False, verbose: bool = False) -> str:
    """One object per source under csrc/build/ (recompiled when the source or any header is newer), then one link: an edit of one
    kernel file costs that file's compile time, not the library's."""
    if not force and not is_stale():
        return OUTPUT
    # -ffp-contract=on: a*b+c fuses only inside one source expression (decided by the front end), never across statements,
    # so that every kernel inlining the same house update / reward expression rounds it identically (hipcc's default
    # "fast" lets the back end fuse depending on the surrounding code)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objects, jobs = [], []
    for src in SOURCES:
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objects.append(obj)
        if force or not os.path.isfile(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in [path] + headers):
            cmd = [_hipcc(), *FLAGS, "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", OUTPUT] + objects
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True, cwd=CSRC)
    return OUTPUT
