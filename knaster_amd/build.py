"""Builds the gfx950 shared library (hipcc, in-tree) -- `python -m knaster_amd.build`.

The library is built with -ffp-contract=off: the compiler never contracts a*b+c
on its own, which is what makes the exact (allow_fma = 0) kernels bit-identical
to the reference's scalar code.  The FMA kernels fuse explicitly.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libknaster_hip.so")
SOURCES = ["kernels_pipe.hip", "kernels_single.hip", "kernels_wide.hip", "kernels_fold.hip", "kernels_interp.hip", "kernels_events.hip", "bank.hip", "jit.hip", "comm.hip"]
# translation units: (source, object, extra flags).  kernels_pipe.hip is built once per pipeline form (its three tables of
# kernels take the longest to compile; side by side they take a third of the time)
UNITS = [("kernels_pipe.hip", "kernels_pipe_mixer.o", ["-DKNH_PIPE_PART=0"]), ("kernels_pipe.hip", "kernels_pipe_fold.o", ["-DKNH_PIPE_PART=1"]),
         ("kernels_pipe.hip", "kernels_pipe_inplace.o", ["-DKNH_PIPE_PART=2"])] + [(s, s.replace(".hip", ".o"), []) for s in SOURCES[1:]]
HELPER = os.path.join(CSRC, "knh_jit_helper")  # the process hiprtc runs in (jit_cache.hpp): host code, links hiprtc only
HEADERS = ["jit_cache.hpp", "stage_table.hpp", "bank_base.hpp", "voice_bank.hpp", "chain_signature.hpp", "voice_stages.hpp", "voice_chain.hpp", "voice_pipe.hpp", "voice_frame.hpp", "voice_dag.hpp", "kernel_registry.hpp", "jit.hpp", "host_shards.hpp", "shard_workers.hpp", "rank_bank.hpp", os.path.join("..", "..", "include", "knaster_hip.h"),
           os.path.join("..", "build.py")]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-Wall",
    "-Wno-unused-value",
]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-lhiprtc", "-lpthread", "-ldl"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(HELPER):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(HELPER))
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS + ["jit_helper.cpp"]]
    return any(os.path.getmtime(d) > t for d in deps)


def build_helper(verbose: bool = True) -> str:
    """csrc/knh_jit_helper: g++ (no device code), hiprtc only -- the compiler's process never holds a GPU."""
    src = os.path.join(CSRC, "jit_helper.cpp")
    if os.path.exists(HELPER) and os.path.getmtime(HELPER) >= max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, "jit_cache.hpp"))):
        return HELPER
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), src, "-o", HELPER + ".tmp",
           "-L" + os.path.join(rocm, "lib"), "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lhiprtc", "-lpthread"]
    if verbose:
        print("[knaster_amd.build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("building knh_jit_helper failed")
    os.replace(HELPER + ".tmp", HELPER)
    return HELPER


def write_jit_source() -> None:
    """csrc/jit_source.inc: voice_stages.hpp + voice_chain.hpp + voice_pipe.hpp + voice_frame.hpp as C++ raw string literals, embedded in the library so
    that chains without a pre-built kernel can be fused at run time by hiprtc (jit.hip), in single-wave or pipelined form."""
    text = open(os.path.join(CSRC, "voice_stages.hpp")).read()
    chain = open(os.path.join(CSRC, "voice_chain.hpp")).read()
    text += "\n" + chain.replace("#pragma once", "").replace('#include "voice_stages.hpp"', "")
    pipe = open(os.path.join(CSRC, "voice_pipe.hpp")).read()
    text += "\n" + pipe.replace("#pragma once", "").replace('#include "voice_chain.hpp"', "")
    frame = open(os.path.join(CSRC, "voice_frame.hpp")).read()
    text += "\n" + frame.replace("#pragma once", "").replace('#include "voice_chain.hpp"', "")
    chunks = [text[i:i + 8000] for i in range(0, len(text), 8000)]
    body = "\n".join('R"KNHJIT(' + c + ')KNHJIT"' for c in chunks) + "\n"
    path = os.path.join(CSRC, "jit_source.inc")
    if not os.path.exists(path) or open(path).read() != body:
        with open(path, "w") as f:
            f.write(body)


def build(force: bool = False, verbose: bool = True, variant: str = "") -> str:
    """Compile every HIP source for gfx950 (one hipcc per translation unit, side by side) and link them into
    csrc/libknaster_hip.so; returns its path.  variant "stamps": the diagnostic build with per-wavefront cycle stamps
    (-DKNH_DAG_STAMPS), kept beside the product as csrc/libknaster_hip_stamps.so (objects in csrc/build_stamps) and loaded
    with KNH_LIB=<path> -- the product library is never the stamped one."""
    variant = variant or os.environ.get("KNH_BUILD_VARIANT", "")
    lib = LIB if not variant else LIB.replace(".so", "_" + variant + ".so")
    if not force and not variant and not needs_build():
        return LIB
    build_helper(verbose)
    write_jit_source()
    extra = ["-DKNH_DAG_STAMPS"] if os.environ.get("KNH_BUILD_STAMPS") == "1" or variant == "stamps" else []  # diagnostic build only
    if os.environ.get("KNH_BUILD_DAG") == "1":  # the experimental five-role pipeline (voice_dag.hpp), not in the default build
        extra.append("-DKNH_WITH_DAG")
    extra += os.environ.get("KNH_EXTRA_FLAGS", "").split()
    objdir = os.path.join(CSRC, "build" + ("_" + variant if variant else ""))
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    objs = []
    # a translation unit is recompiled when its object is missing, older than its source or than any header, or was built
    # with other flags (kept beside it): a change to bank.hip alone costs one compile, not nine
    header_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    header_time = max(header_time, os.path.getmtime(os.path.join(CSRC, "jit_source.inc")))
    for src, objname, unit_flags in UNITS:
        obj = os.path.join(objdir, objname)
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *extra, *unit_flags, "-c", src, "-o", obj]
        stamp = obj + ".flags"
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
                 and os.path.getmtime(obj) >= max(os.path.getmtime(os.path.join(CSRC, src)), header_time))
        if fresh:
            continue
        if os.path.exists(stamp):
            os.remove(stamp)
        if verbose:
            print("[knaster_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), " ".join(cmd)))
    failed = False
    for src, obj, p, cmdline in procs:
        out, _ = p.communicate()
        if p.returncode == 0:
            with open(obj + ".flags", "w") as f:
                f.write(cmdline)
        if p.returncode != 0:
            sys.stderr.write(out)
            failed = True
        elif out.strip() and verbose:
            sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed building libknaster_hip.so")
    cmd = [hipcc, *LINK_FLAGS, "-o", lib + ".tmp", *objs]
    if verbose:
        print("[knaster_amd.build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking libknaster_hip.so")
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, variant=next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")), "")))
