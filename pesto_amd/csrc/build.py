"""In-tree build of libpesto_hip.so for gfx950:  python -m pesto_amd.csrc.build [--force]

hipcc cross-compiles without a GPU, so this also is the CPU-side "does it build" check (__graft_entry__.build).
The .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["pesto_schema.cpp", "pesto_kernels.hip", "pesto_node.hip", "pesto_edge.hip", "pesto_api.hip"]
HEADERS = ["pesto_schema.h", "pesto_kernels.h", "pesto_mfma_common.h", "pesto_fin_rendezvous.inc", "pesto_edge_node_waves.inc", os.path.join("..", "..", "include", "pesto_hip.h")]
OUT = os.path.join(HERE, "libpesto_hip.so")
# host-only structure I/O library (include/pesto_io.h): plain C++, no HIP runtime, safe in forked data-loader workers
IO_SOURCE = "pesto_io.cpp"
IO_OUT = os.path.join(HERE, "libpesto_io.so")
# -fno-slp-vectorize: the SLP vectoriser packs adjacent f32 adds/muls of the edge kernel into v_pk_*_f32 and pays for it with
# ~5x more v_mov_b32 shuffles than it saves (665 -> 138 v_mov, mul+add re-fused into v_fmac) - measured +% in DESIGN.md
# -ffp-contract=on: mul+add fuse only within a source expression (front-end decision), not wherever the back-end finds a
# pair (HIP's default "fast"). The template instantiations of one kernel (tiles per work item, waves per workgroup) then round
# identically, so a structure gives the same bits alone, in a batch or as a trajectory frame; no measurable cost.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize", "-ffp-contract=on"]
# developer variants: PESTO_EXTRA_CXXFLAGS="-DPESTO_PROFILE_PHASES" PESTO_LIB_TAG=prof -> libpesto_hip_prof.so (selected at
# run time with PESTO_LIB=<path>); the default build is what ships
EXTRA = os.environ.get("PESTO_EXTRA_CXXFLAGS", "").split()
TAG = os.environ.get("PESTO_LIB_TAG", "")
if TAG:
    OUT = os.path.join(HERE, f"libpesto_hip_{TAG}.so")


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build" + ("_" + TAG if TAG else ""))
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    jobs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [cc] + FLAGS + EXTRA + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in srcs]
    if force or jobs or _stale(OUT, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    build_io(force, verbose)
    return OUT


def build_io(force=False, verbose=True):
    src = os.path.join(HERE, IO_SOURCE)
    hdr = os.path.join(HERE, "..", "..", "include", "pesto_io.h")
    if force or _stale(IO_OUT, [src, hdr]):
        cxx = shutil.which("g++") or hipcc()
        cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", IO_OUT, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return IO_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
