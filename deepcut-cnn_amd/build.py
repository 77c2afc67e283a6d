"""Build libdeepcut_hip.so (HIP kernels + C-ABI) for gfx950, in-tree.

    python deepcut-cnn_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the
resulting .so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libdeepcut_hip.so")
SOURCES = ["formats.cpp", "hdf5_reader.cpp", "runtime.cpp", "net_init.cpp", "net_lower.cpp", "net_tune.cpp", "net_run.cpp", "net_image.cpp",
           "net_group.cpp", "streams.cpp", "multi_gpu.cpp", "c_api.cpp", "kernels.hip"]
HEADERS = ["formats.h", "net.h", "net_internal.h", "kernels.h", os.path.join("..", "..", "include", "deepcut_hip.h")]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build_lib(force=False, verbose=True):
    """Per-file incremental: an object is rebuilt when its source or any header is newer (kernels.hip takes minutes, the host
    translation units seconds each — they compile in parallel)."""
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", src + ".o")
        objs.append(obj)
        sp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_t, os.path.getmtime(sp)):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
               "-c", sp, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("compilation failed: %s" % ", ".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
