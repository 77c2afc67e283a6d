"""Build libdeepcut_hip.so (HIP kernels + C-ABI) for gfx950, in-tree.

    python deepcut-cnn_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the
resulting .so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libdeepcut_hip.so")
SOURCES = ["formats.cpp", "hdf5_reader.cpp", "runtime.cpp", "net_init.cpp", "net_lower.cpp", "net_tune.cpp", "net_run.cpp", "net_image.cpp",
           "net_group.cpp", "streams.cpp", "multi_gpu.cpp", "c_api.cpp", "kernels.hip", "wino_f16.hip", "stream1x1.hip", "stem_f16.hip", "stream1x1_f32.hip"]
HEADERS = ["formats.h", "net.h", "net_internal.h", "kernels.h", os.path.join("..", "..", "include", "deepcut_hip.h")]


KERNEL_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm"]
ASM = os.path.join(HERE, "lib", "kernels.gfx950.s")


def _asm_key():
    """What the device assembly of kernels.hip depends on: the two sources, the flags, the compiler."""
    h = hashlib.sha256()
    for f in ("kernels.hip", "kernels.h"):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(KERNEL_FLAGS).encode())
    h.update(_hipcc_version())
    return h.hexdigest()


def _asm_fresh():
    try:
        return os.path.getsize(ASM) > 0 and open(ASM + ".key").read().strip() == _asm_key()
    except OSError:
        return False


def _asm_job():
    """The gfx950 assembly of kernels.hip (device side only, the library's own flags), as a process: what
    tools/check_asm_hazards.py reads.  Started beside the object compile of build_lib so that the CPU test suite
    (tests/test_asm_hazards.py) finds it ready instead of compiling the 2 300-line translation unit a second time."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(ASM), exist_ok=True)
    for f in (ASM, ASM + ".key"):
        if os.path.exists(f):
            os.remove(f)
    return subprocess.Popen([hipcc] + KERNEL_FLAGS + ["--offload-device-only", "-S", os.path.join(CSRC, "kernels.hip"), "-o", ASM + ".tmp"])


def _asm_finish(proc, key):
    if proc.wait() != 0:
        raise RuntimeError("device assembly of kernels.hip failed")
    os.replace(ASM + ".tmp", ASM)
    with open(ASM + ".key", "w") as f:
        f.write(key + "\n")


def device_asm():
    """Path of the gfx950 assembly of kernels.hip, compiled now unless the cached one matches the sources."""
    if not _asm_fresh():
        key = _asm_key()
        _asm_finish(_asm_job(), key)
    return ASM


_HIPCC_VERSION = None


def _hipcc_version():
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        try:
            out = subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            # the compiler's identity only: the version lines.  (The rest of the output names the host — `InstalledDir`, configuration
            # files, on a GPU box the detected agents — and would make the prebuilt library look stale on the box it travels to.)
            _HIPCC_VERSION = b"\n".join(ln for ln in out.splitlines() if b"version" in ln.lower())
        except OSError:
            _HIPCC_VERSION = b""
    return _HIPCC_VERSION


def _obj_key(src):
    """What an object depends on, by CONTENT: its source, every header, the flags, the compiler.  (Round 5 compared mtimes: a checkout
    that rewinds sources under a newer .so shipped a stale library silently — the .so travels prebuilt to the GPU box.)"""
    h = hashlib.sha256()
    for f in [src] + HEADERS:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(KERNEL_FLAGS).encode())
    h.update(_hipcc_version())
    return h.hexdigest()


def _lib_key(keys):
    return hashlib.sha256("\n".join(keys).encode()).hexdigest()


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def _stale():
    if not os.path.exists(OUT):
        return True
    return _read(OUT + ".key") != _lib_key([_obj_key(s) for s in SOURCES])


def build_lib(force=False, verbose=True):
    """Per-file incremental, keyed by content hashes (kernels.hip takes a minute, the host translation units seconds each — they
    compile in parallel): an object is rebuilt when the hash of its source + the headers + the flags + the compiler differs from
    the one recorded beside it, the library is re-linked when any object's key changed."""
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs, jobs, keys = [], [], []
    asm = None
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", src + ".o")
        objs.append(obj)
        key = _obj_key(src)
        keys.append(key)
        sp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and _read(obj + ".key") == key:
            continue
        if os.path.exists(obj + ".key"):
            os.remove(obj + ".key")
        cmd = [hipcc] + KERNEL_FLAGS + ["-Wall", "-Wno-unused-function", "-c", sp, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((src, obj, key, subprocess.Popen(cmd)))
        if src == "kernels.hip" and not _asm_fresh():
            asm = (_asm_job(), _asm_key())
    failed = []
    for src, obj, key, p in jobs:
        if p.wait() != 0:
            failed.append(src)
        else:
            with open(obj + ".key", "w") as f:
                f.write(key + "\n")
    if failed:
        raise RuntimeError("compilation failed: %s" % ", ".join(failed))
    if os.path.exists(OUT + ".key"):
        os.remove(OUT + ".key")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(OUT + ".key", "w") as f:
        f.write(_lib_key(keys) + "\n")
    if asm:
        _asm_finish(*asm)
    return OUT


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
