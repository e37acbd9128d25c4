"""Build libcbgx.so (hipcc, gfx950) in-tree: cbgbench_amd/lib/libcbgx.so.

The library is a plain C-ABI shared object (include/cbgx.h); it links against the HIP runtime by
SONAME (libamdhip64.so.7), which is the one PyTorch-ROCm has already loaded when the Python host
imports it, so both share one runtime and torch streams can be passed straight through.
"""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libcbgx.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cbgx.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libcbgx.so")
    return exe


def build_native(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libcbgx.so. Returns the library path."""
    if not force and not _stale():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    # -munsafe-fp-atomics: fp32 atomicAdd of the backward kernels compiles to the hardware global_atomic_add_f32
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-munsafe-fp-atomics"]
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cbgx.h")]
    newest_header = max(os.path.getmtime(p) for p in headers)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            continue
        cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", LIBPATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIBPATH + ".tmp", LIBPATH)
    return LIBPATH


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
