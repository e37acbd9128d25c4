"""Build libcbgx.so (hipcc, gfx950) in-tree: cbgbench_amd/lib/libcbgx.so, and the test-only cross-check build
cbgbench_amd/lib/libcbgx_xcheck.so (same sources + the first-generation VALU kernels of tests/xcheck/csrc/, -DCBGX_XCHECK;
include/cbgx_xcheck.h).

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
XCHECK_LIBPATH = os.path.join(LIBDIR, "libcbgx_xcheck.so")
ARCH = "gfx950"
# first-generation kernels: test-only sources (tests/xcheck/csrc/), compiled into libcbgx_xcheck.so only
XCHECK_CSRC = os.path.join(HERE, "..", "tests", "xcheck", "csrc")


def sources(xcheck=False):
    src = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    return src + sorted(glob.glob(os.path.join(XCHECK_CSRC, "*.hip"))) if xcheck else src


def _headers():
    inc = os.path.join(HERE, "..", "include")
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(inc, "cbgx.h"), os.path.join(inc, "cbgx_xcheck.h")]


def _stale(xcheck=False):
    lib = XCHECK_LIBPATH if xcheck else LIBPATH
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(p) > t for p in sources(xcheck) + _headers())


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libcbgx.so")
    return exe


def build_native(force=False, verbose=False, xcheck=False, ablate=False):
    """Compile the HIP sources for gfx950 and link libcbgx.so (xcheck=True: libcbgx_xcheck.so; ablate=True:
    libcbgx_ablate.so with the timing-ablation switches of scripts/abl_bwd.sh, wrong results by design).
    Returns the library path."""
    lib = os.path.join(LIBDIR, "libcbgx_ablate.so") if ablate else (XCHECK_LIBPATH if xcheck else LIBPATH)
    if not force and not ablate and not _stale(xcheck):
        return lib
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(CSRC, "build", "ablate" if ablate else ("xcheck" if xcheck else "product"))
    os.makedirs(objdir, exist_ok=True)
    objs = []
    # -munsafe-fp-atomics: fp32 atomicAdd of the backward kernels compiles to the hardware global_atomic_add_f32
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-munsafe-fp-atomics", "-I" + CSRC] + (["-DCBGX_XCHECK"] if xcheck else []) + (["-DCBGX_ABLATE"] if ablate else [])
    newest_header = max(os.path.getmtime(p) for p in _headers())
    jobs = []
    for src in sources(xcheck):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            continue
        cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append(subprocess.Popen(cmd))
    for j in jobs:
        if j.wait() != 0:
            raise subprocess.CalledProcessError(j.returncode, j.args)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
    print(build_native(force=True, verbose=True, xcheck=True))
