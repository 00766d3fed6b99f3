"""Builds libfsf_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

No torch headers are involved: the library is plain HIP behind `include/fsf_hip.h` and is loaded with
ctypes (`fullysparsefusion_amd._lib`).  hipcc cross-compiles without a GPU, so this runs in the build
container; the resulting .so travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libfsf_hip.so")
ARCH = "gfx950"

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # integer outputs must be bit-exact: no silent fma contraction
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP extension cannot be built")
    return exe


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "fsf_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile_one(src, obj, verbose):
    extra = os.environ.get("FSF_EXTRA_HIPCC_FLAGS", "").split()  # ablation builds (profiling ablations), never set in product runs
    cmd = [_hipcc(), *HIPCC_FLAGS, *extra, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libfsf_hip.so.  Returns the library path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m)
        if stale:
            jobs.append((src, obj))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(lambda j: _compile_one(j[0], j[1], verbose), jobs))
    relink = force or jobs or not os.path.exists(LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs)
    if relink:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
