"""Build libmrn_hip.so (the gfx950 kernel library behind the C ABI of include/mrn_hip.h) in-tree with hipcc.

The library cross-compiles on a GPU-less host; the resulting .so travels with the repository snapshot to
the MI355X box.  `python -m mrn_amd.build` rebuilds it; `build_library()` is what __graft_entry__.build() calls.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libmrn_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-fno-gpu-rdc",
         "-Wno-unused-result"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(path):
    h = hashlib.sha1()
    with open(path, "rb") as f:
        h.update(f.read())
    for hdr in sorted(os.listdir(CSRC)):
        if hdr.endswith((".hpp", ".h")):
            with open(os.path.join(CSRC, hdr), "rb") as f:
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    stamp = obj + ".sha1"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC, "-x", "hip", *FLAGS, "-I", CSRC, "-c", path, "-o", obj]   # (absolute paths stay out of the digest)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build_library(force=False, verbose=True, jobs=4):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = sources()
    for f in os.listdir(OBJ):                     # objects of sources that no longer exist (tools/build_probe.sh links the directory)
        if f.endswith(".o") and f[:-2] not in srcs:
            os.remove(os.path.join(OBJ, f))
            if os.path.exists(os.path.join(OBJ, f + ".sha1")):
                os.remove(os.path.join(OBJ, f + ".sha1"))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = [s for s, (_, r) in zip(srcs, results) if r]
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[mrn_amd.build] {LIB} ({'rebuilt: ' + ', '.join(rebuilt) if rebuilt else 'up to date'})")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
