"""In-tree build of libcirs_hip.so: hipcc --offload-arch=gfx950 over csrc/*.hip (cross-compiles without a GPU).

Every translation unit is compiled to an object file in parallel (csrc/_obj/, git-ignored) and only when it or a header it
can include changed; one link step produces the shared library."""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(_HERE, "libcirs_hip.so")
# -amdgpu-mfma-vgpr-form: MFMA accumulators / operands stay in the architectural VGPRs.  With the default heuristics the fused head
# backward kernel kept its accumulators in AGPRs and spent 140 of ~720 VALU instructions per tile on v_accvgpr_read / _write copies
# (VALU cannot address AGPRs, and on this part VALU time adds to MFMA time: DESIGN.md section 4).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "cirs_hip.h")]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in sources() + _headers() if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers() if os.path.exists(h))
    todo = [s for s in sources() if force or not os.path.exists(_obj(s)) or os.path.getmtime(_obj(s)) < max(os.path.getmtime(s), hdr_t)]

    def compile_one(src):
        cmd = [hipcc] + CFLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    live = {_obj(s) for s in sources()}
    for o in glob.glob(os.path.join(OBJ, "*.o")):   # a removed source must not linger in the library
        if o not in live:
            os.remove(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + sorted(live) + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force=True)
