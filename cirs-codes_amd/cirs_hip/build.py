"""In-tree build of libcirs_hip.so: hipcc --offload-arch=gfx950 over csrc/*.hip (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
OUT = os.path.join(_HERE, "libcirs_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale():
    if not os.path.exists(OUT):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "cirs_hip.h")]
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + sources() + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force=True)
