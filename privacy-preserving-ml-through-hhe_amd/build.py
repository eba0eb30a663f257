"""Build libhhe_gfx950.so in-tree with hipcc (gfx950 only; cross-compiles without a GPU)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libhhe_gfx950.so")
SOURCES = ["hhe_kernels.hip", "hhe_api.cpp", "hhe_context.cpp", "hhe_pasta_public.cpp", "hhe_client.cpp", "hhe_seal_wire.cpp"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hip")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
