"""In-tree build of the two shared libraries (and, optionally, the oracle).

    python -m raytracer_amd.build            # librtx_host.so + librt_hip.so (gfx950)
    python -m raytracer_amd.build --oracle   # also oracle/liboracle.so and, where /root/reference
                                             # exists, oracle/_ref/ref_parity + ref_perf

hipcc cross-compiles gfx950 without a GPU.  Outputs land in raytracer_amd/lib/ (git-ignored; they
travel to the GPU box with the tree).  Flags that matter for parity:
  -ffp-contract=off   no FMA contraction in traversal / intersection / shading / SAH build
  (hipcc default)     correctly rounded f32 divide and sqrt, denormals kept
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(REPO, "include")

CLANGXX = os.environ.get("RTX_CLANGXX", "/opt/rocm/lib/llvm/bin/clang++")
HIPCC = os.environ.get("RTX_HIPCC", shutil.which("hipcc") or "/opt/rocm/bin/hipcc")

HOST_FLAGS = ["-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall", "-Wextra",
              "-Wno-missing-field-initializers"]
HIP_FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
             "-Wall", "-Wno-unused-function"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_host(force=False):
    out = os.path.join(LIB, "librtx_host.so")
    srcs = [os.path.join(CSRC, "host", "rt_host.cpp"), os.path.join(INC, "rt_host.h"), os.path.join(INC, "rtx.h")]
    if force or _newer(out, srcs):
        os.makedirs(LIB, exist_ok=True)
        _run([CLANGXX, *HOST_FLAGS, "-I" + INC, srcs[0], "-o", out, "-lpthread", "-ldl"])
    return out


def build_hip(force=False, extra=()):
    out = os.path.join(LIB, "librt_hip.so")
    srcs = [os.path.join(CSRC, "rt_hip.hip"), os.path.join(CSRC, "rtx_device.h"), os.path.join(CSRC, "kd_build.h"), os.path.join(INC, "rtx.h")]
    if force or _newer(out, srcs):
        os.makedirs(LIB, exist_ok=True)
        _run([HIPCC, *HIP_FLAGS, *extra, "-I" + INC, "-I" + CSRC, srcs[0], "-o", out])
    return out


def build_oracle(force=False):
    odir = os.path.join(REPO, "oracle")
    if force:
        subprocess.call(["make", "-C", odir, "clean"])
    _run(["make", "-C", odir, "oracle", "ref"])


def main(argv):
    force = "--force" in argv
    build_host(force)
    build_hip(force)
    if "--oracle" in argv:
        build_oracle(force)


if __name__ == "__main__":
    main(sys.argv[1:])
