"""Build libhipad.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python hip-ad_amd/build.py [--force]

The .so lands in hip-ad_amd/csrc/libhipad.so (git-ignored, travels to the GPU box with the
gpurun snapshot).  hipcc cross-compiles for gfx950 without a GPU.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libhipad.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in sources():
        obj = src[:-4] + ".o"
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), *[os.path.getmtime(h) for h in
                                          glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))]):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-fno-fast-math", "-Wno-pass-failed",
                   "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv))
