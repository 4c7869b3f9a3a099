"""Build librift_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librift_hip.so")


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
                  + [os.path.join(os.path.dirname(HERE), "include", "rift_hip.h")])


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sources()
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    # engine.hip, dec_w.hip, nat_l2w.hip, enc_w.hip, pe_w.hip and fo_w.hip are separate translation units; the wave-private streaming kernels are built with
    # `-fno-honor-nans -mno-amdgpu-ieee`: no IEEE-mode canonicalisation (`v_max_f32 x, x, x` in front of every fmaxf on an MFMA result); they test no NaN.
    common = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    obj = os.path.join(HERE, "_obj")
    os.makedirs(obj, exist_ok=True)
    steps = [common + ["-c", os.path.join(CSRC, "engine.hip"), "-o", os.path.join(obj, "engine.o")],
             common + ["-c", "-fno-honor-nans", "-mno-amdgpu-ieee", os.path.join(CSRC, "dec_w.hip"), "-o", os.path.join(obj, "dec_w.o")],
             common + ["-c", "-fno-honor-nans", "-mno-amdgpu-ieee", os.path.join(CSRC, "nat_l2w.hip"), "-o", os.path.join(obj, "nat_l2w.o")],
             common + ["-c", "-fno-honor-nans", "-mno-amdgpu-ieee", os.path.join(CSRC, "enc_w.hip"), "-o", os.path.join(obj, "enc_w.o")],
             common + ["-c", "-fno-honor-nans", "-mno-amdgpu-ieee", os.path.join(CSRC, "pe_w.hip"), "-o", os.path.join(obj, "pe_w.o")],
             common + ["-c", "-fno-honor-nans", "-mno-amdgpu-ieee", os.path.join(CSRC, "fo_w.hip"), "-o", os.path.join(obj, "fo_w.o")],
             common + ["-shared", os.path.join(obj, "engine.o"), os.path.join(obj, "dec_w.o"), os.path.join(obj, "nat_l2w.o"), os.path.join(obj, "enc_w.o"), os.path.join(obj, "pe_w.o"), os.path.join(obj, "fo_w.o"), "-o", LIB]]
    for cmd in steps:
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True)
