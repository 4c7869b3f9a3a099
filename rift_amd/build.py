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
    # -amdgpu-spill-sgpr-to-vgpr=0: the decoder kernel sits at the 256-VGPR limit; with SGPR spills parked in VGPR lanes (the
    # default) AND VGPR spills in the same kernel, ROCm 7.2's backend produced wrong results whenever the SGPR spill count
    # rose (reproduced three times; identical source is correct with SGPR spills sent to scratch).  Cost: < 2 % on that kernel.
    # dec_w.hip (the wave-private decoder kernel) is its own translation unit WITHOUT that flag: its LDS-DMA weight stream must not meet the
    # `s_waitcnt vmcnt(0)` of a scratch reload at every group boundary, and it has no VGPR spills for the SGPR lanes to collide with.
    common = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    obj = os.path.join(HERE, "_obj")
    os.makedirs(obj, exist_ok=True)
    steps = [common + ["-c", "-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0", os.path.join(CSRC, "engine.hip"), "-o", os.path.join(obj, "engine.o")],
             # no IEEE-mode canonicalisation (`v_max_f32 x, x, x` in front of every fmaxf on an MFMA result): this kernel tests no NaN
             common + ["-c", "-fno-honor-nans", "-mno-amdgpu-ieee", os.path.join(CSRC, "dec_w.hip"), "-o", os.path.join(obj, "dec_w.o")],
             common + ["-c", "-fno-honor-nans", "-mno-amdgpu-ieee", os.path.join(CSRC, "nat_l2w.hip"), "-o", os.path.join(obj, "nat_l2w.o")],
             common + ["-shared", os.path.join(obj, "engine.o"), os.path.join(obj, "dec_w.o"), os.path.join(obj, "nat_l2w.o"), "-o", LIB]]
    for cmd in steps:
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True)
