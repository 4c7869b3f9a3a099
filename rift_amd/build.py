"""Build librift_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library holds TWO builds of the engine -- bf16 and fp16 MFMA operands (csrc/opfmt.h: -DRIFT_OP_F16=0/1, kernels in namespace
rift_bf / rift_hf) -- behind one set of exported `rift_*` symbols (csrc/abi.cpp, generated from include/rift_hip.h by
tools/gen/abi_trampolines.py): a context is created for one format and every call is routed to the build that owns it.
"""
import glob
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librift_hip.so")
LIB_STATS = os.path.join(HERE, "librift_hip_stats.so")     # diagnostic twin (build_stats): the bf16 build with the drop-decision counters of csrc/dropstats.h
OBJ = os.path.join(HERE, "_obj")

# Translation units of one engine build.  The wave-private streaming kernels are built with `-fno-honor-nans -mno-amdgpu-ieee`: no
# IEEE-mode canonicalisation (`v_max_f32 x, x, x` in front of every fmaxf on an MFMA result); they test no NaN themselves -- a NaN / Inf
# in the residual stream is caught by the bit-pattern test of the policy-head kernels (rift_check_finite).
# Spill rule (check_resources): a kernel with VGPR spills whose SGPR spills the backend parks in VGPR lanes was miscompiled by ROCm 7.2
# (round 1, DESIGN.md "A compiler hazard").  No kernel may be in that state: either kind of spill alone is fine (pe_w_kernel: SGPR spills
# only since its body became a template on the polyline length; dec_kv_frag_kernel: VGPR spills only), both at once fail the build --
# remove the spills or build that unit with SPILL_SAFE (SGPR spills to scratch memory: correct, and measured 45 % slower on pe_w_kernel).
FAST = ["-fno-honor-nans", "-mno-amdgpu-ieee"]
SPILL_SAFE = ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"]
UNITS = [("engine", []), ("dec_w", FAST), ("nat_l01w", FAST), ("nat_l2w", FAST), ("enc_w", FAST), ("pe_w", FAST), ("fo_w", FAST), ("enc112", FAST)]
FORMATS = [("bf", 0), ("hf", 1)]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("HIPCC_EXTRA", "").split()      # (HIPCC_EXTRA: diagnostic defines, e.g. -DRIFT_DEC_ARR=1)


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp"))
                  + [os.path.join(os.path.dirname(HERE), "include", "rift_hip.h"), os.path.abspath(__file__)])


def compile_commands(extra=(), stats=False, variant=None):
    """[(object path, command)] of every translation unit of the library; `extra` is appended to each hipcc command.
    stats=True: the diagnostic twin -- the bf16-operand build only, compiled with -DRIFT_DROP_STATS=1.
    variant=(tag, [defines]): an A/B build of the whole library under another name (build_variant)."""
    cmds = []
    sfx, defs = ("_st", ["-DRIFT_DROP_STATS=1"]) if stats else ("", [])
    if variant:
        sfx, defs = "_" + variant[0], list(variant[1])
    for tag, f16 in (FORMATS[:1] if stats else FORMATS):
        for unit, flags in UNITS:
            o = os.path.join(OBJ, f"{unit}_{tag}{sfx}.o")
            if stats and unit == "dec_w":      # the decision counters push the train-mode decoder past 256 VGPRs: the diagnostic twin takes the safe spill path (spill rule above)
                flags = flags + SPILL_SAFE
            cmds.append((o, [hipcc()] + COMMON + defs + [f"-DRIFT_OP_F16={f16}", "-c"] + flags + [os.path.join(CSRC, unit + ".hip"), "-o", o] + list(extra)))
    o = os.path.join(OBJ, f"abi{sfx}.o")
    cmds.append((o, [hipcc()] + COMMON + (["-DRIFT_ABI_BF16_ONLY=1"] if stats else []) + ["-x", "hip", "-c", os.path.join(CSRC, "abi.cpp"), "-o", o] + list(extra)))
    return cmds


def check_resources(remarks: str, spill_safe: bool):
    """Raise when a kernel of a translation unit built WITHOUT SPILL_SAFE has VGPR spills and SGPR spills at once (spill rule above)."""
    bad, cur = [], {}
    for line in remarks.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size"):      # last remark of a kernel
                ss, vs = int(cur.get("SGPRs Spill", "0") or 0), int(cur.get("VGPRs Spill", "0") or 0)
                if ss > 0 and vs > 0 and not spill_safe:
                    bad.append(f"{cur['name']}: {vs} VGPR spills and {ss} SGPR spills parked in VGPR lanes")
    if bad:
        raise RuntimeError("kernel resource check failed (build the unit with SPILL_SAFE or remove the spills):\n  " + "\n  ".join(bad))


def build_stats(force: bool = False, verbose: bool = True) -> str:
    """librift_hip_stats.so: what tests/test_gpu_dropstats.py loads to check the train-mode dropout / DropPath / state-dropout decisions."""
    return build(force, verbose, stats=True)


def build_variant(tag: str, defines, force: bool = False, verbose: bool = False) -> str:
    """librift_hip_<tag>.so = the library compiled with extra defines, for same-box A/B runs (RIFT_LIB=<path> python bench.py ...)."""
    return build(force, verbose, variant=(tag, list(defines)))


def build(force: bool = False, verbose: bool = True, jobs: int = 0, stats: bool = False, variant=None) -> str:
    srcs = sources()
    LIB = LIB_STATS if stats else globals()["LIB"]
    if variant:
        LIB = os.path.join(HERE, f"librift_hip_{variant[0]}.so")
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cmds = compile_commands(extra=["-Rpass-analysis=kernel-resource-usage"], stats=stats, variant=variant)

    def run(item):
        o, cmd = item
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stderr}")
        check_resources(r.stderr, "-amdgpu-spill-sgpr-to-vgpr=0" in cmd)

    with ThreadPoolExecutor(max_workers=jobs or min(len(cmds), os.cpu_count() or 4)) as ex:
        list(ex.map(run, cmds))
    link = [hipcc(), "--offload-arch=gfx950", "-shared"] + [o for o, _ in cmds] + ["-o", LIB]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    build(force=True)
    build_stats(force=True)
