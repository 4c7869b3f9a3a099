"""Scan the gfx950 ISA of every translation unit (both operand-format builds) for operand hazards that hipcc does not pad INSIDE or right behind inline asm:
  * a VALU write of a VGPR followed within fewer than 2 wait states by an MFMA reading it as A / B / C, by v_permlane*_swap or by a DPP
    instruction reading it (tools/ubench/cvt_mfma_hazard.hip: the MFMA case measured on MI355X -- 0 or 1 states read the OLD register);
  * v_readfirstlane writing an SGPR followed within fewer than 5 wait states by global_load_lds / buffer / global instructions using it.
  * (round 5) an MFMA result read by a VALU instruction INSIDE an asm statement within fewer than 12 wait states: the states between an MFMA
    write and a VALU read are software's to insert, hipcc pads its own instructions only (a packed-fp16 GELU that converted accumulators in an
    asm statement read stale registers -> NaN).  Conversions fed by accumulators go through common.h: pack_h2c (hipcc's own instruction).
The wave-private kernels hide `v_cvt_pk_bf16_f32` / `v_cvt_pk_f16_f32`, the LDS-DMA and the group GEMM in asm statements; this scan is how their padding is checked.
    python tools/checks/isa_hazard_scan.py            # compiles (no GPU needed) and scans; exit status 1 on a finding"""
import os, re, subprocess, sys, tempfile
from collections import Counter
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from rift_amd import build as b


def regs(tok, kind="v"):
    tok = tok.rstrip(",")
    m = re.match(kind + r"\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(kind + r"(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def states(l):
    return int(l.split()[1]) + 1 if l.startswith("s_nop") else 1


def scan(path):
    ins, in_asm_of, in_asm = [], [], False
    for l in open(path).read().split("\n"):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        if l.startswith("\t") and t and not t.startswith(";") and not t.startswith("."):
            ins.append(t); in_asm_of.append(in_asm)
    bad = []
    for k, l in enumerate(ins):          # MFMA result -> VALU read inside an asm statement
        toks = l.split()
        if not toks[0].startswith("v_mfma") or len(toks) < 2:
            continue
        dst, st = regs(toks[1]), 0
        for d in range(1, 14):
            if k + d >= len(ins) or st >= 12:
                break
            l2 = ins[k + d]
            op2, t2 = l2.split()[0], l2.split()[1:]
            if in_asm_of[k + d] and op2.startswith("v_") and not op2.startswith("v_mfma") and dst & set().union(*[regs(t) for t in t2[1:]] or [set()]):
                bad.append(("mfma->asm-valu", st, l, l2))
            if op2.startswith("v_") and dst & regs(t2[0] if t2 else ""):      # the register is rewritten: the MFMA's value is gone
                break
            st += states(l2)
    for k, l in enumerate(ins):
        toks = l.split()
        op = toks[0]
        if op.startswith("v_readfirstlane") and len(toks) > 1:
            dst, st = regs(toks[1], "s"), 0
            for d in range(1, 7):
                if k + d >= len(ins) or st >= 5:
                    break
                l2 = ins[k + d]
                if l2.split()[0].startswith(("global_", "buffer_", "scratch_")) and dst & set().union(*[regs(t, "s") for t in l2.split()[1:]]):
                    bad.append(("sgpr->vmem", st, l, l2))
                st += states(l2)
            continue
        if not op.startswith("v_") or op.startswith(("v_mfma", "v_cmp")) or len(toks) < 2:
            continue
        dst = regs(toks[1])
        if not dst:
            continue
        st = 0
        for d in range(1, 4):
            if k + d >= len(ins) or st >= 2:
                break
            l2 = ins[k + d]
            op2, t2 = l2.split()[0], l2.split()[1:]
            if op2.startswith("v_mfma") and dst & set().union(*[regs(t) for t in t2[1:4]]):
                bad.append(("valu->mfma", st, l, l2))
            if "permlane" in op2 and dst & set().union(*[regs(t) for t in t2[0:2]]):
                bad.append(("valu->permlane", st, l, l2))
            if ("quad_perm" in l2 or "row_" in l2 or "_dpp" in op2) and dst & set().union(*[regs(t) for t in t2[1:3]]):
                bad.append(("valu->dpp", st, l, l2))
            st += states(l2)
    return bad


def main():
    """Both builds of the engine (bf16 and fp16 MFMA operands) with the flags build.py compiles them with."""
    from concurrent.futures import ThreadPoolExecutor
    total = 0
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for obj, cmd in b.compile_commands():
            if obj.endswith("abi.o"):
                continue
            out = os.path.join(tmp, os.path.basename(obj)[:-2] + ".s")
            cmd = [x for x in cmd if x != "-c" and x != "-fPIC"]
            cmd[cmd.index("-o") + 1] = out
            jobs.append((out, cmd + ["-S", "--cuda-device-only", "-w"]))
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
            list(ex.map(lambda j: subprocess.check_call(j[1]), jobs))
        for out, _ in jobs:
            bad = scan(out)
            print(f"{os.path.basename(out)}: {sum(1 for _ in open(out))} lines, findings {dict(Counter(x[0] for x in bad))}")
            for x in bad[:10]:
                print("   ", x)
            total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
