"""Compile dec_w.hip (bf16 build) alone and print each kernel's register figures."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from rift_amd import build as b
unit = sys.argv[1] if len(sys.argv) > 1 else "dec_w"
for o, cmd in b.compile_commands(extra=["-Rpass-analysis=kernel-resource-usage"]):
    if unit + "_bf" in os.path.basename(o) or (unit + ".") in os.path.basename(o) and "bf" in os.path.basename(o):
        cmd = [c for c in cmd]
        cmd[cmd.index("-o") + 1] = "/tmp/decw_probe.o"
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode: print(r.stderr[-3000:]); sys.exit(1)
        cur = {}
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (.*?) \[-Rpass", line)
            if not m: continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"): cur = {"name": t.split(":", 1)[1].strip()}
            elif ":" in t:
                k, v = t.split(":", 1); cur[k.strip()] = v.strip()
                if k.strip().startswith("LDS Size"):
                    print(cur["name"][:60], "VGPR", cur.get("VGPRs"), "SGPR", cur.get("SGPRs"), "vspill", cur.get("VGPRs Spill"), "sspill", cur.get("SGPRs Spill"), "scratch", cur.get("ScratchSize [bytes/lane]"))
        break
else:
    print([os.path.basename(o) for o, _ in b.compile_commands()])
