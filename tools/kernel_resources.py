"""Per-kernel register / LDS / spill table of librift_hip.so from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).
    python tools/kernel_resources.py [filter]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rift_amd import build as b

out = ""
for tu, extra in (("engine.hip", []), ("dec_w.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"]), ("nat_l2w.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"]), ("nat_l01w.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"]),
                          ("enc_w.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"]),
                          ("pe_w.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"]),
                          ("fo_w.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"]),
                          ("enc112.hip", ["-fno-honor-nans", "-mno-amdgpu-ieee"])):
    cmd = [b.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Rpass-analysis=kernel-resource-usage"] + extra + \
          [os.path.join(b.CSRC, tu), "-o", "/tmp/_kr.o"] + sys.argv[2:]
    out += subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void rift::", "").replace("rift::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print(f"{r['name'][:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPR Spill','?'):>6s} "
          f"{r.get('SGPR Spill', r.get('SGPRs Spill','?')):>6s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
