"""Timing ablation of the row-resident GEMM kernel on representative shapes (diagnostic only)."""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

SHAPES = [(64, 128, 128, False, 0, True), (18432, 128, 128, False, 0, True), (64, 128, 512, True, 2, False),
          (18432, 128, 512, True, 2, False), (327680, 32, 96, True, 0, False), (81920, 128, 384, True, 0, False),
          (18432, 512, 128, False, 0, True), (184320, 256, 256, False, 0, False)]


def run(dbg):
    import torch
    from rift_amd import _ffi
    eng = _ffi.Engine("cuda:0")
    res = {}
    g = torch.Generator().manual_seed(0)
    for (M, K, N, ln, act, resid) in SHAPES:
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        lw = torch.ones(K).cuda() if ln else None
        lb = torch.zeros(K).cuda() if ln else None
        r = torch.randn(M, N, generator=g).cuda() if resid else None
        us = eng.op_linear_bench(x, w, b, ln_w=lw, ln_b=lb, act=act, residual=r, reps=50)
        res[f"{M}x{N}x{K}{':ln' if ln else ''}{':gelu' if act == 2 else ''}{':res' if resid else ''}"] = round(us, 1)
    print(dbg, json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]))
    else:
        for d in [int(x) for x in os.environ.get("DBGS", "0,1,2,4,8,16,31").split(",")]:
            subprocess.run([sys.executable, __file__, str(d)], env=dict(os.environ, RIFT_GEMM_DBG=str(d)))
