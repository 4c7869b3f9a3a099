"""Decoder kernel time against the number of 16-row query tiles it processes (RIFT_DEC_MT diagnostic variants): T = a + b * MT separates
the per-phase latency (a: ~110 barrier-separated phases) from the per-tile work (b).  All scenes are generated with R <= r_max so that
every variant is valid on the same batch."""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if len(sys.argv) > 1:
    mt, rmax = int(sys.argv[1]), int(sys.argv[2])
    os.environ["RIFT_DEC_MT"] = str(mt)
    import torch
    from rift_amd import _ffi, synthetic as syn
    from tests import helpers as H
    sd = H.weights()
    batch = syn.collate_scenes([syn.make_scene(i, 64, 20, 1, rmax) for i in range(256)])
    eng = _ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in sd.items()})
    for _ in range(3):
        eng.forward(batch["cur_pluto_feature_torch"], train=True, seed=3, bn_update=False)
    eng.prof_enable(True)
    for _ in range(10):
        eng.forward(batch["cur_pluto_feature_torch"], train=True, seed=3, bn_update=False)
    rep = eng.prof_report()
    print(json.dumps({"mt": mt, "rmax": rmax, "dec_us": rep["dec_fused_kernel"]["ms"] * 100, "enc_us": rep["enc_fused_kernel"]["ms"] * 100}))
else:
    for rmax, mts in ((1, (1, 2, 3, 4, 5)), (2, (2, 3, 5)), (4, (3, 4, 5)), (6, (5,))):
        for mt in mts:
            print(subprocess.run([sys.executable, __file__, str(mt), str(rmax)], capture_output=True, text=True).stdout.strip(), flush=True)
