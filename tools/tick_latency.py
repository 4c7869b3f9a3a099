"""Rollout-side tick latency: `get_action` of the plain PLUTO policy (eval forward with every output -> candidate trimming
-> PID) and of RIFTPluto in train mode (that + the reference model's logits + the device-side group advantage) for K CBVs of one
environment at the CARLA shapes (49 agent slots, 60 polygon slots, 1..6 reference lines): host wall time per tick, median of N ticks, and
where it goes (collate + H2D | forward issue | first read-back = device time | per-CBV decisions).
    python tools/tick_latency.py [--ticks 60] [--cbvs 1,2,4,8]"""
import argparse, json, os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from rift_amd import synthetic as syn
from rift_amd.planning import CBV_POLICY_LIST
from rift_amd.planning.pluto.feature_builder.pluto_feature import PlutoFeature
from rift_amd.planning.pluto.pluto import CBVStateSource, CenterState
from tests import helpers as H


class Recorded(CBVStateSource):
    def center_state(self, env_id, cbv_id):
        return CenterState(10.0 + cbv_id, -5.0, 0.3, 6.0 + 0.1 * cbv_id, 2.0, 4.6)

    def nearby_actor_states(self, env_id, cbv_id):
        return H.other_vehicle_inputs(seed=100 + cbv_id, N=4)

    def off_road_raster(self, env_id, cbv_id):
        mask = np.ones((400, 400), dtype=np.uint8)
        mask[150:250, :300] = 0
        return mask, (10.0 + cbv_id, -5.0, 0.3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=60)
    ap.add_argument("--cbvs", default="1,2,4,8")
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--profile", default="", help="policy/K to run under cProfile instead, e.g. rift_pluto/4")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    sd = H.weights()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, mode in (("pluto", "eval"), ("rift_pluto", "train")):
            cfg = {'num_scenario': 1, 'ROOT_DIR': tmp, 'model_path': 'ckpt', 'device': 'cuda:0', 'state_source': Recorded(),
                   'compute_precision': args.precision}
            pol = CBV_POLICY_LIST[name](cfg, None)
            pol.pluto_model.load_state_dict(sd)
            if hasattr(pol, "ref_model") and pol.ref_model is not None:
                pol.ref_model.load_state_dict(sd)
            pol.set_mode(mode)
            if args.profile and args.profile.split("/")[0] != name:
                continue
            for K in ([int(args.profile.split("/")[1])] if args.profile else [int(k) for k in args.cbvs.split(",")]):
                ids = list(range(1, K + 1))
                if args.profile:
                    import cProfile, pstats
                    ticks = []
                    for t in range(25):
                        feats = {c: syn.make_scene(7000 + 16 * t + c, num_agents=49, num_polygons=60, r_min=1, r_max=6)["feature"] for c in ids}
                        ticks.append({c: {'raw_pluto_feature': PlutoFeature(data=feats[c])} for c in ids})
                    for obs in ticks[:5]:
                        pol.get_action([obs], [{'env_id': 0}], deterministic=False)
                    pr = cProfile.Profile()
                    pr.enable()
                    for obs in ticks[5:]:
                        pol.get_action([obs], [{'env_id': 0}], deterministic=False)
                    pr.disable()
                    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
                    return
                times = []
                for t in range(args.ticks + 5):
                    feats = {c: syn.make_scene(7000 + 16 * t + c, num_agents=49, num_polygons=60, r_min=1, r_max=6)["feature"] for c in ids}
                    obs = {c: {'raw_pluto_feature': PlutoFeature(data=feats[c])} for c in ids}
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    pol.get_action([obs], [{'env_id': 0}], deterministic=False)
                    times.append(time.perf_counter() - t0)
                ms = np.array(times[5:]) * 1e3
                out[f"{name}/{mode}/K={K}"] = {"median_ms": round(float(np.median(ms)), 3), "p90_ms": round(float(np.percentile(ms, 90)), 3),
                                                "min_ms": round(float(ms.min()), 3)}
                print(f"{name:10s} {mode:5s} K={K}: median {np.median(ms):.3f} ms  p90 {np.percentile(ms, 90):.3f}  min {ms.min():.3f}", flush=True)
            pol.pluto_model.release_engine() if hasattr(pol.pluto_model, "release_engine") else None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
