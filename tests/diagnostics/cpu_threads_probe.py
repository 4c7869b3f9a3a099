import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rift_amd import synthetic as syn
from oracle import pluto_ref
from tests import helpers as H
sd = H.weights()
scenes = [syn.make_scene(i) for i in range(128)]
batch = syn.collate_scenes(scenes)
data = batch["cur_pluto_feature_torch"]
print("cpus", os.cpu_count())
for nt in (4, 8, 16, 32, 64):
    torch.set_num_threads(nt)
    pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=True)
    t = time.perf_counter()
    pluto_ref.planning_model_forward(sd, data, train_bn=True, need_traj=True)
    dt = time.perf_counter() - t
    print(nt, "threads:", round(128 / dt, 1), "scenes/s fwd", flush=True)
