"""Size-independent properties of the HIP forward at the benchmark's full size (256 scenes x 64 agents x 32 polygons x 6 reference lines,
BASELINE.json configs[1]) and at the dense-traffic size, checked without the CPU oracle so that every kernel variant can afford them:

* ISOLATION: an eval forward has no cross-scene term (pluto_model.py:160-218; train-mode BatchNorm and the r2r padding quirk are the only
  couplings, and the quirk is void when every scene has the same number of valid reference lines).  Replacing every OTHER scene of the batch
  must leave a scene's outputs bit-identical -- a tile grid, a workgroup -> scene mapping or a scratch layout that leaks across scenes
  breaks this, in either arithmetic mode;
* PERMUTATION: permuting the scenes permutes the outputs -- bit for bit in fp32 mode and when the agents of a scene fill whole attention
  tiles; otherwise to bf16 rounding: the wave-private NAT kernels put 3 agents x 5 steps in one 16-column score tile, the agent's slot
  follows from its global index, and a softmax sum / P V product over columns {5..9} associates differently from one over {0..4};
* PADDING: agents and polygons that are invalid at every step are key-padded everywhere (pluto_model.py:176-183): appending them changes
  the tile grids, the token count N (and past 96 tokens the kernel variants) but the valid rows only to rounding (the key columns of the
  polygons move), and not at all when they move by whole 16-key tiles."""
import copy

import pytest
import torch

from rift_amd import synthetic as syn
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from rift_amd import _ffi
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    _ffi.load_library()
    eng = _ffi.Engine("cuda:0")
    eng.load_state_dict({k: v.clone() for k, v in H.weights().items()})
    yield eng
    eng.close()


def _forward(eng, scenes, fp32):
    data = syn.collate_scenes(scenes)["cur_pluto_feature_torch"]
    out = eng.forward(data, train=False, no_drop=True, need_traj=True, fp32=fp32, bn_update=False)
    return out["probability"].cpu(), out["trajectory"].cpu()


@pytest.mark.parametrize("agents,polygons,n", [(64, 32, 256), (128, 40, 24), (17, 9, 32)])
@pytest.mark.parametrize("fp32", [False, True])
def test_a_scene_does_not_see_the_other_scenes_of_its_batch(engine, fp32, agents, polygons, n):
    a = [syn.make_scene(9000 + i, num_agents=agents, num_polygons=polygons, r_min=6, r_max=6) for i in range(n)]
    b = [syn.make_scene(7000 + i, num_agents=agents, num_polygons=polygons, r_min=6, r_max=6) for i in range(n)]
    keep = list(range(0, n, 3))
    for k in keep:
        b[k] = a[k]
    pa, ta = _forward(engine, a, fp32)
    pb, tb = _forward(engine, b, fp32)
    assert torch.isfinite(pa).all() and torch.isfinite(ta).all()
    assert torch.equal(pa[keep], pb[keep]) and torch.equal(ta[keep], tb[keep])
    rest = [i for i in range(n) if i not in keep]
    assert not torch.equal(pa[rest], pb[rest])


@pytest.mark.parametrize("agents,fp32,exact", [(64, True, True), (63, False, True), (64, False, False)])
def test_eval_forward_is_scene_permutation_equivariant_at_benchmark_size(engine, agents, fp32, exact):
    scenes = [syn.make_scene(9000 + i, num_agents=agents, num_polygons=32, r_min=6, r_max=6) for i in range(256)]
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(5)).tolist()
    p0, t0 = _forward(engine, scenes, fp32)
    p1, t1 = _forward(engine, [scenes[i] for i in perm], fp32)
    assert torch.isfinite(p0).all() and torch.isfinite(t0).all()
    if exact:
        assert torch.equal(p1, p0[perm]) and torch.equal(t1, t0[perm])
    else:
        assert float((p1 - p0[perm]).abs().max()) < 5e-2
        assert float((t1 - t0[perm]).abs().max()) < 0.5 * max(1.0, float(t0.abs().max()))


def _without_tail(scene, agents, polygons):
    s = copy.deepcopy(scene)
    a, m = s["feature"]["agent"], s["feature"]["map"]
    for k in a:
        a[k] = a[k][:agents].clone()
    for k in m:
        m[k] = m[k][:polygons].clone()
    return s


@pytest.mark.parametrize("agents,polygons,keep_a,keep_p,exact_bf16", [(64, 32, 48, 24, True), (100, 40, 60, 30, False), (20, 12, 12, 8, False)])
@pytest.mark.parametrize("fp32", [False, True])
def test_all_invalid_agents_and_polygons_do_not_change_the_outputs(engine, fp32, agents, polygons, keep_a, keep_p, exact_bf16):
    """(64, 32) -> (48, 24): every token moves by a whole key tile, bit-identical.  (100, 40) -> (60, 30) crosses the 96-token limit: the
    padded batch runs the dense-traffic encoder, the trimmed one the standard one."""
    full, trimmed = [], []
    for i in range(48):
        s = syn.make_scene(9500 + i, num_agents=agents, num_polygons=polygons, r_min=3, r_max=6)
        s["feature"]["agent"]["valid_mask"][keep_a:] = False
        s["feature"]["map"]["valid_mask"][keep_p:] = False
        full.append(s)
        trimmed.append(_without_tail(s, keep_a, keep_p))
    pf, tf = _forward(engine, full, fp32)
    pt, tt = _forward(engine, trimmed, fp32)
    same_kernels = (1 + agents + polygons <= 96) == (1 + keep_a + keep_p <= 96)
    if same_kernels and (fp32 or exact_bf16):
        assert torch.equal(pf, pt) and torch.equal(tf, tt)
    else:
        valid = pt > -1e5
        assert torch.equal(pf > -1e5, valid)
        assert float((pf - pt)[valid].abs().max()) < (1e-4 if fp32 else 5e-2)
        assert float((tf - tt)[valid].abs().max()) < (2e-3 if fp32 else 0.5) * max(1.0, float(tt[valid].abs().max()))


@pytest.mark.gpu
def test_pipeline_streams_sit_on_different_dispatch_pipes():
    """The update pipeline's three streams and the caller's stream must dispatch beside each other: two hardware queues of one dispatch pipe hand
    out one grid at a time, and a trainer with such a pair ran the 256-scene step at 0.60 - 0.73 instead of 0.575 ms (1 fresh process in ~14
    before round 5's probe; profiles/NOTES_r05.md).  `_dispatches_beside` = the fraction of a chip-filling kernel on a that is still ahead when a
    tiny kernel on b finishes: ~0.8 on different pipes, ~0.1 on one pipe, ~0 on one queue."""
    import itertools
    import torch
    from rift_amd.planning.fine_tuner.rlft import trainer as T
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream(dev)
    chosen = T._pipeline_streams(dev, main, probe=True)
    assert len(chosen) == 3 and len({s.cuda_stream for s in chosen} | {main.cuda_stream}) == 4
    for _ in range(3):
        T._dispatches_beside(main, chosen[0], dev)                 # (warm-up: the first probes of a process are off)
    worst = min(min(T._dispatches_beside(a, b, dev), T._dispatches_beside(b, a, dev)) for a, b in itertools.combinations([main] + list(chosen), 2))
    T._PROBE_BUF.clear()
    print(f"worst pair of the pipeline's four streams: {worst:.2f} of the chip-filling kernel still ahead")
    assert worst > 0.5
