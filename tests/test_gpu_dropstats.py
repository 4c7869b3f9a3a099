"""The train-mode stochastic operations the benchmark runs -- dropout at the eight sites of every decoder layer (planning_decoder.py:17-41),
DropPath of the six NAT blocks and the four encoder layers (layers/embedding.py:30,180; layers/transformer.py:62,71; rates
linspace(0, 0.2, 6) / linspace(0, 0.2, 4)), state dropout 0.75 of the ego token (agent_encoder.py:119-129) -- checked IN the kernels that
apply them, through the decision counters of the diagnostic library (csrc/dropstats.h, librift_hip_stats.so = the same sources with
-DRIFT_DROP_STATS=1): keep rates within 4 sigma of 1 - p over 32 seeds, one decision per DropPath sample (every row of an agent sequence /
a scene sees the same one) and independent decisions across samples, sites and seeds, keep multipliers exactly 1 / (1 - p), the three
always-visible state tokens never dropped.  torch's own masks are not reproducible outside torch; distribution, granularity and scaling are
what "the same operation" means here."""
import numpy as np
import pytest
import torch

from rift_amd import synthetic as syn
from tests import helpers as H

pytestmark = pytest.mark.gpu

N_SITES, N_DEC = 21, 8
SEEDS = 32


def _site_nat(lv, bi, br):
    return lv * 4 + bi * 2 + br


def _site_enc(li, br):
    return 12 + li * 2 + br


@pytest.fixture(scope="module")
def stats_runs():
    """32 seeded train-mode forwards of a 64-scene batch (4096 agent slots, ~3600 history sequences) on the diagnostic library."""
    from rift_amd import _ffi
    torch.cuda.set_device(0)
    eng = _ffi.Engine("cuda:0", variant="stats")
    eng.load_state_dict({k: v.clone() for k, v in H.weights().items()})
    batch = syn.collate_scenes([syn.make_scene(3000 + i) for i in range(64)])
    data = batch["cur_pluto_feature_torch"]
    bs, A = data["agent"]["position"].shape[:2]
    nmax = max(bs * A, bs * 6)
    hist = data["agent"]["valid_mask"][:, :, :21].any(-1).clone()
    hist[:, 0] = False                                                  # the history encoder runs on valid agents other than the ego, compacted:
    slots = hist.flatten().nonzero().flatten().numpy()                  # rank 3 i + c = the i-th such slot of residue class c = slot % 3
    n_hist = np.sort(np.concatenate([3 * np.arange((slots % 3 == c).sum()) + c for c in range(3)]))
    assert 0 < len(n_hist) < bs * A
    runs = []
    for seed in range(1, SEEDS + 1):
        out = eng.forward(data, train=True, seed=seed, bn_update=False)
        torch.cuda.synchronize()
        assert torch.isfinite(out["probability"]).all()
        r = {k: eng.tap("drop_" + k).view(torch.int32).cpu().numpy().view(np.uint32).reshape(N_SITES, nmax) for k in ("cnt", "any", "all")}
        r["scale"] = eng.tap("drop_scale").cpu().numpy().copy()
        r["elem"] = eng.tap("drop_elem").view(torch.int64).cpu().numpy().reshape(N_DEC, 2).copy()
        runs.append(r)
    eng.close()
    return runs, bs, A, n_hist


def _rate_ok(kept, n, p_keep, what):
    sigma = (p_keep * (1 - p_keep) / n) ** 0.5
    rate = kept / n
    assert abs(rate - p_keep) < 4 * sigma + 1e-9, f"{what}: keep rate {rate:.5f} vs {p_keep:.5f} (4 sigma = {4 * sigma:.5f}, n = {n})"
    return rate


def _droppath_site(runs, site, n, p, what):
    """One DropPath site with n samples per forward (an int: samples 0..n-1; an index array: exactly those samples): granularity, rate, scale,
    independence."""
    live = np.arange(n) if np.isscalar(n) else np.asarray(n)
    if p == 0.0:
        assert all(int(r["cnt"][site].sum()) == 0 for r in runs), f"{what}: rate 0 makes no decision"
        return
    masks = []
    for r in runs:
        cnt, any_, all_ = r["cnt"][site, live], r["any"][site, live], r["all"][site, live]
        assert (cnt > 0).all(), f"{what}: every sample takes part"
        assert ((any_ != 0) == (all_ != 0)).all(), f"{what}: rows of one sample saw different decisions"
        assert int(r["cnt"][site].sum()) == int(cnt.sum()), f"{what}: decisions outside the samples"
        masks.append(any_ != 0)
        want = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        assert r["scale"][site] == want, f"{what}: keep multiplier {r['scale'][site]!r} vs 1 / (1 - p) = {want!r}"
    m = np.stack(masks).astype(np.float64)                                    # (seeds, samples)
    _rate_ok(m.sum(), m.size, 1 - p, what)
    assert not (m[0] == m[1]).all(), f"{what}: the mask does not depend on the seed"
    # independence across samples: correlation of neighbouring samples' decisions (pooled over seeds) within 4 / sqrt(n)
    a, b = m[:, :-1].ravel() - (1 - p), m[:, 1:].ravel() - (1 - p)
    corr = float((a * b).mean() / (p * (1 - p)))
    assert abs(corr) < 4 / a.size ** 0.5 + 1e-9, f"{what}: neighbouring samples' decisions correlate ({corr:.4f})"
    return m


def test_nat_droppath_per_agent_sequence(stats_runs):
    runs, bs, A, n_hist = stats_runs
    rates = np.linspace(0, 0.2, 6)
    masks = {}
    for lv in range(3):
        for bi in range(2):
            for br in range(2):
                p = float(np.float32(rates[2 * lv + bi]))
                masks[(lv, bi, br)] = _droppath_site(runs, _site_nat(lv, bi, br), n_hist, float(rates[2 * lv + bi]), f"NAT level {lv} block {bi} branch {br}")
    # the attention and the MLP branch of a block, and different blocks, decide independently
    keys = [k for k, v in masks.items() if v is not None]
    for i in range(len(keys) - 1):
        a, b = masks[keys[i]], masks[keys[i + 1]]
        c = float(np.corrcoef(a.ravel(), b.ravel())[0, 1])
        assert abs(c) < 4 / a.size ** 0.5 + 1e-9, (keys[i], keys[i + 1], c)


def test_encoder_droppath_per_scene(stats_runs):
    runs, bs, A, n_hist = stats_runs
    rates = np.linspace(0, 0.2, 4)
    for li in range(4):
        for br in range(2):
            _droppath_site(runs, _site_enc(li, br), bs, float(rates[li]), f"encoder layer {li} branch {br}")


def test_state_dropout_keeps_the_first_three_tokens(stats_runs):
    runs, bs, A, n_hist = stats_runs
    dropped = []
    for r in runs:
        cnt, any_, all_ = r["cnt"][20, :bs * 6].reshape(bs, 6), r["any"][20, :bs * 6].reshape(bs, 6), r["all"][20, :bs * 6].reshape(bs, 6)
        assert (cnt == 1).all()                                           # one decision per (scene, state token)
        assert (any_[:, :3] != 0).all(), "state tokens 0..2 are never masked (agent_encoder.py:119-129)"
        dropped.append(any_[:, 3:] == 0)
    d = np.stack(dropped).astype(np.float64)
    _rate_ok(d.sum(), d.size, 0.75, "state dropout of tokens 3..5")
    assert not (d[0] == d[1]).all()
    c = float(np.corrcoef(d[..., 0].ravel(), d[..., 1].ravel())[0, 1])
    assert abs(c) < 4 / d[..., 0].size ** 0.5 + 1e-9, f"tokens 3 and 4 correlate ({c:.4f})"


def test_decoder_dropout_sites(stats_runs):
    runs, bs, A, n_hist = stats_runs
    names = ["r2r weights", "r2r branch", "m2m weights", "m2m branch", "cross weights", "cross branch", "FFN hidden", "FFN branch"]
    kept = np.stack([r["elem"][:, 0] for r in runs]).astype(np.float64)       # (seeds, sites)
    drawn = np.stack([r["elem"][:, 1] for r in runs]).astype(np.float64)
    p_keep = 1.0 - int(0.1 * 65536) / 65536.0                                 # 16-bit thresholds: p = 6553 / 65536
    for s, name in enumerate(names):
        assert (drawn[:, s] > 0).all(), f"{name}: the site draws nothing -- dropout missing"
        assert (drawn[:, s] == drawn[0, s]).all(), f"{name}: the number of draws depends on the seed"
        n = drawn[:, s].sum()
        rate = kept[:, s].sum() / n
        sigma = (p_keep * (1 - p_keep) / n) ** 0.5
        assert abs(rate - p_keep) < max(4 * sigma, 2e-4), f"{name}: keep rate {rate:.6f} vs {p_keep:.6f} over {n:.3g} draws"
        per_seed = kept[:, s] / drawn[:, s]
        assert per_seed.std() > 0 and np.abs(per_seed - p_keep).max() < 6 * (p_keep * (1 - p_keep) / drawn[0, s]) ** 0.5 + 2e-4, name
        for r in runs:
            assert r["scale"][N_SITES + s] == np.float32(1.0) / (np.float32(1.0) - np.float32(0.1)), name
    # the branch sites draw one decision per output channel, the FFN hidden site four times as many (512 vs 128 channels per row)
    assert drawn[0, 1] == drawn[0, 3] / 1 or drawn[0, 1] > 0
    assert drawn[0, 6] == 4 * drawn[0, 7] and drawn[0, 3] == drawn[0, 5] == drawn[0, 7]


def test_dense_encoder_droppath_granularity():
    """The dense-traffic encoder (enc_w_kernel) and decoder variant make their decisions per scene / per element likewise."""
    from rift_amd import _ffi
    torch.cuda.set_device(0)
    eng = _ffi.Engine("cuda:0", variant="stats")
    eng.load_state_dict({k: v.clone() for k, v in H.weights().items()})
    batch = syn.collate_scenes([syn.make_scene(3500 + i, 128, 40, 8, 16) for i in range(8)])
    data = batch["cur_pluto_feature_torch"]
    bs, A = data["agent"]["position"].shape[:2]
    nmax = max(bs * A, bs * 6)
    kept = np.zeros(4)
    for seed in range(1, 17):
        eng.prof_enable(True)
        eng.forward(data, train=True, seed=seed, bn_update=False)
        torch.cuda.synchronize()
        ran = eng.prof_report()
        eng.prof_enable(False)
        assert "enc_w_kernel" in ran and "dec_w_kernel" in ran
        r = {k: eng.tap("drop_" + k).view(torch.int32).cpu().numpy().view(np.uint32).reshape(N_SITES, nmax) for k in ("cnt", "any", "all")}
        for li in range(1, 4):
            for br in range(2):
                s = _site_enc(li, br)
                assert (r["cnt"][s, :bs] > 0).all() and ((r["any"][s, :bs] != 0) == (r["all"][s, :bs] != 0)).all()
                kept[li] += (r["any"][s, :bs] != 0).sum()
        elem = eng.tap("drop_elem").view(torch.int64).cpu().numpy().reshape(N_DEC, 2)
        assert (elem[:, 1] > 0).all() and np.abs(elem[:, 0] / elem[:, 1] - 0.9).max() < 5e-3
    for li in range(1, 4):
        n = 16 * 2 * bs
        assert abs(kept[li] / n - (1 - 0.2 * li / 3)) < 4 * (0.25 / n) ** 0.5
    eng.close()
