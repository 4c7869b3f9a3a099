"""PlutoFeatureBuilder: the mirror's tensorisation of recorded CARLA readings against
tests/golden/feature_builder.npz = the REFERENCE's own PlutoFeatureBuilder / PlutoFeature.normalize / CostMapManager run on the same
readings (tests/golden/gen_golden.py: gen_feature_builder).  CPU only; no CARLA, shapely or OpenCV involved on either side."""
import os

import numpy as np
import pytest

from rift_amd.planning.pluto.feature_builder.pluto_feature_builder import PlutoFeatureBuilder
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), "golden", "feature_builder.npz")


def _flat(prefix, data, out):
    for k, v in data.items():
        if isinstance(v, dict):
            _flat(f"{prefix}/{k}", v, out)
        elif isinstance(v, (np.ndarray, np.generic, float, int)):
            out[f"{prefix}/{k}"] = np.asarray(v)
        elif isinstance(v, list) and all(isinstance(x, (str, int, np.integer)) for x in v):
            out[f"{prefix}/{k}"] = np.asarray([str(x) for x in v])
    return out


@pytest.mark.parametrize("case", ["busy", "alone"])
def test_builder_matches_the_reference_on_recorded_readings(case):
    gold = {k: v for k, v in np.load(GOLD).items() if k.startswith(case + "/")}
    w = H.feature_builder_world(case)                       # names as plain strings: the mirror never sees the reference's enums
    b = PlutoFeatureBuilder(w.config, w.planner, w.provider, fill_polygon=H.bbox_fill, fill_convex_polygon=H.bbox_fill)
    feature, route_ids, lines, elements, wp = b.build_feature(w.center, w.nearby, mode=w.mode)
    got = _flat(case, feature.data, {})
    got[f"{case}/route_road_ids"] = np.asarray(route_ids["road_ids"])
    got[f"{case}/n_lines"] = np.asarray(len(lines))
    assert set(got) == set(gold), set(got) ^ set(gold)
    for k, ref in gold.items():
        v = got[k]
        assert v.shape == ref.shape and v.dtype == ref.dtype, (k, v.shape, v.dtype, ref.shape, ref.dtype)
        assert np.array_equal(v, ref), (k, float(np.abs(v.astype(np.float64) - ref.astype(np.float64)).max()) if v.dtype.kind == "f" else "differs")
    assert elements == ["route-elements"] and wp == ["interaction-wp"]
    d = feature.data
    if case == "busy":
        assert d["agent"]["position"].shape == (5, 21, 2) and d["cost_maps"].shape == (200, 200, 1) and d["cost_maps"].dtype == np.float16
        assert d["agent_tokens"] == ["ego", 201, 202, 200, 203]                       # neighbours by current distance
        assert d["agent"]["valid_mask"][3 - 0].sum() in (7, 21)                       # the 7-step actor fills steps 0..6 only
        assert d["map"]["point_position"].shape[0] == 8                               # the lane 500 m away is cropped by normalize
        assert d["reference_line"]["valid_mask"].sum(-1).tolist() == [15, 120, 2]     # every 4th point, capped at 120
    else:
        assert d["agent"]["position"].shape == (1, 21, 2) and "cost_maps" not in d
        assert not d["reference_line"]["valid_mask"].any() and not feature.is_valid


def test_builder_collates_and_feeds_the_policy_schema():
    """Two built observations collate like any other PlutoFeature (the batch `get_action` runs): padded agent / map / reference-line
    groups, float32 tensors, bool masks, int8 categories."""
    import torch
    from rift_amd.planning.pluto.feature_builder.pluto_feature import PlutoFeature
    feats = []
    for case in ("busy", "alone"):
        w = H.feature_builder_world(case)
        b = PlutoFeatureBuilder(w.config, w.planner, w.provider, fill_polygon=H.bbox_fill, fill_convex_polygon=H.bbox_fill)
        f = b.build_feature(w.center, w.nearby, mode="eval")[0]
        f.data.pop("agent_tokens")
        feats.append(f.to_feature_tensor())
    batch = PlutoFeature.collate(feats).data
    assert batch["agent"]["position"].shape == (2, 5, 21, 2) and batch["agent"]["position"].dtype == torch.float32
    assert batch["agent"]["valid_mask"].dtype == torch.bool and batch["agent"]["category"].dtype == torch.int8
    assert batch["map"]["point_position"].shape == (2, 8, 3, 20, 2) and batch["reference_line"]["position"].shape == (2, 3, 120, 2)
    assert batch["current_state"].shape == (2, 7) and float(batch["current_state"][:, :3].abs().max()) == 0.0


def test_cost_maps_need_the_raster_fills():
    w = H.feature_builder_world("busy")
    b = PlutoFeatureBuilder(w.config, w.planner, w.provider)
    with pytest.raises(RuntimeError, match="raster fills"):
        b.build_feature(w.center, w.nearby, mode="train_cbv")
