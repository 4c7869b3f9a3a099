"""Range of the 16-bit MFMA operand formats, from the oracle (CPU): the largest magnitude any contraction operand of the policy forward
takes on inputs of CARLA magnitude -- the half of tests/test_gpu_parity.py::test_16bit_operands_stay_in_range_on_carla_magnitudes that
needs no GPU."""
import numpy as np

from tests import helpers as H
from tests.diagnostics import precision_study as PS


def test_every_contraction_operand_is_far_inside_the_fp16_range():
    batch = H.carla_magnitude_batch(n=2)
    data = batch["cur_pluto_feature_torch"]
    assert float(data["agent"]["position"].abs().max()) >= 400.0 and float(data["agent"]["velocity"].norm(dim=-1).max()) > 39.0
    assert float(data["map"]["polygon_speed_limit"].max()) > 30.0 and float(data["reference_line"]["position"].abs().max()) > 200.0
    try:
        rng, (prob, qf) = PS.operand_range(H.weights(), batch)
        rng_bn, _ = PS.operand_range(H.weights(), batch, train_bn=True)
    finally:
        PS.uninstall()
    for r in (rng, rng_bn):
        assert {"nat", "pe", "fourier", "enc", "dec0", "dec3", "ego"} <= set(r)
        assert all(np.isfinite(v) for v in r.values())
        assert max(r.values()) < 0.25 * 65504                          # two binades below the fp16 maximum ...
        assert max(r.values()) <= 501.0                                 # ... in fact the raw coordinates themselves are the largest operands
        assert max(v for k, v in r.items() if k in ("nat", "enc", "dec0", "dec1", "dec2", "dec3", "heads")) < 64
    assert bool(np.isfinite(prob.numpy()[~(~data["reference_line"]["valid_mask"].any(-1)).numpy()]).all())
    # the hooks are gone again: the oracle's own functions are back
    from oracle import pluto_ref
    import torch.nn.functional as F
    assert pluto_ref.F is F and pluto_ref.mha.__module__ == "oracle.pluto_ref"
