"""SFTPluto -- update side of rift/cbv/planning/fine_tuner/sft/sft_pluto.py:250-286 and sft_datamodule.py:18-103.

The reference fits `sft_trainer.LightningTrainer` (teacher cross entropy, sft_trainer.py:123-199) on a 90 / 10 split of the full
`CBVRolloutBuffer`, batch 256, 16 epochs, `planning_decoder.pi_head` trainable (sft/config/sft_training.yaml,
datamodule/sft_datamodule.yaml); a sample is (the CBV's PlutoFeature, its `CBVs_teacher_infos` row of five floats -- the teacher's
target speed first, sft_pluto.py:229-241).  Here the update is `RLFTPluto.train` with loss kind "sft": the teacher rows go to HBM once
per update as one (n, 5) column and every minibatch gathers its rows by the minibatch's device indices; the teacher label itself is
built on the device (`rift_sft_teacher_mode`).

The rollout side of the reference's SFT family asks a CARLA autopilot teacher (sft/teacher/autopilot.py) for `CBVs_teacher_infos` at
every tick; that teacher is CARLA-bound and out of scope -- this class trains on whatever buffer rows carry the key."""
from typing import Dict

import numpy as np
import torch

from rift_amd.planning.fine_tuner.rlft.rlft_pluto import RLFTPluto


def teacher_column(buffer, device) -> torch.Tensor:
    """(n, 5) float32 `CBVs_teacher_infos` of the full buffer on `device` (SFTCollate, sft_datamodule.py:33-38: torch.stack of the rows)."""
    rows = [torch.as_tensor(np.asarray(r), dtype=torch.float32).view(-1) for r in buffer.get_key_data('CBVs_teacher_infos')]
    col = torch.stack(rows, dim=0)
    if col.shape[1] != 5:
        raise ValueError(f"CBVs_teacher_infos rows have {col.shape[1]} entries, expected 5 (sft_pluto.py:229-241)")
    return col.to(device)


class SFTPluto(RLFTPluto):          # fine_tuner/sft/sft_pluto.py:36
    name, type, kind = 'sft_pluto', 'learnable', 'sft'

    def __init__(self, config, logger):
        super().__init__(config, logger)
        self.cfg.update(config.get('sft', {}))          # sft/config/*_training.yaml + datamodule/*.yaml overrides (same schedule keys as the RLFT family)
        self.initial_lr = self.cfg["lr"]

    def preprocess_buffer(self, trainer, replay) -> Dict[str, torch.Tensor]:
        return {"teacher_infos": teacher_column(self.buffer, self.device)}
