"""RewardShapingPluto -- update side of fine_tuner/sft/rs_pluto/rs_pluto.py:21-60 and rs_datamodule.py:19-127.

`preprocess_buffer` (rs_datamodule.py:109-127): reward = CBVs_reward + reward_lambda * CBVs_teacher_rewards (0.2,
datamodule/rs_datamodule.yaml), then the discounted-return scan over the whole buffer with the episode ends of CBVs_done
(compute_return, rs_datamodule.py:19-39, gamma 0.98; its closing normalisation acts on a scalar and changes nothing).  The scan runs on
the device (`rift_discounted_return`, fp64).  The objective (rs_trainer.py:120-170) is REINFORCE's line for line: loss kind "rs"."""
from typing import Dict

import numpy as np
import torch

from rift_amd.planning.fine_tuner.sft.sft_pluto import SFTPluto


def _column(buffer, key) -> torch.Tensor:
    return torch.as_tensor(np.asarray([float(np.asarray(v).reshape(-1)[0]) for v in buffer.get_key_data(key)]), dtype=torch.float64)


class RewardShapingPluto(SFTPluto):
    name, type, kind = 'rs_pluto', 'learnable', 'rs'

    def __init__(self, config, logger):
        super().__init__(config, logger)
        self.cfg.setdefault("reward_lambda", 0.2)

    def shaped_rewards(self) -> torch.Tensor:
        return _column(self.buffer, 'CBVs_reward') + self.cfg["reward_lambda"] * _column(self.buffer, 'CBVs_teacher_rewards')

    def preprocess_buffer(self, trainer, replay) -> Dict[str, torch.Tensor]:
        dones = _column(self.buffer, 'CBVs_done').float()
        ret = trainer.engine.discounted_return(self.shaped_rewards(), dones, self.cfg["gamma"])
        return {"returns": ret.float()}
