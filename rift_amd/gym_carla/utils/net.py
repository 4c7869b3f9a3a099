"""CriticPPO -- parameter-container mirror of rift/gym_carla/utils/net.py:355-371,420-431 (state_dict keys net.{0,2,4}.{weight,bias},
state_avg, state_std, value_avg, value_std; orthogonal(std 0.5) output layer, bias 1e-6).  forward runs on the HIP engine
(rift_critic_forward); there is no CPU path."""
from typing import List

import torch
import torch.nn as nn

from rift_amd import _ffi


def build_mlp(dims: List[int]) -> nn.Sequential:   # net.py:436-451 (ReLU between, raw output)
    layers = []
    for i in range(len(dims) - 1):
        layers.extend([nn.Linear(dims[i], dims[i + 1]), nn.ReLU()])
    del layers[-1]
    return nn.Sequential(*layers)


class CriticPPO(nn.Module):
    def __init__(self, dims: List[int], state_dim: int, action_dim: int):
        super().__init__()
        if list(dims) != [256, 256] or state_dim != 128:
            raise NotImplementedError("the HIP critic is built for the reference configuration dims=[256, 256], state_dim=128 "
                                      "(planning/config/ppo_pluto.yaml:43-45)")
        self.state_dim, self.action_dim = state_dim, action_dim
        self.net = build_mlp([state_dim, *dims, 1])
        torch.nn.init.orthogonal_(self.net[-1].weight, 0.5)
        torch.nn.init.constant_(self.net[-1].bias, 1e-6)
        self.state_avg = nn.Parameter(torch.zeros((state_dim,)), requires_grad=False)
        self.state_std = nn.Parameter(torch.ones((state_dim,)), requires_grad=False)
        self.value_avg = nn.Parameter(torch.zeros((1,)), requires_grad=False)
        self.value_std = nn.Parameter(torch.ones((1,)), requires_grad=False)
        self._engine = None

    def bind(self, engine: "_ffi.Engine"):
        self._engine = engine
        return self

    def forward(self, state: torch.Tensor) -> torch.Tensor:
        if self._engine is None or not state.is_cuda:
            raise RuntimeError("CriticPPO.forward needs the HIP engine (bind(engine)) and device tensors; there is no CPU path")
        return self._engine.critic_forward(dict(self.named_parameters()), state)
