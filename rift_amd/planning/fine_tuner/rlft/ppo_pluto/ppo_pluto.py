"""PPOPlutoModel -- mirror of fine_tuner/rlft/ppo_pluto/ppo_pluto.py:24-37: the Pluto PlanningModel plus the PPO critic."""
from rift_amd.gym_carla.utils.net import CriticPPO
from rift_amd.planning.pluto.model.pluto_model import PlanningModel


class PPOPlutoModel(PlanningModel):
    def __init__(self, radius: float, state_dim: int = 128, action_dim: int = 3, hidden_dim=(256, 256), clip_epsilon: float = 0.2,
                 lambda_entropy: float = 0.01):
        super().__init__(radius=radius)
        self.clip_epsilon, self.lambda_entropy = clip_epsilon, lambda_entropy
        self.value_net = CriticPPO(dims=list(hidden_dim), state_dim=state_dim, action_dim=action_dim)
