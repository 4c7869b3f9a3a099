"""CBVRolloutBuffer -- host mirror of rift/gym_carla/buffer/cbv_rollout_buffer.py:16-138 (+ base_buffer.py).

Same interface and semantics (per-key deque(maxlen=capacity), per-CBV trajectory staging until `CBVs_done`,
trajectories of <= 5 steps dropped, full at `buffer_capacity`), pure Python: the buffer is filled by the CARLA
rollout loop.  The policy update uploads it once into the HBM replay arena (rift_amd.replay.DeviceReplay)."""
from collections import defaultdict, deque

import numpy as np


class BaseBuffer:
    name = 'base'

    def __init__(self, num_scenario, mode, logger):
        self.num_scenario, self.mode, self.logger = num_scenario, mode, logger
        self.buffer_capacity = 2000
        self.buffer_pos = 0
        self.buffer_full = False
        self.buffer_data = None
        self.temp_buffer = None

    def __len__(self):
        return self.buffer_pos


class CBVRolloutBuffer(BaseBuffer):
    name = 'CBVRolloutBuffer'

    def __init__(self, num_scenario, mode, cbv_config, logger=None):
        super().__init__(num_scenario, mode, logger)
        assert self.mode == 'train_cbv', f'Only initialize {self.name} when training the rl-based onpolicy cbv agent'
        self.buffer_capacity = cbv_config['buffer_capacity']
        self.data_keys = cbv_config['data_keys']
        self.reset_buffer()

    def reset_buffer(self):
        self.buffer_pos = 0
        self.buffer_full = False
        self.buffer_data = {key: deque(maxlen=self.buffer_capacity) for key in self.data_keys}
        self.temp_buffer = {key: defaultdict(list) for key in self.buffer_data}

    def process_data_dict(self, data_dict):
        processed = {key: [] for key in self.buffer_data.keys()}
        lengths = set(len(data) for key, data in data_dict.items() if key in self.buffer_data.keys())
        assert len(lengths) == 1, 'all the data in the data dict should have same length'
        n = lengths.pop()
        ids_list = [ids for ids in data_dict['CBV_ids']]
        for i in range(n):
            for cbv_id in ids_list[i]:
                for key, value in self.temp_buffer.items():
                    value[cbv_id].append(data_dict[key][i][cbv_id])
                if data_dict['CBVs_done'][i][cbv_id]:
                    for key, value in processed.items():
                        value.extend(self.temp_buffer[key].pop(cbv_id))
        dl = set(len(d) for d in processed.values())
        assert len(dl) == 1, 'the data in the processed data dict should have same length'
        return processed, dl.pop()

    def store(self, data_dict):
        processed, n = self.process_data_dict(data_dict)
        if n > 5:
            if self.buffer_pos + n >= self.buffer_capacity:
                for i in range(n):
                    if self.buffer_pos < self.buffer_capacity:
                        for key, data in self.buffer_data.items():
                            data.append(processed[key][i])
                        self.buffer_pos += 1
                    else:
                        break
                self.buffer_full = True
            else:
                for key, data in self.buffer_data.items():
                    data.extend(processed[key])
                self.buffer_pos += n

    def get_all_np_data(self):
        assert self.buffer_pos == self.buffer_capacity, 'only get the data when the buffer is full'
        return {key: np.stack(d).reshape(self.buffer_capacity, -1) for key, d in self.buffer_data.items()}

    def add_extra_data(self, data_dict: dict):
        assert self.buffer_full, 'only add data when the buffer is full'
        assert all(len(v) == self.buffer_capacity for v in data_dict.values())
        self.buffer_data.update(data_dict)

    def get_key_data(self, key: str):
        assert self.buffer_full, 'only get the data when the buffer is full'
        return self.buffer_data[key]

    def sample(self, idx):
        assert self.buffer_full, 'only sample the data when the buffer is full'
        indices = idx if isinstance(idx, (list, tuple)) else [idx]
        assert all(0 <= i < self.buffer_capacity for i in indices)
        return {key: [d[i] for i in indices] if len(indices) > 1 else d[indices[0]] for key, d in self.buffer_data.items()}
