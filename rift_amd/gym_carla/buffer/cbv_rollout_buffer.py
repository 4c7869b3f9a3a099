"""On-policy replay for the CBV update, built around the HBM replay arena.

Interface of the reference's `CBVRolloutBuffer` (rift/gym_carla/buffer/cbv_rollout_buffer.py:16-138: `store`, `sample`,
`add_extra_data`, `get_key_data`, `get_all_np_data`, `reset_buffer`, `buffer_full`, `buffer_pos`, `buffer_data`), different
construction: transitions are kept ROW-major.  Every CBV owns an open episode (a list of transition rows); a finished episode is
committed as a block of rows into one flat row store, and the per-key columns that the reference keeps in deques are views built
from the rows on demand.  A committed row is exactly the unit `rift_amd.replay.DeviceReplay` uploads into the arena (`rows()` /
`rift_amd.planning.fine_tuner.rlft.rlft_pluto.buffer_to_scenes`), so the update never walks per-key containers.

Semantics kept from the reference (they decide which transitions reach the update):
  * a transition of CBV `c` in environment `i` is `{key: data_dict[key][i][c]}` for the configured `data_keys`;
  * an episode is committed when `CBVs_done[i][c]` is true;
  * the episodes finished by ONE `store()` call are committed together, and dropped together when they hold <= 5 transitions in total
    (cbv_rollout_buffer.py:80);
  * committing stops at `buffer_capacity`; the buffer counts as full as soon as a commit reaches or would pass it (:81-90).

Streaming (round 4): a committed row whose observation is a PlutoFeature is ALSO laid into a page-locked structure-of-arrays mirror of the
HBM arena right away (`rift_amd.replay.HostReplay`, one per streamed observation key: 'CBVs_obs' always, 'CBVs_next_obs' when a policy
asks for it -- PPO's second sweep), i.e. during the rollout.  `RLFTPluto.train` then uploads that mirror with ~30 asynchronous copies
instead of walking 4096 x 25 Python objects (`host_replay(key)`); the row store stays what `sample` / `get_key_data` serve.
"""
from typing import Dict, Hashable, Iterable, List, Sequence

import numpy as np

MIN_COMMIT = 6     # a store() call must finish at least this many transitions for them to be kept


class _Column(Sequence):
    """Read-only per-key view over the committed rows (what the reference exposes as a deque per key)."""

    def __init__(self, rows: List[dict], key: str):
        self._rows, self._key = rows, key

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [r[self._key] for r in self._rows[i]]
        return self._rows[i][self._key]

    def __iter__(self):
        k = self._key
        return (r[k] for r in self._rows)


class _Columns(dict):
    """`buffer_data`: key -> column.  Configured keys are views over the rows; `add_extra_data` attaches whole extra columns."""

    def __init__(self, rows: List[dict], keys: Iterable[str]):
        super().__init__((k, _Column(rows, k)) for k in keys)


class CBVRolloutBuffer:
    name = 'CBVRolloutBuffer'

    def __init__(self, num_scenario, mode, cbv_config, logger=None):
        if mode != 'train_cbv':
            raise AssertionError(f'Only initialize {self.name} when training the rl-based onpolicy cbv agent')
        self.num_scenario, self.mode, self.logger = num_scenario, mode, logger
        self.buffer_capacity = int(cbv_config['buffer_capacity'])
        self.data_keys = list(cbv_config['data_keys'])
        obs_cfg = cbv_config.get('obs') or {}
        self._host_caps = {"A": int(obs_cfg['max_agent']) + 1} if 'max_agent' in obs_cfg else {}    # rift_pluto.yaml:35: the CBV + max_agent others
        self._host_caps.update(cbv_config.get('host_caps') or {})       # initial capacities of the ragged dimensions (they double on demand)
        self._host: Dict[str, object] = {}          # observation key -> HostReplay (False: rows of this key are not PlutoFeatures)
        self._host_keys = [k for k in ('CBVs_obs',) if k in self.data_keys and cbv_config.get('stream_to_host', True)]
        self.reset_buffer()

    # ---- state ------------------------------------------------------------------------------------------------------
    def reset_buffer(self):
        self._rows: List[dict] = []
        self._open: Dict[Hashable, List[dict]] = {}
        self.buffer_full = False
        self.buffer_data = _Columns(self._rows, self.data_keys)
        for h in self._host.values():
            if h:
                h.reset()

    def reset_buffer_reversibly(self):
        """reset_buffer() that can be undone: the committed rows, the open episodes and the host mirrors' batch dimensions move into the
        returned token instead of being dropped.  `restore(token)` puts them back (a failed update keeps its generation of rollout data, as
        the reference does: its reset follows a successful fit, rlft_pluto.py:244-250); dropping the token is what frees the rows."""
        token = {"rows": self._rows, "open": self._open, "full": self.buffer_full, "data": self.buffer_data,
                 "host": {k: (dict(h.dims), h.has_ref_logits) for k, h in self._host.items() if h}}
        self.reset_buffer()
        return token

    def restore(self, token):
        """Undo reset_buffer_reversibly (nothing may have been stored in between: the pinned mirror's slots still hold the old generation)."""
        assert not self._rows and not self._open, "rows were stored after the reset: the host mirror's slots are overwritten"
        self._rows, self._open, self.buffer_full, self.buffer_data = token["rows"], token["open"], token["full"], token["data"]
        for k, (dims, has_ref) in token["host"].items():
            h = self._host.get(k)
            if h:
                h.dims, h.has_ref_logits = dict(dims), has_ref
                h.version += 1

    # ---- the pinned host mirror of the HBM arena ---------------------------------------------------------------------------
    def attach_host_replay(self, obs_keys: Iterable[str]):
        """Stream these observation keys as well (rows already committed are laid in now)."""
        for k in obs_keys:
            if k in self.data_keys and k not in self._host_keys:
                self._host_keys.append(k)
                self._stream(0, self._rows, only=k)

    def host_replay(self, key: str = 'CBVs_obs'):
        """The HostReplay holding the committed rows of `key`, or None (not streamed / not PlutoFeature observations)."""
        h = self._host.get(key)
        return h if h else None

    @staticmethod
    def _row_scene(row: dict, key: str):
        """(per-scene feature dict, RLFT extras) of a committed row, or None when the observation is not a PlutoFeature."""
        obs = row.get(key)
        pf = obs.get('raw_pluto_feature') if isinstance(obs, dict) else None
        data = getattr(pf, 'data', None)
        if not isinstance(data, dict) or 'agent' not in data or 'reference_line' not in data:
            return None
        ex = {}
        if key == 'CBVs_obs':
            a, o, r = row.get('CBVs_group_advantage'), row.get('CBVs_actions_old_group_logits'), row.get('CBVs_actions_ref_group_logits')
            if isinstance(a, dict):
                ex["group_advantage"], ex["group_advantage_mask"] = np.asarray(a['advantage']), np.asarray(a['valid_mask'])
            if isinstance(o, dict):
                ex["old_group_logits"] = np.asarray(o['logits'])
            if isinstance(r, dict):
                ex["ref_group_logits"] = np.asarray(r['logits'])
        return data, ex

    def _stream(self, start: int, block: List[dict], only: str = None):
        for key in ([only] if only else self._host_keys):
            if self._host.get(key) is False:
                continue
            for j, row in enumerate(block):
                scene = self._row_scene(row, key)
                if scene is None:
                    self._host[key] = False
                    break
                if key not in self._host:
                    from rift_amd.replay import HostReplay
                    self._host[key] = HostReplay(self.buffer_capacity, caps=self._host_caps)
                self._host[key].put(start + j, *scene)

    @property
    def buffer_pos(self) -> int:
        return len(self._rows)

    def __len__(self):
        return len(self._rows)

    def rows(self) -> List[dict]:
        """The committed transitions, oldest first (one dict per transition, keys = data_keys)."""
        return self._rows

    # ---- writing ----------------------------------------------------------------------------------------------------
    def _finished_rows(self, data_dict) -> List[dict]:
        keys = self.data_keys
        n_env = {len(data_dict[k]) for k in keys if k in data_dict}
        if len(n_env) != 1:
            raise AssertionError('all the data in the data dict should have same length')
        done_rows: List[dict] = []
        for i, ids in enumerate(data_dict['CBV_ids']):
            for c in ids:
                episode = self._open.setdefault(c, [])
                episode.append({k: data_dict[k][i][c] for k in keys})
                if data_dict['CBVs_done'][i][c]:
                    done_rows += self._open.pop(c)
        return done_rows

    def store(self, data_dict):
        block = self._finished_rows(data_dict)
        if len(block) < MIN_COMMIT:
            return
        room = self.buffer_capacity - len(self._rows)
        if len(block) >= room:
            block = block[:max(room, 0)]
            self.buffer_full = True
        start = len(self._rows)
        self._rows += block
        self._stream(start, block)

    def add_extra_data(self, data_dict: dict):
        """Attach whole extra columns (PPO / REINFORCE preprocessing results), one entry per committed transition."""
        self._require_full('only add data when the buffer is full')
        if any(len(v) != self.buffer_capacity for v in data_dict.values()):
            raise AssertionError('extra columns must have one entry per buffer slot')
        self.buffer_data.update(data_dict)

    # ---- reading ----------------------------------------------------------------------------------------------------
    def _require_full(self, msg):
        if not self.buffer_full:
            raise AssertionError(msg)

    def get_key_data(self, key: str):
        self._require_full('only get the data when the buffer is full')
        return self.buffer_data[key]

    def get_all_np_data(self):
        if len(self._rows) != self.buffer_capacity:
            raise AssertionError('only get the data when the buffer is full')
        return {k: np.stack(list(col)).reshape(self.buffer_capacity, -1) for k, col in self.buffer_data.items()}

    def sample(self, idx):
        """One transition (int index -> {key: value}) or several (list / tuple -> {key: [values]})."""
        self._require_full('only sample the data when the buffer is full')
        many = isinstance(idx, (list, tuple))
        picks = list(idx) if many else [idx]
        if not all(0 <= i < self.buffer_capacity for i in picks):
            raise AssertionError('sample index out of range')
        if many and len(picks) > 1:
            return {k: [col[i] for i in picks] for k, col in self.buffer_data.items()}
        return {k: col[picks[0]] for k, col in self.buffer_data.items()}
