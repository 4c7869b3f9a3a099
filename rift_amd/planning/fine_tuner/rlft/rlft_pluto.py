"""RLFT policies -- host mirror of the reference's update-loop driver
(fine_tuner/rlft/rlft_pluto.py:32-300, fine_tuner/training_builder.py:45-180 and the four *_pluto.py variants)
without hydra / Lightning / wandb: the same `CBVBasePolicy` surface, checkpoint naming, learning-rate decay across
updates, 16-epoch fit with a 90/10 split, batch 256, clip 0.5, per-epoch WarmupCosLR and top-1 checkpointing
by validation loss -- every forward / loss / backward on the HIP engine, batches gathered on device.

Rollout side: `get_action` (rlft_pluto.py:84-204, rift_pluto.py:28-161, grpo_pluto.py:33-170) on top of
rift_amd.planning.pluto.pluto.PLUTO -- one eval forward per environment and tick, candidate trimming, PID control, and in train mode
the replay columns the update consumes: old log-prob + chosen (r, m) (PPO / REINFORCE), old / reference group logits and the
group-relative advantage of every candidate (RIFT / GRPO; TrajEvaluator on the device).  Live CARLA state comes through the injected
state source (rift_amd.planning.pluto.pluto.CBVStateSource).
"""
import contextlib
import math
import os
import re
import time
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from rift_amd.gym_carla.buffer.cbv_rollout_buffer import CBVRolloutBuffer
from rift_amd.planning.fine_tuner.rlft.trainer import RLFTTrainer, split_minibatch
from rift_amd.planning.pluto.model.pluto_model import PlanningModel
from rift_amd.planning.pluto.pluto import (PLUTO, CBVBasePolicy, CBVStateSource, Candidates, CenterState, NoFlagSource,   # noqa: F401 (re-exported)
                                            capped_host_threads, paused_cyclic_gc)
from rift_amd.replay import DeviceReplay

DEFAULT_CFG = {   # fine_tuner/rlft/config/{rift,grpo,ppo,reinforce}_training.yaml + datamodule/*.yaml + lightning/custom_lightning.yaml
    "epochs": 16, "warmup_epochs": 3, "lr": 1e-4, "cl_lr_decay": 0.9, "min_lr": 1e-6, "weight_decay": 1e-5,
    "trainable_layers": ["planning_decoder.pi_head"], "train_batch_size": 256, "val_batch_size": 256, "shuffle": True,
    "train_ratio": 0.9, "gamma": 0.98, "lambda_gae_adv": 0.98, "gradient_clip_val": 0.5,
    "host_threads": 4,          # cap of torch's intra-op pool during an update (capped_host_threads; 0 = leave it alone)
    "pause_gc": True,           # no cyclic-GC pass inside an update (paused_cyclic_gc: a full collection there is a 100 ms stall)
}


def buffer_to_scenes(buffer: CBVRolloutBuffer) -> List[Dict]:
    """Replay entries (cbv_rollout_buffer.py:77-95; keys of planning/config/rift_pluto.yaml:8-16) -> arena scenes."""
    n = buffer.buffer_capacity
    obs = buffer.get_key_data('CBVs_obs')
    scenes = []
    has = lambda k: k in buffer.buffer_data  # noqa: E731
    for i in range(n):
        pf = obs[i]['raw_pluto_feature']
        ex = {}
        if has('CBVs_group_advantage'):
            a = buffer.buffer_data['CBVs_group_advantage'][i]
            ex["group_advantage"] = torch.as_tensor(np.asarray(a['advantage']), dtype=torch.float64)
            ex["group_advantage_mask"] = torch.as_tensor(np.asarray(a['valid_mask']), dtype=torch.bool)
        if has('CBVs_actions_old_group_logits'):
            o = buffer.buffer_data['CBVs_actions_old_group_logits'][i]
            ex["old_group_logits"] = torch.as_tensor(np.asarray(o['logits']), dtype=torch.float32)
            ex["old_group_logits_mask"] = torch.as_tensor(np.asarray(o['valid_mask']), dtype=torch.bool)
        if has('CBVs_actions_ref_group_logits'):
            ex["ref_group_logits"] = torch.as_tensor(np.asarray(buffer.buffer_data['CBVs_actions_ref_group_logits'][i]['logits']),
                                                     dtype=torch.float32)
        R = pf.data["reference_line"]["position"].shape[0]
        ex.setdefault("group_advantage", torch.zeros(R, 12, dtype=torch.float64))
        ex.setdefault("group_advantage_mask", torch.ones(R, 12, dtype=torch.bool))
        ex.setdefault("old_group_logits", torch.zeros(R, 12))
        ex.setdefault("old_group_logits_mask", torch.ones(R, 12, dtype=torch.bool))
        scenes.append({"feature": pf.data, "extras": ex})
    return scenes


class RLFTPluto(PLUTO):
    name = 'rlft_pluto'
    type = 'rlft'
    kind = 'rift'
    EXTRA_COLUMNS = ('CBVs_actions_old_log_prob', 'CBVs_actions_mode')      # rlft_pluto.py:131-135

    def __init__(self, config, logger):
        super().__init__(config, logger)
        self.model_path = Path(config.get('ROOT_DIR', '.')) / config.get('model_path', 'model_ckpt')
        self.cbv_recog, self.seed = config.get('cbv_recog', 'rule'), config.get('seed', 0)
        self.pretrain_seed = config.get('pretrain_seed', self.seed)
        self.cfg = dict(DEFAULT_CFG)
        self.cfg.update(config.get('rlft', {}))
        self.initial_lr = self.cfg["lr"]
        self.checkpoint: Optional[str] = None
        self.continue_episode, self.current_epoch = 0, 0
        ego = config.get('ego_policy', 'pdm_lite') if config.get('mode', 'train_cbv') == 'train_cbv' else config.get('pretrain_ego', 'pdm_lite')
        sd = self.seed if config.get('mode', 'train_cbv') == 'train_cbv' else self.pretrain_seed
        self.load_agent_info = f"{ego}-{self.cbv_recog}-seed{sd}"
        self.save_agent_info = f"{config.get('ego_policy', 'pdm_lite')}-{self.cbv_recog}-seed{self.seed}"
        self.train_model: Optional[PlanningModel] = None
        self.buffer: Optional[CBVRolloutBuffer] = None
        self.last_fit: Dict = {}

    def _log(self, msg, color=None):
        if self.logger is not None and hasattr(self.logger, "log"):
            self.logger.log(msg, color) if color else self.logger.log(msg)

    # ---- API surface of rlft_pluto.py ---------------------------------------------------------------
    STREAM_KEYS = ('CBVs_obs',)         # observation columns the rollout buffer mirrors into pinned host arenas as it commits them

    def set_buffer(self, buffer, total_routes=None):
        self.buffer = buffer
        self.total_routes = total_routes
        if hasattr(buffer, 'attach_host_replay'):
            buffer.attach_host_replay(self.STREAM_KEYS)

    def _arena(self, key: str = 'CBVs_obs') -> DeviceReplay:
        """The replay in HBM.  Product path: the pinned structure-of-arrays mirror the buffer filled at store() time goes up as it lies
        (DeviceReplay.upload: <= 30 asynchronous copies, the update starts behind them; the device arena is kept across updates).
        A buffer without that mirror (foreign buffer class, non-PlutoFeature rows, stream_to_host: False) is packed scene by scene."""
        host = self.buffer.host_replay(key) if hasattr(self.buffer, 'host_replay') else None
        if host is None:
            if key == 'CBVs_obs':
                return DeviceReplay(buffer_to_scenes(self.buffer), self.device)
            return DeviceReplay([{"feature": o['raw_pluto_feature'].data, "extras": replay_dummy_extras(o['raw_pluto_feature'].data)}
                                 for o in self.buffer.get_key_data(key)], self.device)
        cache = self.__dict__.setdefault("_arenas", {})
        if key not in cache:
            cache[key] = DeviceReplay(None, self.device)
        cache[key].upload(host, self.buffer.buffer_capacity)
        return cache[key]

    def set_mode(self, mode):
        self.mode = mode
        if mode == 'train':
            self._mem_ckpt = None           # (a fresh training model: its trunk is NOT the one the snapshot belongs to)
            self.train_model = PlanningModel(radius=self.radius).to(self.device)
            self.train_model.compute_precision = self.compute_precision
            self.train_model.train()
            self.pluto_model.eval()
        elif mode == 'eval':
            self.pluto_model.eval()
        else:
            raise ValueError(f'Unknown mode {mode}')

    def _per_cbv(self, env_id, cbv_id, obs, data, out, index, state, decision):
        """rlft_pluto.py:172-176: log of the chosen candidate's softmax score and its (r, m) -- integer, bit-exact."""
        flat = int(decision.flat_index[decision.best])
        return {'CBVs_actions_old_log_prob': np.log(decision.score[decision.best] + 1e-12),
                'CBVs_actions_mode': (flat // decision.n_mode, flat % decision.n_mode)}

    def save_model(self, episode):  # checkpoints are written by train(); rlft_pluto.py:295-296
        pass

    def finish(self):
        pass

    def load_model(self, resume=True):
        self._mem_ckpt = None               # whatever gets loaded below, the in-memory snapshot of the last update no longer describes it
        load_dir = self.model_path / self.load_agent_info
        files = list(load_dir.glob("*.ckpt"))
        if resume and files:
            latest = max(files, key=lambda f: int(re.search(r"carla_episode=(\d+)", f.stem).group(1)))
            self.checkpoint = latest.as_posix()
            self.continue_episode = int(re.search(r"carla_episode=(\d+)", latest.stem).group(1))
            self.current_epoch = len(files)
            self._log(f">> Loading {self.name} model from {latest.name}", 'yellow')
        else:
            if not resume:
                for f in files:
                    f.unlink()
            self.checkpoint = self._ckpt_path
            self.continue_episode, self.current_epoch = 0, 0
        # the reference always loads (rlft_pluto.py:293) and fails loudly on a bad path; only a policy configured WITHOUT any checkpoint
        # (ckpt_path unset: synthetic benchmarks, tests) keeps its seeded initialisation
        if self.checkpoint:
            if not Path(self.checkpoint).exists():
                raise FileNotFoundError(f"{self.name}: checkpoint {self.checkpoint} does not exist")
            self.pluto_model.load_state_dict(self.load_infer_checkpoint(self.checkpoint, self.device))
            self._infer_bound = (self.checkpoint, self.pluto_model._tensor_version())

    def update_training_ckpt(self):
        load_dir = self.model_path / self.load_agent_info
        pat = re.compile(r"carla_episode=(\d+)")
        files = sorted(load_dir.glob("*.ckpt"), key=lambda f: int(pat.search(f.stem).group(1)), reverse=True)
        self.current_epoch = len(files)
        if files:
            self.checkpoint = files[0].as_posix()

    @staticmethod
    def _trunk_version(model):
        """Identity + in-place version of every FROZEN tensor of `model` (everything an update does not move): unchanged as long as
        nobody loaded, re-initialised or re-allocated the trunk."""
        # (walks the modules' own tables: named_parameters() + state_dict() of the 438 tensors cost 4.5 ms a call, twice per update)
        mods = list(model.modules())
        frozen = [p for m in mods for p in m._parameters.values() if p is not None and not p.requires_grad]
        if len(frozen) == sum(1 for m in mods for p in m._parameters.values() if p is not None):      # nothing trainable: the buffers do not move either
            frozen += [b for m in mods for b in m._buffers.values() if b is not None]
        return tuple((id(t), t.data_ptr(), t._version) for t in frozen)

    @staticmethod
    def _moving_keys(model) -> List[str]:
        """state_dict keys an update changes: the trainable parameters (after RLFTTrainer's freeze) and every buffer (BatchNorm running
        statistics and batch counters of the train-mode forward)."""
        return [n for n, p in model.named_parameters() if p.requires_grad] + [n for n, _ in model.named_buffers()]

    def _write_checkpoint(self, path: Path, base_cpu: Dict[str, torch.Tensor], snapshot: Dict[str, torch.Tensor], epoch: int, e_i):
        """{'state_dict': {'model.<key>': tensor}} (pluto.py:130-137): the frozen tensors from the CPU copy the update started from, the
        moving ones from the best epoch's snapshot."""
        sd = dict(base_cpu)
        sd.update({k: v.cpu() for k, v in snapshot.items()})
        torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "epoch": epoch, "carla_episode": e_i}, path)
        base_cpu.update({k: sd[k] for k in snapshot})

    # ---- the policy update (rlft_pluto.py:206-247) ---------------------------------------------------
    def preprocess_buffer(self, trainer: RLFTTrainer, replay: DeviceReplay) -> Dict[str, torch.Tensor]:
        """No-op for RIFT / GRPO (rift_datamodule.py:97-98)."""
        return {}

    def train(self, e_i, process_group=None):
        """One policy update = RLFTPluto.train (rlft_pluto.py:206-247): 16 epochs over a 90/10 split of the full buffer, batch 256,
        clip 0.5, per-epoch WarmupCosLR, top-1 checkpoint by validation loss, inference-model refresh, buffer reset.

        `process_group` (torch.distributed, nccl = RCCL): data-parallel update.  Every rank holds the same replay and
        draws the same minibatch order; each 256-scene minibatch is split contiguously over the ranks (32 scenes per GPU at 8 GPUs) and
        the exchanges of RLFTTrainer make the sharded step equal the single-process one -- same losses, gradients, BatchNorm running
        statistics and therefore the same checkpoint on every rank; rank 0 writes it."""
        assert self.buffer is not None and self.buffer.buffer_full, 'The buffer should be full before training'
        with capped_host_threads(self.cfg.get("host_threads", 4)), paused_cyclic_gc(self.cfg.get("pause_gc", True)):
            return self._train(e_i, process_group)

    def _train(self, e_i, process_group=None):
        marks = [("start", time.perf_counter())]          # host-side timeline of the update (last_fit["timing"]; bench.py: full_update_e2e)
        mark = lambda name: marks.append((name, time.perf_counter()))  # noqa: E731
        if self.train_model is None:
            self.set_mode('train')
        cfg = self.cfg
        lr = max(self.initial_lr * (cfg["cl_lr_decay"] ** self.current_epoch), cfg["min_lr"])   # rlft_pluto.py:212
        # Where the training model's weights come from (rlft_pluto.py:214-222: the latest checkpoint).  Only the trainable layers and the
        # BatchNorm running statistics move during an update (`_moving_keys`); when the checkpoint to load is the one THIS process wrote at
        # the end of its previous update, the frozen trunk is already in place and the moving tensors are restored from the device
        # snapshot kept with it -- no 17 MB read + 438 host-to-device copies per update.
        mem = self.__dict__.get("_mem_ckpt")
        loaded_from = "file" if self.checkpoint else "inference model"
        base_cpu = None                       # CPU copy of the whole state_dict the update starts from (the frozen part of the next checkpoint)
        if (self.checkpoint and mem is not None and mem["path"] == self.checkpoint and Path(self.checkpoint).exists()
                and mem["model"] is self.train_model and mem["trunk"] == self._trunk_version(self.train_model)):
            with torch.no_grad():
                own = self.train_model.state_dict()
                for k, v in mem["moving"].items():
                    own[k].copy_(v)
            base_cpu = mem["base_cpu"]
            loaded_from = "device snapshot"
        elif self.checkpoint:
            if not Path(self.checkpoint).exists():
                raise FileNotFoundError(f"{self.name}: checkpoint {self.checkpoint} does not exist")
            sd = torch.load(self.checkpoint, map_location="cpu", weights_only=False)["state_dict"]
            base_cpu = {k.replace("model.", "", 1): v for k, v in sd.items()}
            self.train_model.load_state_dict(base_cpu, strict=False)
            # the next checkpoint is the TRAINING MODEL's state_dict (Lightning saves the module, training_builder.py:131-140): keys the
            # source file lacked (loaded with strict=False) come off the device, keys the model does not have are not carried forward
            own = self.train_model.state_dict()
            base_cpu = {k: (base_cpu[k] if k in base_cpu else v.detach().cpu()) for k, v in own.items()}
        else:
            self.train_model.load_state_dict(self.pluto_model.state_dict(), strict=False)
            if cfg.get("checkpoint_every_improvement", False):      # (a file per improvement needs the frozen part before the first epoch ends)
                base_cpu = {k: v.detach().cpu() for k, v in self.train_model.state_dict().items()}
        infer_in_sync = (not self.checkpoint) or self.__dict__.get("_infer_bound") == (self.checkpoint, self.pluto_model._tensor_version())
        mark("load_checkpoint")
        trainer = RLFTTrainer(self.train_model, kind=self.kind, lr=lr, cl_lr_decay=cfg["cl_lr_decay"],
                              weight_decay=cfg["weight_decay"], epochs=cfg["epochs"], warmup_epochs=cfg["warmup_epochs"],
                              trainable_layers=tuple(cfg["trainable_layers"]), gradient_clip_val=cfg["gradient_clip_val"],
                              clip_epsilon=getattr(self, "clip_epsilon", 0.2), lambda_entropy=getattr(self, "lambda_entropy", 0.01),
                              process_group=process_group, seed=int(e_i) + 1)
        rank, world = trainer.rank, trainer.world
        eng = trainer.engine
        mark("trainer")
        replay = self._arena()
        mark("arena")
        extras = self.preprocess_buffer(trainer, replay)
        mark("preprocess")
        n = replay.n
        g = torch.Generator().manual_seed(int(e_i) + 1234)          # same split and order on every rank
        perm = torch.randperm(n, generator=g)                      # random_split(dataset, [0.9, 0.1]), rift_datamodule.py:93
        n_train = int(math.floor(n * cfg["train_ratio"]))
        train_idx, val_idx = perm[:n_train], perm[n_train:]
        save_dir = self.model_path / self.load_agent_info
        save_dir.mkdir(parents=True, exist_ok=True)
        best, best_path, best_epoch, snapshot, history = None, None, None, None, []
        moving = self._moving_keys(self.train_model)

        # every epoch's minibatch order is drawn up front (the same generator sequence as a draw per epoch: validation draws nothing) and goes
        # to the device in ONE upload, with the reference-line count of every minibatch taken off the host copy
        passes = [train_idx[torch.randperm(train_idx.numel(), generator=g)] if cfg["shuffle"] else train_idx for _ in range(cfg["epochs"])]
        passes.append(val_idx)
        flat_dev = torch.cat(passes).to(torch.int32).to(self.device)
        up = torch.cuda.Event()               # the upload is queued on this stream; the prefetch stream's gathers wait for it
        up.record()
        starts = [0]
        for p_ in passes:
            starts.append(starts[-1] + p_.numel())

        def minibatches(which, bs):
            """(device index slice of this rank, R of the whole minibatch, shard descriptor) per minibatch of pass `which`."""
            idx, idx_dev = passes[which], flat_dev[starts[which]:starts[which + 1]]
            r_all = replay.r_count_cpu[idx]
            for s in range(0, idx.numel(), bs):
                m = min(bs, idx.numel() - s)
                if m < world:       # a tail with fewer scenes than ranks cannot give every rank a scene (each one has to join the
                    continue        # exchanges of the forward): dropped on ALL ranks alike -- at most world - 1 scenes per pass
                lo, hi = split_minibatch(m, rank, world) if world > 1 else (0, m)
                yield idx_dev[s + lo:s + hi], int(r_all[s:s + m].max()), ((lo, m) if world > 1 else None), up

        def run(batch, train, out=None):
            idx_dev, R_out, shard, up = batch
            if out is not None:       # validation through the step pipeline (RLFTTrainer.validation_step): its batch is gathered like a training batch
                fb, b = trainer.gather(replay, idx_dev, R_out, ready=up)
                if extras:
                    b = dict(b)
                    for k, v in extras.items():
                        b[k] = v[idx_dev.long()].contiguous()
                return trainer.validation_step(fb, b, shard=shard, out=out)
            # training steps cycle through _ffi.DEFER_SLOTS batch-buffer sets: the tail of step k (policy head .. AdamW) runs beside the trunk of step
            # k + 1, whose batch is gathered on the prefetch stream (RLFTTrainer.gather); validation joins the update stream first and uses slot 0
            fb, b = trainer.gather(replay, idx_dev, R_out, ready=up) if train else replay.collate(eng, idx_dev, R_out, slot=0)
            if extras:
                b = dict(b)
                for k, v in extras.items():
                    b[k] = v[idx_dev.long()].contiguous()
            return trainer.training_step(fb, b, shard=shard) if train else trainer.validation_step(fb, b, shard=shard)

        # Epoch bookkeeping.  Lightning reads the epoch's losses on the host and decides top-1 there (training_builder.py:131-140): two host
        # reads per epoch, each draining the step pipeline (16 x ~1.7 ms of an update).  Single process, no file per improvement: nothing of an
        # epoch comes back to the host -- its mean training loss and validation loss go into a device table, the tensors that move (pi_head,
        # BatchNorm statistics: `moving`) are copied into that epoch's snapshot on the stream, and the whole table is read ONCE behind the last
        # epoch; top-1 = the FIRST epoch with the minimal validation loss, exactly what the strict `<` of the per-epoch decision selects, and
        # the checkpoint is written from that epoch's snapshot.  The non-finite flag is sticky and checked at that one read.
        deferred = world == 1 and not cfg.get("checkpoint_every_improvement", False)
        trunk_after, buffer_reset, undo_reset = None, False, None

        def uncommitted():                  # the update did not commit: the rollout buffer keeps this generation's data
            if undo_reset is not None:
                self.buffer.restore(undo_reset)

        try:
            if deferred:
                own = self.train_model.state_dict()
                src = [own[k] for k in moving]
                snaps = []
                table = torch.zeros(cfg["epochs"], 2, dtype=torch.float64, device=self.device)
                lrs = []
                # validation through the step pipeline: the epoch's validation trunks follow its last training trunk on this stream without
                # waiting for that step's update, their heads / objectives and the epoch's bookkeeping (mean losses, the snapshot of the trained
                # parameters) sit on the update stream between the tails -- this stream never waits for the update stream inside an update.
                # BatchNorm statistics are written by the training trunks on THIS stream: their snapshot stays here
                piped = trainer.pipelined_validation
                val_mbs = list(minibatches(cfg["epochs"], cfg["val_batch_size"]))
                vtab = torch.zeros(cfg["epochs"], max(len(val_mbs), 1), dtype=torch.float64, device=self.device)
                tp = {n for n, p_ in self.train_model.named_parameters() if p_.requires_grad}      # (state_dict() tensors are detached: ask the parameters)
                trained = [i for i, k in enumerate(moving) if k in tp]           # written by AdamW on the update stream
                buffers = [i for i, k in enumerate(moving) if k not in tp]       # BatchNorm running statistics, written by the training trunks
                for epoch in range(cfg["epochs"]):
                    snaps.append([torch.empty_like(t) for t in src])   # (per epoch, behind the steps issued so far: 0.4 ms of host time each, hidden)
                    for mb in minibatches(epoch, cfg["train_batch_size"]):
                        run(mb, True)
                    if piped:
                        for j, mb in enumerate(val_mbs):
                            run(mb, False, out=vtab[epoch, j:j + 1])
                        with trainer.update_stream():
                            trainer.pop_mean_loss_async(table[epoch, 0], in_update_stream=True)
                            table[epoch, 1].copy_(vtab[epoch].mean() if val_mbs else table[epoch, 0])
                            if trained:
                                torch._foreach_copy_([snaps[epoch][i] for i in trained], [src[i] for i in trained])
                        if buffers:
                            torch._foreach_copy_([snaps[epoch][i] for i in buffers], [src[i] for i in buffers])
                    else:
                        trainer.pop_mean_loss_async(table[epoch, 0])
                        vl = [run(mb, False).clone() for mb in val_mbs]
                        table[epoch, 1].copy_(torch.stack(vl).mean() if vl else table[epoch, 0])
                        torch._foreach_copy_(snaps[epoch], src)      # (behind the validation steps on this stream; the next epoch's updates are ordered behind it)
                    trainer.on_epoch_end()
                    lrs.append(trainer.optimizer.param_groups[0]["lr"])
                trainer.wait_update()                                  # (the table and the snapshots of the trained parameters come off the update stream)
                # the host is ~100 ms ahead of the device here: what the end of the update needs from the host and does not depend on the
                # outcome happens now, in the device's shadow -- the frozen trunk's identity (nothing queued writes it) and the buffer reset
                # (4096 committed rows to free, 9 ms; the arena was uploaded from the pinned mirror before the first epoch)
                trunk_after = self._trunk_version(self.train_model)
                # (reversibly: the rows move into a token.  A non-finite flag at the read below, or a checkpoint that cannot be written,
                # puts them back -- the generation's rollout data survives a failed update, as in the reference, whose reset follows the
                # fit; the token is dropped -- the 9 ms of freeing -- by a helper thread once the checkpoint is on disk)
                undo_reset = self.buffer.reset_buffer_reversibly()
                buffer_reset = True
                host = table.cpu()                                     # the update's one host read
                trainer.check_finite()
                for epoch in range(cfg["epochs"]):
                    history.append({"epoch": epoch, "train_loss": float(host[epoch, 0]), "val_loss": float(host[epoch, 1]), "lr": lrs[epoch]})
                best_epoch = min(range(cfg["epochs"]), key=lambda e: (history[e]["val_loss"], e))
                best = history[best_epoch]["val_loss"]
                best_path = save_dir / f"carla_episode={e_i}-epoch={best_epoch}-val_loss={best:.3f}.ckpt"   # training_builder.py:133
                snapshot = dict(zip(moving, snaps[best_epoch]))
            for epoch in (() if deferred else range(cfg["epochs"])):
                for mb in minibatches(epoch, cfg["train_batch_size"]):
                    run(mb, True)
                train_loss = trainer.pop_mean_loss()    # mean of the step losses; also joins the update stream (parameters are final)
                vl = [run(mb, False).clone() for mb in minibatches(cfg["epochs"], cfg["val_batch_size"])]
                trainer.on_epoch_end()
                val_loss = float(torch.stack(vl).mean().item()) if vl else train_loss
                history.append({"epoch": epoch, "train_loss": train_loss, "val_loss": val_loss,
                                "lr": trainer.optimizer.param_groups[0]["lr"]})
                if best is None or val_loss < best:                    # ModelCheckpoint(save_top_k=1, monitor loss/val_loss)
                    # Lightning writes the file at every improvement and deletes the previous best; what survives an update is ONE file,
                    # the best epoch's.  Here an improvement snapshots the tensors that move (a few device-to-device copies) and the file
                    # is written once, after the last epoch -- same name, same content (`checkpoint_every_improvement: True` restores the
                    # write per improvement for those who want the intermediate files to survive a crash mid-update)
                    if rank == 0 and best_path is not None and best_path.exists():
                        best_path.unlink()
                    best, best_epoch = val_loss, epoch
                    best_path = save_dir / f"carla_episode={e_i}-epoch={epoch}-val_loss={val_loss:.3f}.ckpt"   # training_builder.py:133
                    own = self.train_model.state_dict()
                    snapshot = {k: own[k].detach().clone() for k in moving}
                    if rank == 0 and cfg.get("checkpoint_every_improvement", False):
                        self._write_checkpoint(best_path, base_cpu, snapshot, epoch, e_i)
        except BaseException:
            uncommitted()
            raise
        finally:
            trainer.close()             # the data-parallel hooks sit on the model-owned engine: detach them also when an epoch raises
        mark("epochs")
        try:
            if base_cpu is None:               # first update of a policy without a checkpoint: the frozen part comes off the device once
                base_cpu = {k: v.detach().cpu() for k, v in self.train_model.state_dict().items()}
            if rank == 0 and not cfg.get("checkpoint_every_improvement", False):
                self._write_checkpoint(best_path, base_cpu, snapshot, best_epoch, e_i)
        except BaseException:
            uncommitted()
            raise
        mark("write_checkpoint")
        if undo_reset is not None:           # committed: the rows are freed off the critical path (the main thread goes on to the reload)
            import threading
            box, undo_reset = [undo_reset], None
            threading.Thread(target=box.clear, daemon=True).start()
        if process_group is not None:
            torch.distributed.barrier(group=process_group)         # the checkpoint of rank 0 is on disk before anyone reloads
        self.last_fit = {"history": history, "best_val_loss": best, "checkpoint": best_path.as_posix(), "lr": lr}
        self.update_training_ckpt()
        # the snapshot pairs with the file THIS update wrote and with this very training model (set_mode / load_model drop it; a
        # checkpoint directory in which update_training_ckpt resolves to another file never takes the fast path)
        self._mem_ckpt = {"path": best_path.as_posix(), "moving": snapshot, "base_cpu": base_cpu, "model": self.train_model,
                          "trunk": trunk_after if trunk_after is not None else self._trunk_version(self.train_model)}
        # refresh the inference model (rlft_pluto.py:244-246).  Its frozen trunk equals the training model's when both came from the same
        # checkpoint (or the training model was copied from it) and nobody has touched it since: then only the moving tensors are copied,
        # in place, device to device -- and the engine reads exactly those through their pointers, so no re-bind either
        if infer_in_sync and self.checkpoint == best_path.as_posix():
            with torch.no_grad():
                own = self.pluto_model.state_dict()
                for k, v in snapshot.items():
                    if k in own:
                        own[k].copy_(v)
        else:
            self.pluto_model.load_state_dict(self.load_infer_checkpoint(self.checkpoint, self.device))
        self._infer_bound = (self.checkpoint, self.pluto_model._tensor_version())
        mark("reload")
        if not buffer_reset:
            self.buffer.reset_buffer()
        mark("reset_buffer")
        self.last_fit["timing"] = {b[0] + "_s": b[1] - a[1] for a, b in zip(marks, marks[1:])}
        self.last_fit["loaded_from"] = loaded_from
        return self.last_fit


class _GroupRelativePluto(RLFTPluto):
    """Shared rollout side of RIFT and GRPO (rift_pluto.py:74-161): in train mode every tick also yields, per CBV, the raw logits of
    its valid reference lines (the "old policy" of the coming update) and the group-relative advantage of all R x 12 candidates."""
    EXTRA_COLUMNS = ('CBVs_actions_old_group_logits', 'CBVs_group_advantage')

    def __init__(self, config, logger):
        super().__init__(config, logger)
        self._traj_evaluator = None
        self._fused_tick = bool(config.get('fused_tick', True))       # False: the per-CBV evaluator chain (the two are bit-identical, tested)
        self._tick_columns, self._tick_reads = {}, []

    @property
    def traj_evaluator(self):
        if self._traj_evaluator is None:
            from rift_amd.planning.fine_tuner.rlft.traj_eval.traj_evaluator import TrajEvaluator
            self._traj_evaluator = TrajEvaluator(self.pluto_model.engine(), dt=self._step_interval)
        eng = self.pluto_model._engine
        if eng is not None and self._traj_evaluator.engine is not eng:      # the model re-bound (precision / device change): the PID state is torch's, it carries over
            self._traj_evaluator.engine = eng
        return self._traj_evaluator

    def _get_action(self, CBVs_obs_list, infos, deterministic=False):
        self._tick_reads = []                    # (a tick that raised half-way must not leave its read-backs to the next one)
        return super()._get_action(CBVs_obs_list, infos, deterministic)

    def _begin_env(self, env_id, CBVs_obs, data, out, states):
        """Train mode: the group advantages of the environment's CBVs are ISSUED here, ahead of the per-CBV decisions, so that the device
        evaluates them (0.3 ms per CBV, serial through the shared PID state) while the host trims candidates and runs the PIDs of the loop
        behind it.  The model outputs that loop needs are read back first -- a read-back behind the evaluation would wait for it.  In tick
        order (the evaluator's PID state is shared and never reset, as the reference's): one fused call (rift_group_advantage_tick: one C-ABI
        call, one staged upload) when every CBV's valid reference lines are a prefix of its rows (they are: PlutoFeature pads behind them),
        else -- or with config['fused_tick'] = False -- the per-CBV chain of TrajEvaluator.get_grpo_advantage.  The results stay on the
        device until _finish_columns reads the tick back."""
        self._tick_columns = {}
        if self.mode != 'train':
            return
        for key in ("candidate_trajectories", "probability", "output_ref_free_trajectory", "ref_probability") + (("output_prediction",) if self._render and self._use_prediction else ()):
            if key in out:
                self._host(out, key)
        src = self.state_source
        pending = []
        for index, (cbv_id, obs) in enumerate(CBVs_obs.items()):
            # the CBV's reference lines come from its own observation (host memory: the rows the collated batch was built from), so the
            # ragged valid points of every valid line are cut on the host -- the device-side boolean indexing of the collated batch was a
            # synchronisation per line
            rl = obs['raw_pluto_feature'].data["reference_line"]
            valid = np.asarray(rl["valid_mask"]).astype(np.bool_)                   # (R of this CBV, 120)
            keep = valid.any(-1)
            pos, ang = np.asarray(rl["position"]), np.asarray(rl["orientation"])
            lines = np.nonzero(keep)[0]
            raster = src.off_road_raster(env_id, cbv_id)
            actors = src.nearby_actor_states(env_id, cbv_id)
            # (a source without these inputs raises NotImplementedError: the reference's advantage always carries both penalties; zeros
            # only when the source says so explicitly or, for collisions, when the CBV has no neighbours right now)
            if raster is None:
                raise RuntimeError(f"{type(src).__name__}.off_road_raster returned None: return (mask, pose) or NoFlagSource.ALL_CLEAR")
            column = {"advantage": None, "valid_mask": np.ones((int(keep.sum()), 12), dtype=np.bool_)}
            self._tick_columns[cbv_id] = column
            pending.append({
                "batch_index": index, "center_state": states[cbv_id].rollout_tuple(), "lines": lines,
                "ref_pos": [pos[r][valid[r]] for r in lines], "ref_angle": [ang[r][valid[r]] for r in lines],     # ragged: only the valid points of each valid line
                "actors": None if (actors is None or actors is NoFlagSource.ALL_CLEAR) else actors,
                "off_road": None if raster is NoFlagSource.ALL_CLEAR else raster, "column": column})
        ev = self.traj_evaluator
        prefix = all(len(v["lines"]) and int(v["lines"][-1]) == len(v["lines"]) - 1 for v in pending)
        if self._fused_tick and prefix:
            adv = ev.engine.group_advantage_tick(out["trajectory"], pending, ev.pid_state)
            self._tick_reads.append((adv, [(v["column"], k, len(v["lines"])) for k, v in enumerate(pending)]))
            return
        for v in pending:
            kw = {}
            G = len(v["lines"]) * 12
            if v["off_road"] is None:
                kw["off_road_matrix"] = np.zeros((G, 80), dtype=np.bool_)
            else:
                kw["off_road_mask"], kw["center_pose"] = v["off_road"]
            if v["actors"] is None:
                kw["collision_matrix"] = np.zeros((G, 40), dtype=np.bool_)
            else:
                kw["nearby_actor_states"] = v["actors"]
            traj = out["trajectory"][v["batch_index"]]
            if len(v["lines"]) != traj.shape[0] or not prefix:
                traj = traj[v["lines"].tolist()]
            v["column"]["advantage"] = ev.get_grpo_advantage(v["center_state"], traj, v["ref_pos"], v["ref_angle"], to_host=False, **kw)["advantage"]

    def _group_columns(self, env_id, cbv_id, obs, data, out, index, state, decision) -> Dict[str, Any]:
        """The old-policy logits of the CBV's valid lines; its group advantage was issued in _begin_env."""
        logits = decision.probability[self._host_line_mask(obs, out)]
        return {'CBVs_actions_old_group_logits': {'logits': logits, 'valid_mask': np.ones_like(logits, dtype=np.bool_)},
                'CBVs_group_advantage': self._tick_columns[cbv_id]}

    def _finish_columns(self, result):
        for adv, cols in self._tick_reads:           # one read-back per environment of the tick
            host = adv.cpu().numpy()
            for column, k, R in cols:
                column["advantage"] = host[k, :R].copy()
        self._tick_reads = []
        super()._finish_columns(result)

    @staticmethod
    def _host_line_mask(obs, out) -> np.ndarray:
        """Valid reference lines of one CBV as a mask over the rows of the tick's collated batch."""
        keep = np.asarray(obs['raw_pluto_feature'].data["reference_line"]["valid_mask"]).astype(np.bool_).any(-1)
        mask = np.zeros(out["probability"].shape[1], dtype=np.bool_)
        mask[:keep.shape[0]] = keep
        return mask

    def _per_cbv(self, env_id, cbv_id, obs, data, out, index, state, decision):
        if self.mode != 'train':
            return {k: None for k in self.EXTRA_COLUMNS}
        return self._group_columns(env_id, cbv_id, obs, data, out, index, state, decision)


class RIFTPluto(_GroupRelativePluto):        # fine_tuner/rlft/rift_pluto/rift_pluto.py:18
    name, type, kind = 'rift_pluto', 'rlft', 'rift'


class GRPOPluto(_GroupRelativePluto):        # fine_tuner/rlft/grpo_pluto/grpo_pluto.py:18-170
    name, type, kind = 'grpo_pluto', 'rlft', 'grpo'
    EXTRA_COLUMNS = _GroupRelativePluto.EXTRA_COLUMNS + ('CBVs_actions_ref_group_logits',)

    def __init__(self, config, logger):
        super().__init__(config, logger)
        self.ref_model = PlanningModel(radius=self.radius).to(self.device)      # frozen reference policy of the KL term (:24-28)
        self.ref_model.compute_precision = self.compute_precision
        if self._ckpt_path:
            self.ref_model.load_state_dict(self.load_infer_checkpoint(self._ckpt_path, self.device))
        self.ref_model.eval()

    @torch.no_grad()
    def _forward(self, CBVs_obs):
        data, out = super()._forward(CBVs_obs)
        self.ref_model.need_traj = False
        out["ref_probability"] = self.ref_model(data)["probability"]
        return data, out

    def _per_cbv(self, env_id, cbv_id, obs, data, out, index, state, decision):
        cols = super()._per_cbv(env_id, cbv_id, obs, data, out, index, state, decision)
        if self.mode == 'train':
            r_valid = self._host_line_mask(obs, out)
            logits = self._host(out, "ref_probability")[index][r_valid]
            cols['CBVs_actions_ref_group_logits'] = {'logits': logits, 'valid_mask': np.ones_like(logits, dtype=np.bool_)}
        return cols


class ReinforcePluto(RLFTPluto):   # fine_tuner/rlft/reinforce_pluto/reinforce_pluto.py
    name, type, kind = 'reinforce_pluto', 'rlft', 'reinforce'

    def preprocess_buffer(self, trainer, replay):
        """reinforce_datamodule.py:112-124: discounted return scan over the whole buffer (device)."""
        rewards = torch.as_tensor(np.stack(self.buffer.get_key_data('CBVs_reward'), axis=0)).double().view(-1)
        dones = torch.as_tensor(np.stack(self.buffer.get_key_data('CBVs_done'), axis=0)).float().view(-1)
        ret = trainer.engine.discounted_return(rewards, dones, self.cfg["gamma"])
        return {"returns": ret.float()}


class PPOPluto(RLFTPluto):         # fine_tuner/rlft/ppo_pluto/ppo_pluto.py:40-120
    name, type, kind = 'ppo_pluto', 'learnable', 'ppo'
    STREAM_KEYS = ('CBVs_obs', 'CBVs_next_obs')          # the second buffer sweep reads the next observations (ppo_datamodule.py:141-150)

    def __init__(self, config, logger):
        super().__init__(config, logger)
        ppo = dict(config.get('ppo', {}))
        self.hidden_dim = list(ppo.get('hidden_dim', [256, 256]))      # planning/config/ppo_pluto.yaml:42-48
        self.state_dim, self.action_dim = ppo.get('state_dim', 128), ppo.get('action_dim', 3)
        self.clip_epsilon, self.lambda_entropy = ppo.get('clip_epsilon', 0.2), ppo.get('lambda_entropy', 0.01)
        self.cfg["trainable_layers"] = ["planning_decoder.pi_head", "value_net"]   # ppo_training.yaml:26-28

    def set_mode(self, mode):
        self.mode = mode
        if mode == 'train':
            from rift_amd.planning.fine_tuner.rlft.ppo_pluto.ppo_pluto import PPOPlutoModel
            self._mem_ckpt = None
            self.train_model = PPOPlutoModel(radius=self.radius, state_dim=self.state_dim, action_dim=self.action_dim,
                                             hidden_dim=self.hidden_dim, clip_epsilon=self.clip_epsilon,
                                             lambda_entropy=self.lambda_entropy).to(self.device)
            self.train_model.compute_precision = self.compute_precision
            self.train_model.train()
            self.pluto_model.eval()
        elif mode == 'eval':
            self.pluto_model.eval()
        else:
            raise ValueError(f'Unknown mode {mode}')

    def _sweep(self, trainer, replay: DeviceReplay, seed0: int):
        """hidden (n,128) and value (n) of every scene of `replay`, in chunks of the train batch size, model in train mode under
        no_grad as in ppo_datamodule.py:127-150."""
        eng, bs = trainer.engine, self.cfg["train_batch_size"]
        hidden, value = [], []
        for s in range(0, replay.n, bs):
            idx = torch.arange(s, min(s + bs, replay.n), dtype=torch.int32, device=self.device)
            fb, _ = replay.collate(eng, idx, int(replay.r_count_cpu[s:s + bs].max()))
            h = trainer.forward_hidden(fb, seed0 + s // bs)
            hidden.append(h.clone())
            value.append(self.train_model.value_net(h))
        return torch.cat(hidden), torch.cat(value)

    def preprocess_buffer(self, trainer, replay):
        """ppo_datamodule.py:117-174: two full-buffer sweeps (current / next observation) -> GAE (gamma = lambda = 0.98) ->
        reward_sum = advantage + value -> buffer-wide normalisation; all on the device."""
        buf, eng = self.buffer, trainer.engine
        rewards = torch.as_tensor(np.stack(buf.get_key_data('CBVs_reward'), axis=0)).double().view(-1)
        undones = 1.0 - torch.as_tensor(np.stack(buf.get_key_data('CBVs_done'), axis=0)).float().view(-1)
        unterm = 1.0 - torch.as_tensor(np.stack(buf.get_key_data('CBVs_terminated'), axis=0)).float().view(-1)
        old_log_prob = torch.as_tensor(np.stack(buf.get_key_data('CBVs_actions_old_log_prob'), axis=0)).float().view(-1)
        action_mode = torch.as_tensor(np.stack(buf.get_key_data('CBVs_actions_mode'), axis=0)).long().view(-1, 2)
        state, value = self._sweep(trainer, replay, 1 << 20)
        _, next_value = self._sweep(trainer, self._arena('CBVs_next_obs'), 1 << 21)
        adv = eng.gae(rewards, undones, value, next_value, unterm, self.cfg["gamma"], self.cfg["lambda_gae_adv"])
        reward_sum = adv + value
        adv = eng.normalize_advantage_(adv.clone())
        dev = self.device
        return {"state": state, "advantage": adv, "reward_sum": reward_sum, "old_log_prob": old_log_prob.to(dev),
                "action_mode": action_mode.to(dev)}


def replay_dummy_extras(feature):
    R = feature["reference_line"]["position"].shape[0]
    return {"group_advantage": torch.zeros(R, 12, dtype=torch.float64), "group_advantage_mask": torch.ones(R, 12, dtype=torch.bool),
            "old_group_logits": torch.zeros(R, 12), "old_group_logits_mask": torch.ones(R, 12, dtype=torch.bool)}


CBV_POLICY_LIST = {   # rift/cbv/planning/__init__.py:21-34 (the Pluto family; the MLP-RL and rule-based entries are other policies)
    'pluto': PLUTO, 'rift_pluto': RIFTPluto, 'grpo_pluto': GRPOPluto, 'reinforce_pluto': ReinforcePluto, 'ppo_pluto': PPOPluto,
}
