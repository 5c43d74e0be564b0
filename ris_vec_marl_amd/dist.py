"""Multi-GPU layer: envs are independent, so they shard by contiguous env-index blocks
with NO data-path collective; the single exchange is an all-gather of the joint
observation for a replicated global critic (global_sac_critic.py:47-62 consumes a
[B, 5V] joint state).  One process per GPU, `torch.distributed` backend "nccl"
(= RCCL over xGMI on ROCm); "gloo" on CPU for tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as td


def shard_range(n_envs_global: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [start, start+count) of global env ids owned by `rank`.
    The first (n mod world) ranks get one extra env."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside [0, %d)" % (rank, world_size))
    base, extra = divmod(int(n_envs_global), int(world_size))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    `python -m torch.distributed.run`).  Returns (rank, world_size, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hooks (several ranks on a one-GPU box): RISVEC_DIST_BACKEND=gloo keeps RCCL out
    # of the picture, RISVEC_DEVICE_INDEX pins every rank to one device
    if "RISVEC_DEVICE_INDEX" in os.environ:
        local = int(os.environ["RISVEC_DEVICE_INDEX"])
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("RISVEC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class JointObsGather:
    """All-gather of the per-rank observation block obs_local [E_local, V, 5] into the joint
    observation [E_global, 5V] (the input a replicated global critic consumes).

    The env overwrites its `obs` tensor every step, so `start()` first snapshots it into a
    private staging buffer on the caller's stream (one 20V-byte-per-env device copy), then
    issues the RCCL all-gather from that snapshot on a side stream into one of two output
    buffers: the collective overlaps the following env steps instead of sitting on their
    critical path, and nothing it reads or writes is touched by them.  `wait()` makes the
    result visible to the caller's stream.  xGMI is point-to-point, so the gather is per-link
    bound ((N-1) x message bytes into every GPU): pick its cadence accordingly.  Equal shard
    sizes are required (use an E divisible by the world size).

    A single process (no process group, or a group of one rank) runs the SAME staging / side-stream
    / double-buffer sequence with the collective replaced by a device copy (a one-rank group still
    issues the collective), so the single-GPU tests exercise the logic the multi-GPU run relies on."""

    def __init__(self, n_envs_local: int, n_veh: int, device, group=None, n_buffers: int = 2):
        self.group = group
        self.collective = td.is_available() and td.is_initialized()
        self.world = td.get_world_size(group) if self.collective else 1
        self.backend = td.get_backend(group) if self.collective else "none"
        self.device = torch.device(device)
        if self.collective and self.backend == "nccl" and self.device.type != "cuda":
            raise ValueError("JointObsGather: the nccl (RCCL) backend needs a cuda device, got %s" % self.device)
        if n_buffers < 1:
            raise ValueError("JointObsGather: n_buffers must be >= 1")
        self.n_envs_local = int(n_envs_local)
        self.width = 5 * int(n_veh)
        self.stage = torch.empty(self.n_envs_local, self.width, dtype=torch.float32, device=self.device)
        self.bufs = [torch.empty(self.world * self.n_envs_local, self.width, dtype=torch.float32, device=self.device)
                     for _ in range(n_buffers)]
        self.i = 0
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._work = None
        self.n_started = 0

    def start(self, obs_local: torch.Tensor) -> torch.Tensor:
        """Snapshot `obs_local` and launch its gather; returns the buffer it will land in
        (valid after `wait()`)."""
        if obs_local.shape[0] != self.n_envs_local or obs_local.numel() != self.n_envs_local * self.width:
            raise ValueError("JointObsGather.start: obs_local must be [%d, V, 5] with 5V = %d, got %s"
                             % (self.n_envs_local, self.width, tuple(obs_local.shape)))
        self.wait()                                   # the previous gather still reads `stage`
        out = self.bufs[self.i]
        self.i = (self.i + 1) % len(self.bufs)
        self.n_started += 1
        src = obs_local.reshape(self.n_envs_local, self.width)
        self.stage.copy_(src, non_blocking=True)      # caller's stream: ordered after the step that wrote obs
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                self._issue(out)
        else:
            self._issue(out)
        return out

    def _issue(self, out: torch.Tensor) -> None:
        if self.collective:
            self._work = td.all_gather_into_tensor(out, self.stage, group=self.group, async_op=True)
        else:
            out.copy_(self.stage, non_blocking=True)

    def wait(self) -> None:
        """Make the last started gather visible to the current stream."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)


def gather_joint_obs(obs_local: torch.Tensor, group=None) -> torch.Tensor:
    """Blocking convenience form: [E_local, V, 5] -> [E_global, 5V]."""
    g = JointObsGather(obs_local.shape[0], obs_local.shape[1], obs_local.device, group=group, n_buffers=1)
    out = g.start(obs_local)
    g.wait()
    return out


def combine_episode_summary(summary: torch.Tensor, n_envs_local: int, group=None) -> torch.Tensor:
    """`EpisodeMeter.summary` [3, C] (mean / min / max over this rank's envs) -> the same over the envs
    of every rank: the mean weighted by the shard sizes, the min of mins, the max of maxes.  Three
    C-element all-reduces once per EPISODE (the per-step path has no collective); the envs' own sums
    never leave their GPU.  Returns a new tensor; a single process gets a copy."""
    out = summary.clone()
    if not td.is_initialized() or td.get_world_size(group) == 1:
        return out
    tot = torch.cat([summary[0] * float(n_envs_local), summary.new_tensor([float(n_envs_local)])])
    td.all_reduce(tot, op=td.ReduceOp.SUM, group=group)
    td.all_reduce(out[1], op=td.ReduceOp.MIN, group=group)
    td.all_reduce(out[2], op=td.ReduceOp.MAX, group=group)
    out[0] = tot[:-1] / tot[-1]
    return out
