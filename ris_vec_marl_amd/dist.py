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

    The env overwrites its `obs` tensor every step, so `start()` first snapshots it into a private
    staging buffer on the caller's stream (one 20V-byte-per-env device copy), then issues the RCCL
    all-gather from that snapshot on a side stream into an output buffer: the collective overlaps the
    following env steps instead of sitting on their critical path, and nothing it reads or writes is
    touched by them.  There are `n_buffers` (stage, output) SLOTS used round-robin, so up to `n_buffers`
    gathers can be in flight: `start()` only ever waits -- on the device, not on the host -- for the gather
    that last used the slot it is about to refill (round 2 had one staging buffer and began every `start()`
    by waiting for the previous gather: with a gather every step the step stream stalled on the links).
    `wait()` makes the most recent gather (and, the side stream being in order, every earlier one) visible
    to the caller's stream; the tensor `start()` returned is valid from then until its slot is reused,
    `n_buffers` starts later.  xGMI is point-to-point, so the gather is per-link bound ((N-1) x message
    bytes into every GPU): pick its cadence accordingly.  Equal shard sizes are required (use an E
    divisible by the world size).

    A single process (no process group, or a group of one rank) runs the SAME staging / side-stream /
    slot sequence with the collective replaced by a device copy (a one-rank group still issues the
    collective), so the single-GPU tests exercise the logic the multi-GPU run relies on."""

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
        mk = lambda n: torch.empty(n, self.width, dtype=torch.float32, device=self.device)   # noqa: E731
        self.stages = [mk(self.n_envs_local) for _ in range(n_buffers)]
        self.bufs = [mk(self.world * self.n_envs_local) for _ in range(n_buffers)]
        self.i = 0
        on_gpu = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if on_gpu else None
        self._done = [torch.cuda.Event() if on_gpu else None for _ in range(n_buffers)]     # side stream: slot's gather finished
        self._busy = [False] * n_buffers            # slot has a gather whose completion nobody has ordered against yet
        self._work = [None] * n_buffers             # host-side handles (gloo on CPU: no stream to order on)
        self._last = None                           # slot of the most recent start()
        self.n_started = 0

    @property
    def stage(self) -> torch.Tensor:                # the staging buffer of the slot the next start() fills
        return self.stages[self.i]

    def in_flight(self) -> int:
        """Gathers started and not yet ordered against by a `wait()` / a slot reuse."""
        return sum(self._busy)

    def start(self, obs_local: torch.Tensor) -> torch.Tensor:
        """Snapshot `obs_local` and launch its gather; returns the buffer it will land in
        (valid after `wait()`, until the slot is reused `n_buffers` starts later)."""
        if obs_local.shape[0] != self.n_envs_local or obs_local.numel() != self.n_envs_local * self.width:
            raise ValueError("JointObsGather.start: obs_local must be [%d, V, 5] with 5V = %d, got %s"
                             % (self.n_envs_local, self.width, tuple(obs_local.shape)))
        k = self.i
        self.i = (self.i + 1) % len(self.bufs)
        self._release(k)                              # only the gather that last used THIS slot still reads stage[k]
        out, stage = self.bufs[k], self.stages[k]
        self.n_started += 1
        self._last = k
        src = obs_local.reshape(self.n_envs_local, self.width)
        stage.copy_(src, non_blocking=True)           # caller's stream: ordered after the step that wrote obs
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                if self.collective:
                    # async_op: the communicator's own stream runs the collective; wait() here orders the SIDE stream
                    # behind it (a device-side dependency), so the event below marks the gather's completion
                    td.all_gather_into_tensor(out, stage, group=self.group, async_op=True).wait()
                else:
                    out.copy_(stage, non_blocking=True)
                self._done[k].record(self.stream)
        elif self.collective:
            self._work[k] = td.all_gather_into_tensor(out, stage, group=self.group, async_op=True)
        else:
            out.copy_(stage)
        self._busy[k] = True
        return out

    def _release(self, k: int) -> None:
        """Order the caller's stream (or the host, without streams) behind slot k's gather."""
        if not self._busy[k]:
            return
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_event(self._done[k])
        elif self._work[k] is not None:
            self._work[k].wait()
            self._work[k] = None
        self._busy[k] = False

    def wait(self) -> None:
        """Make every gather started so far visible to the current stream."""
        if self.stream is not None:
            if self._last is not None and self._busy[self._last]:
                torch.cuda.current_stream(self.device).wait_event(self._done[self._last])   # in-order side stream: covers all
            self._busy = [False] * len(self._busy)
        else:
            for k in range(len(self._busy)):
                self._release(k)


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
