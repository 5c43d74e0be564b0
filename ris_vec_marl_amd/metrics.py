"""f4: the per-episode metrics sink of the reference's driver (`marl_train_bcd.py`, TRAIN).

Around every `env.step` the driver adds the env's `last_*` scalars, the clipped per-user rewards and
the equivalent powers to Python floats (TRAIN:1611-1662, 1714-1753, 1769) and at the end of the
episode writes their means / sums to TensorBoard (TRAIN:1824-1865, 1927-2048).  `EpisodeMeter` keeps
those sums for all E envs on the device (`risvec_episode_*`, float64, one launch per step) and reduces
the episode's scalars over the envs in a fixed order; `ScalarSink` writes them under the reference's
tags to a TensorBoard event file (written directly: TFRecord framing + the two protobuf messages a
scalar needs; the tensorboard package is not required) and/or a JSON-lines file.

No CPU path: the sums live in HBM and every method launches HIP kernels through the C ABI.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import time
from typing import Dict, Iterator, List, Optional, Tuple

import torch

from . import _native as N

# column -> TensorBoard tag(s) of TRAIN:1927-2048; columns as RISVEC_EP_* in include/risvec.h
COLUMNS: Tuple[str, ...] = (
    "reward/global_avg", "traffic/offload_kbit_ep", "traffic/local_kbit_ep", "queue/mec_cycles",
    "queue/backlog_kbit_ep_mean", "delay/local_ep_mean", "delay/edge_queue_ep_mean", "delay/edge_compute_ep_mean",
    "delay/tx_ep_mean", "queue/mec_util_ep_mean", "cpu/local_util_ep_mean", "qos/violation_rate_ep_mean",
    "delay/episode_mean", "energy/episode_mean", "power/offload_avg", "power/local_avg", "power/total_avg",
    "reward/min_user", "reward/var_user", "reward/jain", "reward/best_global",
)
assert len(COLUMNS) == N.EP_COLS
# the same values under the second name the driver also logs them as (TRAIN:1930-1934)
ALIASES: Dict[str, Tuple[str, float]] = {
    "power/total_W": ("power/total_avg", 1.0), "power/local_W": ("power/local_avg", 1.0),
    "power/offload_W": ("power/offload_avg", 1.0), "abs/delay_ms": ("delay/episode_mean", 1000.0),
    "abs/energy_J": ("energy/episode_mean", 1.0),
}


class EpisodeMeter:
    """Per-env episode accumulators for E envs of V users.

        meter = EpisodeMeter(env)            # or EpisodeMeter(n_envs=E, n_veh=V, device=...)
        meter.begin_episode()                # where the driver zeroes its ep_* sums (TRAIN:1278-1300)
        for each step:  env.step(...);  meter.accumulate(env)      # or the launcher from bind(env)
        scalars = meter.end_episode()        # {tag: mean over envs}; .per_env [E,21], .summary [3,21]
    """

    def __init__(self, env=None, n_envs: Optional[int] = None, n_veh: Optional[int] = None, device=None,
                 user_clip: float = 5.0):
        N.load()
        if env is not None:
            n_envs, n_veh, device = env.n_envs, env.n_veh, env.device
        self.device = N.resolve_device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("ris_vec_marl_amd needs a HIP device; there is no CPU fallback")
        self.n_envs, self.n_veh, self.user_clip = int(n_envs), int(n_veh), float(user_clip)
        if self.n_envs < 1 or not 1 <= self.n_veh <= N.MAX_VEH:
            raise ValueError("EpisodeMeter: n_envs=%d n_veh=%d" % (self.n_envs, self.n_veh))
        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=self.device)   # noqa: E731
        self.acc = z(N.EP_FIXED + self.n_veh, self.n_envs)     # one row per accumulated quantity, envs contiguous
        self.per_env = z(self.n_envs, N.EP_COLS)
        self.summary = z(3, N.EP_COLS)                     # mean / min / max over the envs
        self._partial = z(int(N.load().risvec_episode_partial_rows(self.n_envs)), 3, N.EP_COLS)
        self.n_steps = 0
        self._last_metrics: Optional[torch.Tensor] = None
        self.begin_episode()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def begin_episode(self) -> None:
        N.check(N.load().risvec_episode_clear(self.n_envs, self.n_veh, self.acc.data_ptr(), self._stream()))
        self.n_steps = 0

    def _check(self, metrics, reward, power_w):
        E, V = self.n_envs, self.n_veh

        def ok(t, shape):
            return (t.dtype == torch.float32 and t.device == self.device and t.is_contiguous()
                    and tuple(t.shape) == shape)
        if not ok(metrics, (E, N.METRICS)) or not ok(reward, (E, V)) or (power_w is not None and not ok(power_w, (E, 2, V))):
            raise ValueError("EpisodeMeter: metrics [E,%d], reward [E,V], power_w [E,2,V] must be contiguous float32 "
                             "tensors on %s" % (N.METRICS, self.device))

    def bind(self, env=None, metrics=None, reward=None, power_w=None):
        """Validate once; returns `launch()`, one C-ABI call adding the CURRENT contents of the env's
        metrics / reward / power_w tensors (read in place) to the episode sums."""
        if env is not None:
            t = env.tensors
            metrics, reward, power_w = t["metrics"], t["reward"], t["power_w"]
        self._check(metrics, reward, power_w)
        fn, check = N.load().risvec_episode_accumulate, N.check
        args = (self.n_envs, self.n_veh, metrics.data_ptr(), reward.data_ptr(),
                None if power_w is None else power_w.data_ptr(), self.user_clip, self.acc.data_ptr())
        stream = self._stream()
        self._last_metrics = metrics

        def launch() -> None:
            check(fn(*args, stream))
            self.n_steps += 1

        launch.keepalive = (metrics, reward, power_w)
        return launch

    def accumulate(self, env=None, metrics=None, reward=None, power_w=None) -> None:
        self.bind(env, metrics, reward, power_w)()

    def summarize(self, metrics: Optional[torch.Tensor] = None) -> None:
        """Fill `per_env` [E,21] and `summary` [3,21] (mean / min / max over envs) on the device;
        asynchronous (two launches), nothing is copied to the host."""
        m = metrics if metrics is not None else self._last_metrics
        if m is None or self.n_steps < 1:
            raise RuntimeError("EpisodeMeter: no step was accumulated")               # ep_steps == 0 -> nan (TRAIN:1853)
        N.check(N.load().risvec_episode_summary(self.n_envs, self.n_veh, self.n_steps, self.acc.data_ptr(), m.data_ptr(),
                                               self.per_env.data_ptr(), self._partial.data_ptr(),
                                               self.summary.data_ptr(), self._stream()))

    def end_episode(self, metrics: Optional[torch.Tensor] = None) -> Dict[str, float]:
        """`summarize()` and return {tag: mean over envs} (this call synchronises)."""
        self.summarize(metrics)
        mean = self.summary[0].tolist()
        out = dict(zip(COLUMNS, mean))
        for alias, (src, scale) in ALIASES.items():
            out[alias] = out[src] * scale
        return out

    def spread(self) -> Dict[str, Tuple[float, float]]:
        """{tag: (min, max) over the envs} of the last `end_episode`."""
        lo, hi = self.summary[1].tolist(), self.summary[2].tolist()
        return {c: (lo[i], hi[i]) for i, c in enumerate(COLUMNS)}

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self) -> Dict[str, object]:
        return {"acc": self.acc.detach().cpu().clone(), "n_steps": self.n_steps, "user_clip": self.user_clip}

    def load_state_dict(self, sd) -> None:
        if tuple(sd["acc"].shape) != tuple(self.acc.shape):
            raise ValueError("EpisodeMeter.load_state_dict: acc %s != %s" % (tuple(sd["acc"].shape), tuple(self.acc.shape)))
        self.acc.copy_(sd["acc"].to(self.device))
        self.n_steps, self.user_clip = int(sd["n_steps"]), float(sd["user_clip"])


# ---------------------------------------------------------------------------------------------------
# TensorBoard event files, written directly.  A file is a sequence of TFRecords
#   u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)
# whose payloads are serialised `Event` messages: wall_time = 1 (double), step = 2 (int64),
# file_version = 3 (string, first record only), summary = 5 { value = 1 { tag = 1 (string),
# simple_value = 2 (float) } }.
def _crc32c_table() -> List[int]:
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _crc32c_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1                    # int64 on the wire: two's complement, 10 bytes when negative
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int, scalars: Optional[List[Tuple[str, float]]] = None,
                 file_version: Optional[str] = None) -> bytes:
    ev = b"\x09" + struct.pack("<d", float(wall_time)) + b"\x10" + _varint(int(step))
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if scalars:
        summary = b"".join(_ld(1, _ld(1, tag.encode()) + b"\x15" + struct.pack("<f", float(v))) for tag, v in scalars)
        ev += _ld(5, summary)
    return ev


def frame_record(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", _masked_crc(head)) + payload + struct.pack("<I", _masked_crc(payload))


def _read_varint(buf: bytes, i: int) -> Tuple[int, int]:
    n = shift = 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, i
        shift += 7


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, i = _read_varint(buf, i)
        elif wire == 1:
            v, i = buf[i:i + 8], i + 8
        elif wire == 2:
            n, i = _read_varint(buf, i)
            v, i = buf[i:i + n], i + n
        elif wire == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield field, wire, v


def read_events(path: str) -> List[Tuple[float, int, str, float]]:
    """[(wall_time, step, tag, value)] of an event file, checking every record's CRCs."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        (c1,) = struct.unpack("<I", data[i + 8:i + 12])
        payload = data[i + 12:i + 12 + n]
        (c2,) = struct.unpack("<I", data[i + 12 + n:i + 16 + n])
        if c1 != _masked_crc(head) or c2 != _masked_crc(payload) or len(payload) != n:
            raise ValueError("%s: corrupt record at byte %d" % (path, i))
        i += 16 + n
        wall, step, summary = 0.0, 0, None
        for field, _, v in _fields(payload):
            if field == 1:
                (wall,) = struct.unpack("<d", v)
            elif field == 2:
                step = v - (1 << 64) if v >> 63 else v
            elif field == 5:
                summary = v
        if summary is None:
            continue
        for field, _, val in _fields(summary):
            if field != 1:
                continue
            tag, x = "", None
            for f2, _, v2 in _fields(val):
                if f2 == 1:
                    tag = v2.decode()
                elif f2 == 2:
                    (x,) = struct.unpack("<f", v2)
            if x is not None:
                out.append((wall, step, tag, x))
    return out


class ScalarSink:
    """`SummaryWriter`-shaped scalar writer (add_scalar / flush / close) for the driver's
    `writer.add_scalar(tag, value, i_episode)` calls (TRAIN:1927-2048): a TensorBoard event file in
    `log_dir` and, if `jsonl` is set, one JSON object per `add_scalars` call."""

    def __init__(self, log_dir: str, jsonl: bool = True, filename_suffix: str = ""):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        now = time.time()
        name = "events.out.tfevents.%010d.%s.%d.0%s" % (int(now), socket.gethostname(), os.getpid(), filename_suffix)
        self.path = os.path.join(log_dir, name)
        self._f = open(self.path, "wb")
        self._f.write(frame_record(encode_event(now, 0, file_version="brain.Event:2")))
        self._j = open(os.path.join(log_dir, "scalars.jsonl"), "a") if jsonl else None

    def add_scalar(self, tag: str, value: float, step: int, walltime: Optional[float] = None) -> None:
        self.add_scalars({tag: value}, step, walltime)

    def add_scalars(self, scalars: Dict[str, float], step: int, walltime: Optional[float] = None) -> None:
        wall = time.time() if walltime is None else walltime
        items = [(k, float(v)) for k, v in scalars.items()]
        self._f.write(frame_record(encode_event(wall, step, items)))
        if self._j is not None:
            self._j.write(json.dumps({"step": int(step), "wall_time": wall, **dict(items)}) + "\n")

    def write_episode(self, meter: EpisodeMeter, i_episode: int, extra: Optional[Dict[str, float]] = None) -> Dict[str, float]:
        """end_episode() of the meter -> the reference's tags, plus `extra` (tau, losses, ...)."""
        scalars = meter.end_episode()
        if extra:
            scalars.update(extra)
        self.add_scalars(scalars, i_episode)
        return scalars

    def flush(self) -> None:
        self._f.flush()
        if self._j is not None:
            self._j.flush()

    def close(self) -> None:
        self.flush()
        self._f.close()
        if self._j is not None:
            self._j.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
