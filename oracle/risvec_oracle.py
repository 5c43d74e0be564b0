"""CPU ORACLE for the RIS-VEC hot path -- TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy restatement of the reference simulator
`Simulation-MARL-BCD/Environment.py` (abbreviated ENV below), vectorised over a
leading batch axis of E independent environments.  It exists to CHECK the HIP
kernels in `ris_vec_marl_amd/csrc/`; it is never the thing measured or shipped.

    Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
    `bench.py` may import it.  The product package must not (and does not).

Parity status: PINNED.  `tools/capture_golden.py` imports the reference itself in
the CPU container, records every random draw it consumes, and stores inputs,
draws and outputs under `tests/golden/`; `tests/test_oracle_golden.py` checks each
function below against those captures (<= 1e-12 relative, float64).

Every random draw is an explicit argument (the reference consumes the global
MT19937 stream, ENV:384-409, 427-512, 299, 10, 17, 23-24, 717-719); the
counter-based Philox generator the kernels use in production is restated at the
bottom so its integer outputs can be compared bit for bit.

Array conventions (E envs, V vehicles, M RIS elements):
    pos [E,V,2] f64, direc [E,V] int (0='u',1='d',2='l',3='r'), vel [E,V] f64,
    theta [E,M] c128, h_r [E,V,M] c128, b [M] c128, gain [E,V] f64,
    action [E,2,V] f64, partner [E,V] int, n_groups [E] int.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# module constants (ENV:29-42)
# --------------------------------------------------------------------------
RIS_XYZ = (220.0, 220.0, 25.0)       # ENV:29
BS_XYZ = (0.0, 0.0, 25.0)            # ENV:32
RO = 10 ** -2                        # ENV:34
LAMB = 1                             # ENV:37
D_ELEM = 0.5                         # ENV:38
SIGMA = 10 ** (-7)                   # ENV:40
ALPHA1 = 2.2                         # ENV:41
ALPHA2 = 2.5                         # ENV:42
VEH_HEIGHT = 1.5                     # ENV:246, 305

DIR_U, DIR_D, DIR_L, DIR_R = 0, 1, 2, 3
DIR_CHARS = "udlr"

# partner encoding for batched NOMA groups (build-defined)
PARTNER_SINGLE = -1        # vehicle is alone in a 1-element group (OMA)
PARTNER_NONE = -2          # vehicle is in no group / a group of another size -> rate 0
PARTNER_SECOND = 1 << 16   # added to the partner index when the vehicle is listed 2nd


def default_lanes() -> Dict[str, list]:
    """Lane coordinates the reference driver passes in (marl_train_bcd.py:446-449)."""
    up = [i / 2.0 for i in [400 + 3.5 / 2, 400 + 3.5 + 3.5 / 2, 800 + 3.5 / 2, 800 + 3.5 + 3.5 / 2]]
    down = [i / 2.0 for i in [400 - 3.5 - 3.5 / 2, 400 - 3.5 / 2, 800 - 3.5 - 3.5 / 2, 800 - 3.5 / 2]]
    return dict(up=up, down=down, left=list(up), right=list(down))


@dataclasses.dataclass
class OracleParams:
    """Physics parameters; defaults are the class defaults of ENV:57-190."""
    bandwidth: float = 1.0                      # MHz, ENV:72
    noise_power: float = 10 ** ((-174 - 30) / 10) * 1.0e6   # ENV:74-76
    P_max: float = 1.0                          # ENV:125
    power_scale: float = 0.7                    # ENV:555
    qos_enable: bool = True                     # ENV:79
    R_min_bpsHz: float = 0.20                   # ENV:80
    D_max_s: float = 0.10                       # ENV:81
    qos_penalty: float = 5.0                    # ENV:82
    time_slow: float = 0.1                      # ENV:101
    time_fast: float = 0.001                    # ENV:102
    k: float = 1e-28                            # ENV:104
    f_local_max: float = 1.0e9                  # ENV:108
    f_edge_max: float = 2.0e9                   # ENV:109
    cycles_per_bit: float = 500.0               # ENV:111
    cpu_share_floor: float = 0.10               # ENV:113
    w_d: float = 0.5                            # ENV:138
    w_e: float = 3.0                            # ENV:139
    reward_clip: float = 50.0                   # ENV:143
    rate: float = 3.0                           # ENV:156
    fc_GHz: float = 3.5                         # ENV:186
    shadow_std_los: float = 4.0                 # ENV:187
    shadow_std_nlos: float = 7.0                # ENV:188
    rician_K_dB: float = 0.0                    # ENV:189
    vehAntGain: float = 3.0                     # ENV:96
    width: float = 400.0
    height: float = 400.0

    @staticmethod
    def yaml_effective() -> "OracleParams":
        """Values in force after the shipped config.yaml + Config overlay
        (marl_train_bcd.py:505-508, 563-594, 750-753; config.yaml mec/phy/env)."""
        p = OracleParams()
        p.bandwidth = 5.0
        p.noise_power = 10 ** ((-174 - 30) / 10) * 5.0e6
        p.P_max = 2.0
        p.f_local_max = 3.0e9
        p.cycles_per_bit = 300.0
        p.rate = 1.0
        p.w_d = 1.0
        p.w_e = 1.0
        p.R_min_bpsHz = 0.15
        p.D_max_s = 0.12
        p.qos_penalty = 1.5
        return p


# --------------------------------------------------------------------------
# a3: constructor-time constants (ENV:169-179)
# --------------------------------------------------------------------------
def distance_B_R() -> float:
    """ENV:175-176."""
    return math.sqrt((BS_XYZ[0] - RIS_XYZ[0]) ** 2 + (BS_XYZ[1] - RIS_XYZ[1]) ** 2
                     + (BS_XYZ[2] - RIS_XYZ[2]) ** 2)


def phase_R(M: int) -> np.ndarray:
    """RIS->BS steering vector, ENV:177-179: exp(+j 2 pi/lamb d angle_BR m)."""
    angle_BR = (RIS_XYZ[0] - BS_XYZ[0]) / distance_B_R()
    m = np.arange(M, dtype=np.float64)
    ph = 2 * (math.pi / LAMB) * D_ELEM * angle_BR * m
    return np.cos(ph) + 1j * np.sin(ph)


def possible_angles(control_bit: int) -> np.ndarray:
    """ENV:169."""
    return np.linspace(0, 2 * math.pi, 2 ** control_bit, endpoint=False)


# --------------------------------------------------------------------------
# a4: reset  (make_new_game ENV:733-737 + add_new_vehicles_by_number ENV:381-410)
# --------------------------------------------------------------------------
def reset(spawn_ints: np.ndarray, buf0: np.ndarray, lanes: Dict[str, Sequence[float]]
          ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """spawn_ints [E,V,3] = (aux, coord, velocity) per vehicle, buf0 [E] = the ONE
    randint(5, 9) draw shared by all vehicles of an env (ENV:737).

    Vehicles come in rounds of four, in list order d,u,l,r (ENV:384-400): aux of the
    'd' vehicle is the down-lane index drawn at ENV:384 (aux of u,l,r is unused).
    The V%4 extras (ENV:402-407) use aux = lane_index + 4*direction_choice, where
    direction_choice indexes 'dulr' (ENV:382, 404).
    Returns pos [E,V,2], direc [E,V], vel [E,V], DataBuf [E,V].
    """
    spawn_ints = np.asarray(spawn_ints)
    E, V, _ = spawn_ints.shape
    pos = np.zeros((E, V, 2))
    direc = np.zeros((E, V), dtype=np.int64)
    vel = spawn_ints[:, :, 2].astype(np.float64)
    down, up, left, right = (np.asarray(lanes[k], dtype=np.float64) for k in ("down", "up", "left", "right"))
    n_round = V // 4
    for v in range(V):
        aux = spawn_ints[:, v, 0]
        coord = spawn_ints[:, v, 1].astype(np.float64)
        if v < 4 * n_round:
            slot = v % 4
            if slot == 0:      # 'd' at (down_lanes[ind], randint(220,230))   ENV:386-388
                pos[:, v, 0] = down[aux]; pos[:, v, 1] = coord; direc[:, v] = DIR_D
            elif slot == 1:    # 'u' at (up_lanes[0], randint(170,180))       ENV:390-392
                pos[:, v, 0] = up[0]; pos[:, v, 1] = coord; direc[:, v] = DIR_U
            elif slot == 2:    # 'l' at (randint(220,230), left_lanes[0])     ENV:394-396
                pos[:, v, 0] = coord; pos[:, v, 1] = left[0]; direc[:, v] = DIR_L
            else:              # 'r' at (randint(170,180), right_lanes[0])    ENV:398-400
                pos[:, v, 0] = coord; pos[:, v, 1] = right[0]; direc[:, v] = DIR_R
        else:                  # extras, ENV:402-407
            lane = aux % 4
            choice = aux // 4                       # index into 'dulr'
            pos[:, v, 0] = down[lane]; pos[:, v, 1] = coord
            direc[:, v] = np.array([DIR_D, DIR_U, DIR_L, DIR_R])[choice]
    data_buf = (np.asarray(buf0) / 2.0)[:, None] * np.ones((1, V))      # ENV:737
    return pos, direc, vel, data_buf


# --------------------------------------------------------------------------
# a5: mobility  (renew_positions ENV:412-542)
# --------------------------------------------------------------------------
def mobility(pos: np.ndarray, direc: np.ndarray, vel: np.ndarray, u_turn: np.ndarray,
             lanes: Dict[str, Sequence[float]], width: float, height: float,
             time_slow: float = 0.1) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One `renew_positions()` call for every (env, vehicle).

    u_turn [E,V,8]: uniform draws, consumed left to right, ONE per detected lane
    crossing (ENV:427,438,452,464,479,489,502,512).  Returns (pos, direc, n_used).
    Plain per-vehicle loops: this is a branchy scalar routine in the reference.
    """
    pos = np.array(pos, dtype=np.float64, copy=True)
    direc = np.array(direc, dtype=np.int64, copy=True)
    E, V = direc.shape
    n_used = np.zeros((E, V), dtype=np.int64)
    up, down, left, right = (list(lanes[k]) for k in ("up", "down", "left", "right"))
    for e in range(E):
        for v in range(V):
            x, y = pos[e, v]
            d = int(direc[e, v])
            dd = vel[e, v] * time_slow                       # ENV:419
            turned = False
            nd = 0
            if d == DIR_U:                                   # ENV:421-446
                for lane in left:
                    if y <= lane and (y + dd) >= lane:
                        u = u_turn[e, v, nd]; nd += 1
                        if u < 0.4:
                            x, y, d, turned = x - (dd - (lane - y)), lane, DIR_L, True
                            break
                if not turned:
                    for lane in right:
                        if y <= lane and (y + dd) >= lane:
                            u = u_turn[e, v, nd]; nd += 1
                            if u < 0.4:                      # NB '+' overshoot, ENV:439-440
                                x, y, d, turned = x + (dd + (lane - y)), lane, DIR_R, True
                                break
                if not turned:
                    y += dd
            # NB: the reference tests `direction == X and change_direction == False`
            # with the *updated* direction; a turned vehicle has turned==True so it
            # skips every later block (ENV:447, 474, 497).
            if d == DIR_D and not turned:                    # ENV:447-473
                for lane in left:
                    if y >= lane and (y - dd) <= lane:
                        u = u_turn[e, v, nd]; nd += 1
                        if u < 0.4:
                            x, y, d, turned = x - (dd - (y - lane)), lane, DIR_L, True
                            break
                if not turned:
                    for lane in right:
                        if y >= lane and (y - dd) <= lane:
                            u = u_turn[e, v, nd]; nd += 1
                            if u < 0.4:                      # NB '+', ENV:465-466
                                x, y, d, turned = x + (dd + (y - lane)), lane, DIR_R, True
                                break
                if not turned:
                    y -= dd
            if d == DIR_R and not turned:                    # ENV:474-496
                for lane in up:
                    if x <= lane and (x + dd) >= lane:
                        u = u_turn[e, v, nd]; nd += 1
                        if u < 0.4:
                            x, y, d, turned = lane, y + (dd - (lane - x)), DIR_U, True
                            break
                if not turned:
                    for lane in down:
                        if x <= lane and (x + dd) >= lane:
                            u = u_turn[e, v, nd]; nd += 1
                            if u < 0.4:
                                x, y, d, turned = lane, y - (dd - (lane - x)), DIR_D, True
                                break
                if not turned:
                    x += dd
            if d == DIR_L and not turned:                    # ENV:497-519
                for lane in up:
                    if x >= lane and (x - dd) <= lane:
                        u = u_turn[e, v, nd]; nd += 1
                        if u < 0.4:
                            x, y, d, turned = lane, y + (dd - (x - lane)), DIR_U, True
                            break
                if not turned:
                    for lane in down:
                        if x >= lane and (x - dd) <= lane:
                            u = u_turn[e, v, nd]; nd += 1
                            if u < 0.4:
                                x, y, d, turned = lane, y - (dd - (x - lane)), DIR_D, True
                                break
                    # the reference's final advance sits INSIDE this `if` (ENV:518-519);
                    # equivalent, since turned==False here implies the up-loop did not turn
                    if not turned:
                        x -= dd
            # exit handling, ENV:522-540
            if x < 0 or y < 0 or x > width or y > height:
                if d == DIR_U:
                    d, y = DIR_R, right[-1]
                elif d == DIR_D:
                    d, y = DIR_L, left[0]
                elif d == DIR_L:
                    d, x = DIR_U, up[0]
                elif d == DIR_R:
                    d, x = DIR_D, down[-1]
            pos[e, v, 0], pos[e, v, 1] = x, y
            direc[e, v] = d
            n_used[e, v] = nd
    return pos, direc, n_used


# --------------------------------------------------------------------------
# a6: geometry  (compute_parms ENV:241-253)
# --------------------------------------------------------------------------
def geometry(pos: np.ndarray, M: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Returns dist_R [E,V], ang_R [E,V], h_r [E,V,M] (= phases_R_i)."""
    x = pos[..., 0]
    y = pos[..., 1]
    dist = np.sqrt((x - RIS_XYZ[0]) ** 2 + (y - RIS_XYZ[1]) ** 2 + (VEH_HEIGHT - RIS_XYZ[2]) ** 2)
    ang = (x - RIS_XYZ[0]) / dist
    m = np.arange(M, dtype=np.float64)
    ph = (-2 * (math.pi / LAMB) * D_ELEM * ang)[..., None] * m          # ENV:253
    h_r = np.cos(ph) + 1j * np.sin(ph)
    return dist, ang, h_r


def pathloss_factor(dist: np.ndarray) -> np.ndarray:
    """ro^2 / (d_Rv^alpha1 * d_BR^alpha2): the real factor in ENV:270-272."""
    return (RO / (np.sqrt(dist ** ALPHA1) * math.sqrt(distance_B_R() ** ALPHA2))) ** 2


# --------------------------------------------------------------------------
# a7: RIS cascaded gain, "free" model (update_channel_gains ENV:263-273)
# --------------------------------------------------------------------------
def gain_free(theta: np.ndarray, h_r: np.ndarray, b: np.ndarray, dist: np.ndarray,
              h_d: np.ndarray | None = None) -> np.ndarray:
    """gain[e,v] = | ro * sum_m theta[e,m] h_r[e,v,m] b[m] / (sqrt(d^a1) sqrt(dBR^a2)) |^2.
    h_d (optional, zero in the reference) is a direct-link amplitude added to the
    scaled cascade before the modulus (north_star's h_d + G^H diag(theta) h_r)."""
    img = np.einsum("em,evm,m->ev", theta, h_r, b)
    casc = (RO * img) / (np.sqrt(dist ** ALPHA1) * math.sqrt(distance_B_R() ** ALPHA2))
    if h_d is not None:
        casc = casc + h_d
    return np.abs(casc) ** 2


# --------------------------------------------------------------------------
# a8: 3GPP TR 38.901-style gains (update_channel_gains ENV:275-327)
# --------------------------------------------------------------------------
def gain_3gpp(pos: np.ndarray, mode: str, u_los: np.ndarray, z_shadow: np.ndarray,
              small: np.ndarray, p: OracleParams) -> np.ndarray:
    """mode in {'3gpp_umi','3gpp_uma', other->pl 0 dB (ENV:315-317)}.
    u_los ~ U[0,1) (ENV:299); z_shadow ~ N(0,1), scaled by the LOS/NLOS std here
    (ENV:10, 322); small = the small-scale POWER (ENV:13-25), i.e. Exp(1) for
    Rayleigh or the Rice power built by `rice_power` below."""
    fc = float(p.fc_GHz)
    dx = np.abs(pos[..., 0] - BS_XYZ[0])
    dy = np.abs(pos[..., 1] - BS_XYZ[1])
    dz = abs(BS_XYZ[2] - VEH_HEIGHT)
    d2d = np.hypot(dx, dy)
    d3d = np.sqrt(d2d * d2d + dz * dz)
    los = u_los < 0.7 * np.exp(-d2d / 200.0)                   # ENV:296-299
    dm = np.maximum(d3d, 1.0)
    if mode == "3gpp_umi":
        pl_los = 32.4 + 21.0 * np.log10(fc) + 20.0 * np.log10(dm)     # ENV:281
        pl_nlos = 36.7 + 22.7 * np.log10(fc) + 26.0 * np.log10(dm)    # ENV:285
    elif mode == "3gpp_uma":
        pl_los = 28.0 + 22.0 * np.log10(fc) + 20.0 * np.log10(dm)     # ENV:289
        pl_nlos = 13.54 + 39.08 * np.log10(dm) + 20.0 * np.log10(fc) - 0.6 * p.vehAntGain  # ENV:293
    else:
        pl_los = pl_nlos = np.zeros_like(dm)
    pl_db = np.where(los, pl_los, pl_nlos)
    large = 10 ** (-pl_db / 10.0)
    std = np.where(los, p.shadow_std_los, p.shadow_std_nlos)
    shadow = 10 ** ((z_shadow * std) / 10.0)
    return large * shadow * small


def rice_power(z_re: np.ndarray, z_im: np.ndarray, rician_K_dB: float) -> np.ndarray:
    """ENV:19-25 with standard-normal inputs z_re, z_im."""
    K = 10 ** (rician_K_dB / 10.0)
    s = np.sqrt(K / (K + 1.0))
    sig = 1.0 / np.sqrt(2.0 * (K + 1.0))
    hr = s + sig * z_re
    hi = sig * z_im
    return hr * hr + hi * hi


# --------------------------------------------------------------------------
# a9: BCD sweep (optimize_phase_shift ENV:208-220, objective ENV:222-231)
# --------------------------------------------------------------------------
def bcd_objective(theta: np.ndarray, h_r: np.ndarray, b: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """ENV:222-231 for a batch.  NB the reference sums the WHOLE [V,M] product
    (ENV:226, no vehicle index), so `img` is shared by all vehicles."""
    img = np.einsum("em,evm,m->e", theta, h_r, b)
    casc = (RO * img)[:, None] / (np.sqrt(dist ** ALPHA1) * math.sqrt(distance_B_R() ** ALPHA2))
    return np.sum((np.abs(casc) ** 2) / SIGMA ** 2, axis=1)


def bcd_sweep(theta: np.ndarray, h_r: np.ndarray, b: np.ndarray, dist: np.ndarray,
              control_bit: int) -> Tuple[np.ndarray, np.ndarray]:
    """One coordinate-ascent sweep m = 0..M-1 over the 2^b phases.

    Incremental form: with c[e,m] = (sum_v h_r[e,v,m]) b[m] the objective is
    Kc[e] * |sum_m theta_m c_m|^2 with Kc > 0, so the argmax over candidates is that
    of |S - theta_m c_m + cand c_m|^2.  Ties and the strict `best < x` test with
    best = 0 (ENV:210-218) are kept: first candidate wins; if no candidate scores
    above 0 the element is set to the integer 0 (ENV:211, 220).
    Returns (theta_new [E,M], idx [E,M] chosen candidate or -1)."""
    theta = np.array(theta, dtype=np.complex128, copy=True)
    E, M = theta.shape
    ang = possible_angles(control_bit)
    cand = np.cos(ang) + 1j * np.sin(ang)
    c = h_r.sum(axis=1) * b[None, :]
    kc = np.sum(pathloss_factor(dist), axis=1) / SIGMA ** 2
    S = np.sum(theta * c, axis=1)
    idx = np.full((E, M), -1, dtype=np.int64)
    for m in range(M):
        rest = S - theta[:, m] * c[:, m]
        best = np.zeros(E)
        best_phase = np.zeros(E, dtype=np.complex128)
        for k in range(cand.shape[0]):
            x = kc * np.abs(rest + cand[k] * c[:, m]) ** 2
            better = best < x
            best = np.where(better, x, best)
            best_phase = np.where(better, cand[k], best_phase)
            idx[:, m] = np.where(better, k, idx[:, m])
        theta[:, m] = best_phase
        S = rest + best_phase * c[:, m]
    return theta, idx


def bcd_sweep_literal(theta, h_r, b, dist, control_bit):
    """Structure-faithful O(M^2 2^b V) form of ENV:208-231 for ONE env (small cases
    only): re-evaluates the full objective for every candidate, like the reference."""
    theta = np.array(theta, dtype=np.complex128, copy=True)
    M = theta.shape[0]
    ang = possible_angles(control_bit)
    dBR = math.sqrt(distance_B_R() ** ALPHA2)
    for m in range(M):
        best = 0
        best_phase = 0
        for phase in ang:
            theta[m] = complex(math.cos(phase), math.sin(phase))
            img = np.sum(theta[None, :] * h_r * b[None, :])
            x = 0
            for v in range(h_r.shape[0]):
                casc = (RO * img) / (math.sqrt(dist[v] ** ALPHA1) * dBR)
                x += (np.abs(casc) ** 2) / SIGMA ** 2
            if best < x:
                best = x
                best_phase = complex(math.cos(phase), math.sin(phase))
        theta[m] = best_phase
    return theta


# --------------------------------------------------------------------------
# batched NOMA-group encoding (build-defined boundary)
# --------------------------------------------------------------------------
def encode_groups(noma_groups: Sequence[Sequence[int]], V: int) -> Tuple[np.ndarray, int]:
    """list-of-lists (ENV:330, 339-369) -> (partner [V], n_groups)."""
    partner = np.full(V, PARTNER_NONE, dtype=np.int64)
    for g in noma_groups:
        if len(g) == 1:
            partner[g[0]] = PARTNER_SINGLE
        elif len(g) == 2:
            partner[g[0]] = g[1]
            partner[g[1]] = g[0] + PARTNER_SECOND
    return partner, len(noma_groups)


# --------------------------------------------------------------------------
# a11: data rate (compute_data_rate ENV:331-372)
# --------------------------------------------------------------------------
def data_rate(p_off: np.ndarray, gain: np.ndarray, partner: np.ndarray,
              n_groups: np.ndarray, noise_power: float) -> np.ndarray:
    """p_off [E,V] offload power in W, returns rate [E,V] in bit/s/Hz."""
    E, V = gain.shape
    frac = 1.0 / np.maximum(1, np.asarray(n_groups))[:, None]          # ENV:341-342
    partner = np.asarray(partner)
    is_single = partner == PARTNER_SINGLE
    is_pair = partner >= 0
    second = is_pair & (partner >= PARTNER_SECOND)
    pidx = np.where(is_pair, partner % PARTNER_SECOND, 0)
    g_p = np.take_along_axis(gain, pidx, axis=1)
    p_p = np.take_along_axis(p_off, pidx, axis=1)
    # near = u1 if gain1 > gain2 else u2 (ENV:355-360)
    near = np.where(second, ~(g_p > gain), gain > g_p)
    sinr_single = (p_off * gain) / noise_power                          # ENV:347-348, 367-368
    sinr_far = (p_off * gain) / (p_p * gain + noise_power)              # ENV:362-364
    sinr = np.where(is_pair & ~near, sinr_far, sinr_single)
    rate = frac * np.log2(1 + sinr)
    return np.where(is_single | is_pair, rate, 0.0)


# --------------------------------------------------------------------------
# a12: step (ENV:547-731)
# --------------------------------------------------------------------------
METRIC_NAMES = (
    "global_reward", "last_off_kbit_sum", "last_local_kbit_sum", "last_mec_queue_cycles",
    "last_backlog_kbit_mean", "last_delay_local_mean", "last_delay_edge_q_mean",
    "last_delay_edge_c_mean", "last_t_tx_mean", "last_mec_utilization",
    "last_local_util_mean", "last_qos_violation", "last_delay_mean", "last_energy_mean",
)


def effective_floor(floor: float) -> float:
    """ENV:574-577."""
    f = float(floor)
    if not np.isfinite(f):
        f = 0.10
    return max(0.0, min(f, 0.95))


def step(data_buf: np.ndarray, mec_q: np.ndarray, gain: np.ndarray, action: np.ndarray,
         partner: np.ndarray, n_groups: np.ndarray, arrivals: np.ndarray,
         p: OracleParams) -> Dict[str, np.ndarray]:
    """One `Environ.step` for E envs.  data_buf [E,V] kbit, mec_q [E] cycles,
    action [E,2,V] (row 0 offload power, row 1 local share), arrivals [E,V] ints
    (the Poisson draws of ENV:718).  Returns a dict with the 7-tuple members,
    vehicle_rate, obs (marl_train_bcd.py:819-827), post-state and the metrics."""
    a = np.asarray(action, dtype=np.float64)
    B = np.asarray(data_buf, dtype=np.float64)
    Q0 = np.asarray(mec_q, dtype=np.float64)
    E, V = B.shape
    eps = 1e-12
    # (1) power projection, ENV:555-561
    proj = np.clip(a, 0.0, None) * p.power_scale
    s = proj.sum(axis=1, keepdims=True)
    proj = np.where(s > 1.0, proj / (s + 1e-12), proj)
    power_W = proj * p.P_max
    # (2) rate, ENV:565-570
    rate = data_rate(power_W[:, 0, :], gain, partner, n_groups, p.noise_power)
    data_t = rate * p.time_fast * p.bandwidth * 1000.0
    # (3) cpu share, ENV:572-580
    cpu = np.maximum(np.clip(a[:, 1, :], 0.0, 1.0), effective_floor(p.cpu_share_floor))
    f_local = cpu * p.f_local_max
    Cpb = float(p.cycles_per_bit)
    # (4) local processing, ENV:585-592
    bc = B * 1000.0 * Cpb
    cap = f_local * p.time_fast
    used = np.minimum(cap, bc)
    data_p = used / (Cpb * 1000.0)
    # (5) offload, ENV:595-601
    remaining = np.maximum(0.0, B - data_p)
    off = np.minimum(data_t, remaining)
    thr = rate * p.bandwidth * 1000.0
    t_tx = off / (thr + 1e-12)
    # (6) MEC queue, ENV:604-610
    ein = off * 1000.0 * Cpb
    ein_sum = ein.sum(axis=1)
    Q = Q0 + ein_sum
    svc = np.minimum(p.f_edge_max * p.time_fast, Q)
    Q = Q - svc
    # (7) backlog, ENV:617-618
    B_new = np.maximum(0.0, B - (data_p + off))
    # (8) delays, ENV:622-633
    d_loc = np.maximum(0.0, bc - ein) / (f_local + eps)
    share = ein / (ein_sum[:, None] + eps)
    d_q = share * (Q0 / (p.f_edge_max + eps))[:, None]
    d_c = ein / (p.f_edge_max + eps)
    delay = d_loc + t_tx + d_q + d_c
    # (9) energy, ENV:659-666
    E_tx = power_W[:, 0, :] * t_tx
    E_loc = p.k * (f_local ** 2) * used
    energy = E_tx + E_loc
    last_power_W = np.stack([E_tx / p.time_fast, E_loc / p.time_fast], axis=1)
    # (10) QoS, ENV:669-677
    if p.qos_enable:
        viol = (rate < float(p.R_min_bpsHz)) | (delay > float(p.D_max_s))
        pen = float(p.qos_penalty) * viol.astype(np.float64)
    else:
        viol = np.zeros_like(rate, dtype=bool)
        pen = np.zeros_like(rate)
    # (11) reward, ENV:696-703
    cost = p.w_d * delay + p.w_e * energy
    reward = np.clip(-cost - pen, -p.reward_clip, p.reward_clip)
    # (12) arrivals, ENV:717-719
    arr = np.asarray(arrivals, dtype=np.float64)
    B_new = B_new + arr * p.time_fast * 1000
    # (13) ENV:721-729
    g_reward = reward.mean(axis=1)
    over_power = np.maximum(0.0, (power_W[:, 0, :] + power_W[:, 1, :]) - p.P_max)
    over_data = np.zeros_like(B)
    metrics = np.stack([
        g_reward,                                      # 0  ENV:721
        off.sum(axis=1),                               # 1  ENV:612
        data_p.sum(axis=1),                            # 2  ENV:613
        Q,                                             # 3  ENV:614
        B.mean(axis=1),                                # 4  ENV:649 (pre-action mean)
        d_loc.mean(axis=1),                            # 5  ENV:643
        d_q.mean(axis=1),                              # 6  ENV:644
        d_c.mean(axis=1),                              # 7  ENV:645
        t_tx.mean(axis=1),                             # 8  ENV:646
        svc / (p.f_edge_max * p.time_fast + 1e-12),    # 9  ENV:652-653
        (used / (cap + 1e-12)).mean(axis=1),           # 10 ENV:656
        viol.astype(np.float64).mean(axis=1),          # 11 ENV:677
        delay.mean(axis=1),                            # 12 ENV:710
        energy.mean(axis=1),                           # 13 ENV:711
    ], axis=1)
    obs = np.stack([B_new / 10, data_t / 10, data_p / 10, over_data / 10, rate / 20], axis=2)
    return dict(reward=reward, global_reward=g_reward, data_buf=B_new, data_t=data_t,
                data_p=data_p, over_power=over_power, over_data=over_data,
                vehicle_rate=rate, mec_q=Q, metrics=metrics, last_power_W=last_power_W,
                obs=obs, delay=delay, viol=viol, power_W=power_W, ein_sum=ein_sum,
                # distances to the discontinuities, for near-threshold tagging in tests
                margin=dict(rate=rate - p.R_min_bpsHz, delay=delay - p.D_max_s,
                            s=s[:, 0, :] - 1.0, raw_reward=-cost - pen))


def action_from_policy(policy_out: np.ndarray, floor: float) -> np.ndarray:
    """a14, marl_train_bcd.py:1601-1608: policy [E,V,2] in [-1,1] -> env [E,2,V]."""
    c = np.clip(policy_out, -0.999, 0.999)
    a = (c + 1) / 2
    a = np.transpose(a, (0, 2, 1)).copy()
    a[:, 1, :] = np.maximum(a[:, 1, :], effective_floor(floor))
    return a


# --------------------------------------------------------------------------
# Philox4x32-10 + the samplers the kernels use (bit-exact restatement of
# ris_vec_marl_amd/csrc/rng_philox.hpp; integer outputs must match exactly)
# --------------------------------------------------------------------------
_PH_M0 = np.uint64(0xD2511F53)
_PH_M1 = np.uint64(0xCD9E8D57)
_PH_W0 = 0x9E3779B9
_PH_W1 = 0xBB67AE85

SITE_ARRIVALS, SITE_TURN_A, SITE_TURN_B, SITE_SPAWN, SITE_BUF0, SITE_3GPP, SITE_PHASE = range(7)


def philox4x32(c0, c1, c2, c3, seed: int) -> Tuple[np.ndarray, ...]:
    """Counter (c0..c3) uint32 arrays (broadcastable), key = 64-bit seed."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
                                           for c in (c0, c1, c2, c3)))
    k0 = seed & 0xFFFFFFFF
    k1 = (seed >> 32) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PH_M0 * c0
        p1 = _PH_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _PH_W0) & 0xFFFFFFFF
        k1 = (k1 + _PH_W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01(x: np.ndarray) -> np.ndarray:
    """uint32 -> float32 in [0,1): top 24 bits, exact."""
    return ((x >> np.uint32(8)).astype(np.float32)) * np.float32(2.0 ** -24)


POISSON_TABLE = 64


def poisson_cdf_table(lam: float) -> np.ndarray:
    """float32 CDF table of Poisson(lam), computed in float64 on the host and shipped
    in the kernel params, so the device-side inversion is pure compares."""
    if lam <= 0:
        return np.ones(POISSON_TABLE, dtype=np.float32)
    k = np.arange(POISSON_TABLE)
    logp = -lam + k * math.log(lam) - np.array([math.lgamma(i + 1) for i in k])
    cdf = np.cumsum(np.exp(logp))
    return np.minimum(cdf, 1.0).astype(np.float32)


def poisson_from_u(u: np.ndarray, cdf: np.ndarray) -> np.ndarray:
    """count of table entries <= u, i.e. inversion by sequential search."""
    return (u[..., None] >= cdf).sum(axis=-1).astype(np.int32)


def philox_arrivals(env_ids: np.ndarray, V: int, step: int, seed: int, lam: float) -> np.ndarray:
    """arrivals [E,V] exactly as the step kernels draw them: counter =
    (global env id, vehicle, step, SITE_ARRIVALS), lane .x of the Philox block."""
    e = np.asarray(env_ids, dtype=np.uint64)[:, None]
    v = np.arange(V, dtype=np.uint64)[None, :]
    r0, _, _, _ = philox4x32(e, v, np.uint64(step & 0xFFFFFFFF), np.uint64(SITE_ARRIVALS), seed)
    return poisson_from_u(u01(r0), poisson_cdf_table(lam))


def randint_from_u32(x: np.ndarray, low: int, high: int) -> np.ndarray:
    """low + floor(x * (high-low) / 2^32)  (multiply-shift; no modulo bias loop)."""
    span = np.uint64(high - low)
    return (low + ((x.astype(np.uint64) * span) >> np.uint64(32))).astype(np.int64)


def philox_reset(env_ids: np.ndarray, V: int, counter: int, seed: int, height: int = 400,
                 n_lanes: int = 4) -> Tuple[np.ndarray, np.ndarray]:
    """(spawn_ints [E,V,3], buf0 [E]) exactly as k_reset draws them."""
    e = np.asarray(env_ids, dtype=np.uint64)[:, None]
    v = np.arange(V, dtype=np.uint64)[None, :]
    r0, r1, r2, r3 = philox4x32(e, v, np.uint64(counter), np.uint64(SITE_SPAWN), seed)
    E = e.shape[0]
    spawn = np.zeros((E, V, 3), dtype=np.int64)
    lane = randint_from_u32(r0, 0, n_lanes)
    n4 = 4 * (V // 4)
    for vv in range(V):
        if vv < n4:
            slot = vv % 4
            spawn[:, vv, 0] = lane[:, vv]
            lo, hi = (220, 230) if slot in (0, 2) else (170, 180)
            spawn[:, vv, 1] = randint_from_u32(r1[:, vv], lo, hi)
            spawn[:, vv, 2] = randint_from_u32(r2[:, vv], 10, 15)
        else:
            spawn[:, vv, 0] = lane[:, vv] + 4 * randint_from_u32(r3[:, vv], 0, 4)
            spawn[:, vv, 1] = randint_from_u32(r1[:, vv], 0, height)
            spawn[:, vv, 2] = randint_from_u32(r2[:, vv], 15, 20)
    b0, _, _, _ = philox4x32(e[:, 0], np.uint64(0), np.uint64(counter), np.uint64(SITE_BUF0), seed)
    return spawn, randint_from_u32(b0, 5, 9)


def philox_turn_draws(env_ids: np.ndarray, V: int, counter: int, seed: int) -> np.ndarray:
    """u_turn [E,V,8] float32 exactly as k_mobility would consume them (block A then B)."""
    e = np.asarray(env_ids, dtype=np.uint64)[:, None]
    v = np.arange(V, dtype=np.uint64)[None, :]
    a = philox4x32(e, v, np.uint64(counter), np.uint64(SITE_TURN_A), seed)
    b = philox4x32(e, v, np.uint64(counter), np.uint64(SITE_TURN_B), seed)
    return np.stack([u01(x) for x in a + b], axis=-1)


def philox_phase_idx(env_ids: np.ndarray, M: int, counter: int, seed: int, control_bit: int) -> np.ndarray:
    """idx [E,M] exactly as k_random_phase draws them."""
    e = np.asarray(env_ids, dtype=np.uint64)[:, None]
    m = np.arange(M, dtype=np.uint64)[None, :]
    r0, _, _, _ = philox4x32(e, m, np.uint64(counter), np.uint64(SITE_PHASE), seed)
    return randint_from_u32(r0, 0, 2 ** control_bit)


def bcd_margin(theta, h_r, b, control_bit):
    """For each (env, m) decision of a sweep: relative gap between the best and the
    second-best candidate score (tests skip decisions with a tiny gap)."""
    theta = np.array(theta, dtype=np.complex128, copy=True)
    E, M = theta.shape
    ang = possible_angles(control_bit)
    cand = np.cos(ang) + 1j * np.sin(ang)
    c = h_r.sum(axis=1) * b[None, :]
    S = np.sum(theta * c, axis=1)
    gap = np.zeros((E, M))
    for m in range(M):
        rest = S - theta[:, m] * c[:, m]
        x = np.abs(rest[:, None] + cand[None, :] * c[:, m][:, None]) ** 2
        xs = np.sort(x, axis=1)
        gap[:, m] = (xs[:, -1] - xs[:, -2]) / np.maximum(xs[:, -1], 1e-300) if x.shape[1] > 1 else 1.0
        k = np.argmax(x, axis=1)
        theta[:, m] = np.where(xs[:, -1] > 0, cand[k], 0)
        S = rest + theta[:, m] * c[:, m]
    return gap


# --------------------------------------------------------------------------
# f1: the single-agent (SARL) environment variant -- Simulation-SARL/Environment.py (SENV)
# --------------------------------------------------------------------------
@dataclasses.dataclass
class SarlParams:
    """SENV:66-83 class defaults."""
    time_fast: float = 0.001       # SENV:67
    bandwidth: float = 1.0         # SENV:69
    k: float = 1e-28               # SENV:70
    L: float = 500.0               # SENV:71
    rate: float = 3.0              # SENV:78
    t_factor1: float = 1.0         # SENV:80
    t_factor2: float = 0.6         # SENV:81
    penalty1: float = 2.0          # SENV:82
    penalty2: float = 2.0          # SENV:83


def sarl_step(data_buf: np.ndarray, gain: np.ndarray, action_power: np.ndarray, arrivals: np.ndarray,
              p: SarlParams) -> Dict[str, np.ndarray]:
    """SENV:321-359 for E envs, given the cascaded gain |ro img / (sqrt(d^a1) sqrt(dBR^a2))|^2 of
    SENV:149-157 (identical to the MARL 'free' gain; the phases come from the agent through
    get_next_phase, SENV:133-139).  action_power [E,2,V]: row 0 offload power, row 1 local
    power, used as given (no projection).  Returns the 6-tuple members + rate."""
    a = np.asarray(action_power, dtype=np.float64)
    B = np.asarray(data_buf, dtype=np.float64)
    p0, p1 = a[:, 0, :], a[:, 1, :]
    rate = np.log(1 + p0 * gain / SIGMA ** 2)                              # SENV:159 (natural log)
    data_t = rate * p.time_fast * p.bandwidth * 1000                      # SENV:329
    data_p = np.power(p1 / p.k, 1.0 / 3.0) * p.time_fast / p.L / 1000     # SENV:330
    Bn = B - (data_t + data_p)                                            # SENV:333
    neg = Bn < 0
    need = np.fmax(0, Bn + data_p)                                        # SENV:336
    proc_rev = np.power(need * 1000 * p.L / p.time_fast, 3.0) * p.k       # SENV:318-319
    over_power = np.where(neg, p1 - proc_rev, 0.0)                        # SENV:336
    over_data = np.where(neg, -Bn, 0.0)                                   # SENV:337, 340
    Bn = np.where(neg, 0.0, Bn)                                           # SENV:338
    base = -(p.t_factor1 * (p0 + p1)) - p.t_factor2 * Bn                  # SENV:344-352
    reward = np.where(Bn > 0, base - p.penalty1, np.where(over_data > 2, base - p.penalty2, base))
    arr = np.asarray(arrivals, dtype=np.float64)
    B_out = Bn + arr * p.time_fast * 1000                                 # SENV:354-356
    return dict(reward_mean=reward.mean(axis=1), reward=reward, data_buf=B_out, data_t=data_t, data_p=data_p,
                over_power=over_power, over_data=over_data, vehicle_rate=rate,
                margin=dict(buf=B - (data_t + data_p), over=over_data - 2.0))


def sarl_obs(theta_real: np.ndarray, data_buf, data_t, data_p, over_data, rate) -> np.ndarray:
    """ddpg_train.py:47-73: per agent [theta slice (M//V), DataBuf/10, data_t/10, data_p/10,
    over_data/10, rate/20] -> [E, V, M//V + 5]."""
    E, V = data_buf.shape
    tn = theta_real.shape[1] // V
    th = theta_real[:, :tn * V].reshape(E, V, tn)
    tail = np.stack([data_buf / 10, data_t / 10, data_p / 10, over_data / 10, rate / 20], axis=2)
    return np.concatenate([th, tail], axis=2)


def sarl_action_map(action: np.ndarray, V: int, M: int) -> Tuple[np.ndarray, np.ndarray]:
    """ddpg_train.py:149-158: agent output [E, 2V+M] in [-1,1] -> (action_power [E,2,V],
    action_phase [E,M] in [0, 2 pi))."""
    a = np.clip(action, -0.999, 0.999)
    power = np.stack([(a[:, :V] + 1) / 2, (a[:, V:2 * V] + 1) / 2], axis=1)
    phase = ((a[:, 2 * V:2 * V + M] + 1) / 2) * math.pi * 2
    return power, phase
