import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from ris_vec_marl_amd import Environ, reference_lanes
L = reference_lanes()
env = Environ(L["down_lanes"], L["up_lanes"], L["left_lanes"], L["right_lanes"], 400, 400, 8, 40, 3)
env.make_new_game(); env.renew_positions(); env.compute_parms(); env.optimize_phase_shift(); env.update_channel_gains()
rng = np.random.default_rng(0)
groups = [[0, 1], [2, 3], [4], [5], [6], [7]]
a = rng.uniform(0, 1, (2, 8))
for _ in range(50): env.step(a, groups)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 500
for _ in range(n):
    r = env.step(a, groups)
    s = (env.DataBuf, env.data_t, env.data_p, env.over_data, env.vehicle_rate)     # marl_get_state reads these
t1 = time.perf_counter()
print("facade step + state read: %.1f us" % ((t1 - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(100):
    env.renew_positions(); env.compute_parms(); env.optimize_phase_shift(); env.update_channel_gains(); g = env.get_channel_gains()
print("refresh (positions+parms+bcd+gains+read): %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
