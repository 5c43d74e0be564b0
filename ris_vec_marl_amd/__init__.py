"""MI355X-native vectorised RIS-VEC environment (hot path of 20242204033/RIS-VEC-MARL's
`Simulation-MARL-BCD/Environment.py`), behind the reference's `Environ` class surface.

    from ris_vec_marl_amd import Environ, VecEnviron, apply_yaml_config

All computation runs in hand-written HIP kernels (csrc/, C ABI in include/risvec.h);
importing the package does not need a GPU, computing does.
"""
from .params import EnvParams, apply_yaml_config, load_yaml, reference_lanes, poisson_cdf_table
from .vec_env import VecEnviron
from .compat import Environ, Vehicle, encode_noma_groups
from .sarl import SarlEnviron, SarlParams, sarl_action_map, sarl_observe
from .noma import NomaConfig, NomaGrouper, anneal_topk
from .replay import VecReplayBuffer, marshal_actions
from .policy import BatchedPolicy
from .metrics import EpisodeMeter, ScalarSink
from . import dist

__all__ = ["EnvParams", "apply_yaml_config", "load_yaml", "reference_lanes", "poisson_cdf_table",
           "VecEnviron", "Environ", "Vehicle", "encode_noma_groups", "SarlEnviron", "SarlParams", "sarl_action_map",
           "sarl_observe", "NomaConfig", "NomaGrouper", "anneal_topk", "VecReplayBuffer",
           "marshal_actions", "BatchedPolicy", "EpisodeMeter", "ScalarSink", "dist"]
