"""Attribute-bag configs with the reference's field names (crowd_nav/configs/icra_benchmark/config.py:
BasePolicyConfig + mp_separate.py / rgl.py), for standalone use without the upstream config modules.
The policies also accept the upstream PolicyConfig objects unchanged -- they only read attributes."""
import numpy as np


class Bag(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def policy_config(name="model_predictive_rl", **over):
    """`over` uses section__key=value, e.g. gcn__num_layer=3, model_predictive_rl__planning_depth=2."""
    c = Bag(name=name)
    c.rl = Bag(gamma=0.9)
    c.om = Bag(cell_num=4, cell_size=1, om_channel_size=3)
    c.action_space = Bag(kinematics='holonomic', speed_samples=5, rotation_samples=16, sampling='exponential',
                         query_env=False, rotation_constraint=np.pi / 3)
    c.gcn = Bag(multiagent_training=True, num_layer=2, X_dim=32, wr_dims=[64, 32], wh_dims=[64, 32],
                final_state_dim=32, gcn2_w1_dim=32, planning_dims=[150, 100, 100, 1],
                similarity_function='embedded_gaussian', layerwise_graph=False, skip_connection=True)
    c.model_predictive_rl = Bag(linear_state_predictor=False, planning_depth=1, planning_width=1,
                                do_action_clip=False, sparse_search=False, motion_predictor_dims=[64, 5],
                                value_network_dims=[32, 100, 100, 1], share_graph_model=False)
    for k, v in over.items():
        sect, key = k.split("__", 1)
        setattr(getattr(c, sect), key, v)
    return c
