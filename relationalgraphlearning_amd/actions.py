"""Action types and the two discrete action tables of the hot path.

Mirrors: crowd_sim/envs/utils/action.py:1-4 (types), ModelPredictiveRL.build_action_space
(crowd_nav/policy/model_predictive_rl.py:155-190, speed-major + sparse-search group ids) and
CADRL.build_action_space (crowd_nav/policy/cadrl.py:91-111, rotation-major).
"""
from collections import namedtuple

import numpy as np

try:                                    # inside the reference's environment use ITS types, so the
    from crowd_sim.envs.utils.action import ActionXY, ActionRot   # simulator's isinstance checks pass
except Exception:                       # standalone: structurally identical namedtuples
    ActionXY = namedtuple("ActionXY", ["vx", "vy"])
    ActionRot = namedtuple("ActionRot", ["v", "r"])


def _samples(v_pref, speed_samples, rotation_samples, kinematics, rotation_constraint):
    speeds = (np.exp((np.arange(speed_samples) + 1) / speed_samples) - 1) / (np.e - 1) * v_pref
    if kinematics == "holonomic":
        rotations = np.linspace(0, 2 * np.pi, rotation_samples, endpoint=False)
    else:
        rotations = np.linspace(-rotation_constraint, rotation_constraint, rotation_samples)
    return speeds, rotations


def _make(kinematics, speed, rotation):
    if kinematics == "holonomic":
        return ActionXY(speed * np.cos(rotation), speed * np.sin(rotation))
    return ActionRot(speed, rotation)


def stop_action(kinematics):
    return ActionXY(0, 0) if kinematics == "holonomic" else ActionRot(0, 0)


def speed_major_table(v_pref, speed_samples, rotation_samples, kinematics, rotation_constraint,
                      sparse_rotation_samples=8):
    """Path M: [stop] + for speed: for rotation.  Group id = (speed band: first 3 speeds | rest) * 8
    + rotation index // 2; the stop action shares group 0."""
    speeds, rotations = _samples(v_pref, speed_samples, rotation_samples, kinematics, rotation_constraint)
    actions, groups = [stop_action(kinematics)], [0]
    for si, s in enumerate(speeds):
        band = 0 if si < 3 else 1
        for ri, r in enumerate(rotations):
            actions.append(_make(kinematics, s, r))
            groups.append(band * sparse_rotation_samples + ri // 2)
    return actions, groups, list(speeds), rotations


def rotation_major_table(v_pref, speed_samples, rotation_samples, kinematics, rotation_constraint):
    """Path G: [stop] + for rotation: for speed."""
    speeds, rotations = _samples(v_pref, speed_samples, rotation_samples, kinematics, rotation_constraint)
    actions = [stop_action(kinematics)]
    for r in rotations:
        for s in speeds:
            actions.append(_make(kinematics, s, r))
    return actions, list(speeds), rotations


def as_array(actions):
    """(A,2) float64: (vx,vy) or (v,r) -- the layout the device action table uses."""
    return np.asarray([[float(a[0]), float(a[1])] for a in actions], dtype=np.float64)
