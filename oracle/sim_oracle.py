"""CPU oracle for the batched simulator step (TEST INFRASTRUCTURE ONLY; see oracle/rgl_oracle.py for the rules).

A sequential, python-float restatement of one environment's time step with `linear` humans, following
crowd_sim/envs/crowd_sim.py:252-368, crowd_sim/envs/utils/agent.py:113-139 and crowd_sim/envs/policy/linear.py:16-22.
Pinned against trajectories recorded from the reference simulator itself (tests/golden/sim.npz)."""
import numpy as np

from oracle.rgl_oracle import point_to_segment_dist

INFO_NOTHING, INFO_DISCOMFORT, INFO_COLLISION, INFO_REACH_GOAL, INFO_TIMEOUT = 0, 1, 2, 3, 4


def linear_action(h_full):
    """h_full = (px, py, vx, vy, radius, gx, gy, v_pref, theta) of one human."""
    theta = np.arctan2(h_full[6] - h_full[1], h_full[5] - h_full[0])
    return np.cos(theta) * h_full[7], np.sin(theta) * h_full[7]


def step(robot, humans, action, global_time, time_step=0.25, time_limit=30, success_reward=1, collision_penalty=-0.25,
         discomfort_dist=0.2, discomfort_penalty_factor=0.5):
    """robot: list of 9 floats, humans: list of 9-float lists (full states), action: (vx, vy).  Holonomic robot.
    Returns (robot', humans', reward, done, info, dmin)."""
    human_actions = [linear_action(h) for h in humans]
    dmin, collision = float("inf"), False
    for h in humans:
        px, py = h[0] - robot[0], h[1] - robot[1]
        vx, vy = h[2] - action[0], h[3] - action[1]
        ex, ey = px + vx * time_step, py + vy * time_step
        d = point_to_segment_dist(px, py, ex, ey, 0, 0) - h[4] - robot[4]
        if d < 0:
            collision = True
            break
        if d < dmin:
            dmin = d
    end = np.array((robot[0] + action[0] * time_step, robot[1] + action[1] * time_step))
    reaching = np.linalg.norm(end - np.array((robot[5], robot[6]))) < robot[4]
    if global_time >= time_limit - 1:
        reward, done, info = 0, True, INFO_TIMEOUT
    elif collision:
        reward, done, info = collision_penalty, True, INFO_COLLISION
    elif reaching:
        reward, done, info = success_reward, True, INFO_REACH_GOAL
    elif dmin < discomfort_dist:
        reward, done, info = (dmin - discomfort_dist) * discomfort_penalty_factor * time_step, False, INFO_DISCOMFORT
    else:
        reward, done, info = 0, False, INFO_NOTHING
    nr = list(robot)
    nr[0], nr[1], nr[2], nr[3] = end[0], end[1], action[0], action[1]
    nh = []
    for h, a in zip(humans, human_actions):
        g = list(h)
        g[0], g[1], g[2], g[3] = h[0] + a[0] * time_step, h[1] + a[1] * time_step, a[0], a[1]
        nh.append(g)
    return nr, nh, reward, done, info, dmin
