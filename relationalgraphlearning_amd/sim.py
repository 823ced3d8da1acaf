"""Batched crowd simulator: B independent environments on the MI355X (float64 state, one kernel per time step).

Mirrors the parts of crowd_sim/envs/crowd_sim.py the rollout needs: seeded scene generation (`reset`, :171-247 with
`generate_human`, :117-169 -- host numpy, reproducing the reference's RNG stream so test case k is the same scene),
`step` / `onestep_lookahead` (:248-368, device: crowd_step_f64) and an Explorer-style episode loop over many
environments at once (crowd_nav/utils/explorer.py:21-111).  Humans follow the reference's `linear` policy
(crowd_sim/envs/policy/linear.py), constant velocity, or externally supplied actions; ORCA (external rvo2) is out of scope.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from .nets import _stream

INFO = {0: "", 1: "Discomfort", 2: "Collision", 3: "Reaching goal", 4: "Timeout", 5: "(finished earlier)"}
HUMAN_POLICY = {"given": 0, "linear": 1, "constant_velocity": 2}
BASE_SEED = {"train": 2000, "val": 0, "test": 1000}       # crowd_sim.py:185-186 with its case capacities


class SimConfig(object):
    """Attribute bag with the reference's EnvConfig defaults (crowd_nav/configs/icra_benchmark/config.py:14-52)."""

    def __init__(self, **over):
        self.time_limit, self.time_step, self.randomize_attributes = 30, 0.25, False
        self.success_reward, self.collision_penalty = 1, -0.25
        self.discomfort_dist, self.discomfort_penalty_factor = 0.2, 0.5
        self.scenario, self.square_width, self.circle_radius, self.human_num = "circle_crossing", 20, 4, 5
        self.human_radius, self.human_v_pref = 0.3, 1
        self.robot_radius, self.robot_v_pref = 0.3, 1
        for k, v in over.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    @staticmethod
    def from_env_config(c):
        """From the reference's EnvConfig object (same attribute names as upstream)."""
        return SimConfig(time_limit=c.env.time_limit, time_step=c.env.time_step,
                         randomize_attributes=c.env.randomize_attributes, success_reward=c.reward.success_reward,
                         collision_penalty=c.reward.collision_penalty, discomfort_dist=c.reward.discomfort_dist,
                         discomfort_penalty_factor=c.reward.discomfort_penalty_factor, scenario=c.sim.test_scenario,
                         square_width=c.sim.square_width, circle_radius=c.sim.circle_radius, human_num=c.sim.human_num,
                         human_radius=c.humans.radius, human_v_pref=c.humans.v_pref, robot_radius=c.robot.radius,
                         robot_v_pref=c.robot.v_pref)


def generate_scene(cfg, phase, case):
    """Initial state of seeded case `case` of `phase`: (robot (9,), humans (H,5), human goals (H,2), human v_pref (H,)),
    float64.  Consumes the legacy numpy stream exactly like CrowdSim.reset/generate_human."""
    rs = np.random.RandomState(BASE_SEED[phase] + case)
    R = cfg.circle_radius
    robot = np.array([0.0, -R, 0.0, 0.0, cfg.robot_radius, 0.0, R, cfg.robot_v_pref, np.pi / 2])
    # agents placed so far as columns (px, py, gx, gy, radius): the rejection tests below run as one vector expression per
    # attempt -- sqrt(dx*dx + dy*dy) element by element, the arithmetic of np.linalg.norm on a pair -- instead of two
    # np.linalg.norm calls per placed agent (20 ms per 19-human scene that way, almost all of it call overhead)
    ag = np.empty((cfg.human_num + 1, 5))
    ag[0] = (robot[0], robot[1], robot[5], robot[6], cfg.robot_radius)
    n = 1
    humans, goals, vprefs = [], [], []

    def clear_of(x, y, cx, cy, radius):        # every distance from (x, y) to the points (cx, cy) at least the margin
        dx, dy = x - cx, y - cy
        return not bool((np.sqrt(dx * dx + dy * dy) < radius + ag[:n, 4] + cfg.discomfort_dist).any())
    for _ in range(cfg.human_num):
        v_pref, radius = cfg.human_v_pref, cfg.human_radius
        if cfg.randomize_attributes:
            v_pref = rs.uniform(0.5, 1.5)
            radius = rs.uniform(0.3, 0.5)
        if cfg.scenario == "circle_crossing":
            while True:
                angle = rs.random_sample() * np.pi * 2
                px_noise = (rs.random_sample() - 0.5) * v_pref
                py_noise = (rs.random_sample() - 0.5) * v_pref
                px = R * np.cos(angle) + px_noise
                py = R * np.sin(angle) + py_noise
                if clear_of(px, py, ag[:n, 0], ag[:n, 1], radius) and clear_of(px, py, ag[:n, 2], ag[:n, 3], radius):
                    break
            gx, gy = -px, -py
        elif cfg.scenario == "square_crossing":
            sign = -1 if rs.random_sample() > 0.5 else 1
            while True:
                px = rs.random_sample() * cfg.square_width * 0.5 * sign
                py = (rs.random_sample() - 0.5) * cfg.square_width
                if clear_of(px, py, ag[:n, 0], ag[:n, 1], radius):
                    break
            while True:
                gx = rs.random_sample() * cfg.square_width * 0.5 * -sign
                gy = (rs.random_sample() - 0.5) * cfg.square_width
                if clear_of(gx, gy, ag[:n, 2], ag[:n, 3], radius):
                    break
        else:
            raise NotImplementedError(cfg.scenario)
        ag[n] = (px, py, gx, gy, radius)
        n += 1
        humans.append([px, py, 0.0, 0.0, radius])
        goals.append([gx, gy])
        vprefs.append(v_pref)
    return robot, np.array(humans), np.array(goals), np.array(vprefs, dtype=np.float64)


class BatchedCrowdSim(object):
    def __init__(self, device, config=None, human_policy="linear", kinematics="holonomic"):
        self.cfg = config or SimConfig()
        self.device = torch.device(device)
        self.human_policy = human_policy
        self.kinematics = kinematics
        self.B = 0
        self._scene_cache = {}

    # -- state ---------------------------------------------------------------------------------------------------
    def reset(self, phase, cases):
        """Load seeded cases (one environment each).  Returns the fp32 observation (robot (B,9), humans (B,H,5))."""
        scenes = []
        for k in cases:                      # scene generation is sequential host work (seeded rejection sampling): memoise
            key = (phase, int(k), self.cfg.scenario, self.cfg.human_num, self.cfg.randomize_attributes, self.cfg.circle_radius,
                   self.cfg.square_width)
            if key not in self._scene_cache:
                self._scene_cache[key] = generate_scene(self.cfg, phase, int(k))
            scenes.append(self._scene_cache[key])
        return self.load(np.stack([s[0] for s in scenes]), np.stack([s[1] for s in scenes]),
                         np.stack([s[2] for s in scenes]), np.stack([s[3] for s in scenes]))

    def load(self, robot, humans, human_goals=None, human_vpref=None):
        dev = self.device
        self.robot = torch.as_tensor(np.asarray(robot, np.float64)).to(dev).contiguous()
        self.humans = torch.as_tensor(np.asarray(humans, np.float64)).to(dev).contiguous()
        self.B, self.H = self.robot.shape[0], self.humans.shape[1]
        self.human_goals = None if human_goals is None else torch.as_tensor(np.asarray(human_goals, np.float64)).to(dev).contiguous()
        self.human_vpref = None if human_vpref is None else torch.as_tensor(np.asarray(human_vpref, np.float64)).to(dev).contiguous()
        self.time = torch.zeros(self.B, dtype=torch.float64, device=dev)
        self.done = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self._r32 = torch.empty(self.B, 9, dtype=torch.float32, device=dev)
        self._h32 = torch.empty(self.B, self.H, 5, dtype=torch.float32, device=dev)
        return self.observe()

    def observe(self):
        with torch.cuda.device(self.device):
            nat.check(nat.lib().crowd_observe_f32(self.robot.data_ptr(), self.humans.data_ptr(), self.B, self.H,
                                                  self._r32.data_ptr(), self._h32.data_ptr(), _stream()), "crowd_observe_f32")
        return self._r32, self._h32

    def _config(self):
        c = nat.CrowdSimConfig()
        c.time_step, c.time_limit = self.cfg.time_step, self.cfg.time_limit
        c.success_reward, c.collision_penalty = self.cfg.success_reward, self.cfg.collision_penalty
        c.discomfort_dist, c.discomfort_penalty_factor = self.cfg.discomfort_dist, self.cfg.discomfort_penalty_factor
        c.kinematics = nat.KINEMATICS[self.kinematics]
        c.human_policy = HUMAN_POLICY[self.human_policy]
        return c

    # -- dynamics ------------------------------------------------------------------------------------------------
    def step(self, robot_actions, human_actions=None, update=True):
        """robot_actions (B,2) float64 (vx,vy)|(v,r).  Returns (obs, reward (B,) fp32, done (B,) bool, info (B,) int32);
        `self.last_dmin` holds the closest approach (Discomfort.min_dist).  Finished environments stay frozen."""
        dev = self.device
        act = torch.as_tensor(robot_actions, dtype=torch.float64).to(dev).contiguous()
        ha = None if human_actions is None else torch.as_tensor(human_actions, dtype=torch.float64).to(dev).contiguous()
        reward = torch.empty(self.B, dtype=torch.float32, device=dev)
        info = torch.empty(self.B, dtype=torch.int32, device=dev)
        dmin = torch.empty(self.B, dtype=torch.float64, device=dev)
        cfg = self._config()
        with torch.cuda.device(dev):
            rc = nat.lib().crowd_step_f64(C.byref(cfg), self.robot.data_ptr(), self.humans.data_ptr(),
                                          None if self.human_goals is None else self.human_goals.data_ptr(),
                                          None if self.human_vpref is None else self.human_vpref.data_ptr(),
                                          act.data_ptr(), None if ha is None else ha.data_ptr(), self.time.data_ptr(),
                                          self.done.data_ptr(), self.B, self.H, int(update), reward.data_ptr(),
                                          info.data_ptr(), dmin.data_ptr(), _stream())
        nat.check(rc, "crowd_step_f64")
        self.last_dmin = dmin
        done = (info >= 2) & (info <= 4) if not update else self.done.bool()
        return self.observe(), reward, done, info

    def onestep_lookahead(self, robot_actions, human_actions=None):
        return self.step(robot_actions, human_actions, update=False)

    def onestep_lookahead_actions(self, actions, env_index=0):
        """CrowdSim.onestep_lookahead (crowd_sim.py:249-250) for EVERY action of a table at once: A copies of environment
        `env_index` are stepped once on the device (the environment itself is untouched).  actions (A,2) float64 ->
        (next observable human states (A,H,5) float64, reward (A,) float32).  What path G's query_env=True asks per action
        (multi_human_rl.py:43-44)."""
        dev = self.device
        act = torch.as_tensor(actions, dtype=torch.float64).to(dev).contiguous()
        A = act.shape[0]
        b = int(env_index)
        robot = self.robot[b:b + 1].repeat(A, 1).contiguous()
        humans = self.humans[b:b + 1].repeat(A, 1, 1).contiguous()
        goals = None if self.human_goals is None else self.human_goals[b:b + 1].repeat(A, 1, 1).contiguous()
        vpref = None if self.human_vpref is None else self.human_vpref[b:b + 1].repeat(A, 1).contiguous()
        time = self.time[b:b + 1].repeat(A).contiguous()
        done = torch.zeros(A, dtype=torch.int32, device=dev)
        reward = torch.empty(A, dtype=torch.float32, device=dev)
        info = torch.empty(A, dtype=torch.int32, device=dev)
        dmin = torch.empty(A, dtype=torch.float64, device=dev)
        cfg = self._config()
        with torch.cuda.device(dev):
            rc = nat.lib().crowd_step_f64(C.byref(cfg), robot.data_ptr(), humans.data_ptr(),
                                          None if goals is None else goals.data_ptr(),
                                          None if vpref is None else vpref.data_ptr(), act.data_ptr(), None,
                                          time.data_ptr(), done.data_ptr(), A, self.H, 1, reward.data_ptr(),
                                          info.data_ptr(), dmin.data_ptr(), _stream())
        nat.check(rc, "crowd_step_f64")
        return humans, reward


def run_episodes(sim, policy, phase, cases, gamma=0.9, max_steps=None):
    """Explorer.run_k_episodes for len(cases) environments in lock-step: the policy decides for every live environment
    at once (`predict_batch`), the simulator advances them together.  Returns per-case outcome codes, times and
    discounted cumulative rewards plus the aggregate statistics the reference logs."""
    robot32, humans32 = sim.reset(phase, cases)
    B = sim.B
    if policy.action_space is None:
        policy.build_action_space(sim.cfg.robot_v_pref)
    from .actions import as_array
    table = torch.tensor(as_array(policy.action_space), dtype=torch.float64, device=sim.device)
    outcome = torch.zeros(B, dtype=torch.int32, device=sim.device)
    cum = torch.zeros(B, dtype=torch.float64, device=sim.device)
    discomfort_steps = torch.zeros(B, dtype=torch.int32, device=sim.device)
    max_steps = max_steps or int(sim.cfg.time_limit / sim.cfg.time_step) + 2
    disc = 1.0
    for t in range(max_steps):
        live = sim.done == 0
        if not bool(live.any()):
            break
        act_idx, _ = policy.predict_batch(robot32, humans32, roots_are_joint_states=True)
        (robot32, humans32), reward, done, info = sim.step(table[act_idx.long()])
        cum += disc * reward.double()
        discomfort_steps += (info == 1).int()
        ended = (info >= 2) & (info <= 4)
        outcome = torch.where(ended, info, outcome)
        disc *= pow(gamma, sim.cfg.time_step * sim.cfg.robot_v_pref)
    outcome_c = outcome.cpu().numpy()
    times = sim.time.cpu().numpy()
    success, collision, timeout = outcome_c == 3, outcome_c == 2, outcome_c == 4
    nav = times[success]
    return {"outcome": outcome_c, "time": np.where(timeout, sim.cfg.time_limit, times), "cumulative_reward": cum.cpu().numpy(),
            "success_rate": float(success.mean()), "collision_rate": float(collision.mean()),
            "timeout_rate": float(timeout.mean()), "unfinished": int((outcome_c == 0).sum()),
            "avg_nav_time": float(nav.mean()) if nav.size else float(sim.cfg.time_limit),
            "discomfort_steps": discomfort_steps.cpu().numpy()}
