"""Policy classes with the reference's Policy.predict() surface, backed by the HIP rollout.

Mirrors (reference paths): crowd_sim/envs/policy/policy.py:6-65 (`Policy`),
crowd_nav/policy/model_predictive_rl.py:15-370 (`ModelPredictiveRL`), crowd_nav/policy/gcn.py:131-157 +
multi_human_rl.py:12-112 + cadrl.py:70-138 (`GCN`), crowd_nav/policy/policy_factory.py:9-13 (registration).

`register(policy_factory)` installs the classes under the reference's keys ('model_predictive_rl', 'gcn'),
so crowd_nav's train.py / test.py pick them up unchanged.
"""
import logging

import numpy as np
import torch

from . import actions as act
from .nets import RGL, ValueEstimator, StatePredictor, LinearStatePredictor, ValueNetwork
from .rollout import TreeSearch, GcnSearch, rotate, prepare_scenes


class Policy(object):
    def __init__(self):
        self.trainable = False
        self.phase = None
        self.model = None
        self.device = None
        self.last_state = None
        self.time_step = None
        self.env = None

    def configure(self, config):
        raise NotImplementedError

    def set_phase(self, phase):
        self.phase = phase

    def set_device(self, device):
        self.device = device

    def set_env(self, env):
        self.env = env

    def set_time_step(self, time_step):
        self.time_step = time_step

    def get_model(self):
        return self.model

    def save_model(self, file):
        torch.save(self.model.state_dict(), file)

    def load_model(self, file):
        self.model.load_state_dict(torch.load(file))

    def get_state_dict(self):
        return self.model.state_dict()

    def load_state_dict(self, state_dict):
        self.model.load_state_dict(state_dict)

    def predict(self, state):
        raise NotImplementedError

    @staticmethod
    def reach_destination(state):
        r = state.robot_state
        return bool(np.linalg.norm((r.py - r.gy, r.px - r.gx)) < r.radius)


def _state_rows(state):
    """JointState -> (robot row of 9 floats, list of human rows of 5 floats)."""
    r = state.robot_state
    robot = [r.px, r.py, r.vx, r.vy, r.radius, r.gx, r.gy, r.v_pref, r.theta]
    humans = [[h.px, h.py, h.vx, h.vy, h.radius] for h in state.human_states]
    return robot, humans


def _check_ready(policy):
    if policy.phase is None or policy.device is None:
        raise AttributeError('Phase, device attributes have to be set!')
    if policy.phase == 'train' and policy.epsilon is None:
        raise AttributeError('Epsilon attribute has to be set in training phase')


class ModelPredictiveRL(Policy):
    """d-step model-predictive policy over the relational graph value/predictor networks (path M)."""

    def __init__(self):
        super().__init__()
        self.name = 'ModelPredictiveRL'
        self.trainable = True
        self.multiagent_training = True
        self.kinematics = None
        self.epsilon = None
        self.gamma = None
        self.speed_samples = None
        self.rotation_samples = None
        self.action_space = None
        self.rotation_constraint = None
        self.speeds = None                  # read by CrowdSim.render (crowd_sim.py:601-602), like rotations
        self.rotations = None
        self.action_values = None           # crowd_sim.py:332
        self.robot_state_dim = 9
        self.human_state_dim = 5
        self.v_pref = 1
        self.share_graph_model = None
        self.value_estimator = None
        self.linear_state_predictor = None
        self.state_predictor = None
        self.planning_depth = None
        self.planning_width = None
        self.do_action_clip = None
        self.sparse_search = None
        self.sparse_rotation_samples = 8
        self.action_group_index = []
        self._traj = None
        self._traj_search = None            # the search whose best branch `traj` describes, read back on demand
        self.contraction_dtype = "f32"      # additive: "f16" = f16-input MFMA for the dense middle-layer products
        self._search = None

    # -- wiring ----------------------------------------------------------------------------------
    def configure(self, config):
        self.set_common_parameters(config)
        mp = config.model_predictive_rl
        self.planning_depth = mp.planning_depth
        self.do_action_clip = mp.do_action_clip
        if hasattr(mp, 'sparse_search'):
            self.sparse_search = mp.sparse_search
        if hasattr(mp, 'contraction_dtype'):
            self.contraction_dtype = mp.contraction_dtype
        self.planning_width = mp.planning_width
        self.share_graph_model = mp.share_graph_model
        self.linear_state_predictor = mp.linear_state_predictor
        value_graph = RGL(config, self.robot_state_dim, self.human_state_dim)
        self.value_estimator = ValueEstimator(config, value_graph)
        if self.linear_state_predictor:
            self.state_predictor = LinearStatePredictor(config, self.time_step)
            self.model = [value_graph, self.value_estimator.value_network]
        elif self.share_graph_model:
            self.state_predictor = StatePredictor(config, value_graph, self.time_step)
            self.model = [value_graph, self.value_estimator.value_network,
                          self.state_predictor.human_motion_predictor]
        else:
            predictor_graph = RGL(config, self.robot_state_dim, self.human_state_dim)
            self.state_predictor = StatePredictor(config, predictor_graph, self.time_step)
            self.model = [value_graph, predictor_graph, self.value_estimator.value_network,
                          self.state_predictor.human_motion_predictor]
        self._search = None
        logging.info('Planning depth: {}'.format(self.planning_depth))
        logging.info('Planning width: {}'.format(self.planning_width))
        logging.info('Sparse search: {}'.format(self.sparse_search))
        if self.planning_depth > 1 and not self.do_action_clip:
            logging.warning('Performing d-step planning without action space clipping!')

    def set_common_parameters(self, config):
        self.gamma = config.rl.gamma
        a = config.action_space
        self.kinematics = a.kinematics
        self.speed_samples = a.speed_samples
        self.rotation_samples = a.rotation_samples
        self.rotation_constraint = a.rotation_constraint

    def set_device(self, device):
        self.device = device
        for m in self.model:
            m.to(device)
        self._search = None

    def set_epsilon(self, epsilon):
        self.epsilon = epsilon

    def set_time_step(self, time_step):
        self.time_step = time_step
        self.state_predictor.time_step = time_step
        self._search = None

    def get_normalized_gamma(self):
        return pow(self.gamma, self.time_step * self.v_pref)

    def get_model(self):
        return self.value_estimator

    @property
    def traj(self):
        """[(state tensors, action, reward), ...] along the best branch of the last decision (model_predictive_rl.py:231,
        298-302).  Read back from the device when somebody asks (CrowdSim.render does), not on every predict()."""
        if self._traj is None and self._traj_search is not None:
            ts, last = self._traj_search
            ts.last = last
            self._traj = [(s, None if a is None else self.action_space[a], r) for s, a, r in ts.best_trajectory(0)]
            self._traj_search = None
        return self._traj

    @traj.setter
    def traj(self, value):
        self._traj, self._traj_search = value, None

    def get_traj(self):
        return self.traj

    # -- checkpoints: same nested layout as upstream ---------------------------------------------
    def get_state_dict(self):
        ve, sp = self.value_estimator, self.state_predictor
        if not sp.trainable:
            return {'graph_model': ve.graph_model.state_dict(), 'value_network': ve.value_network.state_dict()}
        if self.share_graph_model:
            return {'graph_model': ve.graph_model.state_dict(), 'value_network': ve.value_network.state_dict(),
                    'motion_predictor': sp.human_motion_predictor.state_dict()}
        return {'graph_model1': ve.graph_model.state_dict(), 'graph_model2': sp.graph_model.state_dict(),
                'value_network': ve.value_network.state_dict(),
                'motion_predictor': sp.human_motion_predictor.state_dict()}

    def load_state_dict(self, state_dict):
        ve, sp = self.value_estimator, self.state_predictor
        if sp.trainable and not self.share_graph_model:
            ve.graph_model.load_state_dict(state_dict['graph_model1'])
            sp.graph_model.load_state_dict(state_dict['graph_model2'])
        else:
            ve.graph_model.load_state_dict(state_dict['graph_model'])
        ve.value_network.load_state_dict(state_dict['value_network'])
        if sp.trainable:
            sp.human_motion_predictor.load_state_dict(state_dict['motion_predictor'])

    def save_model(self, file):
        torch.save(self.get_state_dict(), file)

    def load_model(self, file):
        self.load_state_dict(torch.load(file, map_location=self.device))

    # -- action space ----------------------------------------------------------------------------
    def build_action_space(self, v_pref):
        actions, groups, speeds, rotations = act.speed_major_table(
            v_pref, self.speed_samples, self.rotation_samples, self.kinematics, self.rotation_constraint,
            self.sparse_rotation_samples)
        self.action_space = actions
        self.action_group_index = groups
        self.speeds = speeds
        self.rotations = rotations
        self._search = None

    def tree_search(self):
        """The device search object for the current configuration (rebuilt when settings change)."""
        key = (self.planning_depth, self.planning_width, bool(self.do_action_clip), bool(self.sparse_search),
               self.kinematics, self.time_step, self.gamma, id(self.state_predictor), id(self.value_estimator),
               self.contraction_dtype)
        if self._search is None or self._search[0] != key:
            if self.action_space is None:
                self.build_action_space(self.v_pref)
            ts = TreeSearch(self.value_estimator, self.state_predictor, act.as_array(self.action_space),
                            self.action_group_index, self.kinematics, self.time_step, self.get_normalized_gamma(),
                            self.planning_depth, self.planning_width, self.do_action_clip, self.sparse_search,
                            self.contraction_dtype)
            self._search = (key, ts)
        return self._search[1]

    # -- decisions -------------------------------------------------------------------------------
    def predict(self, state):
        _check_ready(self)
        if self.reach_destination(state):
            return act.stop_action(self.kinematics)
        if self.action_space is None:
            self.build_action_space(state.robot_state.v_pref)
        probability = np.random.random()          # drawn in every phase, like upstream, to keep RNG streams aligned
        if self.phase == 'train' and probability < self.epsilon:
            max_action = self.action_space[np.random.choice(len(self.action_space))]
            max_traj = None
        else:
            robot, humans = _state_rows(state)
            ts = self.tree_search()
            with torch.no_grad():
                idx = ts.decide(robot, humans)            # captured hipGraph of the whole search for this crowd size
            if idx < 0:
                raise ValueError('Value network is not well trained.')
            max_action = self.action_space[idx]
            max_traj = (ts, ts.last)
        if self.phase == 'train':
            self.last_state = self.transform(state)
        else:
            self._traj, self._traj_search = None, max_traj
        return max_action

    def predict_batch(self, robot, humans, roots_are_joint_states=False):
        """Additive API: robot (B,9), humans (B,H,5) device tensors -> (action index (B,), value (B,))."""
        if self.action_space is None:
            self.build_action_space(self.v_pref)
        with torch.no_grad():
            out = self.tree_search().search(robot, humans, roots_are_joint_states=roots_are_joint_states)
        return out["best_action"], out["best_value"]

    # -- the planner's steps as methods (model_predictive_rl.py:242-357), each backed by the device function the search runs --------
    def _planner_state(self, state):
        """A JointState, or the (robot (1,1,9) | (1,9), humans (1,H,5)) tensor pair the reference's planner passes around
        -> (robot (1,9) fp32, humans (1,H,5) fp32 device tensors, is_joint_state, float64 roots or None)."""
        if isinstance(state, (tuple, list)):
            robot = torch.as_tensor(state[0], dtype=torch.float32).reshape(1, 9).to(self.device).contiguous()
            humans = torch.as_tensor(state[1], dtype=torch.float32).to(self.device)
            return robot, humans.reshape(1, -1, 5).contiguous(), False, None
        rrow, hrows = _state_rows(state)
        r64 = torch.tensor([rrow], dtype=torch.float64, device=self.device)
        h64 = torch.tensor([hrows], dtype=torch.float64, device=self.device).reshape(1, len(hrows), 5)
        return r64.float(), h64.float(), True, (r64, h64)

    def _search_for(self, action_space):
        """The device search object for an action list: the policy's own table -> tree_search(); another list -> a TreeSearch over
        that table (groups only when they still index it)."""
        if self.action_space is None:
            self.build_action_space(self.v_pref)
        if action_space is None or action_space is self.action_space or list(action_space) == list(self.action_space):
            return self.tree_search()
        groups = self.action_group_index if len(self.action_group_index) == len(action_space) else None
        return TreeSearch(self.value_estimator, self.state_predictor, act.as_array(action_space), groups, self.kinematics,
                          self.time_step, self.get_normalized_gamma(), self.planning_depth, self.planning_width,
                          self.do_action_clip, self.sparse_search and groups is not None, self.contraction_dtype)

    def estimate_reward(self, state, action):
        """model_predictive_rl.py:304-357 for one (state, action): the level kernel's float64 reward arithmetic
        (mprl_estimate_reward_f32) on a one-row action table.  A JointState is read in float64, a tensor state through the
        float32 differences tensor_to_joint_state + numpy scalars produce.  The device hands the reward back as float32 -- the
        precision it enters `reward_est + gamma * value` with upstream (a float32 tensor expression)."""
        robot, humans, joint, roots64 = self._planner_state(state)
        a = [action.vx, action.vy] if self.kinematics == 'holonomic' else [action.v, action.r]
        with torch.no_grad():
            _, rew = self.tree_search().estimate_reward(robot, humans, joint, roots64, actions=[a])
        return float(rew[0, 0])

    def action_clip(self, state, action_space, width, depth=1):
        """model_predictive_rl.py:242-269: the `width` actions with the best `reward + gamma_bar * V_planning(next state, depth,
        width)` (one per action group in a sparse search).  depth = 1 (every call upstream makes) is one device level --
        mprl_expand_f32 -- and the selection kernel of the search (mprl_action_clip_f32); deeper values come from V_planning.
        The kept SET is the reference's; the order is descending one-step value (np.argpartition's order is arbitrary)."""
        robot, humans, joint, roots64 = self._planner_state(state)
        ts = self._search_for(action_space)
        with torch.no_grad():
            if roots64 is not None:
                child, reward = ts.estimate_reward(robot, humans, True, roots64)
            o = ts.expand(robot, humans, parents_are_joint_states=joint)
            if roots64 is not None:
                o["reward"] = reward
            if depth == 1:
                child_value = o["child_value"]
            else:
                vals = [float(self.V_planning((o["child_robot"][0, a].reshape(1, 1, 9), o["humans_next"]), depth, width)[0])
                        for a in range(ts.num_actions)]
                child_value = torch.tensor([vals], dtype=torch.float32, device=self.device)
            if not ts.do_action_clip:                      # the method clips whatever the policy's own flag says
                ts = TreeSearch(ts.value_estimator, ts.state_predictor, ts.actions_np, ts.groups_np, ts.kinematics, ts.time_step,
                                ts.gamma_bar, ts.planning_depth, width, True, ts.sparse_search, ts.contraction_dtype)
            value1, keep = ts.action_clip(o["reward"], child_value, width)
        kept = keep[0].tolist()
        if ts.sparse_search and ts.groups_np is not None:
            # fewer distinct groups than `width`: upstream's walk ends with the shorter list (:252-263); the device step pads its
            # fixed-width row by repeating the last kept action -- cut the padding off again
            n_groups = len(set(int(g) for g in ts.groups_np))
            kept = kept[:min(len(kept), n_groups)]
        return [action_space[i] for i in kept]

    def V_planning(self, state, depth, width):
        """model_predictive_rl.py:271-302: (value (1,1) tensor, trajectory [(state, action, reward), ...]) of planning `depth`
        steps ahead from a tensor state (or a JointState).  Every level is the search's own device work -- ValueEstimator forward,
        mprl_expand_f32 (state predictor, next states, rewards, V of the 81 children), mprl_action_clip_f32 -- and the back-up
        `v/d + (d-1)/d (gamma_bar * next + r)` is taken on the host in float32, op for op as the device back-up (rgl_tail.h) and
        the reference's tensor expression do."""
        robot, humans, joint, roots64 = self._planner_state(state)
        state_t = (robot.reshape(1, 1, 9), humans)
        with torch.no_grad():
            v = self.value_estimator((state_t[0], humans))
            if depth == 1:
                return v, [(state_t, None, None)]
            ts = self.tree_search()
            o = ts.expand(robot, humans, parents_are_joint_states=joint)
            if roots64 is not None:
                o["reward"] = ts.estimate_reward(robot, humans, True, roots64)[1]
            if self.do_action_clip:
                kept = ts.action_clip(o["reward"], o["child_value"], width)[1][0].tolist()
            else:
                kept = list(range(ts.num_actions))
            f32 = np.float32
            g, d = f32(self.get_normalized_gamma()), f32(depth)
            c = f32((depth - 1) / depth)
            v_over_d = f32(f32(v.reshape(-1)[0].item()) / d)
            rewards = o["reward"][0].cpu().numpy()
            best, best_ret, best_traj = None, None, None
            for a in kept:
                nxt = (o["child_robot"][0, a].reshape(1, 1, 9).clone(), o["humans_next"].clone())
                nv, ntraj = self.V_planning(nxt, depth - 1, self.planning_width)
                inner = f32(f32(g * f32(nv.reshape(-1)[0].item())) + rewards[a])
                ret = f32(v_over_d + f32(c * inner))
                if best is None or ret > best_ret:          # np.argmax: the first maximum
                    best, best_ret, best_traj = a, ret, [(state_t, self.action_space[a], float(rewards[a]))] + ntraj
        return torch.tensor([[best_ret]], dtype=torch.float32, device=self.device), best_traj

    def _root_tensors(self, state):
        robot, humans = _state_rows(state)
        robot_t = torch.tensor([robot], dtype=torch.float32, device=self.device)
        humans_t = torch.tensor([humans], dtype=torch.float32, device=self.device).reshape(1, len(humans), 5)
        return robot_t, humans_t

    def transform(self, state):
        robot, humans = _state_rows(state)
        return (torch.tensor([robot], dtype=torch.float32, device=self.device),
                torch.tensor(humans, dtype=torch.float32, device=self.device))


class GCN(Policy):
    """One-step lookahead policy over rotated pairwise states with the graph ValueNetwork (path G)."""

    def __init__(self):
        super().__init__()
        self.name = 'GCN'
        self.trainable = True
        self.multiagent_training = None
        self.kinematics = None
        self.epsilon = None
        self.gamma = None
        self.speed_samples = None
        self.rotation_samples = None
        self.query_env = None
        self.action_space = None
        self.rotation_constraint = None
        self.speeds = None
        self.rotations = None
        self.action_values = None
        self.self_state_dim = 6
        self.human_state_dim = 7
        self.joint_state_dim = self.self_state_dim + self.human_state_dim
        self._search = None

    def configure(self, config):
        gc = config.gcn
        self.multiagent_training = gc.multiagent_training
        self.set_common_parameters(config)
        self.model = ValueNetwork(self.input_dim(), self.self_state_dim, gc.num_layer, gc.X_dim, gc.wr_dims, gc.wh_dims,
                                  gc.final_state_dim, gc.gcn2_w1_dim, gc.planning_dims, gc.similarity_function,
                                  gc.layerwise_graph, gc.skip_connection)
        self._search = None
        logging.info('GCN layers: {}'.format(gc.num_layer))
        logging.info('Policy: {}'.format(self.name))

    def set_common_parameters(self, config):
        self.gamma = config.rl.gamma
        a = config.action_space
        self.kinematics = a.kinematics
        self.speed_samples = a.speed_samples
        self.rotation_samples = a.rotation_samples
        self.query_env = a.query_env
        self.rotation_constraint = a.rotation_constraint

    def input_dim(self):
        return self.joint_state_dim        # occupancy maps are never enabled for this policy upstream (gcn.py:131-157)

    def set_device(self, device):
        self.device = device
        self.model.to(device)
        self._search = None

    def set_epsilon(self, epsilon):
        self.epsilon = epsilon

    def load_model(self, file):
        self.model.load_state_dict(torch.load(file, map_location=self.device))

    def get_matrix_A(self):
        return self.model.A

    def build_action_space(self, v_pref):
        actions, speeds, rotations = act.rotation_major_table(v_pref, self.speed_samples, self.rotation_samples,
                                                              self.kinematics, self.rotation_constraint)
        self.action_space = actions
        self.speeds = speeds
        self.rotations = rotations
        self._search = None

    def gcn_search(self):
        mode = getattr(self, "contraction_dtype", "f32")       # additive: "bf16x6" = the graph's weight products on the matrix pipe
        key = (self.kinematics, self.time_step, self.gamma, mode)
        if self._search is None or self._search[0] != key:
            self._search = (key, GcnSearch(self.model, act.as_array(self.action_space), self.kinematics,
                                           self.time_step, self.gamma, mode))
        return self._search[1]

    def predict(self, state):
        _check_ready(self)
        if self.reach_destination(state):
            return act.stop_action(self.kinematics)
        if self.action_space is None:
            self.build_action_space(state.robot_state.v_pref)
        if not state.human_states:
            assert self.phase != 'train'                       # multi_human_rl.py:27-31
            return self.select_greedy_action(state.robot_state)
        probability = np.random.random()
        if self.phase == 'train' and probability < self.epsilon:
            max_action = self.action_space[np.random.choice(len(self.action_space))]
        else:
            robot, humans = _state_rows(state)
            r64 = torch.tensor([robot], dtype=torch.float64, device=self.device)
            h64 = torch.tensor([humans], dtype=torch.float64, device=self.device)
            with torch.no_grad():
                if self.query_env:
                    vals, best = self._search_querying_env(robot, r64)
                else:
                    vals, best = self.gcn_search().search(r64.float(), h64.float(), roots64=(r64, h64))
                    # keep the visual-debug hook: adjacency of the LAST action's graph, as the sequential loop leaves it
                    self._refresh_adjacency(robot, humans)
            self.action_values = [float(v) for v in vals[0].cpu()]
            idx = int(best[0])
            if idx < 0:
                raise ValueError('Value network is not well trained. ')
            max_action = self.action_space[idx]
        if self.phase == 'train':
            self.last_state = self.transform(state)
        return max_action

    def predict_batch(self, robot, humans, roots_are_joint_states=True):
        """Additive API: robot (B,9), humans (B,H,5) device tensors -> (action index (B,), its value (B,))."""
        if self.action_space is None:
            self.build_action_space(float(robot[0, 7]))
        with torch.no_grad():
            search = self.gcn_search()
            vals, best = search.search(robot, humans)
        return best, search.last_best_value

    def _search_querying_env(self, robot, r64):
        """query_env=True (multi_human_rl.py:43-44): next human states and the reward of every action come from the simulator's
        one-step lookahead instead of the constant-velocity guess.  `self.env` must offer
        `onestep_lookahead_actions(actions (A,2) float64) -> (next_humans (A,H,5), reward (A,))` for the scene it holds
        (BatchedCrowdSim does: A copies of the scene stepped once on the device)."""
        if self.env is None or not hasattr(self.env, "onestep_lookahead_actions"):
            raise AttributeError("query_env=True needs policy.set_env(sim) with a simulator offering onestep_lookahead_actions")
        table = torch.tensor(act.as_array(self.action_space), dtype=torch.float64, device=self.device)
        next_humans, reward = self.env.onestep_lookahead_actions(table)              # (A,H,5) float64, (A,) float32
        A, H = table.shape[0], next_humans.shape[1]
        dt = self.time_step
        nr = r64.repeat(A, 1)                                                        # CADRL.propagate, float64
        if self.kinematics == 'holonomic':
            nr[:, 0], nr[:, 1] = r64[0, 0] + table[:, 0] * dt, r64[0, 1] + table[:, 1] * dt
            nr[:, 2], nr[:, 3] = table[:, 0], table[:, 1]
        else:
            th = r64[0, 8] + table[:, 1]
            nr[:, 2], nr[:, 3] = table[:, 0] * torch.cos(th), table[:, 0] * torch.sin(th)
            nr[:, 0], nr[:, 1] = r64[0, 0] + nr[:, 2] * dt, r64[0, 1] + nr[:, 3] * dt
            nr[:, 8] = th
        joint = torch.cat([nr[:, None, :].expand(A, H, 9), next_humans.double()], dim=2).reshape(A * H, 14).float().contiguous()
        v = self.model(rotate(joint, self.kinematics).reshape(A, H, 13))             # leaves .A at the last action, like upstream
        disc = pow(self.gamma, dt * float(robot[7]))
        vals = (reward.double() + disc * v[:, 0].double()).float().reshape(1, A)
        return vals, _first_strict_maximum(vals)

    def compute_reward(self, nav, humans):
        """multi_human_rl.py:73-96 for one (next robot state, next human states) pair: the END-point clearance reward the one-step
        search scores every action with, evaluated by the search's own device step (gcn_prepare_f32, float64) -- the pair is
        handed over as a scene that a zero action and resting humans leave where it is."""
        if len(humans) == 0:
            # no crowd: upstream's loop body never runs -- the goal test decides alone (multi_human_rl.py:88-95; dmin stays inf).
            # The device step needs at least one human row, and there is nothing to evaluate on a device here.
            reaching = np.linalg.norm((nav.px - nav.gx, nav.py - nav.gy)) < nav.radius
            return 1 if reaching else 0
        row = [nav.px, nav.py, 0.0, 0.0, nav.radius, nav.gx, nav.gy, getattr(nav, "v_pref", 1.0), getattr(nav, "theta", 0.0)]
        hrows = [[h.px, h.py, 0.0, 0.0, h.radius] for h in humans]
        r64 = torch.tensor([row], dtype=torch.float64, device=self.device)
        h64 = torch.tensor([hrows], dtype=torch.float64, device=self.device).reshape(1, len(hrows), 5)
        return float(prepare_scenes(r64.float(), h64.float(), [[0.0, 0.0]], self.kinematics, self.time_step, (r64, h64))[2][0])

    def select_greedy_action(self, self_state):
        """Empty crowd (multi_human_rl.py:27-31 -> cadrl.py:193-228): no graph to evaluate -- the table action closest to the
        straight-to-goal velocity, first minimum.  Host arithmetic in float64 like upstream.  Unicycle: the two out-of-view cases;
        the in-view case raises upstream (a misplaced parenthesis hands np.array a float as dtype), so it raises here too."""
        direction = np.arctan2(self_state.gy - self_state.py, self_state.gx - self_state.px)
        distance = np.linalg.norm((self_state.gy - self_state.py, self_state.gx - self_state.px))
        if self.kinematics == 'holonomic':
            speed = min(distance / self.time_step, self_state.v_pref)
            target = np.array((np.cos(direction) * speed, np.sin(direction) * speed))
            min_diff, closest = float('inf'), None
            for action in self.action_space:
                diff = np.linalg.norm(np.array(action) - target)
                if diff < min_diff:
                    min_diff, closest = diff, action
            return closest
        rotation = direction - self_state.theta
        if rotation < self.rotations[0]:
            return act.ActionRot(self.speeds[0], self.rotations[0])
        if rotation > self.rotations[-1]:
            return act.ActionRot(self.speeds[0], self.rotations[-1])
        raise TypeError("select_greedy_action: the in-view unicycle case raises upstream (cadrl.py:222-223)")

    def _refresh_adjacency(self, robot, humans):
        last = self.action_space[-1]
        nxt = self.propagate_robot(robot, last)
        nh = [[h[0] + h[2] * self.time_step, h[1] + h[3] * self.time_step, h[2], h[3], h[4]] for h in humans]
        joint = torch.tensor([list(nxt) + h for h in nh], dtype=torch.float32, device=self.device)
        self.model(rotate(joint, self.kinematics).unsqueeze(0))

    def propagate_robot(self, robot, action):
        dt = self.time_step
        if self.kinematics == 'holonomic':
            return [robot[0] + action.vx * dt, robot[1] + action.vy * dt, action.vx, action.vy] + list(robot[4:9])
        th = robot[8] + action.r
        vx, vy = action.v * np.cos(th), action.v * np.sin(th)
        return [robot[0] + vx * dt, robot[1] + vy * dt, vx, vy] + list(robot[4:8]) + [th]

    def rotate(self, state):
        return rotate(state, self.kinematics)

    def transform(self, state):
        robot, humans = _state_rows(state)
        joint = torch.tensor([robot + h for h in humans], dtype=torch.float32, device=self.device)
        return rotate(joint, self.kinematics)


def _first_strict_maximum(vals):
    """Index of the FIRST maximum of every row, -1 when no value beats -inf (all NaN / -inf): the strict `>` walk of
    multi_human_rl.py:38-64 that leaves max_action None ("Value network is not well trained"), as gcn_argmax_kernel and
    mprl_root_kernel do it.  torch.argmax guarantees neither the first index on ties nor a refusal of NaN rows."""
    clean = torch.where(torch.isnan(vals), torch.full_like(vals, float("-inf")), vals)
    m = clean.max(dim=1, keepdim=True).values
    cols = torch.arange(vals.shape[1], device=vals.device).expand_as(vals)
    first = torch.where(clean == m, cols, torch.full_like(cols, vals.shape[1])).min(dim=1).values
    return torch.where(m[:, 0] > float("-inf"), first, torch.full_like(first, -1)).int()


def register(policy_factory):
    """Install the HIP-backed classes under the reference's registry keys."""
    policy_factory['model_predictive_rl'] = ModelPredictiveRL
    policy_factory['gcn'] = GCN
    return policy_factory
