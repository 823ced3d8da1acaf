"""CPU oracle for the RGL relational-graph forward pass and the model-predictive rollout.

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the product (`relationalgraphlearning_amd`) never does.

What it is: a restatement, in this repo's own words, of the arithmetic of the reference's hot path,
written as pure functions over plain fp32 tensors (torch CPU, because the reference's arithmetic *is*
ATen CPU: linear / matmul / softmax / relu, plus numpy scalar code for rewards).  It exists in two
shapes:

  * a *sequential* planner that walks the action tree one batch-1 forward at a time in exactly the
    order the reference does (`mprl_predict_sequential`, `gcn_predict_sequential`) -- this is "the
    reference PyTorch-CPU path" used as the reference-faithful CPU baseline;
  * a *batched*, level-synchronous planner (`mprl_predict_batched`) doing the same arithmetic for B
    root scenes at once -- used to check the GPU path at sizes the sequential walk cannot reach.

Parity pin: every function here is checked against fixtures produced by importing the reference
itself in the build container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`,
`tests/test_oracle_golden.py`).  The reference ships no golden vectors of its own.

Reference lines each function follows (paths relative to the upstream repo):
  mlp_forward                 crowd_nav/policy/helpers.py:5-13
  similarity_matrix           crowd_nav/policy/graph_model.py:63-97
  rgl_forward                 crowd_nav/policy/graph_model.py:99-130
  value_estimator_forward     crowd_nav/policy/value_estimator.py:11-20
  state_predictor_humans      crowd_nav/policy/state_predictor.py:20-39
  next_robot_state            crowd_nav/policy/state_predictor.py:41-60
  linear_humans               crowd_nav/policy/state_predictor.py:109-118
  mprl_action_space           crowd_nav/policy/model_predictive_rl.py:155-190
  point_to_segment_dist       crowd_sim/envs/utils/utils.py:4-26
  estimate_reward             crowd_nav/policy/model_predictive_rl.py:304-357
  mprl_predict_sequential     crowd_nav/policy/model_predictive_rl.py:192-302
  cadrl_action_space          crowd_nav/policy/cadrl.py:91-111
  rotate_pairwise             crowd_nav/policy/cadrl.py:241-276
  gcn_value_forward           crowd_nav/policy/gcn.py:85-128
  compute_reward_g            crowd_nav/policy/multi_human_rl.py:73-96
  gcn_predict_sequential      crowd_nav/policy/multi_human_rl.py:12-71
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

SIMILARITIES = ("embedded_gaussian", "gaussian", "cosine", "cosine_softmax", "concatenation",
                "squared", "equal_attention", "diagonal")


@dataclass
class OracleConfig:
    # graph
    num_layer: int = 2
    x_dim: int = 32
    similarity: str = "embedded_gaussian"
    layerwise_graph: bool = False
    skip_connection: bool = True
    # world
    kinematics: str = "holonomic"
    time_step: float = 0.25
    gamma: float = 0.9
    v_pref: float = 1.0
    speed_samples: int = 5
    rotation_samples: int = 16
    rotation_constraint: float = float(np.pi / 3)
    # planner (path M)
    planning_depth: int = 1
    planning_width: int = 1
    do_action_clip: bool = False
    sparse_search: bool = False
    linear_state_predictor: bool = False


# --------------------------------------------------------------------------------------------------
# parameter handling: the oracle consumes the reference's own state-dict key names
# --------------------------------------------------------------------------------------------------
def mlp_layers(sd: Dict[str, torch.Tensor], prefix: str):
    """Collect (W, b) of a Sequential[Linear(,ReLU)...] stored as '<prefix><i>.weight'."""
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd
                  if k.startswith(prefix) and k.endswith(".weight")})
    return [(sd["%s%d.weight" % (prefix, i)], sd["%s%d.bias" % (prefix, i)]) for i in idx]


def mlp_forward(x, layers, last_relu):
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, w, b)
        if i != n - 1 or last_relu:
            x = torch.relu(x)
    return x


def graph_weights(gsd, cfg):
    ws = [gsd["Ws.%d" % i] for i in range(cfg.num_layer)] if "Ws.0" in gsd else \
         [gsd["w%d" % (i + 1)] for i in range(cfg.num_layer)]          # path G names them w1, w2
    return ws


# --------------------------------------------------------------------------------------------------
# graph forward
# --------------------------------------------------------------------------------------------------
def similarity_matrix(X, gsd, mode):
    B, N, _ = X.shape
    Xt = X.transpose(1, 2)
    if mode == "embedded_gaussian":
        return torch.softmax(torch.matmul(torch.matmul(X, gsd["w_a"]), Xt), dim=2)
    if mode == "gaussian":
        return torch.softmax(torch.matmul(X, Xt), dim=2)
    if mode in ("cosine", "cosine_softmax"):
        S = torch.matmul(X, Xt)
        m = torch.norm(S, dim=2, keepdim=True)            # row norms of S itself (reference quirk)
        C = S / torch.matmul(m, m.transpose(1, 2))
        return C if mode == "cosine" else torch.softmax(C, dim=2)
    if mode == "concatenation":
        xi = X.unsqueeze(2).expand(B, N, N, X.shape[2])
        xj = X.unsqueeze(1).expand(B, N, N, X.shape[2])
        pairs = torch.cat([xi, xj], dim=3).reshape(B, N * N, -1)
        return mlp_forward(pairs, mlp_layers(gsd, "w_a."), last_relu=True).reshape(B, N, N)
    if mode == "squared":
        S = torch.matmul(X, Xt)
        S2 = S * S
        return S2 / S2.sum(dim=2, keepdim=True)
    if mode == "equal_attention":
        return (torch.ones(N, N) / N).expand(B, N, N)
    if mode == "diagonal":
        return torch.eye(N, N).expand(B, N, N)
    raise NotImplementedError(mode)


def propagate_layers(X, gsd, cfg):
    """Shared by path M (RGL) and path G (ValueNetwork): L x  H <- relu(A H W) (+ H)."""
    ws = graph_weights(gsd, cfg)
    A = None
    if not cfg.layerwise_graph:
        A = similarity_matrix(X, gsd, cfg.similarity)
    A_first = A
    H = X
    for l in range(cfg.num_layer):
        if cfg.layerwise_graph:
            A = similarity_matrix(H, gsd, cfg.similarity)
            if A_first is None:
                A_first = A
        nxt = torch.relu(torch.matmul(torch.matmul(A, H), ws[l]))
        if cfg.skip_connection:
            nxt = nxt + H
        H = nxt
    return H, A_first


def rgl_embed(robot, humans, gsd):
    xr = mlp_forward(robot, mlp_layers(gsd, "w_r."), last_relu=True)
    xh = mlp_forward(humans, mlp_layers(gsd, "w_h."), last_relu=True)
    return torch.cat([xr, xh], dim=1)


def rgl_forward(robot, humans, gsd, cfg):
    """robot (B,1,9), humans (B,H,5) -> (H_L (B,N,X), A (B,N,N))."""
    return propagate_layers(rgl_embed(robot, humans, gsd), gsd, cfg)


def value_estimator_forward(robot, humans, gsd, vsd, cfg):
    H, _ = rgl_forward(robot, humans, gsd, cfg)
    return mlp_forward(H[:, 0, :], mlp_layers(vsd, ""), last_relu=False)      # (B,1)


def state_predictor_humans(robot, humans, gsd, msd, cfg):
    H, _ = rgl_forward(robot, humans, gsd, cfg)
    return mlp_forward(H, mlp_layers(msd, ""), last_relu=False)[:, 1:, :]      # (B,H,5)


def linear_humans(humans):
    nxt = humans.clone()
    nxt[..., 0] = nxt[..., 0] + nxt[..., 2]          # no time-step factor: reference quirk, kept
    nxt[..., 1] = nxt[..., 1] + nxt[..., 3]
    return nxt


def next_robot_state(robot, action, cfg):
    """robot (...,9) fp32 tensor, action = (a0, a1) python/np float64 -> same shape."""
    nxt = robot.clone()
    dt = cfg.time_step
    if cfg.kinematics == "holonomic":
        nxt[..., 0] = nxt[..., 0] + action[0] * dt
        nxt[..., 1] = nxt[..., 1] + action[1] * dt
        nxt[..., 2] = action[0]
        nxt[..., 3] = action[1]
    else:
        # the reference adds the rotation to slot 7 (v_pref), not slot 8 (theta); kept as is
        nxt[..., 7] = nxt[..., 7] + action[1]
        nxt[..., 0] = nxt[..., 0] + torch.cos(nxt[..., 7]) * action[0] * dt
        nxt[..., 1] = nxt[..., 1] + torch.sin(nxt[..., 7]) * action[0] * dt
        nxt[..., 2] = torch.cos(nxt[..., 7]) * action[0]
        nxt[..., 3] = torch.sin(nxt[..., 7]) * action[0]
    return nxt


# --------------------------------------------------------------------------------------------------
# action spaces
# --------------------------------------------------------------------------------------------------
def _speeds_rotations(cfg, v_pref):
    speeds = [(np.exp((i + 1) / cfg.speed_samples) - 1) / (np.e - 1) * v_pref
              for i in range(cfg.speed_samples)]
    if cfg.kinematics == "holonomic":
        rotations = np.linspace(0, 2 * np.pi, cfg.rotation_samples, endpoint=False)
    else:
        rotations = np.linspace(-cfg.rotation_constraint, cfg.rotation_constraint, cfg.rotation_samples)
    return speeds, rotations


def mprl_action_space(cfg, v_pref=1.0, sparse_rotation_samples=8):
    """Path M table: stop action, then speed-major (speed outer, rotation inner).
    Returns (actions float64 (A,2), group ids int (A,))."""
    speeds, rotations = _speeds_rotations(cfg, v_pref)
    holo = cfg.kinematics == "holonomic"
    acts, groups = [(0.0, 0.0)], [0]
    for j, s in enumerate(speeds):
        band = 0 if j < 3 else 1
        for i, r in enumerate(rotations):
            groups.append(band * sparse_rotation_samples + i // 2)
            acts.append((s * np.cos(r), s * np.sin(r)) if holo else (s, r))
    return np.asarray(acts, dtype=np.float64), np.asarray(groups, dtype=np.int64)


def cadrl_action_space(cfg, v_pref=1.0):
    """Path G table: stop action, then rotation-major (rotation outer, speed inner)."""
    speeds, rotations = _speeds_rotations(cfg, v_pref)
    holo = cfg.kinematics == "holonomic"
    acts = [(0.0, 0.0)]
    for r in rotations:
        for s in speeds:
            acts.append((s * np.cos(r), s * np.sin(r)) if holo else (s, r))
    return np.asarray(acts, dtype=np.float64)


# --------------------------------------------------------------------------------------------------
# rewards (numpy scalar arithmetic, dtype follows the inputs exactly like the reference)
# --------------------------------------------------------------------------------------------------
def point_to_segment_dist(x1, y1, x2, y2, x3, y3):
    px, py = x2 - x1, y2 - y1
    if px == 0 and py == 0:
        return np.linalg.norm((x3 - x1, y3 - y1))
    u = ((x3 - x1) * px + (y3 - y1) * py) / (px * px + py * py)
    u = 1 if u > 1 else (0 if u < 0 else u)
    return np.linalg.norm((x1 + u * px - x3, y1 + u * py - y3))


def action_object(actions, i):
    """Row i of an action table as the scalars upstream's action object holds: numpy float64 products for every entry
    (model_predictive_rl.py:183-186) except the stop action, which is `ActionXY(0, 0)` / `ActionRot(0, 0)` -- PYTHON INTS (:166).
    The difference is arithmetic, not cosmetic: under numpy >= 2 promotion (NEP 50) python scalars are weak, so estimate_reward
    of a tensor-born state (numpy float32 scalars) with the stop action runs in float32 from end to end, while a float64 action
    component widens it.  (The reference as it runs in this image, numpy 2.2.6; fixture root_clip.npz, action 0.)"""
    a0, a1 = actions[i][0], actions[i][1]
    if a0 == 0 and a1 == 0:
        return (0, 0)
    return (np.float64(a0), np.float64(a1))


def estimate_reward(robot, humans, action, cfg):
    """robot: 9 scalars, humans: iterable of 5-scalar rows (python floats for a root JointState,
    np.float32 for states that came back from tensors), action: 2 scalars as `action_object` gives them (np.float64, or
    python ints for the stop action).  Scalar code: numpy's promotion rules do the rest, as in the reference."""
    dt = cfg.time_step
    rpx, rpy, _, _, rrad, gx, gy, _, rtheta = robot
    holo = cfg.kinematics == "holonomic"
    dmin, collision = float("inf"), False
    for h in humans:
        px, py = h[0] - rpx, h[1] - rpy
        if holo:
            vx, vy = h[2] - action[0], h[3] - action[1]
        else:
            vx = h[2] - action[0] * np.cos(action[1] + rtheta)
            vy = h[3] - action[0] * np.sin(action[1] + rtheta)
        ex, ey = px + vx * dt, py + vy * dt
        d = point_to_segment_dist(px, py, ex, ey, 0, 0) - h[4] - rrad
        if d < 0:
            collision = True
            break
        if d < dmin:
            dmin = d
    if holo:
        nx, ny = rpx + action[0] * dt, rpy + action[1] * dt
    else:
        th = rtheta + action[1]
        nx, ny = rpx + np.cos(th) * action[0] * dt, rpy + np.sin(th) * action[0] * dt
    reaching = np.linalg.norm(np.array((nx, ny)) - np.array([gx, gy])) < rrad
    if collision:
        return -0.25
    if reaching:
        return 1
    if dmin < 0.2:
        return (dmin - 0.2) * 0.5 * dt
    return 0


def _tensor_state_scalars(robot_t, humans_t):
    r = robot_t.reshape(-1).numpy()
    h = humans_t.reshape(-1, humans_t.shape[-1]).numpy()
    return [r[i] for i in range(9)], [[row[i] for i in range(5)] for row in h]


# --------------------------------------------------------------------------------------------------
# path M planner -- sequential, reference order
# --------------------------------------------------------------------------------------------------
@dataclass
class MprlParams:
    ve_graph: Dict[str, torch.Tensor]
    value_network: Dict[str, torch.Tensor]
    sp_graph: Optional[Dict[str, torch.Tensor]] = None      # None with the linear predictor
    motion_predictor: Optional[Dict[str, torch.Tensor]] = None

    @staticmethod
    def from_checkpoint(ck):
        """`ck` = the dict the reference's ModelPredictiveRL.get_state_dict() returns."""
        if "graph_model1" in ck:
            return MprlParams(ck["graph_model1"], ck["value_network"], ck["graph_model2"],
                              ck["motion_predictor"])
        if "motion_predictor" in ck:
            return MprlParams(ck["graph_model"], ck["value_network"], ck["graph_model"],
                              ck["motion_predictor"])
        return MprlParams(ck["graph_model"], ck["value_network"])


@dataclass
class SeqTrace:
    """What the sequential walk saw (for fixtures / debugging)."""
    n_value_forwards: int = 0
    n_predictor_forwards: int = 0
    root_clip_values: Optional[np.ndarray] = None       # (A,) one-step values of every root action
    root_clip_rewards: Optional[np.ndarray] = None      # (A,) the rewards inside the root's action_clip (tensor-born reading)
    root_rewards: Optional[np.ndarray] = None           # reward of each kept root action (float64 JointState reading)
    root_clipped: Optional[List[int]] = None            # action indices kept at the root
    root_values: Optional[np.ndarray] = None            # value of each kept root action


def _normalized_gamma(cfg):
    return pow(cfg.gamma, cfg.time_step * cfg.v_pref)


class SeqPlanner:
    """The reference's recursive planner, method by method (model_predictive_rl.py:242-302), over tensor states
    (robot (1,1,9), humans (1,H,5)): `clip` = action_clip, `plan` = V_planning (value only), `reward_of` = estimate_reward on a
    tensor-born state (or on the float64 root rows given to the constructor).  mprl_predict_sequential walks it from a root;
    the GPU tests hold ModelPredictiveRL.action_clip / V_planning / estimate_reward against it."""

    def __init__(self, P: MprlParams, cfg: OracleConfig, v_pref=1.0, root_rows=None, trace: SeqTrace = None):
        self.P, self.cfg = P, cfg
        self.actions, self.groups = mprl_action_space(cfg, v_pref)
        self.action_objs = [action_object(self.actions, i) for i in range(len(self.actions))]
        self.gamma = _normalized_gamma(cfg)
        self.root_rows = root_rows                  # (robot9 floats, human rows): the JointState the root tensors came from
        self.trace = trace if trace is not None else SeqTrace()

    def V(self, state):
        self.trace.n_value_forwards += 1
        return value_estimator_forward(state[0], state[1], self.P.ve_graph, self.P.value_network, self.cfg).reshape(())

    def SP(self, state, a):
        self.trace.n_predictor_forwards += 1
        nr = next_robot_state(state[0].reshape(-1), a, self.cfg).reshape(1, 1, 9)
        if self.cfg.linear_state_predictor:
            nh = linear_humans(state[1])
        else:
            nh = state_predictor_humans(state[0], state[1], self.P.sp_graph, self.P.motion_predictor, self.cfg)
        return (nr, nh)

    def reward_of(self, state, a, root):
        """estimate_reward (:304-357) of a planner state.  `root` = the state is the root JointState itself (python floats: the
        final loop of predict, :226); everything action_clip / V_planning see -- the root's own action_clip included, which is
        handed `state.to_tensor(...)` (:216-218) -- is a tensor and goes through tensor_to_joint_state (state.py:82-92):
        float32-born numpy scalars."""
        if root:
            return estimate_reward(self.root_rows[0], self.root_rows[1], a, self.cfg)
        r, h = _tensor_state_scalars(state[0], state[1])
        return estimate_reward(r, h, a, self.cfg)

    def clip(self, state, width, rewards_out=None):
        """action_clip (:242-269) -> (kept action indices in the reference's order, one-step values of every action).
        The state is always a tensor state here (upstream never calls action_clip with anything else)."""
        vals = []
        for a, ao in zip(self.actions, self.action_objs):
            nxt = self.SP(state, a)
            ret = self.V(nxt)
            rew = self.reward_of(state, ao, False)
            if rewards_out is not None:
                rewards_out.append(float(rew))
            vals.append(rew + self.gamma * ret)
        vals_np = np.array([float(v) for v in vals], dtype=np.float32)
        if self.cfg.sparse_search:
            seen, keep = set(), []
            for idx in np.argsort(vals_np)[::-1]:
                if self.groups[idx] not in seen:
                    keep.append(int(idx))
                    seen.add(self.groups[idx])
                    if len(keep) == width:
                        break
        else:
            keep = [int(i) for i in np.argpartition(vals_np, -width)[-width:]]
        return keep, vals_np

    def plan(self, state, depth, width):
        """V_planning (:271-302), the value only."""
        v = self.V(state)
        if depth == 1:
            return v
        keep = self.clip(state, width)[0] if self.cfg.do_action_clip else list(range(len(self.actions)))
        rets = []
        for ai in keep:
            a = self.actions[ai]
            nxt = self.SP(state, a)
            r = self.reward_of(state, self.action_objs[ai], False)
            nv = self.plan(nxt, depth - 1, width)
            rets.append(v / depth + (depth - 1) / depth * (self.gamma * nv + r))
        return rets[int(np.argmax([float(x) for x in rets]))]


def mprl_predict_sequential(robot9, humans, P: MprlParams, cfg: OracleConfig, trace: SeqTrace = None):
    """robot9: 9 python floats; humans: list of 5-float rows (a root JointState's contents).
    Returns (action index, max value float32).  Walks the tree exactly as the reference does."""
    trace = trace if trace is not None else SeqTrace()
    sp = SeqPlanner(P, cfg, robot9[7], (robot9, humans), trace)
    actions, gamma = sp.actions, sp.gamma
    root = (torch.tensor([[robot9]], dtype=torch.float32),
            torch.tensor([humans], dtype=torch.float32).reshape(1, len(humans), 5))
    if cfg.do_action_clip:
        # the root's clipping runs on the float32 TENSOR of the state (:216-218), the values of the kept actions below on the
        # float64 JointState (:226): the same action is priced with two roundings
        rewards = []
        keep, vals_np = sp.clip(root, cfg.planning_width, rewards)
        trace.root_clip_values = vals_np
        trace.root_clip_rewards = np.array(rewards, dtype=np.float64)
    else:
        keep = list(range(len(actions)))
    trace.root_clipped = keep
    best, best_v, root_vals, root_rews = None, float("-inf"), [], []
    for ai in keep:
        a = actions[ai]
        nxt = sp.SP(root, a)
        ret = sp.plan(nxt, cfg.planning_depth, cfg.planning_width)
        rew = sp.reward_of(root, sp.action_objs[ai], True)
        val = rew + gamma * ret
        root_vals.append(float(val))
        root_rews.append(float(rew))
        if val > best_v:
            best_v, best = val, ai
    trace.root_values = np.array(root_vals, dtype=np.float32)
    trace.root_rewards = np.array(root_rews, dtype=np.float64)
    return best, np.float32(best_v)


# --------------------------------------------------------------------------------------------------
# path M planner -- batched, level synchronous (same arithmetic, B roots at once)
# --------------------------------------------------------------------------------------------------
def estimate_reward_batched(robot, humans, actions, cfg, root):
    """robot (P,9), humans (P,H,5) (fp32 tensors, or float64 arrays = genuine float64 JointStates with `root`), actions (A,2)
    float64 -> rewards (P,A) float64.
    Vectorised transcription of `estimate_reward`; `root` selects float64 differences (JointState
    inputs) versus the float32 differences the reference gets from tensor-born states."""
    dt = cfg.time_step
    r = robot.numpy() if torch.is_tensor(robot) else np.asarray(robot)
    h = humans.numpy() if torch.is_tensor(humans) else np.asarray(humans)
    if root:
        r = r.astype(np.float64)
        h = h.astype(np.float64)
    px = (h[:, :, 0] - r[:, None, 0])[:, None, :]                    # (P,1,H)
    py = (h[:, :, 1] - r[:, None, 1])[:, None, :]
    holo = cfg.kinematics == "holonomic"
    if holo:
        avx, avy = actions[None, :, None, 0], actions[None, :, None, 1]
    else:
        th = actions[None, :, None, 1] + r[:, None, None, 8].astype(np.float64)
        avx, avy = actions[None, :, None, 0] * np.cos(th), actions[None, :, None, 0] * np.sin(th)
    vx = h[:, None, :, 2].astype(np.float64) - avx                      # (P,A,H)
    vy = h[:, None, :, 3].astype(np.float64) - avy
    pxd, pyd = px.astype(np.float64), py.astype(np.float64)
    ex, ey = pxd + vx * dt, pyd + vy * dt
    sx, sy = ex - pxd, ey - pyd
    den = sx * sx + sy * sy
    degenerate = (sx == 0) & (sy == 0)
    u = np.where(degenerate, 0.0, (-pxd * sx - pyd * sy) / np.where(degenerate, 1.0, den))
    u = np.clip(u, 0.0, 1.0)
    cx, cy = pxd + u * sx, pyd + u * sy
    dist = np.sqrt(cx * cx + cy * cy)
    closest = dist - h[:, None, :, 4] - r[:, None, None, 4]                # (P,A,H)
    collided = closest < 0
    # first-collision break: dmin only matters when no human collides at all
    collision = collided.any(axis=2)
    dmin = closest.min(axis=2)
    if holo:
        nx = r[:, None, 0] + actions[None, :, 0] * dt
        ny = r[:, None, 1] + actions[None, :, 1] * dt
    else:
        th = r[:, None, 8].astype(np.float64) + actions[None, :, 1]
        nx = r[:, None, 0] + np.cos(th) * actions[None, :, 0] * dt
        ny = r[:, None, 1] + np.sin(th) * actions[None, :, 0] * dt
    gd = np.sqrt((nx - r[:, None, 5]) ** 2 + (ny - r[:, None, 6]) ** 2)
    reaching = gd < r[:, None, 4]
    rew = np.where(dmin < 0.2, (dmin - 0.2) * 0.5 * dt, 0.0)
    rew = np.where(reaching, 1.0, rew)
    rew = np.where(collision, -0.25, rew)
    if not root and r.dtype == np.float32:
        for ai in range(actions.shape[0]):
            if action_object(actions, ai) == (0, 0):
                rew[:, ai] = _stop_reward_f32(r, h, dt)
    return rew


def _stop_reward_f32(r, h, dt):
    """estimate_reward of tensor-born states (P,9) / (P,H,5) float32 under the python-int stop action: float32 from end to end
    (see `action_object`); array form of the scalar code, one float32 operation per numpy scalar operation."""
    f = np.float32
    assert r.dtype == np.float32 and h.dtype == np.float32
    px, py = h[:, :, 0] - r[:, None, 0], h[:, :, 1] - r[:, None, 1]
    ex, ey = px + h[:, :, 2] * f(dt), py + h[:, :, 3] * f(dt)
    sx, sy = ex - px, ey - py
    degenerate = (sx == 0) & (sy == 0)
    u = ((f(0) - px) * sx + (f(0) - py) * sy) / np.where(degenerate, f(1), sx * sx + sy * sy)
    u = np.clip(np.where(degenerate, f(0), u), f(0), f(1))
    cx, cy = px + u * sx, py + u * sy
    closest = np.sqrt(cx * cx + cy * cy) - h[:, :, 4] - r[:, None, 4]
    assert closest.dtype == np.float32
    collision = (closest < 0).any(axis=1)
    dmin = closest.min(axis=1)
    gx, gy = r[:, 0] - r[:, 5], r[:, 1] - r[:, 6]
    reaching = np.sqrt(gx * gx + gy * gy) < r[:, 4]
    rew = np.where(dmin < f(0.2), (dmin - f(0.2)) * f(0.5) * f(dt), f(0))
    assert rew.dtype == np.float32
    rew = np.where(reaching, f(1), rew)
    return np.where(collision, f(-0.25), rew).astype(np.float64)


def _children_robot(robot, actions, cfg):
    """robot (P,9) -> (P,A,9) next robot states with the reference's fp32 update order."""
    P, A = robot.shape[0], actions.shape[0]
    out = robot[:, None, :].repeat(1, A, 1)
    dt = cfg.time_step
    if cfg.kinematics == "holonomic":
        dvx = torch.tensor(actions[:, 0] * dt, dtype=torch.float64).to(torch.float32)
        dvy = torch.tensor(actions[:, 1] * dt, dtype=torch.float64).to(torch.float32)
        out[:, :, 0] = out[:, :, 0] + dvx[None, :]
        out[:, :, 1] = out[:, :, 1] + dvy[None, :]
        out[:, :, 2] = torch.tensor(actions[:, 0]).to(torch.float32)[None, :]
        out[:, :, 3] = torch.tensor(actions[:, 1]).to(torch.float32)[None, :]
    else:
        for ai in range(A):
            out[:, ai, :] = next_robot_state(robot, actions[ai], cfg)
    return out


def select_top(values, width, groups, sparse):
    """values (P,A) fp32 numpy -> (P,width) kept indices, ordered by descending value
    (ties: lower index first; sparse mode reproduces the reference's reversed-argsort walk)."""
    P, A = values.shape
    keep = np.zeros((P, width), dtype=np.int64)
    for p in range(P):
        if sparse:
            seen, k = set(), []
            for idx in np.argsort(values[p])[::-1]:
                if groups[idx] not in seen:
                    k.append(idx)
                    seen.add(groups[idx])
                    if len(k) == width:
                        break
            keep[p, :len(k)] = k
        else:
            order = np.lexsort((np.arange(A), -values[p].astype(np.float64)))
            keep[p] = order[:width]
    return keep


def mprl_expand_batched(robot, humans, P: MprlParams, cfg, actions, root, roots64=None, root_clip=False):
    """One tree level for a batch of parents.
    robot (Pn,9), humans (Pn,H,5) -> dict(next_humans (Pn,H,5), child_robot (Pn,A,9),
    child_value (Pn,A) fp32 = V(child), reward (Pn,A) fp32, value1 (Pn,A) fp32 = r + g*V).
    `root`: the parents are JointStates (float64 reading; `roots64` = the (robot, humans) float64 arrays the fp32 rows were
    rounded from, when they are not float32-representable).  `root_clip`: the level is the root of a clipped search -- the
    one-step values the selection runs on use the TENSOR-BORN reading of the same rows (`reward_clip`), as upstream's root
    action_clip does (model_predictive_rl.py:216-218,246-248); `reward` keeps the float64 reading for the root values (:226)."""
    Pn, A = robot.shape[0], actions.shape[0]
    if cfg.linear_state_predictor:
        nh = linear_humans(humans)
    else:
        nh = state_predictor_humans(robot[:, None, :], humans, P.sp_graph, P.motion_predictor, cfg)
    cr = _children_robot(robot, actions, cfg)
    H = humans.shape[1]
    cv = value_estimator_forward(cr.reshape(Pn * A, 1, 9),
                                 nh[:, None].expand(Pn, A, H, 5).reshape(Pn * A, H, 5),
                                 P.ve_graph, P.value_network, cfg).reshape(Pn, A)
    if root and roots64 is not None:
        rew = torch.tensor(estimate_reward_batched(roots64[0], roots64[1], actions, cfg, True)).to(torch.float32)
    else:
        rew = torch.tensor(estimate_reward_batched(robot, humans, actions, cfg, root)).to(torch.float32)
    gamma = _normalized_gamma(cfg)
    out = dict(next_humans=nh, child_robot=cr, child_value=cv, reward=rew)
    sel = rew
    if root and root_clip:
        sel = torch.tensor(estimate_reward_batched(robot, humans, actions, cfg, False)).to(torch.float32)
        out["reward_clip"] = sel
    out["value1"] = sel + gamma * cv
    return out


def mprl_predict_batched(robot, humans, P: MprlParams, cfg: OracleConfig, return_levels=False, roots64=None,
                         roots_are_joint_states=True):
    """robot (B,9), humans (B,H,5) fp32 tensors -> (best action (B,) int64, best value (B,) fp32,
    root_values (B,W0) fp32, root_kept (B,W0) int64).  W0 = width if clipping else |A|.
    The roots are JointStates (what predict() is given) unless `roots_are_joint_states` is False; `roots64` = (robot (B,9),
    humans (B,H,5)) float64 arrays when the JointStates are not float32-representable (the fp32 tensors are their roundings)."""
    actions, groups = mprl_action_space(cfg, cfg.v_pref)
    A = actions.shape[0]
    gamma = _normalized_gamma(cfg)
    D = cfg.planning_depth
    w = cfg.planning_width if cfg.do_action_clip else A
    B = robot.shape[0]
    levels = []
    pr, ph = robot, humans
    for lvl in range(D):
        ex = mprl_expand_batched(pr, ph, P, cfg, actions, root=(lvl == 0 and roots_are_joint_states), roots64=roots64,
                                 root_clip=bool(cfg.do_action_clip))
        if cfg.do_action_clip:
            keep = select_top(ex["value1"].numpy(), w, groups, cfg.sparse_search)
        else:
            keep = np.tile(np.arange(A), (pr.shape[0], 1))
        ex["keep"] = torch.tensor(keep)
        levels.append(ex)
        if lvl + 1 < D:
            kt = ex["keep"]
            pr = torch.gather(ex["child_robot"], 1, kt[:, :, None].expand(-1, -1, 9)).reshape(-1, 9)
            ph = ex["next_humans"][:, None].expand(-1, w, -1, -1).reshape(-1, ph.shape[1], 5)
    # back-up, deepest level first.  ret(level l parent-slot) for d = D - l ... see V_planning
    kt = levels[D - 1]["keep"]
    nv = torch.gather(levels[D - 1]["child_value"], 1, kt)            # V_planning(child, 1) = V(child)
    for lvl in range(D - 1, 0, -1):
        d = D - lvl + 1                                               # depth argument at this level
        ex = levels[lvl]
        kt = ex["keep"]
        r = torch.gather(ex["reward"], 1, kt)
        v_parent = torch.gather(levels[lvl - 1]["child_value"], 1, levels[lvl - 1]["keep"]).reshape(-1, 1)
        ret = v_parent / d + (d - 1) / d * (gamma * nv + r)
        nv = ret.max(dim=1).values.reshape(-1, w)                     # becomes V_planning(parent, d)
    r0 = torch.gather(levels[0]["reward"], 1, levels[0]["keep"])
    root_vals = r0 + gamma * nv
    # first strict maximum in kept order
    best_slot = torch.tensor(np.argmax(root_vals.numpy(), axis=1))
    best_a = torch.gather(levels[0]["keep"], 1, best_slot[:, None]).reshape(-1)
    best_v = torch.gather(root_vals, 1, best_slot[:, None]).reshape(-1)
    out = (best_a, best_v, root_vals, levels[0]["keep"])
    return out + (levels,) if return_levels else out


# --------------------------------------------------------------------------------------------------
# path G: pairwise rotated features + ValueNetwork + one-step search
# --------------------------------------------------------------------------------------------------
def rotate_pairwise(state14, kinematics="holonomic"):
    """(R,14) [robot 9 | human 5] -> (R,13) agent-centric relation features."""
    s = state14
    dx, dy = s[:, 5] - s[:, 0], s[:, 6] - s[:, 1]
    rot = torch.atan2(dy, dx)
    c, sn = torch.cos(rot), torch.sin(rot)
    dg = torch.norm(torch.stack([dx, dy], dim=1), 2, dim=1)
    vx = s[:, 2] * c + s[:, 3] * sn
    vy = s[:, 3] * c - s[:, 2] * sn
    theta = (s[:, 8] - rot) if kinematics == "unicycle" else torch.zeros_like(dg)
    vx1 = s[:, 11] * c + s[:, 12] * sn
    vy1 = s[:, 12] * c - s[:, 11] * sn
    px1 = (s[:, 9] - s[:, 0]) * c + (s[:, 10] - s[:, 1]) * sn
    py1 = (s[:, 10] - s[:, 1]) * c - (s[:, 9] - s[:, 0]) * sn
    da = torch.norm(torch.stack([s[:, 0] - s[:, 9], s[:, 1] - s[:, 10]], dim=1), 2, dim=1)
    return torch.stack([dg, s[:, 7], theta, s[:, 4], vx, vy, px1, py1, vx1, vy1, s[:, 13], da,
                        s[:, 4] + s[:, 13]], dim=1)


def gcn_value_forward(state13, sd, cfg, self_dim=6):
    """(B,H,13) -> (value (B,1), A (B,N,N)).  Only num_layer in {1,2} exists on path G."""
    xr = mlp_forward(state13[:, 0, :self_dim], mlp_layers(sd, "w_r."), last_relu=True)
    xh = mlp_forward(state13[:, :, self_dim:], mlp_layers(sd, "w_h."), last_relu=True)
    X = torch.cat([xr.unsqueeze(1), xh], dim=1)
    if cfg.num_layer == 1:
        A = similarity_matrix(X, sd, cfg.similarity)
        feat = torch.relu(torch.matmul(torch.matmul(A, X), sd["w1"]))[:, 0, :]   # no skip with one layer
    elif cfg.num_layer == 2:
        A = similarity_matrix(X, sd, cfg.similarity)
        h1 = torch.relu(torch.matmul(torch.matmul(A, X), sd["w1"]))
        if cfg.skip_connection:
            h1 = h1 + X
        A2 = similarity_matrix(h1, sd, cfg.similarity) if cfg.layerwise_graph else A
        h2 = torch.relu(torch.matmul(torch.matmul(A2, h1), sd["w2"]))
        if cfg.skip_connection:
            h2 = h2 + h1
        feat = h2[:, 0, :]
    else:
        raise NotImplementedError
    return mlp_forward(feat, mlp_layers(sd, "value_net."), last_relu=False), A


def compute_reward_g(nav, humans, dt):
    """nav: 9 float64 scalars (propagated robot), humans: rows of 5 float64 (propagated humans)."""
    dmin, collision = float("inf"), False
    for h in humans:
        d = np.linalg.norm((nav[0] - h[0], nav[1] - h[1])) - nav[4] - h[4]
        if d < 0:
            collision = True
            break
        if d < dmin:
            dmin = d
    reaching = np.linalg.norm((nav[0] - nav[5], nav[1] - nav[6])) < nav[4]
    if collision:
        return -0.25
    if reaching:
        return 1
    if dmin < 0.2:
        return (dmin - 0.2) * 0.5 * dt
    return 0


def gcn_predict_sequential(robot9, humans, sd, cfg):
    """One-step lookahead of path G: returns (action index, action_values (A,) float64 list)."""
    actions = cadrl_action_space(cfg, robot9[7])
    dt = cfg.time_step
    gam = pow(cfg.gamma, dt * robot9[7])
    vals = []
    best, best_v = None, float("-inf")
    for ai, a in enumerate(actions):
        if cfg.kinematics == "holonomic":
            nr = [robot9[0] + a[0] * dt, robot9[1] + a[1] * dt, a[0], a[1]] + list(robot9[4:9])
        else:
            th = robot9[8] + a[1]
            nvx, nvy = a[0] * np.cos(th), a[0] * np.sin(th)
            nr = [robot9[0] + nvx * dt, robot9[1] + nvy * dt, nvx, nvy] + list(robot9[4:8]) + [th]
        nh = [[h[0] + h[2] * dt, h[1] + h[3] * dt, h[2], h[3], h[4]] for h in humans]
        rew = compute_reward_g(nr, nh, dt)
        joint = torch.tensor([list(nr) + list(h) for h in nh], dtype=torch.float32)
        rot = rotate_pairwise(joint, cfg.kinematics).unsqueeze(0)
        v = gcn_value_forward(rot, sd, cfg)[0].item()
        val = rew + gam * v
        vals.append(val)
        if val > best_v:
            best_v, best = val, ai
    return best, vals


def gcn_predict_batched(robot, humans, sd, cfg, chunk=4096):
    """Batched restatement of gcn_predict_sequential (the loop of multi_human_rl.py:36-64 for B root scenes at once).
    robot (B,9), humans (B,H,5): arrays / tensors of the ROOT states (float32 values as the fixtures hold them; all arithmetic
    of propagate / compute_reward in float64 like the python floats of the sequential walk, the joint rows rounded to float32
    exactly where it builds its tensor).  Returns (best action (B,) int64, action_values (B,A) float64).
    Pinned against fixture F7 (`path_g.npz`) in tests/test_oracle_golden.py next to the sequential form."""
    robot = np.asarray(robot, dtype=np.float64)
    humans = np.asarray(humans, dtype=np.float64)
    B, H = humans.shape[0], humans.shape[1]
    dt = cfg.time_step
    # the action table depends on v_pref (robot[7]); scenes are grouped by it (the benchmarks and fixtures use one value)
    out_vals, out_best = None, np.zeros(B, dtype=np.int64)
    for vp in np.unique(robot[:, 7]):
        sel = np.nonzero(robot[:, 7] == vp)[0]
        actions = np.asarray(cadrl_action_space(cfg, float(vp)), dtype=np.float64)              # (A,2)
        A = actions.shape[0]
        r = robot[sel]                                                                          # (b,9)
        nr = np.repeat(r[:, None, :], A, axis=1)                                                # (b,A,9)
        if cfg.kinematics == "holonomic":
            nr[:, :, 0] = r[:, None, 0] + actions[None, :, 0] * dt
            nr[:, :, 1] = r[:, None, 1] + actions[None, :, 1] * dt
            nr[:, :, 2] = actions[None, :, 0]
            nr[:, :, 3] = actions[None, :, 1]
        else:
            th = r[:, None, 8] + actions[None, :, 1]
            nr[:, :, 2] = actions[None, :, 0] * np.cos(th)
            nr[:, :, 3] = actions[None, :, 0] * np.sin(th)
            nr[:, :, 0] = r[:, None, 0] + nr[:, :, 2] * dt
            nr[:, :, 1] = r[:, None, 1] + nr[:, :, 3] * dt
            nr[:, :, 8] = th
        h = humans[sel]
        nh = h.copy()
        nh[:, :, 0] = h[:, :, 0] + h[:, :, 2] * dt
        nh[:, :, 1] = h[:, :, 1] + h[:, :, 3] * dt
        # compute_reward (multi_human_rl.py:73-96): END-point clearance; a collision with ANY human wins (the sequential walk
        # breaks at the first one, which gives the same reward), dmin over all humans otherwise
        dx = nr[:, :, None, 0] - nh[:, None, :, 0]
        dy = nr[:, :, None, 1] - nh[:, None, :, 1]
        clear = np.sqrt(dx * dx + dy * dy) - nr[:, :, None, 4] - nh[:, None, :, 4]               # (b,A,H)
        collision = (clear < 0).any(axis=2)
        dmin = clear.min(axis=2)
        reaching = np.sqrt((nr[:, :, 0] - nr[:, :, 5]) ** 2 + (nr[:, :, 1] - nr[:, :, 6]) ** 2) < nr[:, :, 4]
        rew = np.where(collision, -0.25, np.where(reaching, 1.0, np.where(dmin < 0.2, (dmin - 0.2) * 0.5 * dt, 0.0)))
        joint = np.concatenate([np.repeat(nr[:, :, None, :], H, axis=2),
                                np.repeat(nh[:, None, :, :], A, axis=1)], axis=3).astype(np.float32)   # (b,A,H,14)
        joint = torch.tensor(joint.reshape(-1, 14))
        v = np.zeros(len(sel) * A)
        with torch.no_grad():
            for lo in range(0, len(sel) * A, chunk):
                hi = min(len(sel) * A, lo + chunk)
                rot = rotate_pairwise(joint[lo * H:hi * H], cfg.kinematics).reshape(hi - lo, H, 13)
                v[lo:hi] = gcn_value_forward(rot, sd, cfg)[0][:, 0].double().numpy()
        gam = pow(cfg.gamma, dt * float(vp))
        val = rew + gam * v.reshape(len(sel), A)
        if out_vals is None:
            out_vals = np.zeros((B, A))
        out_vals[sel] = val
        out_best[sel] = np.argmax(val, axis=1)          # first maximum, like the strict '>' walk
    return out_best, out_vals
