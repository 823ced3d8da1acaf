"""Generate the golden fixtures by running the upstream reference itself (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes tests/golden/*.npz.  Fixtures are DATA (inputs, weights and the outputs the reference
produced); no reference source travels.  The GPU box never runs this script -- it only reads the
committed .npz files.

Shims applied to the reference *at run time, in memory* (SURVEY.md §8c), never to its files:
  * everything runs under torch.no_grad() (np.array() of grad-tracking tensors raises),
  * the value estimator's output is reshaped to a 0-d tensor (np.argpartition over (1,1) tensors
    raises on current numpy); values are unchanged.
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HERE_SRC = HERE
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

policy_factory = ref_loader.load_reference()

from crowd_nav.policy.graph_model import RGL  # noqa: E402
from crowd_nav.policy.value_estimator import ValueEstimator  # noqa: E402
from crowd_nav.policy.state_predictor import StatePredictor, LinearStatePredictor  # noqa: E402
from crowd_sim.envs.utils.state import FullState, ObservableState, JointState  # noqa: E402
from crowd_sim.envs.utils.action import ActionXY, ActionRot  # noqa: E402
from crowd_sim.envs.utils.utils import point_to_segment_dist  # noqa: E402

SIMS = ["embedded_gaussian", "gaussian", "cosine", "cosine_softmax", "concatenation", "squared",
        "equal_attention", "diagonal"]


def policy_config(name="mp_separate", **over):
    mod = importlib.import_module("crowd_nav.configs.icra_benchmark." + name)
    pc = mod.PolicyConfig()
    for k, v in over.items():
        sect, key = k.split("__")
        setattr(getattr(pc, sect), key, v)
    return pc


def flat(prefix, sd):
    return {prefix + k: v.detach().cpu().numpy().astype(np.float32) for k, v in sd.items()}


def synth_scene(rng, B, H, moving=True):
    """Seeded crowd states: robot on a radius-4 circle heading to the antipode, humans U(-5,5)^2
    kept >= 0.8 apart from everything (mirrors the clearance rule of the simulator)."""
    robot = np.zeros((B, 9), np.float32)
    humans = np.zeros((B, H, 5), np.float32)
    for b in range(B):
        ang = rng.uniform(0, 2 * np.pi)
        p = 4 * np.array([np.cos(ang), np.sin(ang)]) + rng.uniform(-0.5, 0.5, 2)
        v = rng.uniform(-1, 1, 2) if moving else np.zeros(2)
        sp = np.linalg.norm(v)
        if sp > 1:
            v = v / sp
        g = -4 * np.array([np.cos(ang), np.sin(ang)])
        robot[b] = [p[0], p[1], v[0], v[1], 0.3, g[0], g[1], 1.0, np.pi / 2]
        placed = [p]
        for h in range(H):
            for _ in range(10000):
                q = rng.uniform(-5, 5, 2)
                if all(np.linalg.norm(q - o) >= 0.8 for o in placed):
                    break
            placed.append(q)
            hv = rng.uniform(-1, 1, 2) if moving else np.zeros(2)
            humans[b, h] = [q[0], q[1], hv[0], hv[1], 0.3]
    return robot, humans


def crowd_around_robot(rng, robot, humans, dmin, dmax):
    """Re-place every human at a random bearing, dmin..dmax from its scene's robot (in place)."""
    for b in range(robot.shape[0]):
        for h in range(humans.shape[1]):
            d, ang = rng.uniform(dmin, dmax), rng.uniform(0, 2 * np.pi)
            humans[b, h, 0] = robot[b, 0] + d * np.cos(ang)
            humans[b, h, 1] = robot[b, 1] + d * np.sin(ang)


# --------------------------------------------------------------------------------------------------
# master weights: one random-init set (reference constructors under a torch seed) and a
# "trained-like" set (w_a, Ws scaled by 1/sqrt(32) so the softmax is not one-hot)
# --------------------------------------------------------------------------------------------------
def make_master(seed, scale):
    torch.manual_seed(seed)
    pc = policy_config(gcn__num_layer=3)
    g1 = RGL(pc, 9, 5)
    g2 = RGL(pc, 9, 5)
    ve = ValueEstimator(pc, g1)
    sp = StatePredictor(pc, g2, 0.25)
    pcc = policy_config(gcn__num_layer=3, gcn__similarity_function="concatenation")
    gc = RGL(pcc, 9, 5)
    master = {}
    master.update(flat("graph_model1.", g1.state_dict()))
    master.update(flat("graph_model2.", g2.state_dict()))
    master.update(flat("value_network.", ve.value_network.state_dict()))
    master.update(flat("motion_predictor.", sp.human_motion_predictor.state_dict()))
    master.update({"concat_w_a." + k[len("w_a."):]: v.detach().numpy().astype(np.float32)
                   for k, v in gc.state_dict().items() if k.startswith("w_a.")})
    # a fourth GCN layer for both graphs (round 2: L = 4 exercises the tile kernel's deep mode).  Drawn from its own seed
    # AFTER everything above, so the arrays of the round-1 fixtures stay bit-identical.
    torch.manual_seed(seed + 1000)
    pc4 = policy_config(gcn__num_layer=4)
    master["graph_model1.Ws.3"] = RGL(pc4, 9, 5).state_dict()["Ws.3"].numpy().astype(np.float32)
    master["graph_model2.Ws.3"] = RGL(pc4, 9, 5).state_dict()["Ws.3"].numpy().astype(np.float32)
    if scale != 1.0:
        for k in list(master):
            if k.endswith(".w_a") or ".Ws." in k:
                master[k] = (master[k] * scale).astype(np.float32)
    return master


def graph_sd(master, which, L, similarity):
    pre = which + "."
    sd = {k[len(pre):]: torch.tensor(v) for k, v in master.items()
          if k.startswith(pre) and not k[len(pre):].startswith("Ws.")}
    for l in range(L):
        sd["Ws.%d" % l] = torch.tensor(master[pre + "Ws.%d" % l])
    if similarity == "concatenation":
        sd.pop("w_a")
        for k, v in master.items():
            if k.startswith("concat_w_a."):
                sd["w_a." + k[len("concat_w_a."):]] = torch.tensor(v)
    elif similarity != "embedded_gaussian":
        sd.pop("w_a")
    return sd


def sub_sd(master, which):
    pre = which + "."
    return {k[len(pre):]: torch.tensor(v) for k, v in master.items() if k.startswith(pre)}


def build_ref_modules(master, L=2, similarity="embedded_gaussian", layerwise=False, skip=True, share=False):
    pc = policy_config(gcn__num_layer=L, gcn__similarity_function=similarity,
                       gcn__layerwise_graph=layerwise, gcn__skip_connection=skip)
    g1 = RGL(pc, 9, 5)
    g1.load_state_dict(graph_sd(master, "graph_model1", L, similarity))
    if share:
        g2 = g1
    else:
        g2 = RGL(pc, 9, 5)
        g2.load_state_dict(graph_sd(master, "graph_model2", L, similarity))
    ve = ValueEstimator(pc, g1)
    ve.value_network.load_state_dict(sub_sd(master, "value_network"))
    sp = StatePredictor(pc, g2, 0.25)
    sp.human_motion_predictor.load_state_dict(sub_sd(master, "motion_predictor"))
    return pc, g1, g2, ve, sp


# --------------------------------------------------------------------------------------------------
def gen_forward_kats(masters, out):
    rng = np.random.RandomState(11)
    cases = []
    # default-flag sweeps over crowd size / depth of the GCN / weight flavour
    for H, B, L, flavour in [(1, 3, 2, "rand"), (5, 4, 2, "rand"), (5, 4, 2, "trained"), (19, 3, 2, "trained"),
                             (19, 2, 1, "rand"), (49, 2, 3, "trained"), (19, 2, 3, "rand"), (5, 1, 2, "trained"),
                             (5, 64, 2, "trained")]:
        cases.append(dict(H=H, B=B, L=L, flavour=flavour, sim="embedded_gaussian", layerwise=False, skip=True))
    # every similarity x layerwise x skip on a small crowd
    for sim in SIMS:
        for lw in (False, True):
            for sk in (False, True):
                cases.append(dict(H=5, B=3, L=2, flavour="trained", sim=sim, layerwise=lw, skip=sk))
    meta = []
    for ci, c in enumerate(cases):
        robot, humans = synth_scene(rng, c["B"], c["H"])
        pc, g1, g2, ve, sp = build_ref_modules(masters[c["flavour"]], c["L"], c["sim"], c["layerwise"], c["skip"])
        r = torch.tensor(robot).unsqueeze(1)
        h = torch.tensor(humans)
        with torch.no_grad():
            HL = g1((r, h))
            X = torch.cat([g1.w_r(r), g1.w_h(h)], dim=1)
            A = g1.compute_similarity_matrix(X)
            val = ve((r, h))
            nh = sp((r, h), None)[1]
        k = "f%02d." % ci
        out[k + "robot"], out[k + "humans"] = robot, humans
        out[k + "H_L"], out[k + "A"] = HL.numpy(), A.numpy().astype(np.float32)
        out[k + "value"], out[k + "humans_next"] = val.numpy(), nh.numpy()
        meta.append("%d|%d|%d|%s|%s|%d|%d" % (c["H"], c["B"], c["L"], c["flavour"], c["sim"],
                                              int(c["layerwise"]), int(c["skip"])))
    out["forward_cases"] = np.array(meta)


def gen_forward_kats_l4(masters, out):
    """Forward KATs with four GCN layers (same arrays per case as gen_forward_kats; own file so the round-1 file is unchanged)."""
    rng = np.random.RandomState(21)
    cases = [dict(H=7, B=3, L=4, flavour="trained", sim="embedded_gaussian", layerwise=False, skip=True),
             dict(H=19, B=2, L=4, flavour="rand", sim="embedded_gaussian", layerwise=False, skip=True),
             dict(H=5, B=2, L=4, flavour="trained", sim="embedded_gaussian", layerwise=False, skip=False),
             dict(H=12, B=2, L=4, flavour="trained", sim="gaussian", layerwise=True, skip=True)]
    meta = []
    for ci, c in enumerate(cases):
        robot, humans = synth_scene(rng, c["B"], c["H"])
        pc, g1, g2, ve, sp = build_ref_modules(masters[c["flavour"]], c["L"], c["sim"], c["layerwise"], c["skip"])
        r = torch.tensor(robot).unsqueeze(1)
        h = torch.tensor(humans)
        with torch.no_grad():
            HL = g1((r, h))
            X = torch.cat([g1.w_r(r), g1.w_h(h)], dim=1)
            A = g1.compute_similarity_matrix(X)
            val = ve((r, h))
            nh = sp((r, h), None)[1]
        k = "f%02d." % ci
        out[k + "robot"], out[k + "humans"] = robot, humans
        out[k + "H_L"], out[k + "A"] = HL.numpy(), A.numpy().astype(np.float32)
        out[k + "value"], out[k + "humans_next"] = val.numpy(), nh.numpy()
        meta.append("%d|%d|%d|%s|%s|%d|%d" % (c["H"], c["B"], c["L"], c["flavour"], c["sim"],
                                              int(c["layerwise"]), int(c["skip"])))
    out["forward_cases"] = np.array(meta)


def gen_state_predictor_kats(masters, out):
    rng = np.random.RandomState(12)
    robot, humans = synth_scene(rng, 1, 5)
    pc, g1, g2, ve, sp = build_ref_modules(masters["trained"])
    r = torch.tensor(robot).unsqueeze(1)
    h = torch.tensor(humans)
    acts = [ActionXY(0, 0), ActionXY(np.float64(0.3), np.float64(-0.7)), ActionXY(np.float64(-1.0), np.float64(0.123456789))]
    nr = []
    with torch.no_grad():
        for a in acts:
            o = sp((r, h), a)
            nr.append(o[0].numpy().reshape(9))
        lin = LinearStatePredictor(pc, 0.25)
        lo = lin((r, h), acts[1])
    out["sp.robot"], out["sp.humans"] = robot, humans
    out["sp.actions"] = np.array([[a.vx, a.vy] for a in acts], np.float64)
    out["sp.next_robot"] = np.array(nr, np.float32)
    out["sp.linear_next_robot"] = lo[0].numpy().reshape(9)
    out["sp.linear_next_humans"] = lo[1].numpy()
    # unicycle kinematics (pins the slot-7 quirk)
    pcu = policy_config(action_space__kinematics="unicycle")
    spu = StatePredictor(pcu, g2, 0.25)
    ua = ActionRot(np.float64(0.8), np.float64(0.4))
    with torch.no_grad():
        out["sp.unicycle_next_robot"] = spu.compute_next_state(r, ua).numpy().reshape(9)
    out["sp.unicycle_action"] = np.array([ua.v, ua.r], np.float64)
    # restore the class-level shared config attribute the constructor above mutated
    policy_config(action_space__kinematics="holonomic")


def gen_action_spaces(out):
    pc = policy_config()
    pol = policy_factory["model_predictive_rl"]()
    pol.configure(pc)
    pol.build_action_space(1.0)
    out["act.mprl"] = np.array([[a.vx, a.vy] for a in pol.action_space], np.float64)
    out["act.mprl_groups"] = np.array(pol.action_group_index, np.int64)
    pg = policy_factory["gcn"]()
    pg.configure(policy_config("rgl"))
    pg.build_action_space(1.0)
    out["act.gcn"] = np.array([[a.vx, a.vy] for a in pg.action_space], np.float64)
    pol2 = policy_factory["model_predictive_rl"]()
    pol2.configure(pc)
    pol2.build_action_space(0.7)
    out["act.mprl_vpref07"] = np.array([[a.vx, a.vy] for a in pol2.action_space], np.float64)
    # unicycle tables
    pcu = policy_config(action_space__kinematics="unicycle")
    pu = policy_factory["model_predictive_rl"]()
    pu.configure(pcu)
    pu.build_action_space(1.0)
    out["act.mprl_unicycle"] = np.array([[a.v, a.r] for a in pu.action_space], np.float64)
    policy_config(action_space__kinematics="holonomic")


def gen_reward_kats(out):
    pc = policy_config()
    pol = policy_factory["model_predictive_rl"]()
    pol.configure(pc)
    pol.set_time_step(0.25)
    # hand-built states hitting: free, discomfort, collision, goal, degenerate segment, clamped ends
    R = FullState(0.0, 0.0, 0.0, 0.0, 0.3, 3.0, 0.0, 1.0, 0.0)
    cases = [
        ("free", R, [ObservableState(3.0, 3.0, 0.1, 0.0, 0.3)], ActionXY(0.5, 0.0)),
        ("discomfort", R, [ObservableState(0.75, 0.0, 0.0, 0.0, 0.3)], ActionXY(0.0, 0.5)),
        ("collision", R, [ObservableState(0.7, 0.0, -0.5, 0.0, 0.3)], ActionXY(1.0, 0.0)),
        ("collision_first", R, [ObservableState(0.5, 0.0, 0.0, 0.0, 0.3), ObservableState(0.9, 0.3, 0.0, 0.0, 0.3)], ActionXY(0.0, 0.0)),
        ("goal", FullState(2.8, 0.0, 0.0, 0.0, 0.3, 3.0, 0.0, 1.0, 0.0), [ObservableState(-3.0, 3.0, 0.0, 0.0, 0.3)], ActionXY(0.5, 0.0)),
        ("degenerate", R, [ObservableState(0.72, 0.0, 0.0, 0.0, 0.3)], ActionXY(0.0, 0.0)),
        ("clamp_u1", R, [ObservableState(2.0, 0.0, -1.0, 0.0, 0.3)], ActionXY(1.0, 0.0)),
        ("clamp_u0", R, [ObservableState(0.9, 0.0, 1.0, 0.0, 0.3)], ActionXY(-1.0, 0.0)),
        ("two_humans", R, [ObservableState(0.0, 0.78, 0.0, 0.0, 0.3), ObservableState(-0.74, 0.0, 0.0, 0.0, 0.3)], ActionXY(0.2, 0.0)),
    ]
    names, robots, humans, acts, r_joint, r_tensor, nh = [], [], [], [], [], [], []
    for name, rs, hs, a in cases:
        a = ActionXY(np.float64(a.vx), np.float64(a.vy))
        js = JointState(rs, hs)
        r_joint.append(float(pol.estimate_reward(js, a)))
        ts = js.to_tensor(add_batch_size=True)
        r_tensor.append(float(pol.estimate_reward(ts, a)))
        names.append(name)
        robots.append(rs.to_tuple())
        hh = np.zeros((2, 5), np.float64)
        for i, h_ in enumerate(hs):
            hh[i] = h_.to_tuple()
        humans.append(hh)
        nh.append(len(hs))
        acts.append([a.vx, a.vy])
    out["rew.names"] = np.array(names)
    out["rew.robot"] = np.array(robots, np.float64)
    out["rew.humans"] = np.array(humans, np.float64)
    out["rew.n_humans"] = np.array(nh, np.int64)
    out["rew.actions"] = np.array(acts, np.float64)
    out["rew.joint"] = np.array(r_joint, np.float64)
    out["rew.tensor"] = np.array(r_tensor, np.float64)
    # random sweep (tensor-born states, all 81 actions) -- the bulk check for the batched reward
    rng = np.random.RandomState(13)
    pol.build_action_space(1.0)
    robot, hum = synth_scene(rng, 6, 7)
    for b, lo in enumerate([0.45, 0.62, 0.7, 0.85, 1.0, 1.3]):   # crowd so every branch is populated
        crowd_around_robot(rng, robot[b:b + 1], hum[b:b + 1], lo, lo + 1.5)
    sweep = np.zeros((6, len(pol.action_space)), np.float64)
    sweep_joint = np.zeros_like(sweep)
    for b in range(6):
        ts = (torch.tensor(robot[b]).reshape(1, 1, 9), torch.tensor(hum[b]).reshape(1, 7, 5))
        js = JointState(FullState(*[float(x) for x in robot[b]]),
                        [ObservableState(*[float(x) for x in row]) for row in hum[b]])
        for ai, a in enumerate(pol.action_space):
            sweep[b, ai] = pol.estimate_reward(ts, a)
            sweep_joint[b, ai] = pol.estimate_reward(js, a)
    out["rew.sweep_robot"], out["rew.sweep_humans"] = robot, hum
    out["rew.sweep_tensor"], out["rew.sweep_joint"] = sweep, sweep_joint
    # point_to_segment_dist
    pts = np.array([[0, 0, 1, 0, 0.5, 1], [0, 0, 1, 0, 2, 1], [0, 0, 1, 0, -1, -1], [1, 1, 1, 1, 4, 5],
                    [-1, 2, 3, -2, 0, 0], [0.3, 0.1, 0.3, 0.1, 0, 0]], np.float64)
    out["p2s.in"] = pts
    out["p2s.out"] = np.array([point_to_segment_dist(*p) for p in pts], np.float64)
    # path G reward
    pg = policy_factory["gcn"]()
    pg.configure(policy_config("rgl"))
    pg.time_step = 0.25
    g_out = []
    for name, rs, hs, a in cases:
        g_out.append(float(pg.compute_reward(rs, hs)))
    out["rew.g"] = np.array(g_out, np.float64)


# --------------------------------------------------------------------------------------------------
class PlanRecorder:
    """Wraps a reference ModelPredictiveRL so one predict() call yields the numbers a test needs."""

    def __init__(self, pol):
        self.pol = pol
        self.nest = 0
        self.top_returns = []
        self.clipped = None
        self.nv = 0
        self.nsp = 0
        ve_fwd = pol.value_estimator.forward

        def ve(state):
            self.nv += 1
            return ve_fwd(state).reshape(())
        pol.value_estimator.forward = ve
        if not isinstance(pol.state_predictor, LinearStatePredictor):
            sp_fwd = pol.state_predictor.forward

            def spf(state, action, detach=False):
                self.nsp += 1
                return sp_fwd(state, action, detach)
            pol.state_predictor.forward = spf
        vp = pol.V_planning

        def vplan(state, depth, width):
            self.nest += 1
            try:
                ret = vp(state, depth, width)
            finally:
                self.nest -= 1
            if self.nest == 0:
                self.top_returns.append(float(ret[0]))
            return ret
        pol.V_planning = vplan
        ac = pol.action_clip

        def aclip(state, action_space, width, depth=1):
            res = ac(state, action_space, width, depth)
            if self.nest == 0:
                self.clipped = [self._index(a) for a in res]
            return res
        pol.action_clip = aclip

    def _index(self, a):
        for i, b in enumerate(self.pol.action_space):
            if a is b:
                return i
        raise KeyError

    def run(self, js):
        self.top_returns, self.clipped, self.nv, self.nsp = [], None, 0, 0
        with torch.no_grad():
            action = self.pol.predict(js)
        return self._index(action)


def make_ref_mprl(master, D, w, clip, sparse, variant, L=2):
    name = {"separate": "mp_separate", "shared": "mp_detach", "linear": "mp_linear"}[variant]
    pc = policy_config(name, gcn__num_layer=L, model_predictive_rl__planning_depth=D,
                       model_predictive_rl__planning_width=w, model_predictive_rl__do_action_clip=clip)
    pc.model_predictive_rl.sparse_search = sparse
    pol = policy_factory["model_predictive_rl"]()
    pol.configure(pc)
    pol.value_estimator.graph_model.load_state_dict(graph_sd(master, "graph_model1", L, "embedded_gaussian"))
    pol.value_estimator.value_network.load_state_dict(sub_sd(master, "value_network"))
    if variant == "separate":
        pol.state_predictor.graph_model.load_state_dict(graph_sd(master, "graph_model2", L, "embedded_gaussian"))
    if variant != "linear":
        pol.state_predictor.human_motion_predictor.load_state_dict(sub_sd(master, "motion_predictor"))
    pol.set_time_step(0.25)
    pol.set_phase("test")
    pol.set_device(torch.device("cpu"))
    return pol


def joint_state_of(robot_row, human_rows):
    return JointState(FullState(*[float(x) for x in robot_row]),
                      [ObservableState(*[float(x) for x in row]) for row in human_rows])


def gen_planning_kats(masters, scenes, out):
    rng = np.random.RandomState(14)
    r5, h5 = synth_scene(rng, 3, 5)
    r19, h19 = synth_scene(rng, 2, 19)
    # crowd one scene so that reward branches fire inside the tree
    crowd_around_robot(rng, r5[2:3], h5[2:3], 0.7, 2.5)
    sets = {
        "s5": (r5, h5), "s19": (r19, h19),
        "env": (scenes["test_robot"][:3].astype(np.float32), scenes["test_humans"][:3].astype(np.float32)),
    }
    for k, (r, h) in sets.items():
        out["plan.scene.%s.robot" % k], out["plan.scene.%s.humans" % k] = r, h
    plans = [  # (tag, scene set, D, w, clip, sparse, variant, flavour)
        ("d1", "s5", 1, 1, False, False, "separate", "trained"),
        ("d1env", "env", 1, 1, False, False, "separate", "trained"),
        ("d1rand", "s5", 1, 1, False, False, "separate", "rand"),
        ("d2w1", "s5", 2, 1, True, False, "separate", "trained"),
        ("d2w2", "s5", 2, 2, True, False, "separate", "trained"),
        ("d2w2env", "env", 2, 2, True, False, "separate", "trained"),
        ("d2w2n20", "s19", 2, 2, True, False, "separate", "trained"),
        ("d3w2", "s5", 3, 2, True, False, "separate", "trained"),
        ("d2w10", "s5", 2, 10, True, False, "separate", "trained"),
        ("d2w2sparse", "s5", 2, 2, True, True, "separate", "trained"),
        ("d2w4sparse", "s5", 2, 4, True, True, "separate", "trained"),
        ("d2w2shared", "s5", 2, 2, True, False, "shared", "trained"),
        ("d2w2linear", "s5", 2, 2, True, False, "linear", "trained"),
        ("d1linear", "env", 1, 1, False, False, "linear", "trained"),
    ]
    meta = []
    for tag, sk, D, w, clip, sparse, variant, flavour in plans:
        pol = make_ref_mprl(masters[flavour], D, w, clip, sparse, variant)
        rec = RootClipRecorder(pol)
        r, h = sets[sk]
        gamma = pol.get_normalized_gamma()
        acts, maxv, clipvals, kept, rootvals, counts = [], [], [], [], [], []
        for b in range(r.shape[0]):
            js = joint_state_of(r[b], h[b])
            ai = rec.run(js)
            acts.append(ai)
            nA = len(pol.action_space)
            if clip:
                one_step = rec.top_returns[:nA]
                main = rec.top_returns[nA:]
                keep = rec.clipped
                # the values upstream's root action_clip selected on, as IT computed them (round 5: captured, not recomputed --
                # it prices the float32 tensor of the state there, model_predictive_rl.py:216-218,246-248)
                assert rec.clip_values.shape == (nA,) and len(one_step) == nA
                clipvals.append(rec.clip_values)
            else:
                main = rec.top_returns
                keep = list(range(nA))
            rv = [float(np.float32(np.float32(pol.estimate_reward(js, pol.action_space[i])) +
                                   np.float32(gamma) * np.float32(m))) for i, m in zip(keep, main)]
            kept.append(keep)
            rootvals.append(rv)
            maxv.append(max(rv))
            counts.append([rec.nv, rec.nsp])
        k = "plan.%s." % tag
        out[k + "action"] = np.array(acts, np.int64)
        out[k + "max_value"] = np.array(maxv, np.float32)
        out[k + "kept"] = np.array(kept, np.int64)
        out[k + "root_values"] = np.array(rootvals, np.float32)
        out[k + "counts"] = np.array(counts, np.int64)
        if clip:
            out[k + "clip_values"] = np.array(clipvals, np.float32)
        meta.append("%s|%s|%d|%d|%d|%d|%s|%s" % (tag, sk, D, w, int(clip), int(sparse), variant, flavour))
    out["plan_cases"] = np.array(meta)



# --------------------------------------------------------------------------------------------------
class _NumpyTap:
    """Stands in for the name `np` inside the reference planner's module while a RootClipRecorder runs: every attribute is
    numpy's own, except that the array handed to np.argpartition / np.argsort is noted first -- the `values` list upstream's
    action_clip selects on (model_predictive_rl.py:254,265), which no other hook can see."""

    def __init__(self, sink):
        self._sink = sink

    def __getattr__(self, name):
        return getattr(np, name)

    def argpartition(self, a, *args, **kw):
        self._sink(a)
        return np.argpartition(a, *args, **kw)

    def argsort(self, a, *args, **kw):
        self._sink(a)
        return np.argsort(a, *args, **kw)


class RootClipRecorder(PlanRecorder):
    """PlanRecorder that also keeps what happens INSIDE the root's action_clip of one predict() call: the reward upstream's
    estimate_reward returned for each action there (it is handed the float32 TENSOR of the state, :216-218 -> :246-248), the
    values array the selection ran on, and the rewards / raw returns of the final loop over the kept actions (:222-231, where
    estimate_reward reads the float64 JointState)."""

    def __init__(self, pol):
        super().__init__(pol)
        import crowd_nav.policy.model_predictive_rl as mpr
        self.mpr = mpr
        self.in_root_clip = False
        self.clip_rewards, self.clip_values, self.root_rewards, self.top_raw = [], None, [], []
        er = pol.estimate_reward

        def est(state, action):
            r = er(state, action)
            if self.in_root_clip:
                self.clip_rewards.append(float(r))
            elif self.nest == 0:
                self.root_rewards.append(float(r))
            return r
        pol.estimate_reward = est
        ac = pol.action_clip                     # PlanRecorder's wrapper

        def aclip(state, action_space, width, depth=1):
            if self.nest != 0:
                return ac(state, action_space, width, depth)
            self.in_root_clip = True
            mpr.np = _NumpyTap(self._note_values)
            try:
                return ac(state, action_space, width, depth)
            finally:
                mpr.np = np
                self.in_root_clip = False
        pol.action_clip = aclip
        vp = pol.V_planning                      # PlanRecorder's wrapper

        def vplan(state, depth, width):
            ret = vp(state, depth, width)
            if self.nest == 0:
                self.top_raw.append(ret[0])
            return ret
        pol.V_planning = vplan

    def _note_values(self, a):
        if self.clip_values is None:
            self.clip_values = np.array(a)

    def run(self, js):
        self.clip_rewards, self.clip_values, self.root_rewards, self.top_raw = [], None, [], []
        return super().run(js)


def gen_root_clip_kats(masters, out):
    """what the reference computes at the ROOT of a clipped search, captured from inside its own
    action_clip -- on roots whose coordinates are genuine float64 (not float32-representable) with 3-5 humans 0.65-1.1 m from
    the robot, so that the discomfort / collision branches fire and the float32-born and float64 readings of the same state
    give different rewards."""
    rng = np.random.RandomState(41)
    B, H = 24, 5
    r32, h32 = synth_scene(rng, B, H)
    robot = r32.astype(np.float64)
    humans = h32.astype(np.float64)
    robot[:, :4] += rng.uniform(-1e-3, 1e-3, (B, 4))           # genuine float64 positions / velocities
    robot[:, 5:7] += rng.uniform(-1e-3, 1e-3, (B, 2))
    humans[:, :, :4] += rng.uniform(-1e-3, 1e-3, (B, H, 4))
    for b in range(B):
        near = rng.choice(H, size=rng.randint(3, 6), replace=False)
        for h in near:
            d, ang = rng.uniform(0.65, 1.1), rng.uniform(0, 2 * np.pi)
            humans[b, h, 0] = robot[b, 0] + d * np.cos(ang)
            humans[b, h, 1] = robot[b, 1] + d * np.sin(ang)
    out["rootclip.robot64"], out["rootclip.humans64"] = robot, humans
    plans = [("d2w2", 2, 2, False, "separate"), ("d2w2sparse", 2, 2, True, "separate"), ("d1w3", 1, 3, False, "separate"),
             ("d2w2linear", 2, 2, False, "linear")]
    meta = []
    for tag, D, w, sparse, variant in plans:
        pol = make_ref_mprl(masters["trained"], D, w, True, sparse, variant)
        rec = RootClipRecorder(pol)
        gamma = pol.get_normalized_gamma()
        nA = None
        acts, kept, clip_rewards, clip_values, root_rewards, root_values = [], [], [], [], [], []
        for b in range(B):
            js = JointState(FullState(*[float(x) for x in robot[b]]), [ObservableState(*[float(x) for x in row]) for row in humans[b]])
            ai = rec.run(js)
            nA = len(pol.action_space)
            assert len(rec.clip_rewards) == nA and rec.clip_values is not None and rec.clip_values.shape == (nA,)
            assert len(rec.root_rewards) == w and len(rec.top_raw) == nA + w
            acts.append(ai)
            kept.append(rec.clipped)
            clip_rewards.append(rec.clip_rewards)
            clip_values.append(rec.clip_values)
            root_rewards.append(rec.root_rewards)
            # upstream's own expression on the objects it had (:227): python / numpy float + python float * 0-d float32 tensor
            root_values.append([float(rr + gamma * raw) for rr, raw in zip(rec.root_rewards, rec.top_raw[nA:])])
        k = "rootclip.%s." % tag
        out[k + "action"] = np.array(acts, np.int64)
        out[k + "kept"] = np.array(kept, np.int64)
        out[k + "clip_rewards"] = np.array(clip_rewards, np.float64)
        out[k + "clip_values"] = np.array(clip_values)
        out[k + "root_rewards"] = np.array(root_rewards, np.float64)
        out[k + "root_values"] = np.array(root_values, np.float32)
        meta.append("%s|%d|%d|%d|%s" % (tag, D, w, int(sparse), variant))
    out["rootclip_cases"] = np.array(meta)

def gen_tie_kats(masters, out):
    """EXACT ties in action_clip, as the reference resolves them in this image.  The value head's last
    layer is zeroed (V = its bias for every state) and the crowd is far away (reward 0 for every action but the ones that reach
    nothing): all 81 one-step values are bit-equal.  Upstream's sparse walk follows np.argsort(values)[::-1] and np.argpartition
    picks the dense set -- both implementation-defined on ties (numpy's default sort is not stable).  Recorded so that the
    documented deviation of the device search (ties: lower action index first) is visible in a test against real data."""
    rng = np.random.RandomState(77)
    r32, h32 = synth_scene(rng, 2, 5)
    for b in range(2):                                           # humans far from the robot: no reward term fires
        for h in range(5):
            ang = rng.uniform(0, 2 * np.pi)
            h32[b, h, 0] = r32[b, 0] + 3.5 * np.cos(ang) + 0.4 * h
            h32[b, h, 1] = r32[b, 1] + 3.5 * np.sin(ang)
    out["tie.robot"], out["tie.humans"] = r32, h32
    meta = []
    for tag, w, sparse in (("w2sparse", 2, True), ("w4sparse", 4, True), ("w3dense", 3, False)):
        pol = make_ref_mprl(masters["trained"], 2, w, True, sparse, "separate")
        with torch.no_grad():
            last = pol.value_estimator.value_network[-1]
            last.weight.zero_()
        rec = RootClipRecorder(pol)
        kept, vals, acts = [], [], []
        for b in range(2):
            acts.append(rec.run(joint_state_of(r32[b], h32[b])))
            kept.append(rec.clipped)
            vals.append(rec.clip_values)
        vals = np.array(vals)
        assert (vals == vals[:, :1]).all(), "the construction must tie every action exactly"
        out["tie.%s.kept" % tag] = np.array(kept, np.int64)
        out["tie.%s.clip_values" % tag] = vals
        out["tie.%s.action" % tag] = np.array(acts, np.int64)
        meta.append("%s|%d|%d" % (tag, w, int(sparse)))
    out["tie_cases"] = np.array(meta)
    out["tie.groups"] = np.array(pol.action_group_index, np.int64)


# --------------------------------------------------------------------------------------------------
def gen_path_g(out, scenes):
    torch.manual_seed(5)
    pc = policy_config("rgl")
    pol = policy_factory["gcn"]()
    pol.configure(pc)
    sd = pol.model.state_dict()
    with torch.no_grad():
        sd["w_a"].mul_(1 / np.sqrt(32))
        sd["w1"].mul_(1 / np.sqrt(32))
        sd["w2"].mul_(1 / np.sqrt(32))
    out.update(flat("g.weights.", sd))
    pol.set_phase("test")
    pol.set_device(torch.device("cpu"))
    pol.time_step = 0.25
    # rotate KAT
    rng = np.random.RandomState(15)
    s14 = rng.uniform(-3, 3, (7, 14)).astype(np.float32)
    s14[:, 4] = 0.3
    s14[:, 13] = 0.35
    out["g.rotate_in"] = s14
    out["g.rotate_out"] = pol.rotate(torch.tensor(s14)).numpy()
    pol.kinematics = "unicycle"
    out["g.rotate_out_unicycle"] = pol.rotate(torch.tensor(s14)).numpy()
    pol.kinematics = "holonomic"
    # ValueNetwork KATs: L=2 (skip, non-layerwise = shipped config), plus flag variants and L=1
    x13 = rng.uniform(-2, 2, (4, 5, 13)).astype(np.float32)
    out["g.vn_in"] = x13
    variants = [(2, False, True), (2, True, True), (2, False, False), (2, True, False), (1, False, True)]
    meta = []
    for L, lw, sk in variants:
        pcv = policy_config("rgl", gcn__num_layer=L, gcn__layerwise_graph=lw, gcn__skip_connection=sk)
        pv = policy_factory["gcn"]()
        pv.configure(pcv)
        msd = {k: v for k, v in sd.items() if not (L == 1 and k == "w2")}
        pv.model.load_state_dict(msd)
        with torch.no_grad():
            v = pv.model(torch.tensor(x13))
        tag = "L%d_lw%d_sk%d" % (L, int(lw), int(sk))
        out["g.vn_value." + tag] = v.numpy()
        out["g.vn_A0." + tag] = np.asarray(pv.model.A, np.float32)
        meta.append(tag)
    out["g_cases"] = np.array(meta)
    policy_config("rgl", gcn__num_layer=2, gcn__layerwise_graph=False, gcn__skip_connection=True)
    # predict on env scenes + a moving synthetic scene
    rs, hs = scenes["test_robot"][:3], scenes["test_humans"][:3]
    r2, h2 = synth_scene(np.random.RandomState(16), 2, 5)
    rs = np.concatenate([rs, r2.astype(np.float64)])
    hs = np.concatenate([hs, h2.astype(np.float64)])
    acts, avals, A0 = [], [], []
    for b in range(rs.shape[0]):
        js = joint_state_of(rs[b], hs[b])
        with torch.no_grad():
            a = pol.predict(js)
        acts.append([i for i, x in enumerate(pol.action_space) if x is a][0])
        avals.append(pol.action_values)
        A0.append(np.asarray(pol.get_matrix_A(), np.float32))
    out["g.pred_robot"], out["g.pred_humans"] = rs, hs
    out["g.pred_action"] = np.array(acts, np.int64)
    out["g.pred_action_values"] = np.array(avals, np.float64)
    out["g.pred_A_last"] = np.array(A0)


def gen_env_scenes():
    """Initial JointStates of the simulator's seeded cases (pure numpy scene generation)."""
    import gym
    from crowd_sim.envs.utils.robot import Robot
    # scene generation instantiates human policies; ORCA's constructor needs nothing from rvo2
    envc = importlib.import_module("crowd_nav.configs.icra_benchmark.mp_separate").EnvConfig()
    env = gym.make("CrowdSim-v0")
    env.configure(envc)
    robot = Robot(envc, "robot")
    robot.time_step = env.time_step
    pol = policy_factory["model_predictive_rl"]()
    pol.configure(policy_config())
    robot.set_policy(pol)
    env.set_robot(robot)
    scenes = {}
    for phase, n in (("test", 10), ("val", 4)):
        R, Hs = [], []
        for k in range(n):
            ob = env.reset(phase, k)
            R.append(robot.get_full_state().to_tuple())
            Hs.append([o.to_tuple() for o in ob])
        scenes[phase + "_robot"] = np.array(R, np.float64)
        scenes[phase + "_humans"] = np.array(Hs, np.float64)
    return scenes


INFO_CODES = {"": 0, "Discomfort": 1, "Collision": 2, "Reaching goal": 3, "Timeout": 4}


def gen_sim_kats(out):
    """Trajectories of the reference simulator's step() with its `linear` human policy (rvo2-free) under three
    scripted robot behaviours.  In-memory shim: the reference's Linear.predict reads `state.self_state`, an attribute
    JointState no longer has (bit-rot); it is aliased to `robot_state` here, nothing on disk changes."""
    import gym
    from crowd_sim.envs.utils.robot import Robot
    JointState.self_state = property(lambda self_: self_.robot_state)
    mod = importlib.import_module("crowd_nav.configs.icra_benchmark.mp_separate")
    envc = mod.EnvConfig()
    old_policy, old_central = envc.humans.policy, envc.sim.centralized_planning
    envc.humans.policy = "linear"
    envc.sim.centralized_planning = False
    try:
        env = gym.make("CrowdSim-v0")
        env.configure(envc)
        robot = Robot(envc, "robot")
        robot.time_step = env.time_step
        pol = policy_factory["model_predictive_rl"]()
        pol.configure(policy_config())
        robot.set_policy(pol)
        env.set_robot(robot)
        pol.build_action_space(1.0)
        table = np.array([[a.vx, a.vy] for a in pol.action_space], np.float64)
        meta = []
        for script, cases in (("greedy", (0, 1, 2)), ("late", (0, 1, 2)), ("random", (0,)), ("stop", (1,))):
            for case in cases:
                env.reset("test", case)
                rng = np.random.RandomState(100 + case)
                R = [robot.get_full_state().to_tuple()]
                Hs = [[h.get_full_state().to_tuple() for h in env.humans]]
                acts, rew, done_l, info_l, dmin_l, times = [], [], [], [], [], [env.global_time]
                for t in range(140):
                    if script == "greedy" or (script == "late" and t >= 44):
                        goal = np.array([robot.gx - robot.px, robot.gy - robot.py])
                        ai = int(np.argmax(table @ goal))
                    elif script == "random":
                        ai = int(rng.randint(0, len(table)))
                    else:
                        ai = 0
                    a = pol.action_space[ai]
                    a = ActionXY(np.float64(a.vx), np.float64(a.vy))
                    _, reward, done, info = env.step(a)
                    acts.append(ai)
                    rew.append(float(reward))
                    done_l.append(int(done))
                    info_l.append(INFO_CODES[str(info)])
                    dmin_l.append(float(info.min_dist) if str(info) == "Discomfort" else np.nan)
                    R.append(robot.get_full_state().to_tuple())
                    Hs.append([h.get_full_state().to_tuple() for h in env.humans])
                    times.append(env.global_time)
                    if done:
                        break
                k = "sim.%s%d." % (script, case)
                out[k + "robot"] = np.array(R, np.float64)
                out[k + "humans"] = np.array(Hs, np.float64)
                out[k + "actions"] = np.array(acts, np.int64)
                out[k + "reward"] = np.array(rew, np.float64)
                out[k + "done"] = np.array(done_l, np.int64)
                out[k + "info"] = np.array(info_l, np.int64)
                out[k + "dmin"] = np.array(dmin_l, np.float64)
                out[k + "time"] = np.array(times, np.float64)
                meta.append("%s%d|%d" % (script, case, case))
        out["sim_cases"] = np.array(meta)
        out["sim.action_table"] = table
    finally:
        envc.humans.policy, envc.sim.centralized_planning = old_policy, old_central
        del JointState.self_state


def make_goal_master(base):
    """A weight set whose value estimator prefers states closer to the goal (so that episodes end in goals and collisions,
    not only in time-outs): the reference ValueEstimator (trained-like start) regressed for 500 Adam steps onto
    -0.25 * |position - goal| on seeded random states.  Single-threaded CPU training: reproducible bit for bit here."""
    # (The regression runs on the repo's functional restatement of the forward, oracle/rgl_oracle.py: the reference's own
    # RGL.forward adds the skip connection in place, which current torch autograd rejects.  The result is just a weight set;
    # everything recorded with it below is produced by the reference.)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE_SRC)))
    from oracle import rgl_oracle as orc
    torch.manual_seed(7)
    rng = np.random.RandomState(7)
    gsd = {k: v.clone().requires_grad_(True) for k, v in graph_sd(base, "graph_model1", 2, "embedded_gaussian").items()}
    vsd = {k: v.clone().requires_grad_(True) for k, v in sub_sd(base, "value_network").items()}
    opt = torch.optim.Adam(list(gsd.values()) + list(vsd.values()), lr=2e-3)
    cfg = orc.OracleConfig()
    for it in range(500):
        B = 256
        robot = np.zeros((B, 9), np.float32)
        robot[:, 0:2] = rng.uniform(-2.2, 2.2, (B, 2))
        robot[:, 2:4] = rng.uniform(-1, 1, (B, 2)) * rng.uniform(0, 1, (B, 1))
        ang = rng.uniform(0, 2 * np.pi, B)
        robot[:, 4], robot[:, 5], robot[:, 6], robot[:, 7], robot[:, 8] = 0.3, 1.5 * np.cos(ang), 1.5 * np.sin(ang), 1.0, np.pi / 2
        humans = np.zeros((B, 5, 5), np.float32)
        humans[:, :, 0:2] = rng.uniform(-2.2, 2.2, (B, 5, 2))
        humans[:, :, 2:4] = rng.uniform(-1, 1, (B, 5, 2))
        humans[:, :, 4] = 0.3
        target = torch.tensor(-0.25 * np.linalg.norm(robot[:, 0:2] - robot[:, 5:7], axis=1).astype(np.float32)).reshape(B, 1)
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(orc.value_estimator_forward(torch.tensor(robot).unsqueeze(1), torch.tensor(humans),
                                                                        gsd, vsd, cfg), target)
        loss.backward()
        opt.step()
    m = dict(base)
    m.update(flat("graph_model1.", gsd))
    m.update(flat("value_network.", vsd))
    m["goal.final_loss"] = np.array(float(loss.detach()), np.float64)
    return m


def gen_explorer_kats(masters, out):
    """The reference Explorer (crowd_nav/utils/explorer.py:21-140) driving the reference simulator with its `linear` humans and
    the reference ModelPredictiveRL robot (trained-like weights, depth 1): statistics, per-episode outcomes, every decision, and
    the experience tuples it pushes into the reference ReplayMemory.  Same in-memory JointState.self_state alias as gen_sim_kats."""
    import gym
    from crowd_sim.envs.utils.robot import Robot
    from crowd_sim.envs.utils.info import ReachGoal, Collision, Timeout
    from crowd_nav.utils.explorer import Explorer
    from crowd_nav.utils.memory import ReplayMemory
    JointState.self_state = property(lambda self_: self_.robot_state)
    mod = importlib.import_module("crowd_nav.configs.icra_benchmark.mp_separate")
    envc = mod.EnvConfig()
    old_policy, old_central = envc.humans.policy, envc.sim.centralized_planning
    old_radius, old_limit = envc.sim.circle_radius, envc.env.time_limit
    envc.humans.policy = "linear"
    envc.sim.centralized_planning = False
    # a tight arena and a short clock: the weights are not a trained policy, so on the default 4 m circle every episode times
    # out; at 1.5 m the same robot also reaches goals and collides, and the replay memory receives tuples
    envc.sim.circle_radius, envc.env.time_limit = 1.5, 12
    try:
        env = gym.make("CrowdSim-v0")
        env.configure(envc)
        robot = Robot(envc, "robot")
        robot.time_step = env.time_step
        pol = policy_factory["model_predictive_rl"]()
        pol.configure(policy_config())
        m = masters["goal"]
        pol.load_state_dict({"graph_model1": graph_sd(m, "graph_model1", 2, "embedded_gaussian"),
                             "graph_model2": graph_sd(m, "graph_model2", 2, "embedded_gaussian"),
                             "value_network": sub_sd(m, "value_network"),
                             "motion_predictor": sub_sd(m, "motion_predictor")})
        pol.set_device(torch.device("cpu"))
        pol.set_time_step(env.time_step)
        pol.set_epsilon(0.0)
        ve_fwd = pol.value_estimator.forward
        pol.value_estimator.forward = lambda state: ve_fwd(state).reshape(())        # scalar-shape shim (header)
        robot.set_policy(pol)
        env.set_robot(robot)
        memory = ReplayMemory(100000)
        explorer = Explorer(env, robot, torch.device("cpu"), None, memory, 0.9, target_policy=pol)
        table = [(a.vx, a.vy) for a in pol.action_space] if pol.action_space else None
        # record every decision and every episode end by wrapping robot.act / env.step (nothing in the reference changes)
        log = {"actions": [], "ends": []}
        act0, step0 = robot.act, env.step

        def act(ob):
            a = act0(ob)
            tbl = [(x.vx, x.vy) for x in pol.action_space]
            log["actions"].append(tbl.index((a.vx, a.vy)))
            return a

        def step(action, update=True):
            ob, reward, done, info = step0(action, update)
            if done and update:
                code = 3 if isinstance(info, ReachGoal) else (2 if isinstance(info, Collision) else 4)
                log["ends"].append((code, env.global_time, len(log["actions"])))
            return ob, reward, done, info
        robot.act, env.step = act, step
        meta = []
        with torch.no_grad():
            for tag, phase, k, upd in (("val6", "val", 6, False), ("test4", "test", 4, False), ("train8", "train", 8, True)):
                log["actions"], log["ends"] = [], []
                n_before = len(memory.memory)
                stats = explorer.run_k_episodes(k, phase, update_memory=upd, episode=3)
                key = "ex.%s." % tag
                out[key + "stats"] = np.array(stats, np.float64)
                out[key + "actions"] = np.array(log["actions"], np.int64)
                out[key + "outcome"] = np.array([e[0] for e in log["ends"]], np.int64)
                out[key + "time"] = np.array([e[1] for e in log["ends"]], np.float64)
                out[key + "steps_end"] = np.array([e[2] for e in log["ends"]], np.int64)     # cumulative decision count
                if upd:
                    new = memory.memory[n_before:]
                    out[key + "n_tuples"] = np.array(len(new), np.int64)
                    for name, col in (() if not new else (("robot", 0), ("humans", 1), ("value", 2), ("reward", 3), ("next_robot", 4),
                                      ("next_humans", 5))):
                        out[key + "mem_" + name] = np.stack([t[col].numpy() for t in new]).astype(np.float32)
                meta.append("%s|%s|%d|%d" % (tag, phase, k, int(upd)))
        out["explorer_cases"] = np.array(meta)
        out["ex.circle_radius"], out["ex.time_limit"] = np.array(1.5), np.array(12.0)
    finally:
        envc.humans.policy, envc.sim.centralized_planning = old_policy, old_central
        envc.sim.circle_radius, envc.env.time_limit = old_radius, old_limit
        del JointState.self_state


def gen_trainer_kats(masters, out):
    """The reference MPRLTrainer.optimize_batch (crowd_nav/utils/trainer.py:110-161) on the reference modules: three batches of 16
    seeded transitions, Adam, frozen target copy, with and without detach_state_predictor.  skip_connection=False: the reference's
    RGL.forward adds the skip in place (`next_H += H`), which current torch autograd refuses, so upstream training only runs
    without it.  The loader is injected un-shuffled so that the batch order is part of the fixture."""
    from crowd_nav.utils.trainer import MPRLTrainer
    from crowd_nav.utils.memory import ReplayMemory
    from torch.utils.data import DataLoader

    class Writer(object):
        def add_scalar(self, *a, **k):
            pass
    rng = np.random.RandomState(31)
    n = 48
    robot, humans = synth_scene(rng, n, 5)
    robot2, humans2 = synth_scene(rng, n, 5)
    rewards = rng.uniform(-0.25, 1.0, (n, 1)).astype(np.float32)
    out["tr.robot"], out["tr.humans"], out["tr.next_robot"], out["tr.next_humans"], out["tr.rewards"] = robot, humans, robot2, humans2, rewards
    meta = []
    # ("squared", round 5): a non-default similarity function through the reference trainer (graph_model.py:86-89) -- what the tile
    # backward's plain-weight normalisations are held against under RGL_BACKWARD_MFMA=2
    # ("cosine_softmax"): the cosine family (:75-79), on the tile backward since the same round
    for tag, detach, sim in (("plain", False, "embedded_gaussian"), ("detach", True, "embedded_gaussian"), ("squared", False, "squared"),
                             ("cosine_softmax", False, "cosine_softmax")):
        torch.manual_seed(0)
        pc, g1, g2, ve, sp = build_ref_modules(masters["trained"], 2, sim, False, False)
        memory = ReplayMemory(1000)
        for i in range(n):
            memory.push((torch.tensor(robot[i:i + 1]), torch.tensor(humans[i]), torch.zeros(1), torch.tensor(rewards[i]),
                         torch.tensor(robot2[i:i + 1]), torch.tensor(humans2[i])))
        tr = MPRLTrainer(ve, sp, memory, torch.device("cpu"), None, Writer(), 16, "Adam", 5, False, False, detach, False)
        tr.set_learning_rate(1e-3)
        tr.update_target_model(ve)
        tr.data_loader = DataLoader(memory, 16, shuffle=False)
        with torch.enable_grad():
            v_loss, s_loss = tr.optimize_batch(2, 0)            # batch_count > num_batches: three batches are consumed
        out["tr.%s.losses" % tag] = np.array([v_loss, s_loss], np.float64)
        for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                          ("motion_predictor", sp.human_motion_predictor)):
            out.update(flat("tr.%s.%s." % (tag, name), mod.state_dict()))
        meta.append("%s|%d" % (tag, int(detach)))
    out["trainer_cases"] = np.array(meta)
    # Imitation learning, then RL with detach_state_predictor (configs/icra_benchmark/mp_detach.py; train.py runs optimize_epoch
    # before the RL episodes): two optimize_epoch passes train the predictor's graph model (state-predictor update on every 5th
    # batch, trainer.py:87-99), then three RL batches in which StatePredictor(..., detach=True) gives that graph model NO gradient --
    # `.grad` is None after zero_grad(), Adam skips its parameters although their momentum is non-zero by now.  Separate arrays: the
    # older cases stay bit-identical.
    values = rng.uniform(0.0, 1.0, (n, 1)).astype(np.float32)
    out["tr.values"] = values
    torch.manual_seed(0)
    pc, g1, g2, ve, sp = build_ref_modules(masters["trained"], 2, "embedded_gaussian", False, False)
    memory = ReplayMemory(1000)
    for i in range(n):
        memory.push((torch.tensor(robot[i:i + 1]), torch.tensor(humans[i]), torch.tensor(values[i]), torch.tensor(rewards[i]),
                     torch.tensor(robot2[i:i + 1]), torch.tensor(humans2[i])))
    tr = MPRLTrainer(ve, sp, memory, torch.device("cpu"), None, Writer(), 16, "Adam", 5, False, False, True, False)
    tr.set_learning_rate(1e-3)
    tr.update_target_model(ve)
    tr.data_loader = DataLoader(memory, 16, shuffle=False)
    with torch.enable_grad():
        tr.optimize_epoch(2)
        for name, mod in (("graph_model2", sp.graph_model), ("motion_predictor", sp.human_motion_predictor)):
            out.update(flat("tr.il_then_detach.after_il.%s." % name, mod.state_dict()))
        v_loss, s_loss = tr.optimize_batch(2, 0)
    out["tr.il_then_detach.losses"] = np.array([v_loss, s_loss], np.float64)
    for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                      ("motion_predictor", sp.human_motion_predictor)):
        out.update(flat("tr.il_then_detach.%s." % name, mod.state_dict()))
    # train.py's own order (crowd_nav/train.py:134-176): imitation learning at il_learning_rate, NEW optimizers at rl_learning_rate,
    # the target model refreshed, RL batches, the target refreshed again, more RL batches.
    torch.manual_seed(0)
    pc, g1, g2, ve, sp = build_ref_modules(masters["trained"], 2, "embedded_gaussian", False, False)
    memory = ReplayMemory(1000)
    for i in range(n):
        memory.push((torch.tensor(robot[i:i + 1]), torch.tensor(humans[i]), torch.tensor(values[i]), torch.tensor(rewards[i]),
                     torch.tensor(robot2[i:i + 1]), torch.tensor(humans2[i])))
    tr = MPRLTrainer(ve, sp, memory, torch.device("cpu"), None, Writer(), 16, "Adam", 5, False, False, False, False)
    tr.data_loader = DataLoader(memory, 16, shuffle=False)
    with torch.enable_grad():
        tr.set_learning_rate(1e-2)
        tr.optimize_epoch(1)
        tr.set_learning_rate(1e-3)
        tr.update_target_model(ve)
        first = tr.optimize_batch(2, 0)
        tr.update_target_model(ve)
        second = tr.optimize_batch(2, 1)
    out["tr.train_py_order.losses"] = np.array([first[0], first[1], second[0], second[1]], np.float64)
    for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                      ("motion_predictor", sp.human_motion_predictor)):
        out.update(flat("tr.train_py_order.%s." % name, mod.state_dict()))
    # The trainer's other switches through the reference itself (trainer.py:43-61,134-141): SGD (momentum 0.9 on the value side, plain
    # on the predictor side), reduce_sp_update_frequency (NO predictor update on batches 0, 5, 10 ..), freeze_state_predictor.
    for tag, opt, reduce, freeze in (("sgd", "SGD", False, False), ("reduce", "Adam", True, False), ("freeze", "Adam", False, True)):
        torch.manual_seed(0)
        pc, g1, g2, ve, sp = build_ref_modules(masters["trained"], 2, "embedded_gaussian", False, False)
        memory = ReplayMemory(1000)
        for i in range(n):
            memory.push((torch.tensor(robot[i:i + 1]), torch.tensor(humans[i]), torch.zeros(1), torch.tensor(rewards[i]),
                         torch.tensor(robot2[i:i + 1]), torch.tensor(humans2[i])))
        tr = MPRLTrainer(ve, sp, memory, torch.device("cpu"), None, Writer(), 16, opt, 5, reduce, freeze, False, False)
        tr.set_learning_rate(1e-2 if opt == "SGD" else 1e-3)
        tr.update_target_model(ve)
        tr.data_loader = DataLoader(memory, 16, shuffle=False)
        with torch.enable_grad():
            v_loss, s_loss = tr.optimize_batch(2, 0)
        out["tr.%s.losses" % tag] = np.array([v_loss, s_loss], np.float64)
        for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                          ("motion_predictor", sp.human_motion_predictor)):
            out.update(flat("tr.%s.%s." % (tag, name), mod.state_dict()))
    policy_config(gcn__skip_connection=True, gcn__similarity_function="embedded_gaussian")      # restore the class-level config attributes


def gen_vnrl_trainer_kats(pg_fixture, out):
    """The reference VNRLTrainer.optimize_batch (crowd_nav/utils/trainer.py:199-250; path G's trainer) on the reference
    gcn.ValueNetwork with the weights of fixture path_g.npz: three un-shuffled batches of 16 seeded transitions
    (rotated states (5,13), reward, next rotated states), Adam 1e-3, frozen target copy, the reference's own pad_batch collate.
    gcn.ValueNetwork adds its skip connections out of place, so the shipped skip_connection=True configuration trains."""
    from crowd_nav.utils.trainer import VNRLTrainer, pad_batch
    from crowd_nav.utils.memory import ReplayMemory
    from torch.utils.data import DataLoader

    class Writer(object):
        def add_scalar(self, *a, **k):
            pass
    rng = np.random.RandomState(41)
    n = 48
    states = rng.uniform(-2, 2, (n, 5, 13)).astype(np.float32)
    next_states = rng.uniform(-2, 2, (n, 5, 13)).astype(np.float32)
    rewards = rng.uniform(-0.25, 1.0, (n,)).astype(np.float32)
    out["vn.states"], out["vn.next_states"], out["vn.rewards"] = states, next_states, rewards
    meta = []
    for tag, L, lw, sk in (("shipped", 2, False, True), ("layerwise_noskip", 2, True, False)):
        pc = policy_config("rgl", gcn__num_layer=L, gcn__layerwise_graph=lw, gcn__skip_connection=sk)
        pol = policy_factory["gcn"]()
        pol.configure(pc)
        pol.model.load_state_dict({k[len("g.weights."):]: torch.tensor(v) for k, v in pg_fixture.items()
                                   if k.startswith("g.weights.")})
        memory = ReplayMemory(1000)
        for i in range(n):
            memory.push((torch.tensor(states[i]), torch.zeros(1), torch.tensor(rewards[i:i + 1]), torch.tensor(next_states[i])))
        tr = VNRLTrainer(pol.model, memory, torch.device("cpu"), pol, 16, "Adam", Writer())
        tr.set_learning_rate(1e-3)
        tr.update_target_model(pol.model)
        tr.data_loader = DataLoader(memory, 16, shuffle=False, collate_fn=pad_batch)
        with torch.enable_grad():
            loss = tr.optimize_batch(2, 0)                       # batch_count > num_batches: three batches are consumed
        out["vn.%s.loss" % tag] = np.array([loss], np.float64)
        out.update(flat("vn.%s.model." % tag, pol.model.state_dict()))
        meta.append("%s|%d|%d|%d" % (tag, L, int(lw), int(sk)))
    out["vnrl_cases"] = np.array(meta)
    # imitation learning then RL through the reference's VNRLTrainer (trainer.py:199-250; train.py's order): optimize_epoch(2) on
    # seeded values, then optimize_batch(2), shipped configuration.  New arrays only.
    values = rng.uniform(0.0, 1.0, (n,)).astype(np.float32)
    out["vn.values"] = values
    pc = policy_config("rgl", gcn__num_layer=2, gcn__layerwise_graph=False, gcn__skip_connection=True)
    pol = policy_factory["gcn"]()
    pol.configure(pc)
    pol.model.load_state_dict({k[len("g.weights."):]: torch.tensor(v) for k, v in pg_fixture.items() if k.startswith("g.weights.")})
    memory = ReplayMemory(1000)
    for i in range(n):
        memory.push((torch.tensor(states[i]), torch.tensor(values[i:i + 1]), torch.tensor(rewards[i:i + 1]), torch.tensor(next_states[i])))
    tr = VNRLTrainer(pol.model, memory, torch.device("cpu"), pol, 16, "Adam", Writer())
    tr.set_learning_rate(1e-3)
    tr.update_target_model(pol.model)
    tr.data_loader = DataLoader(memory, 16, shuffle=False, collate_fn=pad_batch)
    with torch.enable_grad():
        il = tr.optimize_epoch(2)
        out.update(flat("vn.il_then_rl.after_il.model.", pol.model.state_dict()))
        rl = tr.optimize_batch(2, 0)
    out["vn.il_then_rl.losses"] = np.array([il, rl], np.float64)
    out.update(flat("vn.il_then_rl.model.", pol.model.state_dict()))
    policy_config("rgl", gcn__num_layer=2, gcn__layerwise_graph=False, gcn__skip_connection=True)


def gen_trainer_host_kats(out):
    """The reference trainers (crowd_nav/utils/trainer.py) driving tests/trainer_standins.py's small CPU modules: what the product
    trainers' host logic -- batching (un-shuffled AND shuffled: the loader's draws from torch's global generator), step order,
    optimizers, target model, loss bookkeeping -- is held against in the CPU suite.  Three flows, final parameters and losses."""
    from crowd_nav.utils.trainer import MPRLTrainer, VNRLTrainer, pad_batch
    from crowd_nav.utils.memory import ReplayMemory
    from torch.utils.data import DataLoader
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE_SRC)))
    from tests.trainer_standins import StandInValue, StandInPredictor, StandInPathG, seeded, flat_params

    class Writer(object):
        def add_scalar(self, *a, **k):
            pass
    rng = np.random.RandomState(77)
    n, H = 70, 4                                                    # 70 = 4 batches of 16 + one of 6: a short last batch
    robot = rng.uniform(-2, 2, (n, 1, 9)).astype(np.float32)
    humans = rng.uniform(-2, 2, (n, H, 5)).astype(np.float32)
    robot2 = rng.uniform(-2, 2, (n, 1, 9)).astype(np.float32)
    humans2 = rng.uniform(-2, 2, (n, H, 5)).astype(np.float32)
    values = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    rewards = rng.uniform(-0.25, 1, (n, 1)).astype(np.float32)
    states = rng.uniform(-2, 2, (n, H, 13)).astype(np.float32)
    next_states = rng.uniform(-2, 2, (n, H, 13)).astype(np.float32)
    for k, v in (("robot", robot), ("humans", humans), ("next_robot", robot2), ("next_humans", humans2), ("values", values),
                 ("rewards", rewards), ("states", states), ("next_states", next_states)):
        out["th." + k] = v

    def mprl_memory():
        memory = ReplayMemory(1000)
        for i in range(n):
            memory.push((torch.tensor(robot[i]), torch.tensor(humans[i]), torch.tensor(values[i]), torch.tensor(rewards[i]),
                         torch.tensor(robot2[i]), torch.tensor(humans2[i])))
        return memory
    for tag, shuffle, detach, reduce, opt in (("ordered", False, True, False, "Adam"), ("shuffled", True, False, True, "Adam"),
                                              ("shuffled_sgd", True, False, False, "SGD")):
        ve, sp = seeded(StandInValue, 11), seeded(StandInPredictor, 12)
        tr = MPRLTrainer(ve, sp, mprl_memory(), torch.device("cpu"), None, Writer(), 16, opt, 3, reduce, False, detach, False)
        if not shuffle:
            tr.data_loader = DataLoader(tr.memory, 16, shuffle=False)
        torch.manual_seed(5)                                        # the shuffled loaders draw from here on
        with torch.enable_grad():
            tr.set_learning_rate(1e-2)
            tr.optimize_epoch(2)
            tr.set_learning_rate(1e-3)
            tr.update_target_model(ve)
            first = tr.optimize_batch(2, 0)
            tr.update_target_model(ve)
            second = tr.optimize_batch(3, 1)
        out["th.mprl.%s.losses" % tag] = np.array(list(first) + list(second), np.float64)
        out["th.mprl.%s.params" % tag] = flat_params(ve, sp)
    for tag, shuffle in (("ordered", False), ("shuffled", True)):
        model = seeded(StandInPathG, 13)
        memory = ReplayMemory(1000)
        for i in range(n):
            memory.push((torch.tensor(states[i]), torch.tensor(values[i]), torch.tensor(rewards[i]), torch.tensor(next_states[i])))
        tr = VNRLTrainer(model, memory, torch.device("cpu"), None, 16, "Adam", Writer())
        if not shuffle:
            tr.data_loader = DataLoader(memory, 16, shuffle=False, collate_fn=pad_batch)
        torch.manual_seed(6)
        with torch.enable_grad():
            tr.set_learning_rate(1e-2)
            il = tr.optimize_epoch(2)
            tr.set_learning_rate(1e-3)
            tr.update_target_model(model)
            rl = tr.optimize_batch(3, 0)
        out["th.vnrl.%s.losses" % tag] = np.array([il, rl], np.float64)
        out["th.vnrl.%s.params" % tag] = flat_params(model)


def gen_greedy_kats(out):
    """MultiHumanRL.predict with an EMPTY crowd (multi_human_rl.py:27-31) -> CADRL.select_greedy_action (cadrl.py:193-228): the
    table action closest to the straight-to-goal velocity.  Holonomic (the unicycle branch's last case raises upstream: a
    misplaced parenthesis makes np.array take a float as dtype)."""
    pc = policy_config("rgl")
    pol = policy_factory["gcn"]()
    pol.configure(pc)
    pol.set_phase("test")
    pol.set_device(torch.device("cpu"))
    pol.time_step = 0.25
    rng = np.random.RandomState(51)
    rows, acts = [], []
    for i in range(24):
        px, py, gx, gy = rng.uniform(-4, 4, 4)
        if i % 6 == 0:
            gx, gy = px + rng.uniform(-0.1, 0.1), py + rng.uniform(0.31, 0.4)      # close to the goal: speed = distance / dt
        row = [px, py, rng.uniform(-1, 1), rng.uniform(-1, 1), 0.3, gx, gy, 1.0, rng.uniform(-3, 3)]
        js = JointState(FullState(*row), [])
        a = pol.predict(js)
        rows.append(row)
        acts.append([k for k, x in enumerate(pol.action_space) if x is a][0])
    out["greedy.robot"] = np.array(rows, np.float64)
    out["greedy.action"] = np.array(acts, np.int64)


def gen_query_env_kats(pg_fixture, out):
    """Path G with query_env=True (multi_human_rl.py:43-44) on the reference simulator with `linear` humans: per action the next
    human states and the reward come from env.onestep_lookahead.  States a few steps into seeded test cases; recorded: the full
    simulator state (float64), human goals and preferred speeds, the 81 action values and the chosen action."""
    import gym
    from crowd_sim.envs.utils.robot import Robot
    JointState.self_state = property(lambda self_: self_.robot_state)
    mod = importlib.import_module("crowd_nav.configs.icra_benchmark.mp_separate")
    envc = mod.EnvConfig()
    old_policy, old_central = envc.humans.policy, envc.sim.centralized_planning
    envc.humans.policy = "linear"
    envc.sim.centralized_planning = False
    try:
        env = gym.make("CrowdSim-v0")
        env.configure(envc)
        robot = Robot(envc, "robot")
        robot.time_step = env.time_step
        pol = policy_factory["gcn"]()
        pc = policy_config("rgl")
        pol.configure(pc)
        pol.model.load_state_dict({k[len("g.weights."):]: torch.tensor(v) for k, v in pg_fixture.items() if k.startswith("g.weights.")})
        pol.set_phase("test")
        pol.set_device(torch.device("cpu"))
        pol.time_step = env.time_step
        pol.query_env = True
        robot.set_policy(pol)
        env.set_robot(robot)
        pol.set_env(env)
        R, Hs, goals, vpref, avals, acts, times = [], [], [], [], [], [], []
        for case, steps in ((0, 0), (1, 6), (2, 14), (4, 9)):
            ob = env.reset("test", case)
            for _ in range(steps):
                ob, _, done, _ = env.step(ActionXY(np.float64(0.0), np.float64(0.7)))
            js = JointState(robot.get_full_state(), ob)
            with torch.no_grad():
                a = pol.predict(js)
            R.append(robot.get_full_state().to_tuple())
            Hs.append([h.get_full_state().to_tuple() for h in env.humans])
            goals.append([(h.gx, h.gy) for h in env.humans])
            vpref.append([h.v_pref for h in env.humans])
            avals.append(pol.action_values)
            acts.append([i for i, x in enumerate(pol.action_space) if x is a][0])
            times.append(env.global_time)
        out["qe.robot"] = np.array(R, np.float64)
        out["qe.humans_full"] = np.array(Hs, np.float64)            # (cases, H, 9) FullState of every human
        out["qe.human_goals"] = np.array(goals, np.float64)
        out["qe.human_vpref"] = np.array(vpref, np.float64)
        out["qe.action_values"] = np.array(avals, np.float64)
        out["qe.action"] = np.array(acts, np.int64)
        out["qe.time"] = np.array(times, np.float64)
    finally:
        envc.humans.policy, envc.sim.centralized_planning = old_policy, old_central
        del JointState.self_state
        policy_config("rgl", action_space__query_env=False)


def main():
    global HERE
    if len(sys.argv) > 1:                       # optional output directory (regeneration checks write to a scratch dir)
        HERE = os.path.abspath(sys.argv[1])
        os.makedirs(HERE, exist_ok=True)
    torch.set_num_threads(1)
    masters = {"rand": make_master(1, 1.0), "trained": make_master(2, 1.0 / np.sqrt(32.0))}
    np.savez(os.path.join(HERE, "weights_rand.npz"), **masters["rand"])
    np.savez(os.path.join(HERE, "weights_trained.npz"), **masters["trained"])
    masters["goal"] = make_goal_master(masters["trained"])
    np.savez(os.path.join(HERE, "weights_goal.npz"), **masters["goal"])
    scenes = gen_env_scenes()
    np.savez(os.path.join(HERE, "scenes.npz"), **scenes)
    fw = {}
    gen_forward_kats(masters, fw)
    gen_state_predictor_kats(masters, fw)
    np.savez(os.path.join(HERE, "forward.npz"), **fw)
    fw4 = {}
    gen_forward_kats_l4(masters, fw4)
    np.savez(os.path.join(HERE, "forward_l4.npz"), **fw4)
    misc = {}
    gen_action_spaces(misc)
    gen_reward_kats(misc)
    np.savez(os.path.join(HERE, "actions_rewards.npz"), **misc)
    pl = {}
    gen_planning_kats(masters, scenes, pl)
    np.savez(os.path.join(HERE, "planning.npz"), **pl)
    pg = {}
    gen_path_g(pg, scenes)
    np.savez(os.path.join(HERE, "path_g.npz"), **pg)
    sim = {}
    gen_sim_kats(sim)
    np.savez(os.path.join(HERE, "sim.npz"), **sim)
    ex = {}
    gen_explorer_kats(masters, ex)
    np.savez(os.path.join(HERE, "explorer.npz"), **ex)
    tq = {}
    gen_trainer_kats(masters, tq)
    gen_query_env_kats(pg, tq)
    np.savez(os.path.join(HERE, "training_queryenv.npz"), **tq)
    vt = {}
    gen_greedy_kats(vt)
    gen_vnrl_trainer_kats(pg, vt)
    np.savez(os.path.join(HERE, "vnrl_trainer.npz"), **vt)
    th = {}
    gen_trainer_host_kats(th)                               # round 5; its own file
    np.savez(os.path.join(HERE, "trainer_host.npz"), **th)
    rc = {}
    gen_root_clip_kats(masters, rc)                         # round 5; its own file and its own rng: the earlier fixtures stay bit-identical
    gen_tie_kats(masters, rc)
    np.savez(os.path.join(HERE, "root_clip.npz"), **rc)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-24s %8.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == "__main__":
    main()
