"""Pins the CPU oracle against fixtures produced by the reference itself (tests/golden/make_golden.py).

These run on CPU (`-m "not gpu"`).  Tolerances: fp32 forward 2e-6 abs/rel-ish (measured noise ~1e-7),
rewards 1e-7 (float64 scalar code), action tables exact to 1e-15.
"""
import numpy as np
import pytest
import torch

from oracle import rgl_oracle as orc
from tests import golden_io as gio


def cfg_of(c, **kw):
    return orc.OracleConfig(num_layer=c.get("L", 2), similarity=c.get("sim", "embedded_gaussian"),
                            layerwise_graph=c.get("layerwise", False), skip_connection=c.get("skip", True), **kw)


def close(a, b, tol):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max(), scale)


@pytest.mark.parametrize("c", gio.forward_cases(), ids=lambda c: "f%02d-%s-H%d-L%d-lw%d-sk%d" % (
    c["idx"], c["sim"], c["H"], c["L"], c["layerwise"], c["skip"]))
def test_forward_kats(c):
    fw = gio.load(c["file"])
    k = "f%02d." % c["idx"]
    m = gio.master(c["flavour"])
    cfg = cfg_of(c)
    g1 = gio.graph_sd(m, "graph_model1", c["L"], c["sim"])
    g2 = gio.graph_sd(m, "graph_model2", c["L"], c["sim"])
    robot = torch.tensor(fw[k + "robot"]).unsqueeze(1)
    humans = torch.tensor(fw[k + "humans"])
    H_L, _ = orc.rgl_forward(robot, humans, g1, cfg)
    A = orc.similarity_matrix(orc.rgl_embed(robot, humans, g1), g1, c["sim"])
    val = orc.value_estimator_forward(robot, humans, g1, gio.sub_sd(m, "value_network"), cfg)
    nh = orc.state_predictor_humans(robot, humans, g2, gio.sub_sd(m, "motion_predictor"), cfg)
    close(H_L.numpy(), fw[k + "H_L"], 2e-6)
    close(A.numpy(), fw[k + "A"], 2e-6)
    close(val.numpy(), fw[k + "value"], 2e-6)
    close(nh.numpy(), fw[k + "humans_next"], 2e-6)


def test_state_predictor_kats():
    fw = gio.load("forward")
    cfg = orc.OracleConfig()
    r = torch.tensor(fw["sp.robot"]).reshape(9)
    for a, want in zip(fw["sp.actions"], fw["sp.next_robot"]):
        got = orc.next_robot_state(r, a, cfg).numpy()
        assert np.array_equal(got, want)
    got = orc.linear_humans(torch.tensor(fw["sp.humans"])).numpy()
    assert np.array_equal(got, fw["sp.linear_next_humans"])
    assert np.array_equal(orc.next_robot_state(r, fw["sp.actions"][1], cfg).numpy(), fw["sp.linear_next_robot"])
    ucfg = orc.OracleConfig(kinematics="unicycle")
    got = orc.next_robot_state(r, fw["sp.unicycle_action"], ucfg).numpy()
    close(got, fw["sp.unicycle_next_robot"], 1e-6)
    assert got[8] == r.numpy()[8] and got[7] != r.numpy()[7]     # the slot-7 quirk is pinned


def test_action_spaces():
    ar = gio.load("actions_rewards")
    acts, groups = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    assert np.abs(acts - ar["act.mprl"]).max() < 1e-15
    assert np.array_equal(groups, ar["act.mprl_groups"])
    assert np.abs(orc.mprl_action_space(orc.OracleConfig(), 0.7)[0] - ar["act.mprl_vpref07"]).max() < 1e-15
    assert np.abs(orc.cadrl_action_space(orc.OracleConfig(), 1.0) - ar["act.gcn"]).max() < 1e-15
    ua = orc.mprl_action_space(orc.OracleConfig(kinematics="unicycle"), 1.0)[0]
    assert np.abs(ua - ar["act.mprl_unicycle"]).max() < 1e-15


def test_reward_kats():
    ar = gio.load("actions_rewards")
    cfg = orc.OracleConfig()
    for i, name in enumerate(ar["rew.names"]):
        n = int(ar["rew.n_humans"][i])
        robot = [float(x) for x in ar["rew.robot"][i]]
        humans = [[float(x) for x in row] for row in ar["rew.humans"][i][:n]]
        a = ar["rew.actions"][i]
        assert abs(orc.estimate_reward(robot, humans, a, cfg) - ar["rew.joint"][i]) < 1e-12, name
        r32 = [np.float32(x) for x in robot]
        h32 = [[np.float32(x) for x in row] for row in humans]
        assert abs(orc.estimate_reward(r32, h32, a, cfg) - ar["rew.tensor"][i]) < 1e-12, name
        assert abs(orc.compute_reward_g(robot, humans, 0.25) - ar["rew.g"][i]) < 1e-12, name
    # every branch must actually be present in the KAT set
    assert {-0.25, 1.0, 0.0} <= set(np.round(ar["rew.joint"], 6)) and (ar["rew.joint"] < 0).sum() >= 3


def test_reward_sweep_batched():
    ar = gio.load("actions_rewards")
    cfg = orc.OracleConfig()
    acts, _ = orc.mprl_action_space(cfg, 1.0)
    r = torch.tensor(ar["rew.sweep_robot"])
    h = torch.tensor(ar["rew.sweep_humans"])
    got_t = orc.estimate_reward_batched(r, h, acts, cfg, root=False)
    got_j = orc.estimate_reward_batched(r, h, acts, cfg, root=True)
    assert np.abs(got_t - ar["rew.sweep_tensor"]).max() < 1e-7
    assert np.abs(got_j - ar["rew.sweep_joint"]).max() < 1e-7
    vals = np.round(ar["rew.sweep_tensor"], 6)
    assert (vals == -0.25).any() and (vals == 0).any() and ((vals < 0) & (vals > -0.25)).any()


def test_point_to_segment():
    ar = gio.load("actions_rewards")
    for p, want in zip(ar["p2s.in"], ar["p2s.out"]):
        assert abs(orc.point_to_segment_dist(*p) - want) < 1e-15


@pytest.mark.parametrize("c", gio.plan_cases(), ids=lambda c: c["tag"])
def test_planning_sequential(c):
    """The reference-order walk: same action, same values, same number of forwards."""
    pl = gio.load("planning")
    k = "plan.%s." % c["tag"]
    P = gio.oracle_params(c["flavour"], 2, c["variant"])
    cfg = orc.OracleConfig(planning_depth=c["D"], planning_width=c["w"], do_action_clip=c["clip"],
                           sparse_search=c["sparse"], linear_state_predictor=(c["variant"] == "linear"))
    R = pl["plan.scene.%s.robot" % c["scene"]]
    Hh = pl["plan.scene.%s.humans" % c["scene"]]
    with torch.no_grad():
        for b in range(R.shape[0]):
            tr = orc.SeqTrace()
            a, v = orc.mprl_predict_sequential([float(x) for x in R[b]], [[float(x) for x in row] for row in Hh[b]],
                                               P, cfg, tr)
            assert a == int(pl[k + "action"][b])
            assert abs(float(v) - float(pl[k + "max_value"][b])) < 2e-6 * max(1.0, abs(float(v)))
            assert tr.root_clipped == [int(x) for x in pl[k + "kept"][b]]
            close(tr.root_values, pl[k + "root_values"][b], 2e-6)
            if c["clip"]:
                close(tr.root_clip_values, pl[k + "clip_values"][b], 2e-6)
            assert tr.n_value_forwards == int(pl[k + "counts"][b][0])
            if c["variant"] != "linear":
                assert tr.n_predictor_forwards == int(pl[k + "counts"][b][1])


@pytest.mark.parametrize("c", gio.plan_cases(), ids=lambda c: c["tag"])
def test_planning_batched(c):
    """The level-synchronous restatement gives the reference's decisions and values."""
    pl = gio.load("planning")
    k = "plan.%s." % c["tag"]
    P = gio.oracle_params(c["flavour"], 2, c["variant"])
    cfg = orc.OracleConfig(planning_depth=c["D"], planning_width=c["w"], do_action_clip=c["clip"],
                           sparse_search=c["sparse"], linear_state_predictor=(c["variant"] == "linear"))
    R = torch.tensor(pl["plan.scene.%s.robot" % c["scene"]].astype(np.float32))
    Hh = torch.tensor(pl["plan.scene.%s.humans" % c["scene"]].astype(np.float32))
    with torch.no_grad():
        a, v, rv, kept, levels = orc.mprl_predict_batched(R, Hh, P, cfg, return_levels=True)
    assert np.array_equal(a.numpy(), pl[k + "action"])
    close(v.numpy(), pl[k + "max_value"], 3e-6)
    # kept sets agree as sets (order inside the clipped set is a tie-break detail)
    for b in range(R.shape[0]):
        assert sorted(kept[b].tolist()) == sorted(pl[k + "kept"][b].tolist())
        order = [kept[b].tolist().index(int(i)) for i in pl[k + "kept"][b]]
        close(rv[b].numpy()[order], pl[k + "root_values"][b], 3e-6)
    if c["clip"]:
        close(levels[0]["value1"].numpy(), pl[k + "clip_values"], 3e-6)


@pytest.mark.parametrize("c", gio.root_clip_cases(), ids=lambda c: c["tag"])
def test_root_clip_is_priced_on_the_tensor_state(c):
    """upstream hands the root's action_clip the float32 TENSOR of the state (model_predictive_rl.py:216-218
    -> :246-248 -> tensor_to_joint_state: float32-born scalars) and prices the kept actions on the float64 JointState (:226).
    Fixture root_clip.npz holds what the reference computed INSIDE that action_clip on 24 genuine-float64 crowded roots: the
    reward of every action as estimate_reward returned it there (bit for bit: float64 scalar code), the values array the
    selection ran on (bit for bit: one float32 add of a float32 product), the kept lists, the rewards and values of the final
    loop.  Both restatements -- the sequential walk and the level-synchronous one -- against it."""
    rc = gio.load("root_clip")
    k = "rootclip.%s." % c["tag"]
    P = gio.oracle_params("trained", 2, c["variant"])
    cfg = orc.OracleConfig(planning_depth=c["D"], planning_width=c["w"], do_action_clip=True, sparse_search=c["sparse"],
                           linear_state_predictor=(c["variant"] == "linear"))
    R64, H64 = rc["rootclip.robot64"], rc["rootclip.humans64"]
    assert (R64[:, :2].astype(np.float32).astype(np.float64) != R64[:, :2]).all()       # genuine float64 roots
    n_differ = 0
    with torch.no_grad():
        for b in range(R64.shape[0]):
            tr = orc.SeqTrace()
            a, v = orc.mprl_predict_sequential([float(x) for x in R64[b]], [[float(x) for x in row] for row in H64[b]], P, cfg, tr)
            assert np.array_equal(tr.root_clip_rewards, rc[k + "clip_rewards"][b])              # exact: same float64 scalar code
            close(tr.root_clip_values, rc[k + "clip_values"][b], 2e-6)                            # the network's float32 noise
            assert tr.root_clipped == [int(x) for x in rc[k + "kept"][b]]
            assert np.array_equal(tr.root_rewards, rc[k + "root_rewards"][b])
            close(tr.root_values, rc[k + "root_values"][b], 2e-6)
            assert a == int(rc[k + "action"][b])
            # ... and the float64 reading of the same state prices the actions differently (what the oracle did until round 4)
            joint = np.array([orc.estimate_reward([float(x) for x in R64[b]], [[float(x) for x in row] for row in H64[b]], act, cfg)
                              for act in orc.mprl_action_space(cfg, R64[b, 7])[0]], np.float64)
            n_differ += int((joint != rc[k + "clip_rewards"][b]).any())
        R32, H32 = torch.tensor(R64.astype(np.float32)), torch.tensor(H64.astype(np.float32))
        a, v, rv, kept, levels = orc.mprl_predict_batched(R32, H32, P, cfg, return_levels=True, roots64=(R64, H64))
    assert n_differ >= R64.shape[0] // 2                                                           # the two readings do differ
    assert np.array_equal(a.numpy(), rc[k + "action"])
    assert np.array_equal(levels[0]["reward_clip"].numpy(), rc[k + "clip_rewards"].astype(np.float32))
    close(levels[0]["value1"].numpy(), rc[k + "clip_values"], 3e-6)
    for b in range(R64.shape[0]):
        assert sorted(kept[b].tolist()) == sorted(rc[k + "kept"][b].tolist())
        order = [kept[b].tolist().index(int(i)) for i in rc[k + "kept"][b]]
        close(rv[b].numpy()[order], rc[k + "root_values"][b], 3e-6)
        assert np.array_equal(levels[0]["reward"][b].numpy()[rc[k + "kept"][b]], rc[k + "root_rewards"][b].astype(np.float32))


def test_exact_ties_fixture():
    """Fixture root_clip.npz `tie.*` (the reference on a constant value head, crowd far away): every root's 81 one-step values
    are bit-equal.  The oracle's sparse walk IS numpy's reversed argsort, so it names the reference's representatives on the
    numpy of this image; its dense selection documents its own order (lower index first) where np.argpartition's is
    implementation-defined -- same tied value either way."""
    rc = gio.load("root_clip")
    g = rc["tie.groups"]
    for line in rc["tie_cases"]:
        tag, w, sparse = str(line).split("|")
        v = rc["tie.%s.clip_values" % tag]
        assert (v == v[:, :1]).all()
        kept = orc.select_top(v, int(w), g, bool(int(sparse)))
        if int(sparse):
            assert np.array_equal(kept, rc["tie.%s.kept" % tag])
        else:
            assert kept.tolist() == [list(range(int(w)))] * v.shape[0]
            assert (np.take_along_axis(v, kept, 1) == np.take_along_axis(v, rc["tie.%s.kept" % tag], 1)).all()


def test_forward_counts_match_survey():
    pl = gio.load("planning")
    want = {"d1": 81, "d2w2": 249, "d3w2": 581}
    for tag, n in want.items():
        assert (pl["plan.%s.counts" % tag][:, 0] == n).all()


def test_path_g_rotate_and_value_network():
    g = gio.load("path_g")
    sd = gio.path_g_sd()
    close(orc.rotate_pairwise(torch.tensor(g["g.rotate_in"])).numpy(), g["g.rotate_out"], 1e-6)
    close(orc.rotate_pairwise(torch.tensor(g["g.rotate_in"]), "unicycle").numpy(), g["g.rotate_out_unicycle"], 1e-6)
    x = torch.tensor(g["g.vn_in"])
    for tag in g["g_cases"]:
        tag = str(tag)
        L, lw, sk = int(tag[1]), bool(int(tag[5])), bool(int(tag[9]))
        cfg = orc.OracleConfig(num_layer=L, layerwise_graph=lw, skip_connection=sk)
        v, A = orc.gcn_value_forward(x, sd, cfg)
        close(v.numpy(), g["g.vn_value." + tag], 2e-6)
        close(A[0].numpy(), g["g.vn_A0." + tag], 2e-6)


def test_path_g_predict():
    g = gio.load("path_g")
    sd = gio.path_g_sd()
    cfg = orc.OracleConfig()
    for b in range(g["g.pred_robot"].shape[0]):
        a, vals = orc.gcn_predict_sequential([float(x) for x in g["g.pred_robot"][b]],
                                             [[float(x) for x in row] for row in g["g.pred_humans"][b]], sd, cfg)
        assert a == int(g["g.pred_action"][b])
        close(np.array(vals), g["g.pred_action_values"][b], 2e-6)


def test_path_g_predict_batched_restatement():
    """The batched form the at-size GPU tests compare with: pinned against the reference's own outputs (fixture
    F7) and against the sequential restatement, holonomic and unicycle."""
    g = gio.load("path_g")
    sd = gio.path_g_sd()
    best, vals = orc.gcn_predict_batched(g["g.pred_robot"], g["g.pred_humans"], sd, orc.OracleConfig())
    assert np.array_equal(best, g["g.pred_action"].astype(np.int64))
    close(vals, g["g.pred_action_values"], 2e-6)
    cfg_u = orc.OracleConfig(kinematics="unicycle")
    rng = np.random.RandomState(3)
    robot = g["g.pred_robot"].astype(np.float64).copy()
    humans = np.concatenate([g["g.pred_humans"], g["g.pred_humans"] + rng.uniform(-0.5, 0.5, g["g.pred_humans"].shape)], axis=1)
    humans[:, :, 4] = 0.3
    for cfg in (orc.OracleConfig(), cfg_u):
        bb, bv = orc.gcn_predict_batched(robot, humans, sd, cfg, chunk=100)
        for b in range(robot.shape[0]):
            a, v = orc.gcn_predict_sequential([float(x) for x in robot[b]], [[float(x) for x in row] for row in humans[b]], sd, cfg)
            assert a == int(bb[b])
            close(bv[b], np.array(v), 2e-6)


def test_vnrl_trainer_fixture_with_autograd_on_the_oracle():
    """Fixture vnrl_trainer.npz (the reference VNRLTrainer.optimize_batch, trainer.py:199-250): torch autograd over the oracle's
    path-G forward + Adam reproduces the reference's parameters -- which pins the oracle as the gradient reference the GPU
    backward tests of path G compare with."""
    import torch
    fx = gio.load("vnrl_trainer")
    for case in fx["vnrl_cases"]:
        tag, L, lw, sk = str(case).split("|")
        cfg = orc.OracleConfig(num_layer=int(L), layerwise_graph=bool(int(lw)), skip_connection=bool(int(sk)))
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in gio.path_g_sd().items()}
        target = {k: v.detach().clone() for k, v in sd.items()}
        opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
        gamma_bar = pow(0.9, 0.25)
        losses = 0.0
        for b in range(3):
            sl = slice(16 * b, 16 * b + 16)
            x, x2 = torch.tensor(fx["vn.states"][sl]), torch.tensor(fx["vn.next_states"][sl])
            rew = torch.tensor(fx["vn.rewards"][sl]).unsqueeze(1)
            opt.zero_grad()
            out = orc.gcn_value_forward(x, sd, cfg)[0]
            with torch.no_grad():
                tgt = rew + gamma_bar * orc.gcn_value_forward(x2, target, cfg)[0]
            loss = torch.nn.functional.mse_loss(out, tgt)
            loss.backward()
            opt.step()
            losses += float(loss.detach())
        assert abs(losses / 2 - float(fx["vn.%s.loss" % tag][0])) < 1e-6
        for k, v in sd.items():
            close(v.detach().numpy(), fx["vn.%s.model.%s" % (tag, k)], 2e-6)


def test_env_scene_fixture_matches_survey_probe():
    sc = gio.load("scenes")
    assert np.allclose(sc["test_robot"][0], [0, -4, 0, 0, 0.3, 0, 4, 1, np.pi / 2])
    assert np.allclose(sc["test_humans"][0][0][:2], [-2.66256, -2.83799], atol=1e-5)
    assert sc["test_humans"].shape == (10, 5, 5)
