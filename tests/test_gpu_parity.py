"""GPU parity: the HIP path (through the C ABI) against the golden fixtures and the CPU oracle.

Tolerance (north star): |value_gpu - value_ref| <= 1e-4 in fp32 on identical crowd states; written
here as 1e-4 * max(1, max|ref|): for VALUES the factor is 1 everywhere (|V| < 1 with both weight sets; the
absolute error of the raw random-init set is logged in the parity report), the factor matters for the hidden
features H_L of the raw random-init weights (order 10..100), held to the same relative bar.  Integer results (actions, kept sets) and fp32-exact kinematics must be equal.
"""
import os
import numpy as np
import pytest
import torch

import relationalgraphlearning_amd as rga
from relationalgraphlearning_amd import _native as nat
from relationalgraphlearning_amd.config import policy_config
from oracle import rgl_oracle as orc
from tests import golden_io as gio
from tests.helpers import make_mprl_policy, make_gcn_policy, JS

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Regression-level bounds beside the north-star one (VERDICT r3 "weak" 1): the kernels measure 1.5e-8 (f32 values), 1.2e-6 (f16
# contractions) and 2e-6 of the largest entry (gradients) -- a regression that costs two digits must not pass at 1e-4.  Every bound
# below is >= 8x what is measured today; the north-star assertion stays next to it.
REG_F32 = 1e-6
REG_F16 = 1e-5
REG_GRAD = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def close(a, b, tol=TOL, reg=None):
    """North-star bound `tol` (relative to max(1, max|ref|)); `reg`: the regression-level bound on the same quantity."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, (err, scale)
    if reg is not None:
        assert err <= reg * scale, ("regression-level bound", err, reg, scale)
    return err


def report(line):
    from tests.conftest import PARITY_REPORT
    PARITY_REPORT.append(line)


def check_decisions(tag, act, val, oracle_out, levels, tol=TOL):
    """Decisions against the oracle.  A different root action is accepted only when it is a numerical tie IN THE ORACLE: the
    action the GPU chose must be one the oracle kept at the root with an oracle root value within `tol` of the oracle's best,
    or -- when the root clipping itself kept a different set -- its oracle one-step value must be within `tol` of the
    weakest kept one AND the GPU's value for it within `tol` of the oracle's best.  Returns (and logs) the mismatch count."""
    oa, ov, orv, okept = [x.numpy() if hasattr(x, "numpy") else np.asarray(x) for x in oracle_out]
    act = act.cpu().numpy().astype(np.int64)
    val = val.cpu().numpy().astype(np.float64)
    mism = np.nonzero(act != oa)[0]
    value1 = levels[0]["value1"].numpy() if levels is not None else None
    for b in mism:
        scale = max(1.0, abs(float(ov[b])))
        slot = np.nonzero(okept[b] == act[b])[0]
        if slot.size:
            assert orv[b, slot[0]] >= ov[b] - tol * scale, (tag, int(b), "not a tie in the oracle", float(orv[b, slot[0]]), float(ov[b]))
        else:
            assert value1 is not None, (tag, int(b), "GPU action outside the oracle's kept set")
            weakest = value1[b, okept[b]].min()
            assert value1[b, act[b]] >= weakest - tol * scale, (tag, int(b), "clipping is not a tie in the oracle")
            assert abs(val[b] - ov[b]) <= tol * scale, (tag, int(b))
    report("%s: %d of %d decisions differ from the oracle (all verified as ties in the oracle, tol %.0e); max |dV| = %.2e"
           % (tag, mism.size, act.size, tol, float(np.abs(val - ov).max())))
    return mism.size


def build_modules(c, dev):
    cfg = policy_config(gcn__num_layer=c["L"], gcn__similarity_function=c["sim"],
                        gcn__layerwise_graph=c["layerwise"], gcn__skip_connection=c["skip"])
    m = gio.master(c["flavour"])
    g1 = rga.RGL(cfg, 9, 5)
    g1.load_state_dict(gio.graph_sd(m, "graph_model1", c["L"], c["sim"]))
    g2 = rga.RGL(cfg, 9, 5)
    g2.load_state_dict(gio.graph_sd(m, "graph_model2", c["L"], c["sim"]))
    ve = rga.ValueEstimator(cfg, g1)
    ve.value_network.load_state_dict(gio.sub_sd(m, "value_network"))
    sp = rga.StatePredictor(cfg, g2, 0.25)
    sp.human_motion_predictor.load_state_dict(gio.sub_sd(m, "motion_predictor"))
    return g1.to(dev), ve.to(dev), sp.to(dev)


@pytest.mark.parametrize("c", gio.forward_cases(), ids=lambda c: "f%02d-%s-H%d-L%d-lw%d-sk%d" % (
    c["idx"], c["sim"], c["H"], c["L"], c["layerwise"], c["skip"]))
def test_forward_kats(c, dev):
    fw = gio.load(c["file"])
    k = "f%02d." % c["idx"]
    g1, ve, sp = build_modules(c, dev)
    robot = torch.tensor(fw[k + "robot"]).unsqueeze(1).to(dev)
    humans = torch.tensor(fw[k + "humans"]).to(dev)
    with torch.no_grad():
        H_L = g1((robot, humans))
        A0 = g1.A
        val = ve((robot, humans))
        nr, nh = sp((robot, humans), None)
    close(H_L.cpu().numpy(), fw[k + "H_L"])
    close(A0, fw[k + "A"][0])
    close(val.cpu().numpy(), fw[k + "value"])
    close(nh.cpu().numpy(), fw[k + "humans_next"])
    assert nr is None


def test_state_predictor_robot_update(dev):
    fw = gio.load("forward")
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    _, _, sp = build_modules(c, dev)
    r = torch.tensor(fw["sp.robot"]).unsqueeze(1).to(dev)
    h = torch.tensor(fw["sp.humans"]).to(dev)
    with torch.no_grad():
        for a, want in zip(fw["sp.actions"], fw["sp.next_robot"]):
            nr, _ = sp((r, h), rga.ActionXY(a[0], a[1]))
            assert np.array_equal(nr.cpu().numpy().reshape(9), want)
    lin = rga.LinearStatePredictor(policy_config(), 0.25)
    nr, nh = lin((r, h), rga.ActionXY(*fw["sp.actions"][1]))
    assert np.array_equal(nh.cpu().numpy(), fw["sp.linear_next_humans"])
    assert np.array_equal(nr.cpu().numpy().reshape(9), fw["sp.linear_next_robot"])


def test_expand_level_against_oracle(dev):
    """child robots bit-exact, rewards to 1e-7, child values / predicted humans to tolerance."""
    ar = gio.load("actions_rewards")
    robot = torch.tensor(ar["rew.sweep_robot"])
    humans = torch.tensor(ar["rew.sweep_humans"])
    pol = make_mprl_policy("trained", D=1, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    P = gio.oracle_params("trained")
    cfg = orc.OracleConfig()
    acts, _ = orc.mprl_action_space(cfg, 1.0)
    for joint in (True, False):
        got = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=joint)
        with torch.no_grad():
            want = orc.mprl_expand_batched(robot, humans, P, cfg, acts, root=joint)
        assert np.array_equal(got["child_robot"].cpu().numpy(), want["child_robot"].numpy())
        assert np.abs(got["reward"].cpu().numpy() - want["reward"].numpy()).max() < 1e-7
        close(got["humans_next"].cpu().numpy(), want["next_humans"].numpy())
        close(got["child_value"].cpu().numpy(), want["child_value"].numpy())
        close(got["value1"].cpu().numpy(), want["value1"].numpy())
    # and straight against what the reference itself returned for these states
    got = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=False)
    assert np.abs(got["reward"].cpu().numpy() - ar["rew.sweep_tensor"]).max() < 1e-7
    got = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=True)
    assert np.abs(got["reward"].cpu().numpy() - ar["rew.sweep_joint"]).max() < 1e-7


@pytest.mark.parametrize("kin", ["holonomic", "unicycle"])
def test_reward_step_over_crowd_sizes_and_dense_scenes(kin, dev):
    """Round 6 rewrote the reward step's inner structure (rgl_children.h: one visit per SET BIT of the far-human mask, the python-int stop
    action evaluated by the whole wave over the humans, the goal test on squares away from the boundary, the planner's speed bound).
    None of that may move a bit: estimate_reward + compute_next_state on their own (mprl_estimate_reward_f32) against the oracle's
    vectorised transcription (pinned on the reference's own rewards by tests/test_oracle_golden.py) for crowd sizes around every
    width the masks care about -- 1, 31 / 32 / 33 (two parents per ballot up to 32), 63 / 64 (one ballot per parent), 65 (no masks:
    the per-thread path) -- with DENSE crowds (humans within 0.4-1.6 m: several near humans per parent, collisions, discomfort
    values, robots at their goals), an odd parent count (a partial last wave), tensor-born and joint-state readings."""
    cfg = orc.OracleConfig(kinematics=kin)
    acts, _ = orc.mprl_action_space(cfg, 1.0)
    pol = make_mprl_policy("trained", D=1, device=dev)
    pol.kinematics = kin
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    assert np.array_equal(ts.actions_np, acts)
    rng = np.random.RandomState(606)
    n_coll = n_disc = n_goal = 0
    for H in (1, 2, 31, 32, 33, 63, 64, 65):
        P = 37
        robot = np.zeros((P, 9), np.float32)
        robot[:, 0:2] = rng.uniform(-3, 3, (P, 2))
        robot[:, 2:4] = rng.uniform(-0.5, 0.5, (P, 2))
        robot[:, 4] = 0.3
        robot[:, 5:7] = robot[:, 0:2] + rng.uniform(-0.6, 0.6, (P, 2)) * (rng.rand(P, 1) < 0.3) + rng.uniform(-4, 4, (P, 2)) * (rng.rand(P, 1) < 0.7)
        robot[:, 7] = 1.0
        robot[:, 8] = rng.uniform(-np.pi, np.pi, P)
        humans = np.zeros((P, H, 5), np.float32)
        ang = rng.uniform(0, 2 * np.pi, (P, H))
        rad = np.where(rng.rand(P, H) < 0.25, rng.uniform(0.45, 1.6, (P, H)), rng.uniform(1.6, 6.0, (P, H)))      # a quarter of them close
        humans[:, :, 0] = robot[:, None, 0] + rad * np.cos(ang)
        humans[:, :, 1] = robot[:, None, 1] + rad * np.sin(ang)
        humans[:, :, 2:4] = rng.uniform(-1, 1, (P, H, 2))
        humans[:, :, 4] = 0.3
        rt, ht = torch.tensor(robot), torch.tensor(humans)
        want_child = orc._children_robot(rt, acts, cfg).numpy()
        for joint in (False, True):
            child, reward = ts.estimate_reward(rt.to(dev), ht.to(dev), parents_are_joint_states=joint)
            want = orc.estimate_reward_batched(rt, ht, acts, cfg, root=joint)
            got = reward.cpu().numpy()
            if kin == "holonomic":
                assert np.array_equal(child.cpu().numpy(), want_child), (H, joint)
                assert np.array_equal(got, want.astype(np.float32)), (H, joint, float(np.abs(got - want).max()))
            else:
                # the device's cos / sin are not the host libm's (last-bit differences in the heading terms): the bound of
                # test_unicycle_kinematics_both_paths; a branch of the reward may only differ where the oracle itself is within 1e-6
                # of that branch's threshold
                assert np.abs(child.cpu().numpy() - want_child).max() < 1e-6, (H, joint)
                off = np.abs(got - want) > 1e-6
                assert off.mean() < 2e-3, (H, joint, float(off.mean()))
            n_coll += int((want == -0.25).sum()); n_disc += int(((want < 0) & (want > -0.25)).sum()); n_goal += int((want == 1.0).sum())
    assert n_coll > 100 and n_disc > 100 and n_goal > 100, (n_coll, n_disc, n_goal)          # every branch of the reward is exercised
    report("reward step, %s, H in {1..65}, dense crowds: float32 rewards and child rows %s (%d collisions, %d discomfort "
           "values, %d goals among the pairs)" % (kin, "bit for bit" if kin == "holonomic" else "to 1e-6 (device cos / sin)", n_coll, n_disc, n_goal))


def test_reward_kats(dev):
    ar = gio.load("actions_rewards")
    pol = make_mprl_policy("trained", D=1, device=dev)
    for i, name in enumerate(ar["rew.names"]):
        n = int(ar["rew.n_humans"][i])
        robot = torch.tensor(ar["rew.robot"][i:i + 1].astype(np.float32)).to(dev)
        humans = torch.tensor(ar["rew.humans"][i:i + 1, :n].astype(np.float32)).to(dev)
        ts = rga.TreeSearch(pol.value_estimator, pol.state_predictor, ar["rew.actions"][i:i + 1], None)
        got = ts.expand(robot, humans, parents_are_joint_states=False)
        assert abs(float(got["reward"][0, 0]) - ar["rew.tensor"][i]) < 1e-7, name


@pytest.mark.parametrize("c", gio.plan_cases(), ids=lambda c: c["tag"])
def test_planning_kats(c, dev):
    pl = gio.load("planning")
    k = "plan.%s." % c["tag"]
    pol = make_mprl_policy(c["flavour"], c["D"], c["w"], c["clip"], c["sparse"], c["variant"], device=dev)
    R = torch.tensor(pl["plan.scene.%s.robot" % c["scene"]].astype(np.float32)).to(dev)
    Hh = torch.tensor(pl["plan.scene.%s.humans" % c["scene"]].astype(np.float32)).to(dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    out = ts.search(R, Hh, roots_are_joint_states=True)
    assert np.array_equal(out["best_action"].cpu().numpy().astype(np.int64), pl[k + "action"])
    close(out["best_value"].cpu().numpy(), pl[k + "max_value"])
    kept = out["root_kept"].cpu().numpy()
    rv = out["root_values"].cpu().numpy()
    for b in range(R.shape[0]):
        assert sorted(kept[b].tolist()) == sorted(pl[k + "kept"][b].tolist())
        order = [kept[b].tolist().index(int(i)) for i in pl[k + "kept"][b]]
        close(rv[b][order], pl[k + "root_values"][b])
    if c["clip"]:
        close(ts.level_arrays(0)["value1"].cpu().numpy(), pl[k + "clip_values"])
    assert ts.logical_value_evals_per_root() == int(pl[k + "counts"][0][0])


@pytest.mark.parametrize("c", gio.root_clip_cases(), ids=lambda c: c["tag"])
def test_root_clip_is_priced_on_the_tensor_state(c, dev):
    """VERDICT r4 weak 1 / next 1.  Upstream prices every root action twice: action_clip is handed the float32 TENSOR of the
    state (model_predictive_rl.py:216-218 -> :246-248 -> tensor_to_joint_state, state.py:82-92: float32-born scalars), the
    values of the kept actions read the float64 JointState (:226).  Fixture root_clip.npz holds what the reference computed
    INSIDE its root action_clip on 24 genuine-float64 crowded roots.  The search's level 0 against it: the selection's rewards
    (MprlLevelView::reward_clip_off) and the kept rewards BIT FOR BIT, kept sets equal, values to the network's float32 noise.
    Action 0 is the python-int stop action: float32 from end to end under numpy >= 2 (rgl_children.h stop_reward_f32)."""
    rc = gio.load("root_clip")
    k = "rootclip.%s." % c["tag"]
    pol = make_mprl_policy("trained", c["D"], c["w"], True, c["sparse"], c["variant"], device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    r64, h64 = torch.tensor(rc["rootclip.robot64"], device=dev), torch.tensor(rc["rootclip.humans64"], device=dev)
    out = ts.search(r64.float(), h64.float(), True, roots64=(r64, h64))
    lv0 = ts.level_arrays(0)
    want_clip = rc[k + "clip_rewards"].astype(np.float32)
    got_clip = lv0["reward_clip"].cpu().numpy()
    assert np.array_equal(got_clip, want_clip), np.abs(got_clip.astype(np.float64) - rc[k + "clip_rewards"]).max()
    close(lv0["value1"].cpu().numpy(), rc[k + "clip_values"], reg=REG_F32)
    kept, rv = out["root_kept"].cpu().numpy(), out["root_values"].cpu().numpy()
    rew64 = lv0["reward"].cpu().numpy()
    n_two_readings = int((rew64 != got_clip).sum())
    for b in range(kept.shape[0]):
        want_kept = rc[k + "kept"][b]
        assert sorted(kept[b].tolist()) == sorted(want_kept.tolist())
        order = [kept[b].tolist().index(int(i)) for i in want_kept]
        close(rv[b][order], rc[k + "root_values"][b], reg=REG_F32)
        assert np.array_equal(rew64[b][want_kept], rc[k + "root_rewards"][b].astype(np.float32))
    assert np.array_equal(out["best_action"].cpu().numpy().astype(np.int64), rc[k + "action"])
    assert n_two_readings > 0                                      # the two readings of the same roots do differ in float32
    # the same decisions through the public predict() (one captured search per crowd size, float64 JointStates in)
    for b in range(0, kept.shape[0], 5):
        a = pol.predict(JS(rc["rootclip.robot64"][b], rc["rootclip.humans64"][b]))
        assert a == pol.action_space[int(rc[k + "action"][b])]
    # tensor roots (V_planning's view of a state): one reading, the tensor-born one, for selection and values alike
    ts.search(r64.float(), h64.float(), False)
    lv0 = ts.level_arrays(0)
    assert "reward_clip" not in lv0 and np.array_equal(lv0["reward"].cpu().numpy(), want_clip)
    report("root action_clip (%s): %d x 81 rewards inside upstream's root clip reproduced bit for bit; %d of them differ from the "
           "float64 reading in float32" % (c["tag"], kept.shape[0], n_two_readings))


def test_exact_ties_in_action_clip_against_the_reference_fixture(dev):
    """exact ties.  Fixture root_clip.npz `tie.*`: the reference itself on a policy whose value head's last
    layer is zeroed (V = its bias for every state) with the crowd far away -- all 81 one-step values of a root are BIT-EQUAL.
    Upstream's choice among tied actions comes out of numpy's unstable argsort / argpartition (here: highest indices; sparse
    [80, 78, 76, 74], dense {78, 79, 80}) and is implementation-defined; the device search documents ITS order instead: lower
    action index first, one per group in a sparse search (rgl_tail.h).  Held here: the device reproduces the tie
    exactly (81 bit-equal values, equal to the reference's), keeps the lowest-index member of the first `width` groups / the
    `width` lowest indices, and both kept sets carry the same value -- the deviation is in naming a representative of a tie,
    never in a value."""
    rc = gio.load("root_clip")
    groups = rc["tie.groups"]
    r, h = torch.tensor(rc["tie.robot"]).to(dev), torch.tensor(rc["tie.humans"]).to(dev)
    for line in rc["tie_cases"]:
        tag, w, sparse = str(line).split("|")
        w, sparse = int(w), bool(int(sparse))
        pol = make_mprl_policy("trained", 2, w, True, sparse, device=dev)
        with torch.no_grad():
            pol.value_estimator.value_network[-1].weight.zero_()
        pol.build_action_space(1.0)
        ts = pol.tree_search()
        out = ts.search(r, h, True)
        v1 = ts.level_arrays(0)["value1"].cpu().numpy()
        ref = rc["tie.%s.clip_values" % tag]
        assert (v1 == v1[:, :1]).all() and np.array_equal(v1, ref)            # the same exact tie, bit for bit
        kept = out["root_kept"].cpu().numpy()
        ref_kept = rc["tie.%s.kept" % tag]
        if sparse:
            want = []
            for a in range(len(groups)):                                         # lowest index of each group, groups in index order
                if groups[a] not in [groups[k] for k in want]:
                    want.append(a)
                if len(want) == w:
                    break
            assert len({int(groups[a]) for a in ref_kept[0]}) == w              # upstream: one per group as well, other members
        else:
            want = list(range(w))
        for b in range(kept.shape[0]):
            assert kept[b].tolist() == want, (tag, kept[b], want)
            assert sorted(ref_kept[b].tolist()) != sorted(want)                  # the documented deviation, visible
        rv = out["root_values"].cpu().numpy()
        assert (rv == rv[:, :1]).all()                                          # every kept action of either choice has the same value
        assert int(out["best_action"][0]) == want[0] and int(rc["tie.%s.action" % tag][0]) == int(ref_kept[0][0])
    report("exact ties in action_clip: 81 bit-equal one-step values reproduced; kept representatives differ from numpy's as documented "
           "(device: lowest index per group; reference in this image: highest)")


def test_planner_step_methods_match_the_reference_planner(dev):
    """ModelPredictiveRL.estimate_reward / action_clip / V_planning and GCN.compute_reward (model_predictive_rl.py:242-357,
    multi_human_rl.py:73-96) as Python-callable methods backed by the device functions of the search (VERDICT r3 missing 5):
    rewards against the reference's own KATs (fixture actions_rewards.npz), action_clip against the kept sets / clip values the
    reference returned (planning.npz), V_planning against the oracle's restatement of the recursive planner (SeqPlanner, pinned
    through mprl_predict_sequential on the same fixtures)."""
    from relationalgraphlearning_amd import actions as A_
    ar, pl = gio.load("actions_rewards"), gio.load("planning")
    pol = make_mprl_policy("trained", D=2, w=2, clip=True, device=dev)
    pol.build_action_space(1.0)
    # estimate_reward: tensor states (fp32-difference convention) and JointStates (float64)
    for i, name in enumerate(ar["rew.names"]):
        n = int(ar["rew.n_humans"][i])
        a = A_.ActionXY(*[float(x) for x in ar["rew.actions"][i]])
        st = (torch.tensor(ar["rew.robot"][i:i + 1].astype(np.float32)).reshape(1, 1, 9),
              torch.tensor(ar["rew.humans"][i:i + 1, :n].astype(np.float32)))
        assert abs(pol.estimate_reward(st, a) - float(ar["rew.tensor"][i])) < 1e-7, name
        js = JS(ar["rew.robot"][i], ar["rew.humans"][i, :n])
        assert abs(pol.estimate_reward(js, a) - float(ar["rew.joint"][i])) < 1e-7, name
    # action_clip on the fixture's root scenes: the kept SET and the one-step values the reference computed
    k = "plan.d2w2."
    R, Hh = pl["plan.scene.s5.robot"], pl["plan.scene.s5.humans"]
    P = gio.oracle_params("trained")
    cfg = orc.OracleConfig(planning_depth=2, planning_width=2, do_action_clip=True)
    worst = 0.0
    for b in range(R.shape[0]):
        js = JS(R[b], Hh[b])
        kept = pol.action_clip(js, pol.action_space, 2)
        idx = sorted(pol.action_space.index(a) for a in kept)
        assert idx == sorted(int(x) for x in pl[k + "kept"][b]), (b, idx, pl[k + "kept"][b])
        # V_planning of the kept children against the oracle's recursive planner, depth 1 and 2
        sp = orc.SeqPlanner(P, cfg, 1.0, ([float(x) for x in R[b]], [[float(x) for x in row] for row in Hh[b]]))
        root = (torch.tensor(R[b:b + 1].astype(np.float32)).reshape(1, 1, 9), torch.tensor(Hh[b:b + 1].astype(np.float32)))
        with torch.no_grad():
            for a in kept:
                nxt = sp.SP(root, sp.actions[pol.action_space.index(a)])
                for depth in (1, 2):
                    want = float(sp.plan(nxt, depth, 2))
                    got, traj = pol.V_planning((nxt[0].to(dev), nxt[1].to(dev)), depth, 2)
                    assert tuple(got.shape) == (1, 1) and len(traj) == depth
                    assert traj[-1][1] is None and (depth == 1 or traj[0][1] in pol.action_space)
                    worst = max(worst, abs(float(got) - want))
    assert worst < 1e-6, worst
    # the reference's root value of a kept action = estimate_reward + gamma_bar * V_planning(next, D, w): rebuild one from the methods
    js = JS(R[0], Hh[0])
    out = pol.tree_search().search(torch.tensor(R[:1].astype(np.float32)).to(dev), torch.tensor(Hh[:1].astype(np.float32)).to(dev), True)
    a0 = int(out["root_kept"][0, 0])
    o = pol.tree_search().expand(torch.tensor(R[:1].astype(np.float32)).to(dev), torch.tensor(Hh[:1].astype(np.float32)).to(dev))
    nxt = (o["child_robot"][0, a0].reshape(1, 1, 9), o["humans_next"])
    v, _ = pol.V_planning(nxt, 2, 2)
    rebuilt = pol.estimate_reward(js, pol.action_space[a0]) + pol.get_normalized_gamma() * float(v)
    assert abs(rebuilt - float(out["root_values"][0, 0])) < 1e-6
    # GCN.compute_reward: every branch of multi_human_rl.py:73-96 against a float64 evaluation of the same lines
    gp = make_gcn_policy(device=dev)
    gp.time_step = 0.25

    class S(object):
        def __init__(self, **kw):
            self.__dict__.update(kw)
    rng = np.random.RandomState(3)
    for case in range(40):
        nav = S(px=rng.uniform(-2, 2), py=rng.uniform(-2, 2), radius=0.3, gx=rng.uniform(-2, 2), gy=rng.uniform(-2, 2), v_pref=1.0, theta=0.0)
        if case % 5 == 0:
            nav.gx, nav.gy = nav.px + 0.1, nav.py
        hs = [S(px=nav.px + rng.uniform(-1.5, 1.5), py=nav.py + rng.uniform(-1.5, 1.5), radius=0.3) for _ in range(1 + case % 6)]
        dmin, coll = float("inf"), False
        for h in hs:
            d = np.linalg.norm((nav.px - h.px, nav.py - h.py)) - nav.radius - h.radius
            if d < 0:
                coll = True
                break
            dmin = min(dmin, d)
        reach = np.linalg.norm((nav.px - nav.gx, nav.py - nav.gy)) < nav.radius
        want = -0.25 if coll else (1 if reach else ((dmin - 0.2) * 0.5 * 0.25 if dmin < 0.2 else 0))
        assert abs(gp.compute_reward(nav, hs) - want) < 1e-7, (case, want)
    report("planner-step methods (estimate_reward / action_clip / V_planning / compute_reward): reference KATs and kept sets exact, "
           "V_planning within %.1e of the recursive oracle" % worst)


def test_sparse_search_takes_any_group_ids(dev):
    """the select kernel used to mask group ids with `& 63`, so a direct C-ABI caller with other ids got silent
    aliasing.  It now compares ids like the reference's python set (model_predictive_rl.py:252-263): a relabelling of the groups by
    any injective map -- negative ids, ids beyond 63, ids that are all equal modulo 64 -- must not change a single kept action,
    value or decision, at depth 2 and 3; and the sparse walk against the oracle on seeded scenes."""
    pol = make_mprl_policy("trained", 3, 3, True, True, device=dev)
    pol.build_action_space(1.0)
    robot, humans = seeded_scenes(91, 48, 7)
    r, h = robot.to(dev), humans.to(dev)
    base_groups = np.asarray(pol.action_group_index, dtype=np.int64)

    def run(groups, D):
        ts = rga.TreeSearch(pol.value_estimator, pol.state_predictor, rga.actions.as_array(pol.action_space), groups,
                            pol.kinematics, pol.time_step, pol.get_normalized_gamma(), D, 3, True, True)
        o = ts.search(r, h, True)
        return [o[k].clone() for k in ("best_action", "best_value", "root_values", "root_kept")]
    for D in (2, 3):
        ref = run(base_groups, D)
        for relabel in (lambda g: 64 * g, lambda g: -7 - 1000 * g, lambda g: (g * 2654435761) % (2 ** 31) - 2 ** 30):
            got = run(relabel(base_groups), D)
            for x, y in zip(ref, got):
                assert torch.equal(x, y)
    # two groups only -> a width-3 request can keep just two actions per node: refused widths stay refused, W <= 16 is served
    ts = rga.TreeSearch(pol.value_estimator, pol.state_predictor, rga.actions.as_array(pol.action_space), base_groups,
                        pol.kinematics, pol.time_step, pol.get_normalized_gamma(), 2, 17, True, True)
    with pytest.raises(rga._native.NativeLibraryError, match="RGL_ERR_BAD_MODE"):
        ts.search(r, h, True)
    cfg = orc.OracleConfig(planning_depth=2, planning_width=3, do_action_clip=True, sparse_search=True)
    got = run(base_groups, 2)
    with torch.no_grad():
        oa, ov, orv, okept = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained"), cfg)
    close(got[1].cpu().numpy(), ov.numpy())
    assert np.array_equal(got[3].cpu().numpy().astype(np.int64), okept.numpy())
    report("sparse search, arbitrary group ids: relabelled groups bit-identical at D = 2, 3; kept sets equal to the oracle's")


def test_predict_joint_state_api(dev):
    """Policy.predict(JointState) -> ActionXY, traj of D+1 entries, errors of the reference contract."""
    pl = gio.load("planning")
    pol = make_mprl_policy("trained", 2, 2, True, device=None)
    js = JS(pl["plan.scene.s5.robot"][0], pl["plan.scene.s5.humans"][0])
    with pytest.raises(AttributeError):
        pol.predict(js)                       # device not set
    pol.set_device(dev)
    a = pol.predict(js)
    want = pol.action_space[int(pl["plan.d2w2.action"][0])]
    assert isinstance(a, rga.ActionXY) and a == want
    traj = pol.get_traj()
    assert len(traj) == 3 and traj[-1][1] is None and traj[0][1] == a
    assert traj[0][0][0].shape == (1, 1, 9) and traj[0][0][1].shape == (1, 5, 5)
    pol.set_phase("train")
    with pytest.raises(AttributeError):
        pol.predict(js)                       # epsilon not set
    pol.set_epsilon(0.0)
    assert pol.predict(js) == want
    assert pol.last_state[0].shape == (1, 9) and pol.last_state[1].shape == (5, 5)
    at_goal = JS([0, 4, 0, 0, 0.3, 0, 4, 1, 0], pl["plan.scene.s5.humans"][0])
    assert pol.predict(at_goal) == rga.ActionXY(0, 0)


def test_batched_search_keeps_its_weight_image_and_follows_the_weights(dev):
    """ABI 3: TreeSearch hands the searches a weight image packed once per parameter state (MprlPlanner.children_image).  At a
    size the fused children kernel takes (600 roots x 81 actions): unchanged weights -> no re-pack (same cache entry, same
    buffer); weights updated in place -> re-packed into the SAME buffer and the values follow, exactly as a fresh TreeSearch
    (which packs from scratch) computes them."""
    pol = make_mprl_policy("trained", 1, 1, False, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    robot, humans = seeded_scenes(77, 600, 19)
    r, h = robot.to(dev), humans.to(dev)

    def fresh():
        o = rga.TreeSearch(pol.value_estimator, pol.state_predictor, rga.actions.as_array(pol.action_space), pol.action_group_index,
                           pol.kinematics, pol.time_step, pol.get_normalized_gamma(), pol.planning_depth, pol.planning_width,
                           pol.do_action_clip, pol.sparse_search, pol.contraction_dtype).search(r, h, True)
        return o["best_action"].clone(), o["best_value"].clone()
    o1 = ts.search(r, h, True)
    a1, v1 = o1["best_action"].clone(), o1["best_value"].clone()
    ent1 = ts._images[str(dev)]
    if ent1[1] is None:                  # RGL_CHILDREN_TWO_STAGE=1 (the pair has no weight image): only "values follow the weights"
        assert os.environ.get("RGL_CHILDREN_TWO_STAGE") == "1" and not ts.last["planner"].children_image
        with torch.no_grad():
            pol.value_estimator.graph_model.w_a.add_(0.02)
        o3 = ts.search(r, h, True)
        fa, fv = fresh()
        assert torch.equal(o3["best_value"], fv) and torch.equal(o3["best_action"], fa)
        return
    assert ts.last["planner"].children_image == ent1[1].data_ptr()
    o2 = ts.search(r, h, True)
    assert ts._images[str(dev)] is ent1                                  # unchanged weights: the image was not re-packed
    assert torch.equal(o2["best_value"], v1) and torch.equal(o2["best_action"], a1)
    fa, fv = fresh()
    assert torch.equal(fv, v1) and torch.equal(fa, a1)
    with torch.no_grad():                                                # optimizer-like in-place update
        for p_ in pol.value_estimator.value_network.parameters():
            p_.mul_(1.25)
        pol.value_estimator.graph_model.w_a.add_(0.02)
    o3 = ts.search(r, h, True)
    ent3 = ts._images[str(dev)]
    assert ent3 is not ent1 and ent3[1].data_ptr() == ent1[1].data_ptr()   # re-packed, into the same device buffer
    fa, fv = fresh()
    assert torch.equal(o3["best_value"], fv) and torch.equal(o3["best_action"], fa)
    assert float((o3["best_value"] - v1).abs().max()) > 1e-3             # and the values did move
    report("weight image: packed once per parameter state, %d bytes, values follow in-place updates" % ent3[1].numel())


def test_predict_replays_a_captured_search_and_follows_the_weights(dev):
    """predict() replays a hipGraph of the whole search per crowd size (TreeSearch.decide).  The graph bakes device pointers, so:
    parameters updated IN PLACE (optimizer step, load_state_dict) must show up in the next decision (transposed copies are
    refreshed into the same buffers), and a parameter moved to new storage must trigger a re-capture."""
    pl = gio.load("planning")
    pol = make_mprl_policy("trained", 2, 2, True, device=dev)
    ts = pol.tree_search()
    scenes = [JS(pl["plan.scene.s5.robot"][b], pl["plan.scene.s5.humans"][b]) for b in range(3)]

    def eager(js):
        r = torch.tensor([[getattr(js.robot_state, k) for k in ("px", "py", "vx", "vy", "radius", "gx", "gy", "v_pref", "theta")]],
                         dtype=torch.float64, device=dev)
        h = torch.tensor([[[getattr(x, k) for k in ("px", "py", "vx", "vy", "radius")] for x in js.human_states]],
                         dtype=torch.float64, device=dev)
        o = rga.TreeSearch(pol.value_estimator, pol.state_predictor, rga.actions.as_array(pol.action_space), pol.action_group_index,
                           planning_depth=2, planning_width=2, do_action_clip=True).search(r.float(), h.float(), True, roots64=(r, h))
        return int(o["best_action"][0]), float(o["best_value"][0])
    for js in scenes:
        a = pol.predict(js)
        ea, ev = eager(js)
        assert a == pol.action_space[ea] and abs(float(ts.last["best_value"][0]) - ev) == 0.0
    assert len(ts._decisions) == 1                                   # one capture served all three decisions
    graph0 = next(iter(ts._decisions.values()))["graph"]
    with torch.no_grad():                                            # in-place update: same storage, new version
        for p_ in pol.value_estimator.value_network.parameters():
            p_.mul_(1.5)
        pol.value_estimator.graph_model.w_a.add_(0.01)
    for js in scenes:
        a = pol.predict(js)
        ea, ev = eager(js)
        assert a == pol.action_space[ea] and abs(float(ts.last["best_value"][0]) - ev) == 0.0
    assert next(iter(ts._decisions.values()))["graph"] is graph0      # still the first capture
    sd = pol.get_state_dict()
    sd["value_network"] = {k: v * 0.5 for k, v in sd["value_network"].items()}
    pol.load_state_dict(sd)                                          # copy_ into the same parameters
    a = pol.predict(scenes[0])
    assert a == pol.action_space[eager(scenes[0])[0]]
    gm = pol.value_estimator.graph_model
    gm.w_a = torch.nn.Parameter(gm.w_a.detach().clone() * 1.1)     # new storage: the captured pointers are stale
    a = pol.predict(scenes[1])
    ea, ev = eager(scenes[1])
    assert a == pol.action_space[ea] and abs(float(ts.last["best_value"][0]) - ev) == 0.0
    assert next(iter(ts._decisions.values()))["graph"] is not graph0
    # a larger crowd gets its own capture; the trajectory is read back lazily and belongs to the latest decision
    big = JS(pl["plan.scene.s5.robot"][0], np.concatenate([pl["plan.scene.s5.humans"][0], pl["plan.scene.s5.humans"][1] + 0.37]))
    a = pol.predict(big)
    assert len(ts._decisions) == 2 and pol._traj is None
    traj = pol.get_traj()
    assert len(traj) == 3 and traj[0][1] == a and traj[0][0][1].shape == (1, 10, 5) and pol.traj is traj


def test_root_reward_reads_the_float64_joint_state(dev):
    """estimate_reward of the ROOT is evaluated on the simulator's float64 state in the reference (model_predictive_rl.py:226);
    rounding the root to fp32 first moves clearances by ~1e-7 and flips threshold cases.  A human placed so that its
    clearance under action 0 is -2e-9 in float64 (a collision) but +3e-8 after fp32 rounding."""
    pol = make_mprl_policy("trained", 1, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    acts = rga.actions.as_array(pol.action_space)
    robot = np.array([[0.1, -0.2, 0.0, 0.0, 0.3, 0.0, 4.0, 1.0, np.pi / 2]], np.float64)
    found = None
    for k in range(2000):                                     # search a placement where fp32 rounding crosses the boundary
        cand = np.array([[[robot[0, 0] + 0.6 - 2e-9, -0.2 + k * 1e-4, 0.0, 0.0, 0.3]]], np.float64)
        cand[0, 0, 0] = robot[0, 0] + np.sqrt(max(0.0, (0.6 - 2e-9) ** 2 - (k * 1e-4) ** 2))
        d64 = np.hypot(cand[0, 0, 0] - robot[0, 0], cand[0, 0, 1] - robot[0, 1]) - 0.6
        r32, c32 = robot.astype(np.float32).astype(np.float64), cand.astype(np.float32).astype(np.float64)
        d32 = np.hypot(c32[0, 0, 0] - r32[0, 0], c32[0, 0, 1] - r32[0, 1]) - np.float64(np.float32(0.3)) * 2
        if d64 < 0 <= d32:
            found = cand
            break
    assert found is not None
    r64, h64 = torch.tensor(robot, device=dev), torch.tensor(found, device=dev)
    cfg = orc.OracleConfig()
    want = [orc.estimate_reward([float(x) for x in robot[0]], [[float(x) for x in found[0, 0]]], a, cfg) for a in acts]
    ts.search(r64.float(), h64.float(), True, roots64=(r64, h64))
    got64 = ts.level_arrays(0)["reward"][0].cpu().numpy()
    assert np.abs(got64 - np.array(want, np.float64)).max() < 1e-7
    ts.search(r64.float(), h64.float(), True)                 # without the float64 state the stop action looks collision-free
    got32 = ts.level_arrays(0)["reward"][0].cpu().numpy()
    assert want[0] == -0.25 and got64[0] == -0.25 and got32[0] != -0.25


def test_path_g_query_env(dev):
    """query_env=True (multi_human_rl.py:43-44): per action, the next human states and the reward come from the simulator's
    one-step lookahead.  Against a plain restatement on the same simulator (one action at a time through onestep_lookahead)."""
    from relationalgraphlearning_amd.sim import BatchedCrowdSim
    pol = make_gcn_policy(device=dev)
    pol.query_env = True
    sim = BatchedCrowdSim(dev)
    sim.reset("test", [3])
    for _ in range(6):                                        # a few steps in, so humans move
        sim.step(torch.tensor([[0.0, 0.8]], dtype=torch.float64))
    pol.set_env(sim)
    robot = sim.robot[0].cpu().numpy()
    js = JS(robot, sim.humans[0].cpu().numpy())
    with pytest.raises(AttributeError):
        pol.set_env(None)
        pol.predict(js)
    pol.set_env(sim)
    a = pol.predict(js)
    assert len(pol.action_values) == len(pol.action_space)
    # restatement: action by action
    table = rga.actions.as_array(pol.action_space)
    vals = []
    for i in range(table.shape[0]):
        nh, rew = sim.onestep_lookahead_actions(table[i:i + 1])
        (obs_r, obs_h), r1, _, _ = sim.onestep_lookahead(torch.tensor(table[i:i + 1]))
        assert abs(float(r1[0]) - float(rew[0])) == 0.0       # same reward as the simulator's own lookahead
        nr = list(robot)
        nr[0], nr[1], nr[2], nr[3] = robot[0] + table[i, 0] * 0.25, robot[1] + table[i, 1] * 0.25, table[i, 0], table[i, 1]
        joint = torch.tensor([nr + [float(x) for x in row] for row in nh[0].cpu().numpy()], dtype=torch.float32, device=dev)
        with torch.no_grad():
            v = pol.model(rga.rotate(joint).unsqueeze(0))
        vals.append(float(rew[0]) + pow(0.9, 0.25 * robot[7]) * float(v[0, 0]))
    assert np.abs(np.array(vals) - np.array(pol.action_values)).max() < 1e-5
    assert a == pol.action_space[int(np.argmax(vals))]


def test_checkpoint_roundtrip(dev, tmp_path):
    pol = make_mprl_policy("trained", 1, device=dev)
    f = str(tmp_path / "rl_model.pth")
    pol.save_model(f)
    ck = torch.load(f)
    assert set(ck) == {"graph_model1", "graph_model2", "value_network", "motion_predictor"}
    assert set(ck["graph_model1"]) == {"w_a", "w_r.0.weight", "w_r.0.bias", "w_r.2.weight", "w_r.2.bias",
                                        "w_h.0.weight", "w_h.0.bias", "w_h.2.weight", "w_h.2.bias", "Ws.0", "Ws.1"}
    pol2 = make_mprl_policy("rand", 1, device=dev)
    pol2.load_model(f)
    pl = gio.load("planning")
    R = torch.tensor(pl["plan.scene.s5.robot"].astype(np.float32)).to(dev)
    Hh = torch.tensor(pl["plan.scene.s5.humans"].astype(np.float32)).to(dev)
    a1, v1 = pol.predict_batch(R, Hh)
    a2, v2 = pol2.predict_batch(R, Hh)
    assert torch.equal(a1, a2) and torch.equal(v1, v2)       # cache invalidation on load_state_dict works


@pytest.mark.parametrize("H,L,flavour,skip,P", [(19, 2, "trained", True, 5), (19, 2, "rand", True, 3), (5, 2, "trained", True, 7),
                                                 (4, 2, "trained", False, 4), (1, 2, "trained", True, 3),
                                                 (19, 1, "trained", True, 4), (19, 3, "trained", True, 3),
                                                 (49, 3, "trained", True, 2), (49, 2, "rand", False, 2),
                                                 (63, 2, "trained", True, 2), (30, 3, "rand", True, 2),
                                                 (7, 3, "trained", False, 5), (31, 2, "trained", True, 2),
                                                 (15, 2, "rand", True, 3), (12, 2, "trained", False, 3),
                                                 (7, 2, "rand", True, 4), (31, 1, "trained", True, 2),
                                                 (12, 3, "trained", True, 2), (9, 1, "rand", True, 3),
                                                 (15, 3, "trained", False, 2), (32, 2, "trained", True, 3),
                                                 (39, 3, "rand", False, 2), (54, 3, "trained", True, 2),
                                                 (46, 3, "trained", False, 9), (3, 3, "trained", True, 4),
                                                 (47, 2, "trained", True, 2), (33, 3, "rand", True, 2)])
def test_value_children_mfma_path_vs_general_kernel(H, L, flavour, skip, P, dev):
    """The two-stage MFMA path (mprl_value_children_f32) against the general kernel (module forward)
    and the oracle, on children that share their crowd exactly like the rollout's siblings."""
    pol = make_mprl_policy(flavour, 1, L=L, skip=skip, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    A = ts.num_actions
    robot, humans = seeded_scenes(300 + H + L, P, H)
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())                  # (P,A,9)
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu()
    with torch.no_grad():
        general = pol.value_estimator((cr.reshape(P * A, 1, 9).to(dev),
                                       humans[:, None].expand(P, A, H, 5).reshape(P * A, H, 5).contiguous().to(dev)))
        cfg = orc.OracleConfig(num_layer=L, skip_connection=skip)
        Pm = gio.oracle_params(flavour, L)
        want = orc.value_estimator_forward(cr.reshape(P * A, 1, 9), humans[:, None].expand(P, A, H, 5).reshape(P * A, H, 5),
                                           Pm.ve_graph, Pm.value_network, cfg)
    close(general.cpu().numpy().reshape(P, A), want.numpy().reshape(P, A))
    close(got.numpy(), want.numpy().reshape(P, A))
    if flavour == "rand":     # the tolerance is relative to max |V| for these weights: log what that is in absolute terms
        report("value of children, raw random-init weights H=%d L=%d: max |dV| = %.2e absolute at max |V| = %.1f (north star: 1e-4 "
               "absolute on trained-scale values)" % (H, L, float(np.abs(got.numpy() - want.numpy().reshape(P, A)).max()),
                                                      float(want.abs().max())))


def test_fused_kernel_random_shapes_vs_general_kernel(dev):
    """Differential test of the fused children kernel (forced: RGL_CHILDREN_FUSED=1, child process) against the general VALU kernel
    (the module forward) over seeded random shapes: crowd size 2..32, 1..400 parents, action tables of 3..115 actions (full tiles
    only, partial tile only, both), five similarity functions, skip on / off.  Exercises every register bucket, both row-pass forms,
    the work-item plan (tiles per item, item order, snake passes) and the in-launch scoring of the partial tiles' rows."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from tests import golden_io as gio
from tests.test_gpu_parity import seeded_scenes
from oracle import rgl_oracle as orc
from relationalgraphlearning_amd.config import policy_config
import relationalgraphlearning_amd as rga
dev = torch.device("cuda:0")
rng = np.random.RandomState(20260927)
sims = ["embedded_gaussian", "embedded_gaussian", "embedded_gaussian", "gaussian", "squared", "equal_attention", "diagonal"]
worst = 0.0
for case in range(36):
    H = int(rng.randint(1, 32)); P = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 257, 400]))
    speeds = int(rng.randint(1, 7)); rots = int(rng.randint(2, 20)); skip = bool(rng.randint(2)); sim = sims[rng.randint(len(sims))]
    if case == 0: speeds, rots = 3, 5            # 16 actions: one full tile, no partial
    if case == 1: speeds, rots = 1, 2            # 3 actions: partial tile only
    cfgp = policy_config("model_predictive_rl", action_space__speed_samples=speeds, action_space__rotation_samples=rots,
                         gcn__similarity_function=sim, gcn__skip_connection=skip)
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfgp)
    pol.load_state_dict(gio.checkpoint("trained", 2, "separate", sim))
    pol.set_time_step(0.25); pol.set_phase("test"); pol.set_device(dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    A = ts.num_actions
    assert A == speeds * rots + 1
    robot, humans = seeded_scenes(5000 + case, P, H)
    g = torch.Generator().manual_seed(case)
    cr = robot[:, None, :].expand(P, A, 9).clone()
    cr[:, :, :4] += 0.3 * torch.randn(P, A, 4, generator=g)            # children: perturbed position / velocity of the parent
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    with torch.no_grad():
        ref = pol.value_estimator((cr.reshape(P * A, 1, 9).to(dev),
                                   humans[:, None].expand(P, A, H, 5).reshape(P * A, H, 5).contiguous().to(dev))).cpu().numpy().reshape(P, A)
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    worst = max(worst, err)
    assert np.isfinite(got).all() and err < 2e-5, (case, H, P, A, sim, skip, err)
print("OK worst %.2e" % worst)
'''
    env = dict(os.environ, RGL_CHILDREN_FUSED="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
    report("fused kernel vs general kernel over 36 random shapes: " + out.stdout.strip().splitlines()[-1])


def test_scene_kernel_split_and_unsplit_forced(dev):
    """The one-wave-per-scene kernel splits a scene over its column tiles' waves below a size threshold (rgl_scene.hip, SPLIT).  Both
    organisations over the state-predictor, similarity / layerwise and forward-KAT tests, whatever their batch sizes: the threshold is
    forced to "always" and to "never" in child processes (the switch is read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for below in ("1000000", "0"):
        env = dict(os.environ, RGL_SCENE_SPLIT_BELOW=below)
        out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                              "-k", "state_predictor or cosine_concatenation or forward_kats or path_g_value_network or path_g_at_size"],
                             cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    report("scene kernel: split / unsplit forced over the predictor, similarity and forward-KAT tests: green")


def test_tile_pipeline_forced_over_the_module_and_training_tests(dev):
    """The tile kernels of rgl_backward_mfma.hip are chosen by shape (forward: models outside the shipped shapes) and by batch size
    (backward: from 256 scenes).  Forced on for everything they cover -- RGL_TILES_FORWARD=2, RGL_BACKWARD_MFMA=1 -- the forward KATs,
    the state-predictor tests, the non-default heads, every gradient test and the reference-trainer fixtures must stay green (this
    run found the 6-layer head whose weights alone exceed one half of a CU's LDS)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGL_TILES_FORWARD="2", RGL_BACKWARD_MFMA="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                          "-k", "forward_kats or state_predictor or non_default_value_heads or gradients or training_against or "
                                "training_step_matches or target_model"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    report("tile kernels forced over the forward KATs, predictor, head, gradient and trainer-fixture tests: " + out.stdout.strip().splitlines()[-1])
    # round 6: the wide heads' rows run head_rows_kernel (weights straight from L2); its A/B partner -- mlp_rows_kernel with one tile per
    # workgroup, weights staged in LDS, the form batches beyond 1024 tiles still take -- over the head and gradient tests as well
    env["RGL_HEAD_ROWS_DIRECT"] = "0"
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                          "-k", "non_default_value_heads or gradients or training_step_matches"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    report("the same with the heads' rows on the staged kernel (RGL_HEAD_ROWS_DIRECT=0): " + out.stdout.strip().splitlines()[-1])


def test_tile_kernel_variant_forced(dev):
    """With RGL_CHILDREN_TILE_KERNEL=1 the MFMA tile kernel also handles what the shared-crowd kernels (rank-1: L=2,
    N<=32; deep: L in {2,3}, N<=56) take by default.  The switch is read once per process, so this runs in a child process."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from oracle import rgl_oracle as orc
from tests import golden_io as gio
from tests.helpers import make_mprl_policy
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
for H, skip, L in ((19, True, 2), (5, True, 2), (9, False, 2), (1, True, 2), (12, True, 2), (15, False, 2), (16, True, 2),
                   (31, True, 2), (49, True, 3), (19, False, 3), (40, True, 2), (7, True, 3)):
    pol = make_mprl_policy("trained", 1, L=L, skip=skip, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    A = ts.num_actions
    robot, humans = seeded_scenes(700 + H, 3, H)
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    Pm = gio.oracle_params("trained", L)
    with torch.no_grad():
        want = orc.value_estimator_forward(cr.reshape(3 * A, 1, 9), humans[:, None].expand(3, A, H, 5).reshape(3 * A, H, 5),
                                           Pm.ve_graph, Pm.value_network,
                                           orc.OracleConfig(num_layer=L, skip_connection=skip)).numpy().reshape(3, A)
    err = np.abs(got - want).max()
    assert err < 1e-4 * max(1.0, np.abs(want).max()), (H, skip, L, err)
print("OK")
'''
    env = dict(os.environ, RGL_CHILDREN_TILE_KERNEL="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_fused_tile_kernel_forced(dev):
    """The fused tile-stream kernel (rgl_fused.hip) is picked for large launches only; RGL_CHILDREN_FUSED=1 forces it for the small,
    odd-sized launches here (partial tiles, every register bucket, skip on/off, the plain-weight similarities, action tables whose
    size is / is not a multiple of 16) -- against the oracle, and whole searches against the two-stage pair.  Child process: the
    switch is read once."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from oracle import rgl_oracle as orc
from tests import golden_io as gio
from tests.helpers import make_mprl_policy
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
worst = 0.0
for H, skip, P, sim, flavour in ((19, True, 5, "embedded_gaussian", "trained"), (19, True, 67, "embedded_gaussian", "rand"),
                                 (5, True, 7, "embedded_gaussian", "trained"), (4, False, 33, "embedded_gaussian", "trained"),
                                 (1, True, 3, "embedded_gaussian", "trained"), (15, False, 9, "embedded_gaussian", "rand"),
                                 (16, True, 3, "embedded_gaussian", "trained"), (31, True, 21, "embedded_gaussian", "trained"),
                                 (12, True, 130, "gaussian", "trained"), (19, False, 6, "squared", "trained"),
                                 (7, True, 5, "equal_attention", "trained"), (23, True, 4, "diagonal", "trained"),
                                 # mid-size odd parent counts: several tiles per work item, several dealing passes, the partial
                                 # tiles' rows spread over all workgroups
                                 (19, True, 701, "embedded_gaussian", "trained"), (5, False, 1501, "embedded_gaussian", "trained"),
                                 (9, True, 2311, "embedded_gaussian", "trained")):
    pol = make_mprl_policy(flavour, 1, L=2, skip=skip, similarity=sim, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    A = ts.num_actions
    robot, humans = seeded_scenes(900 + H, P, H)
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    Pm = gio.oracle_params(flavour, 2, similarity=sim)
    with torch.no_grad():
        want = orc.value_estimator_forward(cr.reshape(P * A, 1, 9), humans[:, None].expand(P, A, H, 5).reshape(P * A, H, 5),
                                           Pm.ve_graph, Pm.value_network,
                                           orc.OracleConfig(num_layer=2, skip_connection=skip, similarity=sim)).numpy().reshape(P, A)
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    worst = max(worst, err)
    assert err < 1e-4, (H, skip, P, sim, err)
# sharp attention: w_a scaled so that the similarities span hundreds -- the row weights' exponent differences (msh_i - S_i0) run
# past both ends of the fp32 exp range; the packed row pass caps r = alpha / beta at e^60 (rgl_fused.hip), the oracle softmax is plain
import copy
for scale, H, P in ((25.0, 19, 40), (-40.0, 9, 33), (300.0, 5, 17)):
    ck = copy.deepcopy(gio.checkpoint("trained", 2))
    ck["graph_model1"]["w_a"] = ck["graph_model1"]["w_a"] * scale
    pol = make_mprl_policy("trained", 1, L=2, device=dev)
    pol.load_state_dict(ck)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    A = ts.num_actions
    robot, humans = seeded_scenes(1200 + H, P, H)
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    Pm = orc.MprlParams.from_checkpoint(ck)
    with torch.no_grad():
        want = orc.value_estimator_forward(cr.reshape(P * A, 1, 9), humans[:, None].expand(P, A, H, 5).reshape(P * A, H, 5),
                                           Pm.ve_graph, Pm.value_network, orc.OracleConfig(num_layer=2)).numpy().reshape(P, A)
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    worst = max(worst, err)
    assert np.isfinite(got).all() and err < 1e-4, (scale, H, P, err)
# whole depth-2 searches: action tables of 25 (1 full tile + 9), 96 (6 full tiles, no partial), 97 (6 + 1) and 5 (partial only)
from relationalgraphlearning_amd.config import policy_config
import relationalgraphlearning_amd as rga
for speeds, rots, H in ((3, 8, 19), (5, 19, 5), (6, 16, 19), (1, 4, 9)):
    cfgp = policy_config("model_predictive_rl", action_space__speed_samples=speeds, action_space__rotation_samples=rots,
                         model_predictive_rl__planning_depth=2, model_predictive_rl__planning_width=2,
                         model_predictive_rl__do_action_clip=True)
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfgp)
    pol.load_state_dict(gio.checkpoint("trained", 2))
    pol.set_time_step(0.25)
    pol.set_phase("test")
    pol.set_device(dev)
    robot, humans = seeded_scenes(950 + speeds * rots, 11, H)
    cfg = orc.OracleConfig(speed_samples=speeds, rotation_samples=rots, planning_depth=2, planning_width=2, do_action_clip=True)
    with torch.no_grad():
        oa, ov, _, _ = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained", 2), cfg)
    a, v = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    assert pol.tree_search().num_actions == speeds * rots + 1
    err = float((v.cpu() - ov).abs().max())
    assert err < 1e-4, (speeds, rots, H, err)
# a whole depth-3 search through the fused kernel (packed images once per search) vs the oracle
pol = make_mprl_policy("trained", D=3, w=2, clip=True, device=dev)
robot, humans = seeded_scenes(77, 24, 19)
a, v = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
cfg = orc.OracleConfig(planning_depth=3, planning_width=2, do_action_clip=True)
with torch.no_grad():
    oa, ov, _, _ = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained"), cfg)
assert np.array_equal(a.cpu().numpy().astype(np.int64), oa.numpy()), (a, oa)
assert float((v.cpu() - ov).abs().max()) < 1e-4
print("OK worst relative error %.2e" % worst)
'''
    env = dict(os.environ, RGL_CHILDREN_FUSED="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
    print(out.stdout.strip())


@pytest.mark.parametrize("H,L,flavour,skip,P", [(1, 2, "trained", True, 5), (15, 2, "trained", True, 4), (16, 2, "rand", True, 3),
                                                 (31, 1, "trained", True, 3), (49, 3, "trained", True, 2),
                                                 (63, 2, "trained", False, 2), (19, 3, "rand", True, 3), (7, 4, "trained", True, 3)])
def test_state_predictor_mfma_path_vs_oracle(H, L, flavour, skip, P, dev):
    """mprl_expand_f32's predicted humans (row-MLP embeddings + scene_graph_kernel with the motion head fused)."""
    pol = make_mprl_policy(flavour, 1, L=L, skip=skip, device=dev)
    pol.build_action_space(1.0)
    robot, humans = seeded_scenes(500 + H + L, P, H)
    got = pol.tree_search().expand(robot.to(dev), humans.to(dev))["humans_next"].cpu().numpy()
    Pm = gio.oracle_params(flavour, L)
    cfg = orc.OracleConfig(num_layer=L, skip_connection=skip)
    with torch.no_grad():
        want = orc.state_predictor_humans(robot[:, None, :], humans, Pm.sp_graph, Pm.motion_predictor, cfg).numpy()
    close(got, want)


# ---------------------------------------------------------------------------------------------------
# larger seeded batches against the batched oracle
# ---------------------------------------------------------------------------------------------------
def seeded_scenes(seed, B, H):
    rng = np.random.RandomState(seed)
    robot = np.zeros((B, 9), np.float32)
    humans = np.zeros((B, H, 5), np.float32)
    ang = rng.uniform(0, 2 * np.pi, B)
    robot[:, 0], robot[:, 1] = 4 * np.cos(ang), 4 * np.sin(ang)
    robot[:, 2:4] = rng.uniform(-0.7, 0.7, (B, 2))
    robot[:, 4] = 0.3
    robot[:, 5], robot[:, 6] = -4 * np.cos(ang), -4 * np.sin(ang)
    robot[:, 7] = 1.0
    robot[:, 8] = np.pi / 2
    humans[:, :, 0:2] = rng.uniform(-5, 5, (B, H, 2))
    humans[:, :, 2:4] = rng.uniform(-1, 1, (B, H, 2))
    humans[:, :, 4] = 0.3
    return torch.tensor(robot), torch.tensor(humans)


@pytest.mark.parametrize("H,D,w,clip,B,L", [(4, 1, 1, False, 96, 2), (5, 1, 1, False, 64, 2), (19, 2, 2, True, 24, 2),
                                             (19, 3, 2, True, 6, 2), (49, 2, 2, True, 4, 3), (3, 2, 81, False, 2, 2)])
def test_tree_vs_batched_oracle(H, D, w, clip, B, L, dev):
    robot, humans = seeded_scenes(100 + H + D, B, H)
    pol = make_mprl_policy("trained", D, w, clip, L=L, device=dev)
    cfg = orc.OracleConfig(num_layer=L, planning_depth=D, planning_width=w, do_action_clip=clip)
    with torch.no_grad():
        oa, ov, orv, okept, lv = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained", L), cfg, return_levels=True)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    out = pol.tree_search().last
    close(val.cpu().numpy(), ov.numpy())
    close(out["root_values"].cpu().numpy(), orv.numpy())
    check_decisions("tree vs batched oracle H=%d D=%d w=%d B=%d" % (H, D, w, B), act, val, (oa, ov, orv, okept), lv)


@pytest.mark.parametrize("contraction", ["f32", "bf16x6"])
@pytest.mark.parametrize("H,D,w,clip,B,kin", [(19, 2, 2, True, 24, "holonomic"), (5, 1, 1, False, 64, "holonomic"),
                                               (19, 2, 2, True, 12, "unicycle")])
def test_tree_vs_batched_oracle_over_seeds(H, D, w, clip, B, kin, contraction, dev):
    """test_tree_vs_batched_oracle's check over eight more seeded scene sets per shape, in both arithmetic modes of the search (round
    6: every other search test draws ONE set; a top-k over near-ties is where a rare seed would show)."""
    pol = make_mprl_policy("trained", D, w, clip, L=2, device=dev, kinematics=kin)
    pol.contraction_dtype = contraction
    cfg = orc.OracleConfig(num_layer=2, planning_depth=D, planning_width=w, do_action_clip=clip, kinematics=kin)
    worst, differ = 0.0, 0
    for seed in range(2001, 2009):
        robot, humans = seeded_scenes(seed + H, B, H)
        with torch.no_grad():
            oa, ov, orv, okept, lv = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained", 2), cfg, return_levels=True)
        act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
        close(val.cpu().numpy(), ov.numpy())
        differ += check_decisions("seed %d H=%d D=%d %s %s" % (seed, H, D, kin, contraction), act, val, (oa, ov, orv, okept), lv)
        worst = max(worst, float((val.cpu() - ov).abs().max()))
    report("tree vs batched oracle over 8 seeds, H=%d D=%d %s, %s kernels: %d of %d decisions differ (ties in the oracle), max |dV| %.2e"
           % (H, D, kin, contraction, differ, 8 * B, worst))


@pytest.mark.parametrize("H,sim,L,layerwise", [(64, "embedded_gaussian", 2, False), (99, "embedded_gaussian", 2, False),
                                               (127, "embedded_gaussian", 3, False), (80, "cosine_softmax", 2, True),
                                               (99, "squared", 2, False), (70, "concatenation", 2, False)])
def test_crowds_beyond_64_nodes(H, sim, L, layerwise, dev):
    """VERDICT r2 missing 3 / next 9: the reference has no crowd-size cap (graph_model.py:99-130, test.py --human_num); the ABI
    limit is 128 nodes since round 3.  Module forwards at N = 65 / 100 / 128 against the oracle: RGL.forward with node features and
    adjacency (general kernel), ValueEstimator / StatePredictor forwards (one-wave-per-scene MFMA kernel, eight column tiles split
    over a workgroup's waves; concatenation beyond 64 nodes stays on the general kernel)."""
    c = dict(L=L, sim=sim, layerwise=layerwise, skip=True, flavour="trained")
    g1, ve, sp = build_modules(c, dev)
    B = 6
    robot, humans = seeded_scenes(700 + H, B, H)
    state = (robot.unsqueeze(1).to(dev), humans.to(dev))
    with torch.no_grad():
        H_L = g1(state)
        A0 = g1.A
        val = ve(state)
        _, nh = sp(state, None)
    cfg = orc.OracleConfig(num_layer=L, similarity=sim, layerwise_graph=layerwise, skip_connection=True)
    Pm = gio.oracle_params("trained", L, similarity=sim)
    with torch.no_grad():
        oH, oA = orc.rgl_forward(robot[:, None, :], humans, Pm.ve_graph, cfg)
        ov = orc.value_estimator_forward(robot[:, None, :], humans, Pm.ve_graph, Pm.value_network, cfg)
        onh = orc.state_predictor_humans(robot[:, None, :], humans, Pm.sp_graph, Pm.motion_predictor, cfg)
    e1 = close(H_L.cpu().numpy(), oH.numpy())
    close(A0, oA[0].numpy())
    e2 = close(val.cpu().numpy(), ov.numpy())
    e3 = close(nh.cpu().numpy(), onh.numpy())
    report("N = %d nodes (%s, L=%d%s): |dH| %.1e |dV| %.1e |d humans'| %.1e vs the oracle" % (
        H + 1, sim, L, ", layerwise" if layerwise else "", e1, e2, e3))


@pytest.mark.parametrize("H,D,B", [(79, 2, 6), (127, 1, 4)])
def test_rollout_beyond_64_nodes(H, D, B, dev):
    """A whole search with a dense crowd of 80 / 128 agents: state predictor and every child's graph on the split scene kernel,
    reward / selection / back-up unchanged; against the batched oracle."""
    robot, humans = seeded_scenes(800 + H, B, H)
    pol = make_mprl_policy("trained", D, 2, D > 1, device=dev)
    cfg = orc.OracleConfig(planning_depth=D, planning_width=2, do_action_clip=D > 1)
    with torch.no_grad():
        oa, ov, orv, okept, lv = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained"), cfg, return_levels=True)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    close(val.cpu().numpy(), ov.numpy())
    check_decisions("rollout with N=%d nodes D=%d B=%d" % (H + 1, D, B), act, val, (oa, ov, orv, okept), lv)
    # path G at the same crowd size
    g = make_gcn_policy(device=dev)
    g.build_action_space(1.0)
    vals, best = g.gcn_search().search(robot.to(dev), humans.to(dev))
    ob, ovg = orc.gcn_predict_batched(robot.numpy(), humans.numpy(), gio.path_g_sd(), orc.OracleConfig())
    close(vals.cpu().numpy().astype(np.float64), ovg)
    if H == 127:                                                 # one more agent is beyond the ABI limit: refused, loudly
        r2, h2 = seeded_scenes(1, 2, 128)
        with pytest.raises(rga._native.NativeLibraryError, match="RGL_ERR_BAD_SHAPE|rejected"):
            pol.predict_batch(r2.to(dev), h2.to(dev))


@pytest.mark.parametrize("sim", ["gaussian", "squared", "equal_attention", "diagonal"])
@pytest.mark.parametrize("H,L,D,B", [(19, 2, 2, 12), (5, 2, 1, 40), (49, 3, 2, 3), (40, 2, 1, 4), (19, 1, 1, 8), (12, 3, 1, 5)])
def test_other_similarities_on_the_mfma_path(sim, H, L, D, B, dev):
    """The similarity functions whose row normalisation is a per-row sum -- gaussian (S = X X^T, softmax), squared
    (S^2 / sum S^2), equal_attention (1/N), diagonal (I) -- run on the shared-crowd MFMA kernels (Wa = I built in LDS):
    children's values (rank-1 / deep kernel by shape; L = 1 goes to the tile kernel for gaussian and to the general kernel
    otherwise), state predictor and the whole search against the oracle."""
    pol = make_mprl_policy("trained", D, 2, D > 1, L=L, similarity=sim, device=dev)
    pol.build_action_space(1.0)
    cfg = orc.OracleConfig(num_layer=L, similarity=sim, planning_depth=D, planning_width=2, do_action_clip=D > 1)
    Pm = gio.oracle_params("trained", L, similarity=sim)
    robot, humans = seeded_scenes(500 + H + L, B, H)
    ts = pol.tree_search()
    A = ts.num_actions
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    with torch.no_grad():
        want = orc.value_estimator_forward(cr.reshape(B * A, 1, 9), humans[:, None].expand(B, A, H, 5).reshape(B * A, H, 5),
                                           Pm.ve_graph, Pm.value_network, cfg).numpy().reshape(B, A)
        hn = orc.state_predictor_humans(robot[:, None], humans, Pm.sp_graph, Pm.motion_predictor, cfg)
        oa, ov, orv, okept, lv = orc.mprl_predict_batched(robot, humans, Pm, cfg, return_levels=True)
    close(got, want)
    ex = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=True)
    close(ex["humans_next"].cpu().numpy(), hn.numpy())
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    close(val.cpu().numpy(), ov.numpy())
    check_decisions("similarity %s H=%d L=%d D=%d" % (sim, H, L, D), act, val, (oa, ov, orv, okept), lv)


def test_cosine_concatenation_and_layerwise_rollouts_on_the_scene_kernel(dev):
    """cosine / cosine_softmax / concatenation and layerwise graphs (graph_model.py:70-85,119-122) have no shared-crowd form; their children and
    state predictor run on the one-wave-per-scene MFMA kernel (value rows + robot_head_kernel).  With RGL_REQUIRE_MFMA_CHILDREN=1 the
    library refuses to fall back to the general VALU kernel, so passing here proves the MFMA path ran.  Values, predicted humans
    and whole searches against the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from oracle import rgl_oracle as orc
from tests import golden_io as gio
from tests.helpers import make_mprl_policy
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
worst = 0.0
for sim, layerwise, H, L, D, B, skip in (("cosine", False, 19, 2, 2, 6, True), ("cosine_softmax", False, 19, 2, 2, 6, True),
                                         ("cosine", False, 5, 2, 1, 17, False), ("cosine_softmax", False, 40, 3, 1, 3, True),
                                         ("cosine", True, 19, 2, 1, 5, True), ("cosine_softmax", True, 5, 3, 2, 7, True),
                                         ("embedded_gaussian", True, 19, 2, 2, 6, True), ("gaussian", True, 5, 2, 1, 9, False),
                                         ("squared", True, 19, 3, 1, 4, True), ("embedded_gaussian", True, 49, 2, 1, 2, True),
                                         ("equal_attention", True, 7, 2, 1, 5, True), ("embedded_gaussian", True, 5, 1, 1, 8, True),
                                         ("concatenation", False, 19, 2, 2, 5, True), ("concatenation", False, 5, 2, 1, 11, False),
                                         ("concatenation", True, 19, 2, 1, 3, True), ("concatenation", True, 7, 3, 2, 4, True),
                                         ("concatenation", False, 33, 2, 1, 2, True)):
    pol = make_mprl_policy("trained", D, 2, D > 1, L=L, similarity=sim, layerwise=layerwise, skip=skip, device=dev)
    pol.build_action_space(1.0)
    cfg = orc.OracleConfig(num_layer=L, similarity=sim, layerwise_graph=layerwise, skip_connection=skip, planning_depth=D,
                           planning_width=2, do_action_clip=D > 1)
    Pm = gio.oracle_params("trained", L, similarity=sim)
    robot, humans = seeded_scenes(640 + H + L, B, H)
    ts = pol.tree_search()
    A = ts.num_actions
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    with torch.no_grad():
        want = orc.value_estimator_forward(cr.reshape(B * A, 1, 9), humans[:, None].expand(B, A, H, 5).reshape(B * A, H, 5),
                                           Pm.ve_graph, Pm.value_network, cfg).numpy().reshape(B, A)
        hn = orc.state_predictor_humans(robot[:, None], humans, Pm.sp_graph, Pm.motion_predictor, cfg).numpy()
        oa, ov, orv, okept = orc.mprl_predict_batched(robot, humans, Pm, cfg)
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    assert err < 1e-4, (sim, layerwise, H, L, "children", err)
    ex = ts.expand(robot.to(dev), humans.to(dev), parents_are_joint_states=True)
    e2 = np.abs(ex["humans_next"].cpu().numpy() - hn).max() / max(1.0, np.abs(hn).max())
    assert e2 < 1e-4, (sim, layerwise, H, L, "state predictor", e2)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    e3 = float((val.cpu() - ov).abs().max()) / max(1.0, float(ov.abs().max()))
    assert e3 < 1e-4, (sim, layerwise, H, L, "search", e3)
    worst = max(worst, err, e2, e3)
print("OK worst relative error %.2e" % worst)
'''
    env = dict(os.environ, RGL_REQUIRE_MFMA_CHILDREN="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
    report("scene-kernel rollouts (cosine family, concatenation, layerwise): " + out.stdout.strip().splitlines()[-1])


def test_non_default_embeddings_and_x_dim_on_the_tile_kernels(dev):
    """VERDICT r2 item 9 / missing 3: `X_dim` = 64 and `wr_dims` / `wh_dims` other than the shipped [64, 32] used to run on the general
    VALU kernel (1.8 % of peak).  They now take the tile kernels of rgl_backward_mfma.hip -- MFMA row kernels for any MLP, one
    workgroup per scene for the graph block with x_dim 32 | 64 -- in the module forwards, the state predictor, the children's values
    of a search (sibling scenes share their crowd's embedded rows) and the backward pass.  With RGL_REQUIRE_MFMA_CHILDREN /
    RGL_REQUIRE_MFMA_FORWARD = 1 the library refuses the general kernel, so passing proves the path.  Randomly initialised models
    against the oracle: values, next humans, whole searches, and every parameter gradient against torch autograd over the oracle."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
import relationalgraphlearning_amd as rga
from relationalgraphlearning_amd.config import policy_config
from oracle import rgl_oracle as orc
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
worst, worst_g = 0.0, 0.0
for X, wr, wh, H, L, D, B, sim, lw in ((64, [64, 64], [64, 64], 5, 2, 2, 6, "embedded_gaussian", False), (32, [128, 64, 32], [48, 32], 19, 2, 1, 5, "embedded_gaussian", False),
                                   (64, [32, 64], [100, 64], 33, 2, 1, 2, "embedded_gaussian", False), (32, [64, 32], [32, 32], 5, 1, 2, 7, "gaussian", False),
                                   (64, [64], [256, 64], 12, 2, 2, 4, "gaussian", False),
                                   # round 6: layerwise graphs (an adjacency per layer) of other embedding MLPs on the tile kernels
                                   (32, [128, 64, 32], [48, 32], 7, 2, 2, 4, "embedded_gaussian", True), (32, [48, 32], [64, 64, 32], 19, 3, 1, 3, "gaussian", True)):
    cfgp = policy_config("model_predictive_rl", gcn__num_layer=L, gcn__X_dim=X, gcn__final_state_dim=X, gcn__wr_dims=wr, gcn__wh_dims=wh,
                         gcn__layerwise_graph=lw,
                         gcn__similarity_function=sim, model_predictive_rl__planning_depth=D, model_predictive_rl__planning_width=2,
                         model_predictive_rl__do_action_clip=D > 1, model_predictive_rl__value_network_dims=[X, 100, 100, 1])
    torch.manual_seed(X * 100 + H)
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfgp)
    with torch.no_grad():                       # the reference draws w_a / Ws from randn: scaled to the size trained weights have
        for gm in (pol.value_estimator.graph_model, pol.state_predictor.graph_model):
            for n_, p_ in gm.named_parameters():
                if n_ == "w_a" or n_.startswith("Ws"):
                    p_.mul_(1.0 / X ** 0.5)
    pol.set_time_step(0.25); pol.set_phase("test"); pol.set_device(dev)
    robot, humans = seeded_scenes(900 + H, B, H)
    cfg = orc.OracleConfig(num_layer=L, similarity=sim, planning_depth=D, planning_width=2, do_action_clip=D > 1, layerwise_graph=lw)
    Pm = orc.MprlParams.from_checkpoint({k: ({kk: vv.cpu() for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in pol.get_state_dict().items()})
    r, h = robot.unsqueeze(1).to(dev), humans.to(dev)
    with torch.no_grad():
        ov1 = orc.value_estimator_forward(robot[:, None, :], humans, Pm.ve_graph, Pm.value_network, cfg)
        emb, _ = orc.rgl_forward(robot[:, None, :], humans, Pm.sp_graph, cfg)
        onh = orc.mlp_forward(emb, orc.mlp_layers(Pm.motion_predictor, ""), last_relu=False)[:, 1:, :]
        oa, ov, orv, okept = orc.mprl_predict_batched(robot, humans, Pm, cfg)
        v1 = pol.value_estimator((r, h))
        _, nh = pol.state_predictor((r, h), None)
    e1 = float((v1.cpu() - ov1).abs().max()) / max(1.0, float(ov1.abs().max()))
    e3 = float((nh.cpu() - onh).abs().max()) / max(1.0, float(onh.abs().max()))
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    e2 = float((val.cpu() - ov).abs().max()) / max(1.0, float(ov.abs().max()))
    assert e1 < 1e-4 and e2 < 1e-4 and e3 < 1e-4, (X, wr, wh, H, e1, e2, e3)
    same = (act.cpu().long() == oa).float().mean().item()
    assert same == 1.0 or e2 < 1e-6, (X, wr, wh, same)
    worst = max(worst, e1, e2, e3)
    print("forward / search ok:", X, wr, wh, H, L, D, sim, "layerwise" if lw else "", e1, e2, e3, flush=True)
    # gradients of the value estimator through the tile pipeline against autograd over the oracle
    ve = pol.value_estimator
    wv = torch.linspace(-1.0, 1.5, B).reshape(B, 1)
    (ve((r, h)) * wv.to(dev)).sum().backward()
    gsd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ve.graph_model.state_dict().items()}
    vsd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ve.value_network.state_dict().items()}
    (orc.value_estimator_forward(robot[:, None, :], humans, gsd, vsd, cfg) * wv).sum().backward()
    for mod, sd in ((ve.graph_model, gsd), (ve.value_network, vsd)):
        for k, v in mod.named_parameters():
            scale = max(1e-3, float(sd[k].grad.abs().max()))
            err = float((v.grad.cpu() - sd[k].grad).abs().max()) / scale
            assert err < 2e-4, (X, wr, wh, k, err)
            worst_g = max(worst_g, err)
# path G with other embedding MLPs (x_dim 32): its B x 81 rotated scenes on the tile kernels
for wr, wh, H, B in (([128, 32], [48, 32], 5, 6), ([32], [64, 64, 32], 19, 3)):
    cfgp = policy_config("gcn", gcn__wr_dims=wr, gcn__wh_dims=wh)
    torch.manual_seed(H)
    gp = rga.GCN()
    gp.configure(cfgp)
    with torch.no_grad():
        for n_, p_ in gp.model.named_parameters():
            if n_ in ("w_a", "w1", "w2"):
                p_.mul_(1.0 / 32 ** 0.5)
    gp.time_step = 0.25
    gp.set_phase("test"); gp.set_device(dev)
    gp.build_action_space(1.0)
    robot, humans = seeded_scenes(1700 + H, B, H)
    vals, best = gp.gcn_search().search(robot.to(dev), humans.to(dev))
    ob, ovv = orc.gcn_predict_batched(robot.numpy(), humans.numpy(), {k: v.detach().cpu() for k, v in gp.model.state_dict().items()}, orc.OracleConfig())
    eg = float(np.abs(vals.cpu().numpy().astype(np.float64) - ovv).max()) / max(1.0, float(np.abs(ovv).max()))
    assert eg < 1e-4, ("path G", wr, wh, eg)
    assert (best.cpu().numpy().astype(np.int64) == ob).all() or eg < 1e-6
    worst = max(worst, eg)
    print("path G ok:", wr, wh, H, eg, flush=True)
print("OK worst relative error %.2e (values, next humans, searches, path G), %.2e (gradients)" % (worst, worst_g))
'''
    env = dict(os.environ, RGL_REQUIRE_MFMA_CHILDREN="1", RGL_REQUIRE_MFMA_FORWARD="1", RGL_BACKWARD_MFMA="2")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    report("X_dim 64 / other wr_dims, wh_dims on the tile kernels (general kernel refused): " + out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("H", [1, 2, 3, 14, 15, 16, 17, 30, 31, 32, 47, 48, 62, 63])
def test_tile_kernels_across_node_counts(H, dev, monkeypatch):
    """The tile kernels at every boundary of their tiling -- N = 2 .. 64 nodes: one, two or four node tiles, a last tile with a single
    valid row (N = 17, 33), N a multiple of 4 or not, batches that are not a multiple of 16 rows -- forced on for the shipped shapes
    (RGL_TILES_FORWARD=2, RGL_BACKWARD_MFMA=1): values and next humans against the oracle, every gradient against autograd over it."""
    monkeypatch.setenv("RGL_TILES_FORWARD", "2")
    monkeypatch.setenv("RGL_BACKWARD_MFMA", "1")
    L = 3 if H in (2, 16, 31, 47) else 2
    skip = H % 3 != 0
    sim = "cosine" if H in (3, 17, 48) else ("cosine_softmax" if H in (15, 32, 62) else ("embedded_gaussian" if H % 2 else "gaussian"))
    c = dict(L=L, sim=sim, layerwise=False, skip=skip, flavour="trained")
    g1, ve, sp = build_modules(c, dev)
    B = 7 if H < 40 else 3
    robot, humans = seeded_scenes(1300 + H, B, H)
    r, h = robot.unsqueeze(1).to(dev), humans.to(dev)
    cfg = orc.OracleConfig(num_layer=L, similarity=c["sim"], skip_connection=skip, layerwise_graph=False)
    wv = torch.linspace(-1.0, 1.5, B).reshape(B, 1)
    wm = torch.randn(B, H, 5, generator=torch.Generator().manual_seed(3))
    for p_ in list(ve.parameters()) + list(sp.parameters()):
        p_.grad = None
    out = ve((r, h))
    (out * wv.to(dev)).sum().backward()
    _, nh = sp((r, h), None, detach=False)
    (nh * wm.to(dev)).sum().backward()
    gsd, vsd = _oracle_leafs(ve.graph_model.state_dict()), _oracle_leafs(ve.value_network.state_dict())
    want = orc.value_estimator_forward(robot.unsqueeze(1), humans, gsd, vsd, cfg)
    close(out.detach().cpu().numpy(), want.detach().numpy())
    (want * wv).sum().backward()
    for k, v in ve.graph_model.named_parameters():
        _grad_close(v.grad, gsd[k].grad, "graph." + k)
    for k, v in ve.value_network.named_parameters():
        _grad_close(v.grad, vsd[k].grad, "value." + k)
    gsd, msd = _oracle_leafs(sp.graph_model.state_dict()), _oracle_leafs(sp.human_motion_predictor.state_dict())
    emb, _ = orc.rgl_forward(robot.unsqueeze(1), humans, gsd, cfg)
    wantm = orc.mlp_forward(emb, orc.mlp_layers(msd, ""), last_relu=False)[:, 1:, :]
    close(nh.detach().cpu().numpy(), wantm.detach().numpy())
    (wantm * wm).sum().backward()
    for k, v in sp.graph_model.named_parameters():
        _grad_close(v.grad, gsd[k].grad, "sp_graph." + k)
    for k, v in sp.human_motion_predictor.named_parameters():
        _grad_close(v.grad, msd[k].grad, "motion." + k)


def test_non_default_value_heads_keep_the_mfma_path(dev):
    """VERDICT r2 missing 3: a `value_network_dims` other than the shipped [32, 100, 100, 1] (path G: `planning_dims` other than
    [150, 100, 100, 1]) used to drop the whole search to the general VALU kernel.  robot_head_any_kernel (any depth <= 6, widths
    <= 256) keeps the MFMA stage-1 kernels: with RGL_REQUIRE_MFMA_CHILDREN=1 the library refuses the general kernel, so passing
    proves it.  Module forward, children's values and whole searches against the oracle on randomly initialised heads."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
import relationalgraphlearning_amd as rga
from relationalgraphlearning_amd.config import policy_config
from oracle import rgl_oracle as orc
from tests import golden_io as gio
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
worst = 0.0
for dims, H, L, D, B in (([64, 1], 19, 2, 2, 8), ([32, 100, 100, 100, 100, 1], 5, 2, 1, 20), ([7, 1], 19, 2, 1, 5), ([1], 5, 2, 2, 6),
                         ([256, 3, 130, 1], 49, 3, 1, 2), ([48, 48, 1], 79, 2, 1, 2)):
    cfgp = policy_config("model_predictive_rl", gcn__num_layer=L, model_predictive_rl__planning_depth=D,
                         model_predictive_rl__planning_width=2, model_predictive_rl__do_action_clip=D > 1,
                         model_predictive_rl__value_network_dims=dims)
    torch.manual_seed(len(dims) * 100 + H)
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfgp)
    sd = gio.checkpoint("trained", L)
    sd["value_network"] = {k: v * 0.5 for k, v in pol.value_estimator.value_network.state_dict().items()}
    pol.load_state_dict(sd)
    pol.set_time_step(0.25); pol.set_phase("test"); pol.set_device(dev)
    robot, humans = seeded_scenes(900 + H, B, H)
    cfg = orc.OracleConfig(num_layer=L, planning_depth=D, planning_width=2, do_action_clip=D > 1)
    Pm = orc.MprlParams.from_checkpoint({k: ({kk: vv.cpu() for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in pol.get_state_dict().items()})
    with torch.no_grad():
        ov1 = orc.value_estimator_forward(robot[:, None, :], humans, Pm.ve_graph, Pm.value_network, cfg)
        oa, ov, orv, okept = orc.mprl_predict_batched(robot, humans, Pm, cfg)
        v1 = pol.value_estimator((robot.unsqueeze(1).to(dev), humans.to(dev)))
    e1 = float((v1.cpu() - ov1).abs().max()) / max(1.0, float(ov1.abs().max()))
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    e2 = float((val.cpu() - ov).abs().max()) / max(1.0, float(ov.abs().max()))
    assert e1 < 1e-4 and e2 < 1e-4, (dims, H, e1, e2)
    same = (act.cpu().long() == oa).float().mean().item()
    assert same == 1.0 or e2 < 1e-6, (dims, same)
    worst = max(worst, e1, e2)
print("OK worst relative error %.2e" % worst)
'''
    env = dict(os.environ, RGL_REQUIRE_MFMA_CHILDREN="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    report("non-default value heads on the MFMA path (robot_head_any_kernel): " + out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("speeds,rots,H,L", [(3, 8, 19, 2), (5, 19, 5, 2), (6, 16, 19, 2), (2, 4, 49, 3), (1, 1, 7, 2),
                                             (15, 17, 5, 2)])
def test_non_default_action_spaces(speeds, rots, H, L, dev):
    """Action tables other than 5 x 16 + 1: A = 25 (two child tiles), 96 (exactly six), 97 (beyond the MFMA kernels' 96:
    general kernel), 9, 2, 256 (the ABI maximum) -- whole depth-2 search against the batched oracle."""
    cfgp = policy_config("model_predictive_rl", gcn__num_layer=L, action_space__speed_samples=speeds,
                         action_space__rotation_samples=rots, model_predictive_rl__planning_depth=2,
                         model_predictive_rl__planning_width=2, model_predictive_rl__do_action_clip=True)
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfgp)
    pol.load_state_dict(gio.checkpoint("trained", L))
    pol.set_time_step(0.25)
    pol.set_phase("test")
    pol.set_device(dev)
    B = 6
    robot, humans = seeded_scenes(321 + speeds + rots, B, H)
    cfg = orc.OracleConfig(num_layer=L, speed_samples=speeds, rotation_samples=rots, planning_depth=2, planning_width=2,
                           do_action_clip=True)
    with torch.no_grad():
        oa, ov, orv, okept, lv = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained", L), cfg, return_levels=True)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    assert pol.tree_search().num_actions == speeds * rots + 1
    close(val.cpu().numpy(), ov.numpy())
    if speeds * rots + 1 == nat.MAX_ACTIONS:                     # one more action is refused, not truncated
        too_many = policy_config("model_predictive_rl", gcn__num_layer=L, action_space__speed_samples=speeds,
                                 action_space__rotation_samples=rots + 1)
        big = rga.ModelPredictiveRL()
        big.time_step = 0.25
        big.configure(too_many)
        big.load_state_dict(gio.checkpoint("trained", L))
        big.set_time_step(0.25)
        big.set_phase("test")
        big.set_device(dev)
        with pytest.raises(nat.NativeLibraryError):
            big.predict_batch(robot.to(dev), humans.to(dev))
    check_decisions("action table %dx%d+1, H=%d" % (speeds, rots, H), act, val, (oa, ov, orv, okept), lv)


F16_TOL = 1e-3      # BASELINE configs[4]: f16-input MFMA for the dense middle-layer products, f32 accumulate (measured ~1e-5)


@pytest.mark.parametrize("H,skip,flavour", [(49, True, "trained"), (19, True, "trained"), (33, False, "trained"),
                                            (5, True, "trained"), (55, True, "trained"), (49, True, "rand")])
def test_value_children_f16_contraction(H, skip, flavour, dev):
    """contraction_dtype = "f16" (3-layer graph, shared-crowd deep kernel): values within F16_TOL of the fp32 oracle,
    and not bit-identical to the fp32 path (i.e. the f16 kernel really ran)."""
    L, P = 3, 3
    pol = make_mprl_policy(flavour, 1, L=L, skip=skip, device=dev)
    pol.build_action_space(1.0)
    robot, humans = seeded_scenes(900 + H, P, H)
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())
    got = {}
    for dt in ("f32", "f16"):
        pol.contraction_dtype = dt
        ts = pol.tree_search()
        assert ts.contraction_dtype == dt
        got[dt] = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    A = got["f32"].shape[1]
    Pm = gio.oracle_params(flavour, L)
    with torch.no_grad():
        want = orc.value_estimator_forward(cr.reshape(P * A, 1, 9), humans[:, None].expand(P, A, H, 5).reshape(P * A, H, 5),
                                           Pm.ve_graph, Pm.value_network,
                                           orc.OracleConfig(num_layer=L, skip_connection=skip)).numpy().reshape(P, A)
    close(got["f32"], want)
    close(got["f16"], want, tol=F16_TOL)
    assert not np.array_equal(got["f16"], got["f32"])


def test_deep_kernel_last_node_tile_forms_agree(dev):
    """children_deep_kernel computes a last node tile of <= 4 valid nodes (N = 50, 36, 20 with three layers) on the 4 x 4 x 1 MFMA
    (T4, round 4); RGL_DEEP_T4=0 keeps the 16-row form.  Both against the oracle in child processes (the switch is read once),
    over crowd sizes on both sides of the condition (N = 49..53: 1..4 valid nodes and 5), skip on / off."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from oracle import rgl_oracle as orc
from tests import golden_io as gio
from tests.helpers import make_mprl_policy
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
worst = 0.0
for H, skip in ((48, True), (49, True), (49, False), (50, True), (51, True), (52, True), (35, True), (19, True), (16, False), (33, True)):
    pol = make_mprl_policy("trained", 1, L=3, skip=skip, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    A = ts.num_actions
    robot, humans = seeded_scenes(1500 + H, 3, H)
    acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
    cr = orc._children_robot(robot, acts, orc.OracleConfig())
    got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
    Pm = gio.oracle_params("trained", 3)
    with torch.no_grad():
        want = orc.value_estimator_forward(cr.reshape(3 * A, 1, 9), humans[:, None].expand(3, A, H, 5).reshape(3 * A, H, 5),
                                           Pm.ve_graph, Pm.value_network, orc.OracleConfig(num_layer=3, skip_connection=skip)).numpy().reshape(3, A)
    err = float(np.abs(got - want).max())
    worst = max(worst, err)
    assert err < 1e-6, (H, skip, err)
print("OK worst %.2e" % worst)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for sw, fuse in (("1", "1"), ("0", "1"), ("1", "0")):
        out = subprocess.run([sys.executable, "-c", code], cwd=root,
                             env=dict(os.environ, RGL_DEEP_T4=sw, RGL_DEEP_FUSE_HEAD=fuse, RGL_REQUIRE_MFMA_CHILDREN="1"),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
        lines.append("RGL_DEEP_T4=%s RGL_DEEP_FUSE_HEAD=%s %s" % (sw, fuse, out.stdout.strip().splitlines()[-1]))
    report("deep kernel, last node tile on the 4x4x1 MFMA vs the 16-row form, stage 2 inside the launch vs robot_head_kernel, "
           "N = 17..54, L = 3: " + "; ".join(lines))


def test_deep_kernel_searches_with_stage_two_inside_and_outside_the_launch(dev):
    """Round 4: with the packed weight image at hand children_deep_kernel runs the value head -- and the search's select / back-up /
    root steps -- over the rows of the parents each workgroup owns, behind its last parent (RGL_DEEP_FUSE_HEAD=0: robot_head_kernel
    in a launch of its own, as before).  Whole searches over the configurations the deep kernel serves (three layers; two layers
    beyond 32 agents; depth 1..3; f32 and f16 contraction) against the batched oracle, both ways, in child processes."""
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from oracle import rgl_oracle as orc
from tests import golden_io as gio
from tests.helpers import make_mprl_policy
from tests.test_gpu_parity import seeded_scenes
dev = torch.device("cuda:0")
worst = {"f32": 0.0, "f16": 0.0}
for H, L, D, B, dt in ((49, 3, 2, 12, "f32"), (49, 3, 2, 12, "f16"), (19, 3, 3, 9, "f32"), (40, 2, 2, 7, "f32"), (5, 3, 1, 33, "f32"),
                       (33, 3, 2, 300, "f32"), (49, 3, 1, 700, "f16")):
    pol = make_mprl_policy("trained", D, 2, D > 1, L=L, device=dev)
    pol.contraction_dtype = dt
    robot, humans = seeded_scenes(1700 + H + D, B, H)
    cfg = orc.OracleConfig(num_layer=L, planning_depth=D, planning_width=2, do_action_clip=D > 1)
    Pm = gio.oracle_params("trained", L)
    with torch.no_grad():
        oa, ov, orv, okept = orc.mprl_predict_batched(robot, humans, Pm, cfg)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    err = float((val.cpu() - ov).abs().max())
    same = float((act.cpu().long() == oa).float().mean())
    tol = 1e-3 if dt == "f16" else 1e-4
    assert err < tol and (same == 1.0 or err < 1e-5), (H, L, D, B, dt, err, same)
    worst[dt] = max(worst[dt], err)
print("OK max |dV| f32 %.2e, f16 %.2e" % (worst["f32"], worst["f16"]))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for fuse in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, RGL_DEEP_FUSE_HEAD=fuse, RGL_REQUIRE_MFMA_CHILDREN="1"),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
        lines.append("RGL_DEEP_FUSE_HEAD=%s %s" % (fuse, out.stdout.strip().splitlines()[-1]))
    report("searches on the deep kernel, stage 2 + tail inside the launch vs robot_head_kernel: " + "; ".join(lines))


def test_f16_tree_search_and_refusals(dev):
    """configs[4] shape (N = 50, L = 3, D = 2, w = 2) with the f16 products: root values within F16_TOL, same decisions as
    the fp32 oracle except on numerical ties; configurations without an f16 kernel are refused, not silently run in fp32."""
    H, L, B = 49, 3, 6
    robot, humans = seeded_scenes(77, B, H)
    pol = make_mprl_policy("trained", 2, 2, True, L=L, device=dev)
    pol.contraction_dtype = "f16"
    cfg = orc.OracleConfig(num_layer=L, planning_depth=2, planning_width=2, do_action_clip=True)
    with torch.no_grad():
        oa, ov, orv, okept = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained", L), cfg)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    close(val.cpu().numpy(), ov.numpy(), tol=F16_TOL)
    same = act.cpu().numpy().astype(np.int64) == oa.numpy()
    for b in np.nonzero(~same)[0]:
        assert abs(float(val[b]) - float(ov[b])) < F16_TOL
    pol2 = make_mprl_policy("trained", 1, L=2, device=dev)
    pol2.contraction_dtype = "f16"
    with pytest.raises(nat.NativeLibraryError):
        pol2.predict_batch(robot.to(dev), humans.to(dev))


def test_unicycle_kinematics_both_paths(dev):
    """Unicycle action tables: the rollout (incl. the reference's slot-7 rotation quirk in compute_next_state) against
    the batched oracle, and path G's one-step search against its sequential oracle."""
    robot, humans = seeded_scenes(41, 6, 5)
    pol = make_mprl_policy("trained", 2, 2, True, kinematics="unicycle", device=dev)
    cfg = orc.OracleConfig(kinematics="unicycle", planning_depth=2, planning_width=2, do_action_clip=True)
    with torch.no_grad():
        oa, ov, orv, okept, levels = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained"), cfg, return_levels=True)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    ts = pol.tree_search()
    assert np.abs(ts.level_arrays(0)["child_robot"].cpu().numpy() - levels[0]["child_robot"].numpy()).max() < 1e-6
    assert np.abs(ts.level_arrays(0)["reward"].cpu().numpy() - levels[0]["reward"].numpy()).max() < 1e-6
    close(val.cpu().numpy(), ov.numpy())
    assert np.array_equal(act.cpu().numpy().astype(np.int64), oa.numpy())
    a0 = pol.action_space[1]
    assert isinstance(a0, rga.ActionRot)
    # path G
    from relationalgraphlearning_amd.config import policy_config as pc
    g = rga.GCN()
    g.configure(pc("gcn", action_space__kinematics="unicycle"))
    g.model.load_state_dict(gio.path_g_sd())
    g.time_step = 0.25
    g.set_phase("test")
    g.set_device(dev)
    js = JS(robot[0].numpy(), humans[0].numpy())
    a = g.predict(js)
    want_a, want_vals = orc.gcn_predict_sequential([float(x) for x in robot[0]], [[float(x) for x in r] for r in humans[0]],
                                                    gio.path_g_sd(), orc.OracleConfig(kinematics="unicycle"))
    close(np.array(g.action_values), np.array(want_vals))
    assert a == g.action_space[want_a]


def test_state_containers_roundtrip(dev):
    js = rga.JointState(rga.FullState(0.5, -4, 0.1, 0.2, 0.3, 0, 4, 1, 1.57), [rga.ObservableState(1, 2, 0.3, 0.4, 0.3),
                                                                              rga.ObservableState(-1, 0, 0, 0, 0.3)])
    r, h = js.to_tensor(add_batch_size=True, device=dev)
    assert r.shape == (1, 1, 9) and h.shape == (1, 2, 5) and r.is_cuda
    back = rga.tensor_to_joint_state((r, h))
    assert abs(back.robot_state.gy - 4.0) < 1e-7 and len(back.human_states) == 2
    assert (js.robot_state + js.human_states[0])[9:] == (1, 2, 0.3, 0.4, 0.3)
    pol = make_mprl_policy("trained", 1, device=dev)
    assert isinstance(pol.predict(js), rga.ActionXY)


def test_whole_search_is_graph_capturable(dev):
    """No allocation / sync inside mprl_tree_search_f32: a complete depth-2 search replays from a hipGraph."""
    robot, humans = seeded_scenes(55, 64, 19)
    r, h = robot.to(dev), humans.to(dev)
    pol = make_mprl_policy("trained", 2, 2, True, device=dev)
    pol.build_action_space(1.0)
    ts = pol.tree_search()
    ref = ts.search(r, h)
    ref_a, ref_v = ref["best_action"].clone(), ref["best_value"].clone()
    graph, out = ts.capture(r, h)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["best_action"], ref_a) and torch.equal(out["best_value"], ref_v)
    robot2, humans2 = seeded_scenes(56, 64, 19)                    # new inputs, same buffers
    r.copy_(robot2.to(dev))
    h.copy_(humans2.to(dev))
    graph.replay()
    torch.cuda.synchronize()
    got_a, got_v = out["best_action"].clone(), out["best_value"].clone()
    fresh = ts.search(robot2.to(dev), humans2.to(dev))
    assert torch.equal(got_a, fresh["best_action"]) and torch.equal(got_v, fresh["best_value"])


def test_properties_at_full_size(dev):
    """BASELINE config 3 (N=20, L=2, D=2, w=2, B=2048): size-independent properties."""
    B, H = 2048, 19
    robot, humans = seeded_scenes(7, B, H)
    pol = make_mprl_policy("trained", 2, 2, True, device=dev)
    r, h = robot.to(dev), humans.to(dev)
    a1, v1 = pol.predict_batch(r, h)
    a1, v1 = a1.clone(), v1.clone()
    a2, v2 = pol.predict_batch(r, h)
    assert torch.equal(a1, a2) and torch.equal(v1, v2)                      # deterministic
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
    a3, v3 = pol.predict_batch(r[perm], h[perm])
    assert torch.equal(a3, a1[perm]) and torch.equal(v3, v1[perm])          # roots are independent
    # batch-size independent: 64 roots take the two-stage kernels, 2048 the fused one (size-based dispatch) -- the
    # same arithmetic in another summation order, so values agree to rounding and decisions wherever the values are not tied
    a4, v4 = pol.predict_batch(r[:64], h[:64])
    assert float((v4 - v1[:64]).abs().max()) < 1e-5
    differ = a4 != a1[:64]
    assert int(differ.sum()) <= 1 and float((v4 - v1[:64])[differ].abs().max() if differ.any() else 0.0) < 1e-5
    hp = torch.randperm(H, generator=torch.Generator().manual_seed(2)).to(dev)
    a5, v5 = pol.predict_batch(r, h[:, hp])                                 # humans are an unordered set
    assert float((v5 - v1).abs().max()) < 1e-4
    assert float((a5 == a1).float().mean()) > 0.99
    # value back-up sanity: the chosen value is the max of the reported root values
    out = pol.tree_search().last
    assert torch.equal(out["root_values"].max(dim=1).values, out["best_value"])
    # spot-check 16 roots against the oracle at this size
    cfg = orc.OracleConfig(planning_depth=2, planning_width=2, do_action_clip=True)
    with torch.no_grad():
        oa, ov, _, _ = orc.mprl_predict_batched(robot[:16], humans[:16], gio.oracle_params("trained"), cfg)
    close(v1[:16].cpu().numpy(), ov.numpy())


@pytest.mark.parametrize("tag,H,L,D,B,contraction", [
    ("configs[3] per-GPU share", 19, 2, 3, 512, "f32"),
    ("configs[4] per-GPU share", 49, 3, 2, 256, "f32"),
    ("configs[4] per-GPU share, f16 contractions", 49, 3, 2, 256, "f16")])
def test_properties_of_the_other_baseline_workloads(tag, H, L, D, B, contraction, dev):
    """Size-independent properties at the sizes configs[3] / configs[4] are benchmarked at (configs[2] has
    test_properties_at_full_size): run-to-run determinism, root-permutation equivariance (roots are independent trees; bit-exact:
    a root's arithmetic does not depend on its position), human-permutation invariance within the parity tolerance, and the
    chosen value being the maximum of the reported root values."""
    import bench

    class Args:
        pass
    Args.layers, Args.depth, Args.width, Args.humans, Args.contraction = L, D, 2, H, contraction
    pol = bench.make_policy(Args, dev)
    robot, humans = bench.synth_scenes(1000, B, H)
    r, h = robot.to(dev), humans.to(dev)
    a1, v1 = pol.predict_batch(r, h)
    a1, v1 = a1.clone(), v1.clone()
    out = pol.tree_search().last
    assert torch.equal(out["root_values"].max(dim=1).values, out["best_value"])
    a2, v2 = pol.predict_batch(r, h)
    assert torch.equal(a1, a2) and torch.equal(v1, v2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(dev)
    a3, v3 = pol.predict_batch(r[perm].contiguous(), h[perm].contiguous())
    assert torch.equal(a3, a1[perm]) and torch.equal(v3, v1[perm])
    hp = torch.randperm(H, generator=torch.Generator().manual_seed(4)).to(dev)
    a4, v4 = pol.predict_batch(r, h[:, hp].contiguous())
    tol = TOL if contraction == "f32" else 1e-3
    assert float((v4 - v1).abs().max()) < tol
    assert float((a4 == a1).float().mean()) > 0.98
    report("properties at size, %s: deterministic, root-permutation equivariant (bit-exact), human-permutation invariant to %.1e"
           % (tag, float((v4 - v1).abs().max())))


def test_bf16x6_head_matrix_at_size_and_in_other_shapes(dev):
    """RGL_CONTRACT_BF16X6 (ABI 6): the first 64 input features of the children kernel's 100 x 100 head matrix
    as six bf16 MFMA terms over operands that keep all 24 significand bits (three round-to-nearest bf16 pieces each, no scaling:
    layer_mfma_bx, rgl_mlp_chain.h).  Not a reduced-precision mode -- so it is held to what the f32 kernels are held to:
    (1) configs[2] in full against the batched oracle at the REGRESSION bound of the f32 kernels (1e-6), decisions as for f32;
    (2) against the library's own f32 path; (3) against a float64 evaluation of the same search: its deviation must not exceed the
    f32 kernels' by more than the noise between two f32 summation orders; (4) any finite magnitude (bf16 has f32's exponent
    range: head layers scaled by 3e4 / 2e-5 / 1e-9 match float64 as well as the f32 kernels do on the same inputs); (5) the other shapes of the kernel: one node tile, depth 3, a crowd whose wave scratch leaves no room for the
    larger image (falls back to f32 on an image it packs itself), a non-softmax similarity, another head-size-compatible table."""
    import bench

    class Args:
        pass
    Args.layers, Args.depth, Args.width, Args.humans, Args.contraction = 2, 2, 2, 19, "bf16x6"
    pol = bench.make_policy(Args, dev)
    B, H = 2048, 19
    robot, humans = bench.synth_scenes(1000, B, H)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    assert pol.tree_search().last["planner"].contraction_dtype == nat.CONTRACTION_DTYPES["bf16x6"]
    oracle_out, v1, _ = _oracle_at_size(H, 2, 2, B, robot, humans)
    err = close(val.cpu().numpy(), oracle_out[1].numpy(), reg=REG_F32)
    check_decisions("at size, configs[2] in full, bf16x6 head matrix", act, val, oracle_out, [{"value1": v1}], TOL)
    Args.contraction = "f32"
    pol32 = bench.make_policy(Args, dev)
    a32, v32 = pol32.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    d32 = float((val - v32).abs().max())
    same = float((act == a32).float().mean())
    assert d32 < 1e-6 and same > 0.998, (d32, same)
    if not os.environ.get("RGL_CONTRACT_F32_AS"):
        assert d32 > 0.0                                          # (the mode did run: bit-identical values would mean the f32 kernel)
    n64 = 256
    P64 = orc.MprlParams.from_checkpoint({k: {kk: vv.double() for kk, vv in v.items()} for k, v in gio.checkpoint("trained", 2).items()})
    cfg64 = orc.OracleConfig(planning_depth=2, planning_width=2, do_action_clip=True)
    with torch.no_grad():
        _, v64, _, _ = orc.mprl_predict_batched(robot[:n64].double(), humans[:n64].double(), P64, cfg64)
    e_b6 = float((val[:n64].double().cpu() - v64).abs().max())
    e_32 = float((v32[:n64].double().cpu() - v64).abs().max())
    r_b6 = float((val[:n64].double().cpu() - v64).pow(2).mean().sqrt())
    r_32 = float((v32[:n64].double().cpu() - v64).pow(2).mean().sqrt())
    report("bf16x6 head matrix, configs[2] in full: max |dV| vs the oracle %.2e, vs the f32 kernels %.2e, %.2f %% identical decisions; "
           "deviation from float64 (256 roots): max %.2e rms %.2e against %.2e / %.2e for the f32 kernels"
           % (err, d32, 100 * same, e_b6, r_b6, e_32, r_32))
    assert e_b6 < 1e-7 and r_b6 <= 1.25 * r_32 + 1e-9, (e_b6, e_32, r_b6, r_32)
    ts, ts32 = pol.tree_search(), pol32.tree_search()
    r, h = robot[:64].to(dev), humans[:64].to(dev)
    ex = ts.expand(r, h, parents_are_joint_states=False)
    worst, worst32 = 0.0, 0.0
    for scales in ((3e4, 1.0, 1.0, 1.0), (1.0, 2e-5, 3e4, 1.0), (3e4, 3e4, 3e4, 1e-9), (1e-6, 1e-6, 1.0, 1e6)):
        rels = []
        for p_, t_ in ((pol, ts), (pol32, ts32)):
            lins = [m for m in p_.value_estimator.value_network if isinstance(m, torch.nn.Linear)]
            saved = [(m.weight.detach().clone(), m.bias.detach().clone()) for m in lins]
            with torch.no_grad():
                for m, sc in zip(lins, scales):
                    m.weight.mul_(sc)
                    m.bias.mul_(sc)
            got = t_.value_children(ex["child_robot"], ex["humans_next"]).double().cpu()
            with torch.no_grad():
                Pm = orc.MprlParams.from_checkpoint({k: {kk: vv.double().cpu() for kk, vv in v.items()} for k, v in p_.get_state_dict().items()})
                A = t_.num_actions
                want = orc.value_estimator_forward(ex["child_robot"].double().cpu().reshape(64 * A, 1, 9),
                                                   ex["humans_next"].double().cpu()[:, None].expand(64, A, H, 5).reshape(64 * A, H, 5),
                                                   Pm.ve_graph, Pm.value_network, orc.OracleConfig()).reshape(64, A)
            rels.append(float((got - want).abs().max() / want.abs().max()))
            with torch.no_grad():
                for m, (w, b) in zip(lins, saved):
                    m.weight.copy_(w)
                    m.bias.copy_(b)
        worst, worst32 = max(worst, rels[0]), max(worst32, rels[1])
        # what float32 accumulation leaves of such a head is a few 1e-6 whatever the operand form: the mode must stay within 1.5x
        # of the f32 kernels' own error on the same inputs (and inside 1e-5, what the f16 split is asked for)
        assert rels[0] < 1e-5 and rels[0] <= 1.5 * rels[1] + 5e-7, (scales, rels)
    report("bf16x6 head matrix under extreme layer scales (3e4 / 2e-5 / 1e-9 per layer): worst relative error vs float64 %.1e (f32 "
           "kernels on the same inputs: %.1e)" % (worst, worst32))
    for Hh, D, Bb, simf in ((5, 1, 64, "embedded_gaussian"), (19, 3, 24, "embedded_gaussian"), (30, 2, 12, "embedded_gaussian"),
                            (12, 2, 300, "embedded_gaussian"), (19, 2, 40, "squared")):
        p6 = make_mprl_policy("trained", D, 2, D > 1, device=dev, similarity=simf)
        p6.contraction_dtype = "bf16x6"
        p6.build_action_space(1.0)
        rb, hb = bench.synth_scenes(77 + Hh, Bb, Hh)
        a6, v6 = p6.predict_batch(rb.to(dev), hb.to(dev), roots_are_joint_states=True)
        assert p6.tree_search().last["planner"].contraction_dtype == nat.CONTRACTION_DTYPES["bf16x6"]
        cfg = orc.OracleConfig(planning_depth=D, planning_width=2, do_action_clip=D > 1, similarity=simf)
        with torch.no_grad():
            oa, ov, orv, okept, lv = orc.mprl_predict_batched(rb, hb, gio.oracle_params("trained", 2, "separate", simf), cfg, return_levels=True)
        close(v6.cpu().numpy(), ov.numpy(), reg=REG_F32)
        check_decisions("bf16x6, H=%d D=%d B=%d %s" % (Hh, D, Bb, simf), a6, v6, (oa, ov, orv, okept), lv)
        if simf == "embedded_gaussian" and (Hh, D) in ((5, 1), (19, 3)):
            # admission on the other BASELINE shapes too (configs[1]: N = 6, D = 1; configs[3]: D = 3): deviation from a float64
            # evaluation of the same search next to the f32 kernels' on the same roots
            p32 = make_mprl_policy("trained", D, 2, D > 1, device=dev, similarity=simf)
            p32.build_action_space(1.0)
            _, v32b = p32.predict_batch(rb.to(dev), hb.to(dev), roots_are_joint_states=True)
            nb = min(Bb, 32)
            with torch.no_grad():
                _, v64b, _, _ = orc.mprl_predict_batched(rb[:nb].double(), hb[:nb].double(), P64, cfg)
            e6 = float((v6[:nb].double().cpu() - v64b).abs().max())
            e32 = float((v32b[:nb].double().cpu() - v64b).abs().max())
            report("bf16x6 vs float64, H=%d D=%d (%d roots): max %.2e against %.2e for the f32 kernels" % (Hh, D, nb, e6, e32))
            assert e6 <= 1.5 * e32 + 3e-9, (Hh, D, e6, e32)


@pytest.mark.parametrize("speeds,rots,H,D,clip,sparse", [(3, 8, 19, 2, True, False), (6, 16, 19, 2, True, True), (5, 16, 5, 1, False, False),
                                                         (2, 4, 12, 3, True, False), (15, 17, 5, 2, True, False)])
def test_bf16x6_mode_with_other_action_tables_and_search_settings(speeds, rots, H, D, clip, sparse, dev):
    """The six-term bf16 mode (the arithmetic `bench.py --contraction auto` reports in) through the fused kernel's other code paths: partial tiles of every size (A = 25, 97 -> general kernel,
    81, 9, 256), unclipped depth-1 search (W = A), sparse clipping, depth 3 -- whole searches against the oracle; and the
    single-decision path (`predict(JointState)`: the search of that mode captured in a hipGraph)."""
    cfgp = policy_config("model_predictive_rl", action_space__speed_samples=speeds, action_space__rotation_samples=rots,
                         model_predictive_rl__planning_depth=D, model_predictive_rl__planning_width=2,
                         model_predictive_rl__do_action_clip=clip, model_predictive_rl__sparse_search=sparse)
    pol = rga.ModelPredictiveRL()
    pol.time_step = 0.25
    pol.configure(cfgp)
    pol.load_state_dict(gio.checkpoint("trained", 2))
    pol.set_time_step(0.25)
    pol.set_phase("test")
    pol.set_device(dev)
    pol.contraction_dtype = "bf16x6"
    B = 40
    robot, humans = seeded_scenes(1234 + speeds + H, B, H)
    cfg = orc.OracleConfig(speed_samples=speeds, rotation_samples=rots, planning_depth=D, planning_width=2, do_action_clip=clip,
                           sparse_search=sparse)
    with torch.no_grad():
        oa, ov, orv, okept, lv = orc.mprl_predict_batched(robot, humans, gio.oracle_params("trained"), cfg, return_levels=True)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    assert pol.tree_search().last["planner"].contraction_dtype == nat.CONTRACTION_DTYPES["bf16x6"]
    close(val.cpu().numpy(), ov.numpy(), reg=REG_F32)
    check_decisions("bf16x6, action table %dx%d+1, H=%d D=%d%s" % (speeds, rots, H, D, " sparse" if sparse else ""), act, val,
                    (oa, ov, orv, okept), lv)
    # one decision through predict(): the captured search of this mode
    a = pol.predict(JS(robot[0].numpy(), humans[0].numpy()))
    assert a == pol.action_space[int(act[0])] or abs(float(val[0]) - float(ov[0])) < 1e-6


def test_bf16x6_children_over_odd_shapes_and_sharp_attention(dev):
    """Children's values in the six-term bf16 mode against the oracle over the shapes the f32 fused kernel is stressed with (partial tiles,
    every register bucket, skip on / off, raw random-init weights whose hidden features reach 10-100, odd mid-size parent counts)
    and with SHARP attention: w_a scaled so that the similarities span hundreds.  A product's relative error becomes an absolute
    error of the logit and so a relative error of the softmax weights -- the place where a split mode could lose to f32: the mode's
    deviation from float64 is held next to the f32 kernels' on the same inputs."""
    import copy
    pol_kw = dict(L=2, device=dev)
    worst = 0.0
    for H, skip, P, flavour in ((19, True, 5, "trained"), (19, True, 67, "rand"), (5, True, 7, "trained"), (4, False, 33, "trained"),
                                (1, True, 3, "trained"), (15, False, 9, "rand"), (16, True, 3, "trained"),
                                (19, True, 701, "trained"), (5, False, 1501, "trained"), (9, True, 2311, "rand")):
        pol = make_mprl_policy(flavour, 1, skip=skip, **pol_kw)
        pol.contraction_dtype = "bf16x6"
        pol.build_action_space(1.0)
        ts = pol.tree_search()
        A = ts.num_actions
        robot, humans = seeded_scenes(900 + H, P, H)
        acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
        cr = orc._children_robot(robot, acts, orc.OracleConfig())
        got = ts.value_children(cr.to(dev), humans.to(dev)).cpu().numpy()
        Pm = gio.oracle_params(flavour, 2)
        with torch.no_grad():
            want = orc.value_estimator_forward(cr.reshape(P * A, 1, 9), humans[:, None].expand(P, A, H, 5).reshape(P * A, H, 5),
                                               Pm.ve_graph, Pm.value_network,
                                               orc.OracleConfig(num_layer=2, skip_connection=skip)).numpy().reshape(P, A)
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        worst = max(worst, err)
        assert err < 1e-4, (H, skip, P, flavour, err)
    sharp, sharp32 = 0.0, 0.0
    for scale, H, P in ((25.0, 19, 40), (-40.0, 9, 33), (300.0, 5, 17)):
        ck = copy.deepcopy(gio.checkpoint("trained", 2))
        ck["graph_model1"]["w_a"] = ck["graph_model1"]["w_a"] * scale
        errs = []
        for mode in ("bf16x6", "f32"):
            pol = make_mprl_policy("trained", 1, **pol_kw)
            pol.load_state_dict(ck)
            pol.contraction_dtype = mode
            pol.build_action_space(1.0)
            ts = pol.tree_search()
            A = ts.num_actions
            robot, humans = seeded_scenes(950 + H, P, H)
            acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
            cr = orc._children_robot(robot, acts, orc.OracleConfig())
            got = ts.value_children(cr.to(dev), humans.to(dev)).double().cpu().numpy()
            P64 = orc.MprlParams.from_checkpoint({k: {kk: vv.double() for kk, vv in v.items()} for k, v in ck.items()})
            with torch.no_grad():
                want = orc.value_estimator_forward(cr.double().reshape(P * A, 1, 9),
                                                   humans.double()[:, None].expand(P, A, H, 5).reshape(P * A, H, 5),
                                                   P64.ve_graph, P64.value_network, orc.OracleConfig()).numpy().reshape(P, A)
            errs.append(np.abs(got - want).max() / max(1.0, np.abs(want).max()))
        sharp, sharp32 = max(sharp, errs[0]), max(sharp32, errs[1])
        # logits in the hundreds: an f32 rounding of S is ~1e-5 absolute, whatever the operand form; the mode within 2x of the f32 kernels
        assert errs[0] < 1e-4 and errs[0] <= 2.0 * errs[1] + 1e-6, (scale, H, P, errs)
    report("bf16x6 children over odd shapes / raw random weights: worst relative error %.2e; sharp attention (similarities in the "
           "hundreds) vs float64: %.2e (f32 kernels: %.2e)" % (worst, sharp, sharp32))


def test_clamp_bit_relu_over_sharp_attention_and_large_activations(dev):
    """Round 4: the ReLUs of the row passes sit in the clamp bit of packed FMAs on power-of-two-scaled operands (fused kernel: 2^-110,
    the softmax ratio r = e^(msh - S_c) reaches e^60 there; deep kernel: 2^-64).  Exact while r UW + y < 2^110 (a UW + b y < 2^64): so
    the cases that stretch the range -- similarities in the hundreds (w_a scaled), graph weights scaled until the hidden features
    reach 10^3..10^4, both together -- against the float64 oracle, f32 fused kernel (L = 2), deep kernel (L = 3, f32 and f16
    contraction)."""
    import copy
    worst = {}
    for L, H, P, wa_scale, w_scale, dt in ((2, 19, 40, 25.0, 1.0, "f32"), (2, 9, 33, -40.0, 1.0, "f32"), (2, 5, 17, 300.0, 1.0, "f32"),
                                           (2, 19, 21, 300.0, 1e3, "f32"), (2, 19, 21, 1.0, 1e4, "f32"), (2, 16, 9, -300.0, 30.0, "f32"),
                                           (3, 49, 3, 25.0, 1.0, "f32"), (3, 49, 3, 300.0, 100.0, "f32"), (3, 49, 3, 25.0, 1.0, "f16"),
                                           (3, 19, 5, 300.0, 100.0, "f32"), (3, 5, 5, 40.0, 10.0, "f16")):
        ck = copy.deepcopy(gio.checkpoint("trained", L))
        ck["graph_model1"]["w_a"] = ck["graph_model1"]["w_a"] * wa_scale
        ck["graph_model1"]["Ws.0"] = ck["graph_model1"]["Ws.0"] * w_scale
        pol = make_mprl_policy("trained", 1, L=L, device=dev)
        pol.load_state_dict(ck)
        pol.contraction_dtype = dt
        pol.build_action_space(1.0)
        ts = pol.tree_search()
        A = ts.num_actions
        robot, humans = seeded_scenes(960 + H, P, H)
        acts, _ = orc.mprl_action_space(orc.OracleConfig(), 1.0)
        cr = orc._children_robot(robot, acts, orc.OracleConfig())
        got = ts.value_children(cr.to(dev), humans.to(dev)).double().cpu().numpy()
        P64 = orc.MprlParams.from_checkpoint({k: {kk: vv.double() for kk, vv in v.items()} for k, v in ck.items()})
        with torch.no_grad():
            want = orc.value_estimator_forward(cr.double().reshape(P * A, 1, 9),
                                               humans.double()[:, None].expand(P, A, H, 5).reshape(P * A, H, 5),
                                               P64.ve_graph, P64.value_network, orc.OracleConfig(num_layer=L)).numpy().reshape(P, A)
        assert np.isfinite(got).all(), (L, H, wa_scale, w_scale, dt)
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        worst[dt] = max(worst.get(dt, 0.0), err)
        assert err < (F16_TOL if dt == "f16" else 1e-4), (L, H, P, wa_scale, w_scale, dt, err)
    report("clamp-bit ReLU at the edges of its range (sharp attention, hidden features to 1e4) vs float64: worst relative error "
           + ", ".join("%s %.2e" % kv for kv in sorted(worst.items())))


_ORACLE_AT_SIZE = {}


def _oracle_at_size(H, L, D, B, robot, humans):
    """The batched oracle over bench.py's scenes, walked in chunks (256 roots; 64 for the dense crowds) and memoised per workload.
    Returns ((best action, best value, root values, root kept), level-0 one-step values, [per level {"value1", "keep"}])."""
    key = (H, L, D, B)
    if key not in _ORACLE_AT_SIZE:
        cfg = orc.OracleConfig(num_layer=L, planning_depth=D, planning_width=2, do_action_clip=(D > 1))
        step = 64 if H >= 40 else 256
        outs, lv = [], [[] for _ in range(D)]
        with torch.no_grad():
            for lo in range(0, B, step):
                o = orc.mprl_predict_batched(robot[lo:lo + step], humans[lo:lo + step], gio.oracle_params("trained", L), cfg,
                                             return_levels=True)
                outs.append(o[:4])
                for l in range(D):                       # parents of a level are root-major: chunks concatenate
                    lv[l].append((o[4][l]["value1"], o[4][l]["keep"]))
        levels = [{"value1": torch.cat([x[0] for x in lv[l]]), "keep": torch.cat([x[1] for x in lv[l]])} for l in range(D)]
        _ORACLE_AT_SIZE[key] = ([torch.cat([o[i] for o in outs]) for i in range(4)], levels[0]["value1"], levels)
    return _ORACLE_AT_SIZE[key]


def compare_trees(tag, ts, val, oracle_out, oracle_levels, tol, reg):
    """Level by level against the oracle's tree.  A search is a chain of top-w selections, and a selection between two actions whose
    one-step values differ in the 8th digit may legitimately fall the other way on the GPU (another summation order): from there on
    the two trees hold different nodes, and the root's value may move by the difference between two ALMOST equally good branches --
    1e-4 was seen at 4096 roots, where 28 672 selections are made.  So: the trees are aligned node by node (kept sets are matched as
    sets: the GPU orders them by value, np.argpartition does not); at every aligned node the one-step values of all actions must
    agree to `reg` (regression level) and a kept set may differ only where the oracle itself has a tie -- the action the GPU kept is
    within `tie` of the weakest one the oracle kept; the sub-trees below such a node are not comparable and their roots are held to
    the north-star bound `tol` only.  Every other root's value must agree to `reg`.  Returns (worst aligned |dV|, roots with a
    tie-diverged tree, worst |dV| among those)."""
    tie = 10 * reg
    D = len(oracle_levels)
    B = oracle_out[1].shape[0]
    w = oracle_levels[0]["keep"].shape[1]
    gp = np.arange(B)                                   # oracle parent -> the GPU parent holding the same node, -1: diverged above
    diverged_root = np.zeros(B, bool)
    worst_v1 = 0.0
    for l in range(D):
        arr = ts.level_arrays(l)
        g_v1, g_keep = arr["value1"].cpu().numpy(), arr["keep"].cpu().numpy().astype(np.int64)
        o_v1, o_keep = oracle_levels[l]["value1"].numpy(), oracle_levels[l]["keep"].numpy().astype(np.int64)
        P = o_v1.shape[0]
        per_root = P // B
        ok = gp >= 0
        d1 = np.abs(g_v1[gp[ok]] - o_v1[ok])
        worst_v1 = max(worst_v1, float(d1.max()) if d1.size else 0.0)
        assert worst_v1 <= reg, (tag, "one-step values of aligned nodes, level %d" % l, worst_v1)
        nxt = np.full(P * w, -1, np.int64)
        for p in np.nonzero(ok)[0]:
            gk = g_keep[gp[p]]
            same = sorted(gk.tolist()) == sorted(o_keep[p].tolist())
            if not same:
                weakest = o_v1[p, o_keep[p]].min()
                for a in gk:
                    assert o_v1[p, a] >= weakest - tie, (tag, "level %d parent %d: kept set differs and it is no tie in the oracle" % (l, p),
                                                         float(o_v1[p, a]), float(weakest))
                diverged_root[p // per_root] = True
                continue
            for k in range(w):
                nxt[p * w + k] = gp[p] * w + int(np.nonzero(gk == o_keep[p, k])[0][0])
        gp = nxt
    dv = np.abs(val.cpu().numpy().astype(np.float64) - oracle_out[1].numpy().astype(np.float64))
    aligned = ~diverged_root
    worst_aligned = float(dv[aligned].max()) if aligned.any() else 0.0
    worst_div = float(dv[diverged_root].max()) if diverged_root.any() else 0.0
    assert worst_aligned <= reg, (tag, "values of roots with aligned trees", worst_aligned)
    assert worst_div <= tol, (tag, "values of roots whose tree diverged at a tie", worst_div)
    report("%s: trees aligned node by node -- one-step values within %.1e at every level, %d of %d roots with a selection tied in "
           "the oracle (|dV| <= %.1e there), every other root within %.1e"
           % (tag, worst_v1, int(diverged_root.sum()), B, worst_div, worst_aligned))
    return worst_aligned, int(diverged_root.sum()), worst_div


AT_SIZE_CASES = [
    ("configs[1] in full (N=5)", 4, 2, 1, 512, "f32", TOL),
    ("configs[1] in full (N=6)", 5, 2, 1, 512, "f32", TOL),
    ("configs[1] in full (N=6), the arithmetic bench.py reports it in", 5, 2, 1, 512, "bf16x6", TOL),
    ("configs[2] in full", 19, 2, 2, 2048, "f32", TOL),
    ("configs[3] per-GPU share", 19, 2, 3, 512, "f32", TOL),
    ("configs[3] in full (4096 roots, depth 3)", 19, 2, 3, 4096, "f32", TOL),
    ("configs[3] in full (4096 roots, depth 3), the arithmetic bench.py reports it in", 19, 2, 3, 4096, "bf16x6", TOL),
    ("configs[4] per-GPU share in full (256 roots)", 49, 3, 2, 256, "f32", TOL),
    ("configs[4] per-GPU share in full, f16 contractions", 49, 3, 2, 256, "f16", 1e-3)]


def _bench_policy(L, D, H, contraction, dev):
    import bench

    class Args:
        pass
    Args.layers, Args.depth, Args.width, Args.humans, Args.contraction = L, D, 2, H, contraction
    return bench.make_policy(Args, dev)


@pytest.mark.parametrize("tag,H,L,D,B,contraction,tol", AT_SIZE_CASES)
def test_baseline_workloads_against_the_oracle_at_size(tag, H, L, D, B, contraction, tol, dev):
    """The BASELINE workloads with bench.py's own scenes and weights against the batched CPU oracle: every value within
    tolerance, decisions identical except on numerical ties.  configs[2] is checked in full (2048 roots, 510 k value
    forwards, ~20 s of CPU time on the GPU box); configs[4] at the full 256-root share each GPU is benchmarked at (round 3; the
    oracle walks it in chunks of 64 roots -- 1.3 MFLOP per forward there -- and its outputs are shared by the f32 / f16 cases);
    configs[3] also IN FULL (round 4: 4096 roots at depth 3 -- 16 parents per CU at the deepest level, another dealing plan than
    its 512-root share; 2.4 M value forwards for the oracle, ~70 s of CPU time).
    Two bounds: the north star's (1e-4; 1e-3 with f16 contractions) and the regression-level one (REG_F32 / REG_F16)."""
    import bench
    pol = _bench_policy(L, D, H, contraction, dev)
    robot, humans = bench.synth_scenes(1000, B, H)
    act, val = pol.predict_batch(robot.to(dev), humans.to(dev), roots_are_joint_states=True)
    oracle_out, v1, levels = _oracle_at_size(H, L, D, B, robot, humans)
    reg = REG_F16 if contraction == "f16" else REG_F32          # bf16x6 keeps 24-bit operands: held to the f32 kernels' bound
    worst, n_div, _ = compare_trees("at size, %s" % tag, pol.tree_search(), val, oracle_out, levels, tol, reg)
    err = close(val.cpu().numpy(), oracle_out[1].numpy(), tol=tol, reg=None if n_div else reg)
    check_decisions("at size, %s" % tag, act, val, oracle_out, [{"value1": v1}], tol)
    report("at size, %s: max |dV| = %.2e absolute (max |V| = %.3f)" % (tag, err, float(oracle_out[1].abs().max())))


def test_configs3_in_full_properties(dev):
    """configs[3] in full on one GPU (4096 roots, depth 3): run-to-run determinism, root-permutation equivariance (bit-exact), the
    chosen value = the maximum of the reported root values, and the first 512 roots of the full batch searched on their own -- the
    per-GPU share's launch shapes, another dealing plan at every level -- agree to rounding."""
    import bench
    B, H = 4096, 19
    pol = _bench_policy(2, 3, H, "f32", dev)
    robot, humans = bench.synth_scenes(1000, B, H)
    r, h = robot.to(dev), humans.to(dev)
    a1, v1 = pol.predict_batch(r, h)
    a1, v1 = a1.clone(), v1.clone()
    out = pol.tree_search().last
    assert torch.equal(out["root_values"].max(dim=1).values, out["best_value"])
    a2, v2 = pol.predict_batch(r, h)
    assert torch.equal(a1, a2) and torch.equal(v1, v2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(dev)
    a3, v3 = pol.predict_batch(r[perm].contiguous(), h[perm].contiguous())
    assert torch.equal(a3, a1[perm]) and torch.equal(v3, v1[perm])
    a4, v4 = pol.predict_batch(r[:512].contiguous(), h[:512].contiguous())
    dv = float((v4 - v1[:512]).abs().max())
    assert dv <= REG_F32, dv
    differ = a4 != a1[:512]
    assert int(differ.sum()) <= 2, int(differ.sum())
    report("configs[3] in full: deterministic, root-permutation equivariant (bit-exact), 512-root share vs the same roots inside "
           "the 4096-root batch: max |dV| = %.1e, %d decisions differ" % (dv, int(differ.sum())))


# kernel family forced for a whole process -> (environment, roots of configs[2] it is run on)
FORCED_FAMILIES = {
    "two-stage pair (children_rank1_kernel + robot_head_kernel)": ({"RGL_CHILDREN_TWO_STAGE": "1"}, 2048),
    "fused kernel without its tail (stand-alone select / back-up / root kernels)": ({"RGL_FUSED_NO_TAIL": "1"}, 2048),
    "MFMA tile kernel (children_graph_kernel)": ({"RGL_CHILDREN_TILE_KERNEL": "1"}, 2048),
    "fused kernel forced at every launch size": ({"RGL_CHILDREN_FUSED": "1"}, 2048),
    "general VALU kernel": ({"RGL_FORCE_GENERIC": "1"}, 256),
    "fused kernel with the round-3 register-staged weight image": ({"RGL_FUSED_IMAGE_SYNC": "1"}, 2048),
}


@pytest.mark.parametrize("family", list(FORCED_FAMILIES), ids=lambda f: f.split(" (")[0].replace(" ", "_"))
def test_forced_kernel_families_at_size(family, dev, tmp_path):
    """DESIGN.md section 5 says every kernel family serves the whole path; the driver's `pytest -m gpu` only sees the default
    dispatch.  Here the at-size configs[2] oracle check (bench.py's scenes and weights, all 2048 roots; the first 256 for the
    general VALU kernel) is re-run in a child process under each forcing switch -- the switches are read once per process.  The
    oracle's outputs are computed once in this process (memoised with the default-dispatch test) and handed over as a file."""
    import subprocess
    import sys
    import bench
    env_add, B = FORCED_FAMILIES[family]
    H, L, D = 19, 2, 2
    robot, humans = bench.synth_scenes(1000, 2048, H)
    oracle_out, v1, _ = _oracle_at_size(H, L, D, 2048, robot, humans)
    ref = str(tmp_path / "oracle_c2.npz")
    np.savez(ref, oa=oracle_out[0].numpy()[:B], ov=oracle_out[1].numpy()[:B], orv=oracle_out[2].numpy()[:B],
             okept=oracle_out[3].numpy()[:B], v1=v1.numpy()[:B])
    code = r'''
import sys, numpy as np, torch
import bench
from tests import test_gpu_parity as T
ref = np.load(sys.argv[1])
B = int(sys.argv[2])
dev = torch.device("cuda:0")
pol = T._bench_policy(2, 2, 19, "f32", dev)
robot, humans = bench.synth_scenes(1000, 2048, 19)
act, val = pol.predict_batch(robot[:B].to(dev), humans[:B].to(dev), roots_are_joint_states=True)
err = T.close(val.cpu().numpy(), ref["ov"], tol=T.TOL, reg=T.REG_F32)
oracle_out = [torch.tensor(ref[k]) for k in ("oa", "ov", "orv", "okept")]
n = T.check_decisions("forced", act, val, oracle_out, [{"value1": torch.tensor(ref["v1"])}], T.TOL)
print("OK max |dV| = %.2e, %d of %d decisions differ (ties in the oracle)" % (err, n, B))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, ref, str(B)], cwd=root, env=dict(os.environ, **env_add), capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    report("configs[2] at size (%d roots) with %s forced (%s): %s" % (B, family, " ".join("%s=%s" % kv for kv in env_add.items()),
                                                                      out.stdout.strip().splitlines()[-1]))


# ---------------------------------------------------------------------------------------------------
# path G
# ---------------------------------------------------------------------------------------------------
def test_path_g_rotate(dev):
    g = gio.load("path_g")
    x = torch.tensor(g["g.rotate_in"]).to(dev)
    close(rga.rotate(x).cpu().numpy(), g["g.rotate_out"], 2e-6)
    close(rga.rotate(x, "unicycle").cpu().numpy(), g["g.rotate_out_unicycle"], 2e-6)


def test_path_g_value_network(dev):
    g = gio.load("path_g")
    x = torch.tensor(g["g.vn_in"]).to(dev)
    for tag in g["g_cases"]:
        tag = str(tag)
        L, lw, sk = int(tag[1]), bool(int(tag[5])), bool(int(tag[9]))
        pol = make_gcn_policy(L, lw, sk, device=dev)
        with torch.no_grad():
            v = pol.model(x)
            v2 = pol.model((x, None))                       # (tensor, lengths) form is accepted too
        close(v.cpu().numpy(), g["g.vn_value." + tag])
        assert torch.equal(v, v2)
        close(pol.model.A, g["g.vn_A0." + tag])


def test_path_g_predict(dev):
    g = gio.load("path_g")
    pol = make_gcn_policy(device=dev)
    for b in range(g["g.pred_robot"].shape[0]):
        js = JS(g["g.pred_robot"][b], g["g.pred_humans"][b])
        a = pol.predict(js)
        assert a == pol.action_space[int(g["g.pred_action"][b])]
        close(np.array(pol.action_values), g["g.pred_action_values"][b])
        close(pol.get_matrix_A(), g["g.pred_A_last"][b])
    # batched device API against the sequential oracle
    robot = torch.tensor(g["g.pred_robot"].astype(np.float32)).to(dev)
    humans = torch.tensor(g["g.pred_humans"].astype(np.float32)).to(dev)
    vals, best = pol.gcn_search().search(robot, humans)
    assert np.array_equal(best.cpu().numpy().astype(np.int64), g["g.pred_action"])
    close(vals.cpu().numpy(), g["g.pred_action_values"])


def test_path_g_bf16x6_weight_products_at_size(dev):
    """ABI 8: GcnPlanner.contraction_dtype = RGL_CONTRACT_BF16X6 -- the graph's weight products (Wa, W_0) of the scene kernel's value-rows
    mode as six bf16 MFMA terms over three-piece operands (N = 20: two node tiles).  Held to the f32 bounds against the batched
    oracle, next to the f32 kernels' own deviation; a crowd the mode does not cover (N = 6: one node tile) runs plain f32 bit for bit."""
    import bench
    H, B = 19, 512
    robot, humans = bench.synth_scenes(2000 + H, B, H)
    ob, ov = orc.gcn_predict_batched(robot.numpy(), humans.numpy(), gio.path_g_sd(), orc.OracleConfig())
    out = {}
    for mode in ("f32", "bf16x6"):
        pol = make_gcn_policy(device=dev)
        pol.contraction_dtype = mode
        pol.build_action_space(1.0)
        vals, best = pol.gcn_search().search(robot.to(dev), humans.to(dev))
        out[mode] = (vals.cpu().numpy().astype(np.float64), best.cpu().numpy().astype(np.int64))
    e32 = close(out["f32"][0], ov)
    e6 = close(out["bf16x6"][0], ov)
    assert not np.array_equal(out["f32"][0], out["bf16x6"][0])          # the mode did run
    for b in np.nonzero(out["bf16x6"][1] != ob)[0]:
        assert ov[b, ob[b]] - ov[b, out["bf16x6"][1][b]] <= TOL
    assert e6 <= 2.0 * e32 + 5e-8, (e6, e32)              # the mode keeps 24-bit operands: no worse than the f32 kernels' own deviation
    report("path G, bf16x6 weight products, H=19 B=512: max |d action value| vs the oracle %.2e (f32 kernels: %.2e)" % (e6, e32))
    r5, h5 = bench.synth_scenes(2005, 64, 5)
    got = []
    for mode in ("f32", "bf16x6"):
        pol = make_gcn_policy(device=dev)
        pol.contraction_dtype = mode
        pol.build_action_space(1.0)
        got.append(pol.gcn_search().search(r5.to(dev), h5.to(dev))[0])
    assert torch.equal(got[0], got[1])


@pytest.mark.parametrize("H,B,kin", [(5, 512, "holonomic"), (19, 512, "holonomic"), (19, 96, "unicycle"), (49, 64, "holonomic")])
def test_path_g_at_size_against_the_batched_oracle(H, B, kin, dev):
    """path G beyond the five fixture scenes -- B x 81 rotated scenes at H = 5 / 19 (and a dense H = 49 crowd)
    through gcn_predict_f32 (gcn_prepare_kernel, row_mlp2_pair_kernel<6,7>, scene_graph_kernel, robot_head_kernel<150,100,100>,
    gcn_argmax_kernel) against the batched restatement of the reference's loop, which tests/test_oracle_golden.py pins on fixture F7.
    test_scene_kernel_split_and_unsplit_forced re-runs this test with the scene kernel forced into each of its organisations."""
    import bench
    robot, humans = bench.synth_scenes(2000 + H, B, H)
    pol = make_gcn_policy(device=dev)
    pol.kinematics = kin
    pol.build_action_space(1.0)
    vals, best = pol.gcn_search().search(robot.to(dev), humans.to(dev))
    cfg = orc.OracleConfig(kinematics=kin)
    ob, ov = orc.gcn_predict_batched(robot.numpy(), humans.numpy(), gio.path_g_sd(), cfg)
    got_v, got_a = vals.cpu().numpy().astype(np.float64), best.cpu().numpy().astype(np.int64)
    err = close(got_v, ov)
    differ = np.nonzero(got_a != ob)[0]
    for b in differ:                              # a different action only on a tie in the oracle
        assert ov[b, ob[b]] - ov[b, got_a[b]] <= TOL, (b, ov[b, ob[b]], ov[b, got_a[b]])
    # the value of the chosen action as the argmax kernel itself reports it (ABI 8: gcn_predict_f32's best_value; predict_batch hands
    # it on instead of gathering): bit for bit the entry of the action-value table
    pa, pv = pol.predict_batch(robot.to(dev), humans.to(dev))
    assert torch.equal(pa, best) and torch.equal(pv, vals.gather(1, best.long()[:, None])[:, 0])
    report("path G at size, H=%d B=%d %s: %d of %d decisions differ from the oracle (ties in the oracle); max |d action value| = %.2e"
           % (H, B, kin, len(differ), B, err))


# ---------------------------------------------------------------------------------------------------
# error behaviour: loud, never a silent fallback
# ---------------------------------------------------------------------------------------------------
def test_no_cpu_fallback_and_no_silent_autograd(dev):
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    g1, ve, sp = build_modules(c, dev)
    r = torch.zeros(1, 1, 9)
    h = torch.zeros(1, 3, 5)
    with torch.no_grad(), pytest.raises(nat.NativeLibraryError):
        ve((r, h))                                            # CPU tensors
    big = build_modules(dict(L=3, sim="concatenation", layerwise=True, skip=True, flavour="trained"), dev)[1]
    out = big((r.to(dev), torch.zeros(1, 63, 5, device=dev)))           # forward under grad is fine ...
    with pytest.raises(nat.NativeLibraryError):
        out.sum().backward()                                  # ... but N = 64 activations of the backward exceed one CU's LDS: loud
    with torch.no_grad(), pytest.raises(nat.NativeLibraryError):
        ve((r.to(dev), torch.zeros(1, 128, 5, device=dev)))   # N = 129 > RGL_MAX_NODES (128 since round 3)


# ---------------------------------------------------------------------------------------------------
# training path: gradients through the HIP forward (rgl_graph_backward_f32) against torch autograd on the oracle
# ---------------------------------------------------------------------------------------------------
def _oracle_leafs(module_sd):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in module_sd.items()}


GRAD_WORST = {"rel": 0.0, "name": ""}


def _grad_close(got, want, name, tol=2e-4, reg=REG_GRAD):
    """`tol`: the bound of rounds 1-3 (relative to the gradient's largest entry); `reg`: the regression-level bound (round 4)."""
    got, want = got.detach().cpu().numpy().astype(np.float64), want.detach().numpy().astype(np.float64)
    scale = max(1e-3, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    if reg is not None and err / scale > GRAD_WORST["rel"]:          # the comparisons against autograd over the oracle
        GRAD_WORST["rel"], GRAD_WORST["name"] = err / scale, name
    assert err <= tol * scale, (name, err, scale)
    if reg is not None:
        assert err <= reg * scale, ("regression-level bound", name, err, reg, scale)


@pytest.mark.parametrize("H,L,sim,skip,flavour,B,layerwise", [
    (5, 2, "embedded_gaussian", True, "trained", 7, False),
    (19, 2, "embedded_gaussian", True, "trained", 4, False),
    (5, 3, "embedded_gaussian", False, "trained", 5, False),
    (3, 1, "gaussian", True, "trained", 6, False),
    (5, 2, "embedded_gaussian", True, "rand", 3, False),
    (5, 2, "squared", True, "trained", 5, False),
    (7, 3, "squared", False, "trained", 3, False),
    (5, 2, "equal_attention", True, "trained", 4, False),
    (4, 2, "diagonal", True, "trained", 4, False),
    (5, 2, "cosine", True, "trained", 5, False),
    (7, 3, "cosine", False, "trained", 3, False),
    (5, 2, "cosine_softmax", True, "trained", 5, False),
    (5, 2, "concatenation", True, "trained", 4, False),
    (9, 1, "concatenation", False, "rand", 3, False),
    (5, 2, "embedded_gaussian", True, "trained", 5, True),
    (6, 3, "embedded_gaussian", False, "trained", 3, True),
    (5, 2, "gaussian", True, "trained", 4, True),
    (5, 2, "cosine", True, "trained", 4, True),
    (4, 3, "cosine_softmax", False, "trained", 3, True),
    (5, 2, "concatenation", True, "trained", 4, True),
    (5, 2, "squared", True, "trained", 4, True),
    (19, 2, "concatenation", True, "trained", 2, True),
    (19, 2, "cosine_softmax", True, "trained", 2, False),])
def test_gradients_value_estimator_and_state_predictor(H, L, sim, skip, flavour, B, layerwise, dev):
    c = dict(L=L, sim=sim, layerwise=layerwise, skip=skip, flavour=flavour)
    g1, ve, sp = build_modules(c, dev)
    robot, humans = seeded_scenes(900 + H + L, B, H)
    cfg = orc.OracleConfig(num_layer=L, similarity=sim, skip_connection=skip, layerwise_graph=layerwise)
    wv = torch.linspace(-1.0, 1.5, B).reshape(B, 1)
    # value estimator
    for p_ in ve.parameters():
        p_.grad = None
    out = ve((robot.unsqueeze(1).to(dev), humans.to(dev)))
    (out * wv.to(dev)).sum().backward()
    gsd, vsd = _oracle_leafs(ve.graph_model.state_dict()), _oracle_leafs(ve.value_network.state_dict())
    want = orc.value_estimator_forward(robot.unsqueeze(1), humans, gsd, vsd, cfg)
    close(out.detach().cpu().numpy(), want.detach().numpy())
    (want * wv).sum().backward()
    for k, v in ve.graph_model.named_parameters():
        _grad_close(v.grad, gsd[k].grad, "graph." + k)
    for k, v in ve.value_network.named_parameters():
        _grad_close(v.grad, vsd[k].grad, "value." + k)
    # state predictor, with and without detach
    wm = torch.randn(B, H, 5, generator=torch.Generator().manual_seed(3))
    for detach in (False, True):
        for p_ in sp.parameters():
            p_.grad = None
        _, nh = sp((robot.unsqueeze(1).to(dev), humans.to(dev)), None, detach=detach)
        (nh * wm.to(dev)).sum().backward()
        gsd, msd = _oracle_leafs(sp.graph_model.state_dict()), _oracle_leafs(sp.human_motion_predictor.state_dict())
        emb, _ = orc.rgl_forward(robot.unsqueeze(1), humans, gsd, cfg)
        if detach:
            emb = emb.detach()
        wantm = orc.mlp_forward(emb, orc.mlp_layers(msd, ""), last_relu=False)[:, 1:, :]
        (wantm * wm).sum().backward()
        for k, v in sp.human_motion_predictor.named_parameters():
            _grad_close(v.grad, msd[k].grad, "motion." + k)
        for k, v in sp.graph_model.named_parameters():
            if detach:
                assert v.grad is None or float(v.grad.abs().max()) == 0.0
            else:
                _grad_close(v.grad, gsd[k].grad, "sp_graph." + k)


@pytest.mark.parametrize("H,L,sim,skip,flavour,B", [
    (5, 2, "embedded_gaussian", True, "trained", 7),
    (19, 2, "embedded_gaussian", True, "trained", 4),
    (5, 3, "embedded_gaussian", False, "trained", 5),
    (3, 1, "gaussian", True, "trained", 6),
    (5, 2, "embedded_gaussian", True, "rand", 3),
    (16, 2, "embedded_gaussian", True, "trained", 35),
    (40, 2, "gaussian", False, "trained", 3),
    # round 5 (VERDICT r4 missing 4): the plain-weight similarity functions on the tile pipeline (graph_model.py:86-93)
    (19, 2, "squared", True, "trained", 6),
    (5, 3, "squared", False, "trained", 9),
    (33, 2, "squared", True, "trained", 3),
    (19, 2, "equal_attention", True, "trained", 5),
    (7, 1, "equal_attention", False, "trained", 4),
    (19, 2, "diagonal", True, "trained", 5),
    (12, 3, "diagonal", False, "trained", 4),
    # ... and the cosine family (graph_model.py:70-79: S over the outer product of the norms of S's ROWS, then softmax or not)
    (5, 2, "cosine", True, "trained", 5),
    (7, 3, "cosine", False, "trained", 3),
    (33, 2, "cosine", True, "trained", 3),
    (5, 2, "cosine_softmax", True, "trained", 5),
    (19, 2, "cosine_softmax", True, "trained", 4),
    (4, 3, "cosine_softmax", False, "trained", 6),
    (40, 1, "cosine_softmax", True, "trained", 3),])
def test_gradients_on_the_mfma_backward(H, L, sim, skip, flavour, B, dev, monkeypatch):
    """The tile pipeline of rgl_backward_mfma.hip (the large-batch backward: MFMA row kernels for the MLPs, one wave per scene for the
    graph block) forced on for the small oracle-autograd cases of the test above (it is chosen by batch size otherwise), and for the
    RGL-output / path-G gradients."""
    monkeypatch.setenv("RGL_BACKWARD_MFMA", "2")        # 2: an error instead of the per-scene kernel where the pipeline says "not mine"
    test_gradients_value_estimator_and_state_predictor(H, L, sim, skip, flavour, B, False, dev)
    test_gradients_rgl_output_and_path_g(dev)


@pytest.mark.parametrize("H,L,sim,skip,B", [(5, 2, "embedded_gaussian", True, 5), (6, 3, "embedded_gaussian", False, 3),
                                            (5, 2, "gaussian", True, 4), (19, 2, "embedded_gaussian", True, 3),
                                            (19, 3, "gaussian", False, 2), (31, 1, "embedded_gaussian", True, 2),
                                            (5, 2, "squared", True, 4), (19, 3, "squared", False, 2),
                                            # constant adjacencies: a layerwise graph of these IS the one-adjacency graph
                                            (7, 2, "equal_attention", True, 3), (12, 3, "diagonal", False, 2)])
def test_gradients_of_layerwise_graphs_on_the_mfma_backward(H, L, sim, skip, B, dev, monkeypatch):
    """VERDICT r5 missing 4 / next 7: layerwise graphs (graph_model.py:118-122: an adjacency per layer, A_l = softmax(S(H_l))) on
    the tile pipeline in MUST-RUN mode (RGL_BACKWARD_MFMA=2: an error instead of the per-scene kernel) -- the forward recomputes
    the similarity block from every layer's input, the backward goes through it inside the layer loop (softmax and squared
    normalisations; equal_attention / diagonal adjacencies do not depend on the input).  Every gradient of the value estimator and of
    the state predictor against autograd over the oracle."""
    monkeypatch.setenv("RGL_BACKWARD_MFMA", "2")
    test_gradients_value_estimator_and_state_predictor(H, L, sim, skip, "trained", B, True, dev)


def test_reference_vnrl_trainer_fixture_with_a_layerwise_graph_on_the_tile_backward(dev, monkeypatch):
    """Fixture vnrl_trainer.npz, case layerwise_noskip (the REFERENCE VNRLTrainer's batches on a layerwise path-G model) with the
    tile pipeline in must-run mode, through the raw loop and through the public trainer."""
    monkeypatch.setenv("RGL_BACKWARD_MFMA", "2")
    test_training_against_the_reference_vnrl_trainer_fixture("layerwise_noskip|2|1|0", dev)
    test_product_vnrl_trainer_reproduces_the_reference_fixture("layerwise_noskip|2|1|0", dev)


@pytest.mark.parametrize("tag", ["squared", "cosine_softmax"])
def test_reference_trainer_fixture_with_a_non_default_similarity_on_the_tile_backward(tag, dev, monkeypatch):
    """the reference MPRLTrainer's three Adam batches with the `squared` (graph_model.py:86-89) and the
    `cosine_softmax` (:75-79) similarity functions (fixture training_queryenv.npz, cases of those names) reproduced with the tile
    pipeline in MUST-RUN mode (RGL_BACKWARD_MFMA=2: an error instead of the per-scene kernel) -- through the raw loop and through the
    public trainer's captured step."""
    monkeypatch.setenv("RGL_BACKWARD_MFMA", "2")
    test_training_against_the_reference_trainer_fixture(tag, dev)
    test_product_trainer_reproduces_the_reference_trainer_fixture(tag, dev)


@pytest.mark.parametrize("H,L,B", [(19, 2, 1024), (5, 2, 1500), (49, 3, 96), (31, 1, 200),
                                   (3, 2, 16500)])      # beyond 1024 row tiles of the value head: mlp_rows_kernel's many-tiles form
def test_mfma_backward_at_size(H, L, B, dev, monkeypatch):
    """Batches the vector explorer feeds: every parameter gradient of the value estimator and of the state
    predictor from the tile pipeline against torch autograd over the oracle AND against the per-scene VALU kernel of
    rgl_backward.hip; two runs are bit-identical (fixed tile -> wave -> slab order)."""
    c = dict(L=L, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    g1, ve, sp = build_modules(c, dev)
    robot, humans = seeded_scenes(4000 + H, B, H)
    r, h = robot.unsqueeze(1).to(dev), humans.to(dev)
    cfg = orc.OracleConfig(num_layer=L, similarity="embedded_gaussian", skip_connection=True, layerwise_graph=False)
    gen = torch.Generator().manual_seed(11)
    wv = torch.randn(B, 1, generator=gen)
    wm = torch.randn(B, H, 5, generator=gen)

    def grads(mode):
        monkeypatch.setenv("RGL_BACKWARD_MFMA", mode)
        for p_ in list(ve.parameters()) + list(sp.parameters()):
            p_.grad = None
        (ve((r, h)) * wv.to(dev)).sum().backward()
        _, nh = sp((r, h), None, detach=False)
        (nh * wm.to(dev)).sum().backward()
        return {("ve." if m is ve else "sp.") + k: v.grad.detach().clone() for m in (ve, sp) for k, v in m.named_parameters()}
    tiles, again = grads("1"), grads("1")
    try:
        valu = grads("0")
    except nat.NativeLibraryError:            # N = 50 with three layers: a scene does not fit the per-scene kernel's LDS
        assert H >= 49
        valu = None
    for k in tiles:
        assert torch.equal(tiles[k], again[k]), k
        if valu is not None:
            _grad_close(tiles[k], valu[k].cpu(), "tiles vs per-scene kernel " + k, tol=5e-5, reg=None)
    gsd, vsd = _oracle_leafs(ve.graph_model.state_dict()), _oracle_leafs(ve.value_network.state_dict())
    (orc.value_estimator_forward(robot.unsqueeze(1), humans, gsd, vsd, cfg) * wv).sum().backward()
    worst = 0.0
    for prefix, sd in (("ve.graph_model.", gsd), ("ve.value_network.", vsd)):
        for k, v in sd.items():
            _grad_close(tiles[prefix + k], v.grad, prefix + k)
            worst = max(worst, float((tiles[prefix + k].cpu() - v.grad).abs().max() / max(1e-3, float(v.grad.abs().max()))))
    gsd, msd = _oracle_leafs(sp.graph_model.state_dict()), _oracle_leafs(sp.human_motion_predictor.state_dict())
    emb, _ = orc.rgl_forward(robot.unsqueeze(1), humans, gsd, cfg)
    (orc.mlp_forward(emb, orc.mlp_layers(msd, ""), last_relu=False)[:, 1:, :] * wm).sum().backward()
    for k, v in gsd.items():
        _grad_close(tiles["sp.graph_model." + k], v.grad, "sp.graph_model." + k)
    for k, v in msd.items():
        _grad_close(tiles["sp.human_motion_predictor." + k], v.grad, "sp.human_motion_predictor." + k)
    report("tile-pipeline backward, H=%d L=%d batch %d: every gradient within 2e-4 of autograd over the oracle (worst %.1e of the "
           "largest entry), 5e-5 of the per-scene kernel, bit-identical between runs" % (H, L, B, worst))


def test_gradients_rgl_output_and_path_g(dev):
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    g1, _, _ = build_modules(c, dev)
    robot, humans = seeded_scenes(77, 4, 5)
    wH = torch.randn(4, 6, 32, generator=torch.Generator().manual_seed(5))
    out = g1((robot.unsqueeze(1).to(dev), humans.to(dev)))
    (out * wH.to(dev)).sum().backward()
    gsd = _oracle_leafs(g1.state_dict())
    want, _ = orc.rgl_forward(robot.unsqueeze(1), humans, gsd, orc.OracleConfig())
    (want * wH).sum().backward()
    for k, v in g1.named_parameters():
        _grad_close(v.grad, gsd[k].grad, "rgl." + k)
    # path G value network (L=2 and the skip-less one-layer variant)
    g = gio.load("path_g")
    x = torch.tensor(g["g.vn_in"])
    for L in (2, 1):
        pol = make_gcn_policy(L, False, True, device=dev)
        v = pol.model(x.to(dev))
        v.sum().backward()
        sd = _oracle_leafs(pol.model.state_dict())
        wantv, _ = orc.gcn_value_forward(x, sd, orc.OracleConfig(num_layer=L))
        wantv.sum().backward()
        for k, p_ in pol.model.named_parameters():
            _grad_close(p_.grad, sd[k].grad, "gcn%d.%s" % (L, k))


def test_training_step_matches_cpu_autograd(dev):
    """Five Adam steps of the value-estimator TD regression (the body of MPRLTrainer.optimize_batch,
    crowd_nav/utils/trainer.py:110-161) on the HIP path versus the same loop on the CPU oracle."""
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    _, ve, _ = build_modules(c, dev)
    import copy
    target = copy.deepcopy(ve)
    robot, humans = seeded_scenes(31, 100, 5)
    nrobot, nhumans = seeded_scenes(32, 100, 5)
    rewards = torch.linspace(-0.25, 1.0, 100).reshape(100, 1)
    gamma_bar = 0.9 ** 0.25
    gsd, vsd = _oracle_leafs(ve.graph_model.state_dict()), _oracle_leafs(ve.value_network.state_dict())
    tg = {k: v.detach().clone() for k, v in gsd.items()}
    tv = {k: v.detach().clone() for k, v in vsd.items()}
    cfg = orc.OracleConfig()
    opt_gpu = torch.optim.Adam(ve.parameters(), lr=1e-3)
    opt_cpu = torch.optim.Adam(list(gsd.values()) + list(vsd.values()), lr=1e-3)
    crit = torch.nn.MSELoss()
    losses_gpu, losses_cpu = [], []
    for _ in range(5):
        opt_gpu.zero_grad()
        out = ve((robot.unsqueeze(1).to(dev), humans.to(dev)))
        tgt = rewards.to(dev) + gamma_bar * target((nrobot.unsqueeze(1).to(dev), nhumans.to(dev)))
        loss = crit(out, tgt)
        loss.backward()
        opt_gpu.step()
        losses_gpu.append(float(loss.detach()))
        opt_cpu.zero_grad()
        o2 = orc.value_estimator_forward(robot.unsqueeze(1), humans, gsd, vsd, cfg)
        with torch.no_grad():
            t2 = rewards + gamma_bar * orc.value_estimator_forward(nrobot.unsqueeze(1), nhumans, tg, tv, cfg)
        l2 = crit(o2, t2)
        l2.backward()
        opt_cpu.step()
        losses_cpu.append(float(l2.detach()))
    assert np.allclose(losses_gpu, losses_cpu, rtol=2e-4, atol=1e-6), (losses_gpu, losses_cpu)
    assert losses_gpu[-1] < losses_gpu[0]
    for k, v in ve.graph_model.named_parameters():
        close(v.detach().cpu().numpy(), gsd[k].detach().numpy(), 5e-4)


def test_target_model_deepcopy_after_forward(dev):
    """copy.deepcopy(model) AFTER forwards (descriptor caches populated), then train against the copy as the frozen
    target -- the order crowd_nav/train.py:161,206 + trainer.py:41,187 use (this used to raise)."""
    import copy
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    _, ve, sp = build_modules(c, dev)
    robot, humans = seeded_scenes(41, 16, 5)
    state = (robot.unsqueeze(1).to(dev), humans.to(dev))
    with torch.no_grad():
        v0 = ve(state).clone()
        sp(state, None)
    out = ve(state)                              # a forward under grad as well
    out.sum().backward()
    target, sp_copy = copy.deepcopy(ve), copy.deepcopy(sp)
    with torch.no_grad():
        assert torch.equal(target(state), v0)
        assert torch.equal(sp_copy(state, None)[1], sp(state, None)[1])
    opt = torch.optim.Adam(ve.parameters(), lr=1e-2)
    for _ in range(2):
        opt.zero_grad()
        with torch.no_grad():
            tgt = 0.1 + 0.9 * target(state)
        loss = torch.nn.functional.mse_loss(ve(state), tgt)
        loss.backward()
        opt.step()
    with torch.no_grad():
        assert torch.equal(target(state), v0)                 # the copy owns its parameters and its own descriptors
        assert not torch.equal(ve(state), v0)
    pol = make_gcn_policy(device=dev)
    rot = torch.randn(4, 5, 13, device=dev)
    with torch.no_grad():
        g0 = pol.model(rot).clone()
    g2 = copy.deepcopy(pol.model)
    with torch.no_grad():
        assert torch.equal(g2(rot), g0)


def test_bench_runs_under_torchrun_over_rccl(dev):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, backend nccl = RCCL), with the one
    rank a one-GPU box can host: RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the world-size-2 exchange is
    covered by the gloo test and this executes the RCCL all-gather / barrier / all-reduce path on hardware.  Strong-scaling flags."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RGL_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--scaling", "strong", "--total-roots", "96", "--depth", "3", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(lines[0])
    assert r["scaling"] == "strong" and r["n_gpus"] == 1 and r["config"]["total_roots"] == 96
    assert r["multi_gpu"]["roots_per_rank"] == [96] and r["multi_gpu"]["exchange_ms_per_step_slowest_rank"] >= 0.0
    assert r["value"] > 0 and r["roofline"]["frac"] > 0


def test_backward_refuses_stale_parameters_and_state_gradients(dev):
    """the backward recomputes the forward from the current parameters -- an in-place update between forward and
    backward must raise (as torch does for its saved tensors), and so must inputs that ask for gradients."""
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    _, ve, _ = build_modules(c, dev)
    robot, humans = seeded_scenes(61, 8, 5)
    state = (robot.unsqueeze(1).to(dev), humans.to(dev))
    out = ve(state).sum()
    with torch.no_grad():
        ve.value_network[0].weight.mul_(1.001)
    with pytest.raises(RuntimeError):
        out.backward()
    with pytest.raises(NotImplementedError):
        ve((state[0], state[1].clone().requires_grad_(True)))


@pytest.mark.parametrize("tag", ["plain", "detach", "squared", "cosine_softmax"])
def test_training_against_the_reference_trainer_fixture(tag, dev):
    """Fixture training_queryenv.npz: the REFERENCE MPRLTrainer.optimize_batch (crowd_nav/utils/trainer.py:110-161) ran three
    un-shuffled batches of 16 transitions on the reference modules (Adam 1e-3, frozen target copy; skip_connection=False, the only
    way upstream trains on current torch).  The same loop on the product's modules (forward + rgl_graph_backward_f32 on the GPU) must
    end at the same parameters and report the same losses."""
    import copy
    fx = gio.load("training_queryenv")
    c = dict(L=2, sim=tag if tag in ("squared", "cosine_softmax") else "embedded_gaussian", layerwise=False, skip=False, flavour="trained")
    _, ve, sp = build_modules(c, dev)
    detach = tag == "detach"
    target = copy.deepcopy(ve)
    v_opt = torch.optim.Adam(ve.parameters(), lr=1e-3)
    s_opt = torch.optim.Adam(sp.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    gamma_bar = pow(0.9, 0.25 * 1)
    v_losses = s_losses = 0.0
    for b in range(3):
        sl = slice(16 * b, 16 * b + 16)
        r = torch.tensor(fx["tr.robot"][sl]).unsqueeze(1).to(dev)
        h = torch.tensor(fx["tr.humans"][sl]).to(dev)
        r2 = torch.tensor(fx["tr.next_robot"][sl]).unsqueeze(1).to(dev)
        h2 = torch.tensor(fx["tr.next_humans"][sl]).to(dev)
        rew = torch.tensor(fx["tr.rewards"][sl]).to(dev)
        v_opt.zero_grad()
        out = ve((r, h))
        tgt = rew + gamma_bar * target((r2, h2))
        loss = crit(out, tgt)
        loss.backward()
        v_opt.step()
        v_losses += float(loss.detach())
        s_opt.zero_grad()
        _, nh = sp((r, h), None, detach=detach)
        loss = crit(nh, h2)
        loss.backward()
        s_opt.step()
        s_losses += float(loss.detach())
    want_v, want_s = fx["tr.%s.losses" % tag]                       # the reference divides by num_batches = 2
    assert abs(v_losses / 2 - want_v) <= 1e-5 * max(1.0, abs(want_v)) and abs(s_losses / 2 - want_s) <= 1e-5 * max(1.0, abs(want_s))
    worst = 0.0
    for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                      ("motion_predictor", sp.human_motion_predictor)):
        for k, v in mod.state_dict().items():
            want = fx["tr.%s.%s.%s" % (tag, name, k)]
            err = float(np.abs(v.cpu().numpy() - want).max())
            worst = max(worst, err)
            assert err <= 2e-5, (name, k, err)           # three Adam steps of 1e-3 move parameters by ~3e-3
    report("reference MPRLTrainer.optimize_batch (%s): final parameters within %.1e of the reference's, losses %.6f / %.6f"
           % (tag, worst, v_losses / 2, s_losses / 2))


@pytest.mark.parametrize("case", ["shipped|2|0|1", "layerwise_noskip|2|1|0"])
def test_training_against_the_reference_vnrl_trainer_fixture(case, dev):
    """Fixture vnrl_trainer.npz: the REFERENCE VNRLTrainer.optimize_batch (crowd_nav/utils/trainer.py:199-250, path G's
    trainer) ran three un-shuffled batches of 16 transitions on the reference gcn.ValueNetwork (Adam 1e-3, frozen target copy, its
    own pad_batch collate -> the `(states, lengths)` tuple form).  The same loop on the product's ValueNetwork (forward +
    rgl_graph_backward_f32 on the GPU) must end at the same parameters and report the same loss."""
    import copy
    fx = gio.load("vnrl_trainer")
    assert case in [str(c) for c in fx["vnrl_cases"]]
    tag, L, lw, sk = case.split("|")
    pol = make_gcn_policy(int(L), bool(int(lw)), bool(int(sk)), device=dev)
    model = pol.model
    target = copy.deepcopy(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    gamma_bar = pow(0.9, 0.25 * 1)
    losses = 0.0
    for b in range(3):
        sl = slice(16 * b, 16 * b + 16)
        x = torch.tensor(fx["vn.states"][sl]).to(dev)
        x2 = torch.tensor(fx["vn.next_states"][sl]).to(dev)
        rew = torch.tensor(fx["vn.rewards"][sl]).unsqueeze(1).to(dev)
        lengths = torch.full((16,), 5, dtype=torch.int64)
        opt.zero_grad()
        out = model((x, lengths))
        tgt = rew + gamma_bar * target((x2, lengths))
        loss = crit(out, tgt)
        loss.backward()
        opt.step()
        losses += float(loss.detach())
    want = float(fx["vn.%s.loss" % tag][0])                        # the reference divides by num_batches = 2
    assert abs(losses / 2 - want) <= 1e-5 * max(1.0, abs(want)), (losses / 2, want)
    worst = 0.0
    for k, v in model.state_dict().items():
        err = float(np.abs(v.cpu().numpy() - fx["vn.%s.model.%s" % (tag, k)]).max())
        worst = max(worst, err)
        assert err <= 2e-5, (k, err)
    report("reference VNRLTrainer.optimize_batch (%s): final parameters within %.1e of the reference's, loss %.6f"
           % (tag, worst, losses / 2))


class _Writer(object):
    def __init__(self):
        self.scalars = []

    def add_scalar(self, tag, value, step):
        self.scalars.append((tag, float(value), step))


class _ListDataset(torch.utils.data.Dataset):
    """A dataset WITHOUT as_tensors(): the trainers then draw their batches through torch's DataLoader, as upstream does."""

    def __init__(self, items):
        self.items = items

    def __getitem__(self, i):
        return self.items[i]

    def __len__(self):
        return len(self.items)


@pytest.mark.parametrize("tag", ["plain", "detach", "squared", "cosine_softmax"])
def test_product_trainer_reproduces_the_reference_trainer_fixture(tag, dev):
    """the fast optimisation step through the PUBLIC trainer (relationalgraphlearning_amd.MPRLTrainer: the
    reference's constructor / set_learning_rate / update_target_model / optimize_batch contract, every step after the first a
    replay of one captured hipGraph).  Same fixture as test_training_against_the_reference_trainer_fixture: the REFERENCE
    MPRLTrainer.optimize_batch over three un-shuffled batches of 16 transitions -- same final parameters, same reported losses."""
    fx = gio.load("training_queryenv")
    c = dict(L=2, sim=tag if tag in ("squared", "cosine_softmax") else "embedded_gaussian", layerwise=False, skip=False, flavour="trained")
    _, ve, sp = build_modules(c, dev)
    items = [(torch.tensor(fx["tr.robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.humans"][i]).to(dev),
              torch.zeros(1, device=dev), torch.tensor(fx["tr.rewards"][i]).reshape(1).to(dev),
              torch.tensor(fx["tr.next_robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.next_humans"][i]).to(dev)) for i in range(48)]
    writer = _Writer()
    t = rga.MPRLTrainer(ve, sp, _ListDataset(items), dev, None, writer, 16, "Adam", 5, reduce_sp_update_frequency=False,
                        freeze_state_predictor=False, detach_state_predictor=(tag == "detach"), share_graph_model=False)
    t.set_learning_rate(1e-3)
    t.update_target_model(ve)
    t.data_loader = torch.utils.data.DataLoader(t.memory, 16, shuffle=False)          # the fixture's order
    av, as_ = t.optimize_batch(2, 7)                                                # upstream's off-by-one: three batches
    assert t._capturable and len(t._steps) == 1                                    # one capture, two replays
    want_v, want_s = fx["tr.%s.losses" % tag]
    assert abs(av - want_v) <= 1e-5 * max(1.0, abs(want_v)) and abs(as_ - want_s) <= 1e-5 * max(1.0, abs(want_s)), (av, as_)
    assert [s[0] for s in writer.scalars] == ["RL/average_v_loss", "RL/average_s_loss"] and writer.scalars[0][2] == 7
    worst = 0.0
    for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                      ("motion_predictor", sp.human_motion_predictor)):
        for k, v in mod.state_dict().items():
            err = float(np.abs(v.cpu().numpy() - fx["tr.%s.%s.%s" % (tag, name, k)]).max())
            worst = max(worst, err)
            assert err <= 2e-5, (name, k, err)
    # after the call the modules' packed weights follow the trained parameters (replays do not bump version counters)
    r, h = items[0][0].unsqueeze(0), items[0][1].unsqueeze(0)
    with torch.no_grad():
        v_now = ve((r, h))
        rga.invalidate_packed_weights(ve)
        assert torch.equal(v_now, ve((r, h)))
    report("product MPRLTrainer.optimize_batch (%s; one captured step + two replays): final parameters within %.1e of the "
           "reference trainer's, losses %.6f / %.6f" % (tag, worst, av, as_))


def test_product_trainer_fast_batches_equal_the_dataloaders(dev):
    """The trainer's index sampling over ReplayMemory.as_tensors() against the DataLoader path on a plain dataset: same torch seed
    -> the same shuffled batches -> bit-identical parameters after an IL epoch and two RL calls (both replay captured steps); and
    the eager form (capture switched off) agrees to rounding."""
    import copy
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    robot, humans = seeded_scenes(61, 230, 5)
    robot2, humans2 = seeded_scenes(62, 230, 5)
    rew = torch.rand(230, generator=torch.Generator().manual_seed(9))
    items = [(robot[i:i + 1].to(dev), humans[i].to(dev), rew[i:i + 1].to(dev) * 0.5, rew[i:i + 1].to(dev),
              robot2[i:i + 1].to(dev), humans2[i].to(dev)) for i in range(230)]

    def run(kind):
        _, ve, sp = build_modules(c, dev)
        if kind == "fast":
            mem = rga.ReplayMemory(1000)
            for it in items:
                mem.push(it)
        else:
            mem = _ListDataset(items)
        cls = rga.MPRLTrainer
        if kind == "eager":
            cls = type("EagerTrainer", (rga.MPRLTrainer,), {"capture": False})
        t = cls(ve, sp, mem, dev, None, _Writer(), 100, "Adam", 5, reduce_sp_update_frequency=True,
                freeze_state_predictor=False, detach_state_predictor=True, share_graph_model=False)
        torch.manual_seed(77)
        t.set_learning_rate(1e-3)
        t.optimize_epoch(1)                                  # 100 + 100 + 30: two shapes, predictor every 5th batch
        t.update_target_model(ve)
        t.set_learning_rate(5e-4)
        losses = [t.optimize_batch(1, e) for e in range(2)]  # two batches per call, predictor skipped on batch 0
        t.update_target_model(ve)                            # in place: the captured steps stay
        losses.append(t.optimize_batch(1, 2))
        return ve, sp, losses, t
    ve_f, sp_f, l_f, t_f = run("fast")
    ve_d, sp_d, l_d, t_d = run("loader")
    ve_e, sp_e, l_e, _ = run("eager")
    assert t_f._capturable and t_d._capturable and len(t_f._steps) >= 2
    assert l_f == l_d
    worst = 0.0
    for a, b, e in ((ve_f, ve_d, ve_e), (sp_f, sp_d, sp_e)):
        for (k, pa), (_, pb), (_, pe) in zip(a.state_dict().items(), b.state_dict().items(), e.state_dict().items()):
            assert torch.equal(pa, pb), k
            worst = max(worst, float((pa - pe).abs().max()))
    assert worst <= 1e-5, worst          # capturable Adam + the tile-pipeline backward vs plain Adam + the per-scene kernel, ~8 steps
    for (a, b), (c_, d) in zip(l_f, l_e):
        assert abs(a - c_) <= 1e-6 * max(1.0, abs(c_)) and abs(b - d) <= 1e-6 * max(1.0, abs(d))
    report("product MPRLTrainer: index-sampled batches == DataLoader batches (bit-identical parameters); captured vs eager steps: "
           "parameters within %.1e" % worst)


def test_product_trainer_replays_stay_valid_while_the_memory_grows(dev):
    """a captured RL step gathers its batch with index_select from the memory's stacked fields.  The memory
    grows every episode, so the step must gather from the FULL-CAPACITY fields (ReplayMemory.stacked_capacity_fields): recorded
    against the len(memory)-row views of the day of the capture, a replay would read rows beyond the recorded source extent.
    Push items between optimize_batch calls (same batch shape -> the same captured step is replayed) and hold the captured
    trainer against the eager one on the same seeds."""
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    robot, humans = seeded_scenes(71, 420, 5)
    robot2, humans2 = seeded_scenes(72, 420, 5)
    rew = torch.rand(420, generator=torch.Generator().manual_seed(19))
    items = [(robot[i:i + 1].to(dev), humans[i].to(dev), rew[i:i + 1].to(dev) * 0.5, rew[i:i + 1].to(dev),
              robot2[i:i + 1].to(dev), humans2[i].to(dev)) for i in range(420)]

    def run(capture):
        _, ve, sp = build_modules(c, dev)
        mem = rga.ReplayMemory(400)                          # the last pushes wrap around the ring
        for it in items[:150]:
            mem.push(it)
        cls = rga.MPRLTrainer if capture else type("EagerTrainer", (rga.MPRLTrainer,), {"capture": False})
        t = cls(ve, sp, mem, dev, None, _Writer(), 100, "Adam", 5, reduce_sp_update_frequency=False,
                freeze_state_predictor=False, detach_state_predictor=True, share_graph_model=False)
        torch.manual_seed(5)
        t.set_learning_rate(1e-3)
        t.update_target_model(ve)
        losses, lo = [], 150
        for episode in range(4):
            losses.append(t.optimize_batch(1, episode))      # two batches of 100 (upstream's off-by-one), the second shorter at first
            for it in items[lo:lo + 90]:                     # the memory grows: 150 -> 240 -> 330 -> 400 (full, wrapped)
                mem.push(it)
            lo += 90
        return ve, sp, losses, t
    ve_c, sp_c, l_c, t_c = run(True)
    ve_e, sp_e, l_e, _ = run(False)
    assert t_c._capturable and any(st.graph is not None for st in t_c._steps.values())
    key_sizes = sorted(k[-1] for k in t_c._steps)            # the source extent each captured step was recorded with
    assert key_sizes and all(n == 400 for n in key_sizes), key_sizes
    worst = 0.0
    for a, e in ((ve_c, ve_e), (sp_c, sp_e)):
        for (k, pa), (_, pe) in zip(a.state_dict().items(), e.state_dict().items()):
            worst = max(worst, float((pa - pe).abs().max()))
    # eight Adam steps (value + predictor each) of two arithmetic paths -- capturable Adam + tile backward vs plain Adam + per-scene
    # backward: measured 1.6e-5; a replay that gathered the wrong rows would be off by the learning rate per step (1e-3) and more
    assert worst <= 1e-4, worst
    for (a, b), (c_, d) in zip(l_c, l_e):
        assert abs(a - c_) <= 1e-5 * max(1.0, abs(c_)) and abs(b - d) <= 1e-5 * max(1.0, abs(d))      # losses of parameters 1.6e-5 apart
    report("product MPRLTrainer: captured steps replayed across a growing (and wrapping) replay memory stay within %.1e of the eager "
           "trainer's parameters" % worst)


@pytest.mark.parametrize("case", ["shipped|2|0|1", "layerwise_noskip|2|1|0"])
def test_product_vnrl_trainer_reproduces_the_reference_fixture(case, dev):
    """relationalgraphlearning_amd.VNRLTrainer (path G's trainer, crowd_nav/utils/trainer.py:164-250) over the fixture of
    test_training_against_the_reference_vnrl_trainer_fixture, with upstream's pad_batch collate: same parameters, same loss."""
    fx = gio.load("vnrl_trainer")
    tag, L, lw, sk = case.split("|")
    pol = make_gcn_policy(int(L), bool(int(lw)), bool(int(sk)), device=dev)
    items = [(torch.tensor(fx["vn.states"][i]).to(dev), torch.zeros(1, device=dev), torch.tensor(fx["vn.rewards"][i]).reshape(1).to(dev),
              torch.tensor(fx["vn.next_states"][i]).to(dev)) for i in range(48)]
    from relationalgraphlearning_amd.trainer import pad_batch
    t = rga.VNRLTrainer(pol.model, _ListDataset(items), dev, pol, 16, "Adam", _Writer())
    t.set_learning_rate(1e-3)
    t.update_target_model(pol.model)
    t.data_loader = torch.utils.data.DataLoader(t.memory, 16, shuffle=False, collate_fn=pad_batch)
    loss = t.optimize_batch(2)
    want = float(fx["vn.%s.loss" % tag][0])
    assert abs(loss - want) <= 1e-5 * max(1.0, abs(want)), (loss, want)
    worst = 0.0
    for k, v in pol.model.state_dict().items():
        err = float(np.abs(v.cpu().numpy() - fx["vn.%s.model.%s" % (tag, k)]).max())
        worst = max(worst, err)
        assert err <= 2e-5, (k, err)
    report("product VNRLTrainer.optimize_batch (%s): final parameters within %.1e of the reference's, loss %.6f" % (tag, worst, loss))


def test_training_step_replayed_from_a_captured_graph(dev):
    """A whole MPRLTrainer-style optimisation step (value forward, frozen-target forward, backward, Adam; state-predictor forward,
    backward, Adam: crowd_nav/utils/trainer.py:110-161) captured once into a hipGraph and replayed: the library allocates nothing
    outside torch's allocator, never synchronises, and repacks its descriptors on the capture stream, so the capture is legal -- and
    four replays must leave the parameters where four eager steps leave them.  (Batch 100: 1.2-1.5 ms eager, 0.40-0.45 ms replayed,
    profiles/r03_final_train_step_graph.jsonl.)  Replays do not bump autograd's version counters: invalidate_packed_weights()
    afterwards, then an eager forward sees the trained weights."""
    import copy
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=True, flavour="trained")
    robot, humans = seeded_scenes(31, 100, 5)
    robot2, humans2 = seeded_scenes(32, 100, 5)
    r, h, r2, h2 = robot.unsqueeze(1).to(dev), humans.to(dev), robot2.unsqueeze(1).to(dev), humans2.to(dev)
    rew = torch.rand(100, 1, generator=torch.Generator().manual_seed(5)).to(dev)
    crit = torch.nn.MSELoss()

    def make():
        _, ve, sp = build_modules(c, dev)
        target = copy.deepcopy(ve)
        v_opt = torch.optim.Adam(ve.parameters(), lr=1e-3, capturable=True)
        s_opt = torch.optim.Adam(sp.human_motion_predictor.parameters(), lr=1e-3, capturable=True)

        def step():
            v_opt.zero_grad()
            out = ve((r, h))
            with torch.no_grad():
                tgt = rew + 0.9 * target((r2, h2))
            crit(out, tgt).backward()
            v_opt.step()
            s_opt.zero_grad()
            _, nh = sp((r, h), None, detach=True)
            crit(nh, h2).backward()
            s_opt.step()
        return ve, sp, step
    ve_a, sp_a, step_a = make()
    for _ in range(5):
        step_a()
    ve_b, sp_b, step_b = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step_b()                                              # one real step (warm-up of the capture recipe)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step_b()                                              # recorded, not executed
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    worst = 0.0
    for ma, mb in ((ve_a, ve_b), (sp_a, sp_b)):
        for (k, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            err = float((pa - pb).abs().max())
            worst = max(worst, err)
            assert err <= 1e-6, (k, err)
    rga.invalidate_packed_weights(ve_b, sp_b)
    with torch.no_grad():
        va, vb = ve_a((r, h)), ve_b((r, h))
    assert float((va - vb).abs().max()) <= 1e-6
    report("training step replayed from a captured hipGraph: parameters after 1 + 4 steps within %.1e of five eager steps" % worst)


def test_path_g_query_env_against_the_reference_fixture(dev):
    """Fixture training_queryenv.npz: the reference MultiHumanRL.predict with query_env=True on the reference CrowdSim (linear
    humans), four states a few steps into seeded test cases: 81 action values and the chosen action."""
    from relationalgraphlearning_amd.sim import BatchedCrowdSim
    fx = gio.load("training_queryenv")
    pol = make_gcn_policy(device=dev)
    pol.query_env = True
    sim = BatchedCrowdSim(dev)
    worst = 0.0
    for i in range(fx["qe.robot"].shape[0]):
        full = fx["qe.humans_full"][i]                                   # (H, 9) FullStates of the humans
        sim.load(fx["qe.robot"][i:i + 1], full[None, :, :5], fx["qe.human_goals"][i:i + 1], fx["qe.human_vpref"][i:i + 1])
        sim.time[:] = float(fx["qe.time"][i])
        pol.set_env(sim)
        a = pol.predict(JS(fx["qe.robot"][i], full[:, :5]))
        err = float(np.abs(np.array(pol.action_values) - fx["qe.action_values"][i]).max())
        worst = max(worst, err)
        assert err < 1e-5, (i, err)
        assert a == pol.action_space[int(fx["qe.action"][i])], i
    report("path G query_env=True vs the reference on its simulator: max |d action value| = %.2e" % worst)


def test_library_reports_target():
    assert nat.lib().rgl_build_target() == b"gfx950"


# ---------------------------------------------------------------------------------------------------------------------------------
# ABI 7: the element-wise ends of an optimisation step (rgl_train.hip)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_fields,n_index", [(5, 100), (1, 1), (11, 4096), (3, 0)])
def test_gather_rows_is_index_select(n_fields, n_index, dev):
    """rgl_gather_rows_f32 against torch's index_select, bit for bit: the replay memory's fields (rows of 9, H x 5, 1 ... floats), more
    fields than one launch carries, repeated and unsorted indices, an empty batch; an index outside a field leaves a row of NaN."""
    import ctypes
    from relationalgraphlearning_amd import trainer as tr
    g = torch.Generator().manual_seed(7 + n_fields)
    cap = 5000
    shapes = [(cap, 1, 9), (cap, 5, 5), (cap, 1), (cap, 1), (cap, 19, 5), (cap, 7), (cap, 1, 1), (cap, 64), (cap, 3), (cap, 2, 2), (cap, 33)]
    fields = [torch.randn(*s, generator=g).to(dev) for s in shapes[:n_fields]]
    idx = torch.randint(0, cap, (n_index,), generator=g).to(dev)
    got = tr._gather_fields(fields, idx)
    for f, o in zip(fields, got):
        assert o.shape == (n_index,) + tuple(f.shape[1:]) and torch.equal(o, f.index_select(0, idx))
    if n_index:
        jobs = (nat.RglGatherJob * 1)()
        out = torch.zeros(n_index, 9, device=dev)
        jobs[0].src, jobs[0].dst, jobs[0].row_floats, jobs[0].src_rows = fields[0].data_ptr(), out.data_ptr(), 9, 10      # a 10-row window
        bad = idx.clone()
        bad[0] = 3
        nat.check(nat.lib().rgl_gather_rows_f32(jobs, 1, bad.data_ptr(), n_index, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "rgl_gather_rows_f32")
        torch.cuda.synchronize()
        inside = (bad < 10)
        assert torch.equal(out[inside], fields[0].reshape(cap, 9)[bad[inside]]) and bool(torch.isnan(out[~inside]).all())
    assert nat.lib().rgl_gather_rows_f32(None, 1, idx.data_ptr(), 5, None) == -3           # RGL_ERR_NULL
    report("rgl_gather_rows_f32 == index_select over %d fields x %d rows" % (n_fields, n_index))


@pytest.mark.parametrize("shape", [(100, 1), (16, 5, 5), (4096, 19, 5), (1, 1), (1031, 3)])
def test_mse_step_is_torchs_loss_and_backward(shape, dev):
    """rgl_mse_step_f32 against nn.MSELoss + backward: the gradient bit for bit (same (float)(2 / n) * (out - target)), the reported
    loss within float32 summation noise, added to the running float64 sum; with the bootstrapped target r + gamma V' formed in the
    kernel the gradient equals torch's on `rewards + gamma_bar * values` bit for bit (two roundings, no fused multiply-add)."""
    import ctypes
    g = torch.Generator().manual_seed(11)
    out = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
    tgt = torch.randn(*shape, generator=g).to(dev)
    rew = torch.rand(*shape, generator=g).to(dev)
    nxt = torch.randn(*shape, generator=g).to(dev)
    gamma_bar = pow(0.9, 0.25)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    acc = torch.full((2,), 0.5, dtype=torch.float64, device=dev)
    ws = torch.zeros(nat.MSE_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)            # zeroed once; every launch leaves it ready
    for mode, target in (("target", tgt), ("bootstrap", rew + gamma_bar * nxt)):
        out.grad = None
        loss = torch.nn.MSELoss()(out, target)
        loss.backward()
        seen = []
        for rep in range(3):                                  # several workgroups above 8192 floats: same bits every time
            grad = torch.empty_like(out)
            before = float(acc[1])
            rc = nat.lib().rgl_mse_step_f32(out.data_ptr(), tgt.data_ptr() if mode == "target" else None,
                                            None if mode == "target" else rew.data_ptr(), None if mode == "target" else nxt.data_ptr(),
                                            0.0 if mode == "target" else gamma_bar, out.numel(), grad.data_ptr(), acc.data_ptr() + 8,
                                            ws.data_ptr(), st)
            nat.check(rc, "rgl_mse_step_f32")
            torch.cuda.synchronize()
            assert torch.equal(grad, out.grad), (mode, float((grad - out.grad).abs().max()))
            seen.append(np.float32(float(acc[1]) - before))
            acc[1] = 0.5
        assert abs(float(seen[0]) - float(loss.detach())) <= 2e-6 * max(1.0, abs(float(loss.detach()))), (mode, seen, float(loss.detach()))
        assert seen[0] == seen[1] == seen[2] and float(acc[0]) == 0.5            # repeatable; the neighbouring slot is not touched
    assert nat.lib().rgl_mse_step_f32(out.data_ptr(), None, None, None, 0.0, 4, out.data_ptr(), acc.data_ptr(), ws.data_ptr(), st) == -3
    assert nat.lib().rgl_mse_step_f32(out.data_ptr(), tgt.data_ptr(), None, None, 0.0, 0, out.data_ptr(), acc.data_ptr(), ws.data_ptr(), st) == -1
    if out.numel() > 8192:
        assert nat.lib().rgl_mse_step_f32(out.data_ptr(), tgt.data_ptr(), None, None, 0.0, out.numel(), grad.data_ptr(), acc.data_ptr(),
                                          None, st) == -4                           # RGL_ERR_WORKSPACE
    report("rgl_mse_step_f32 == MSELoss + backward on %s (gradient bit-exact, both target forms)" % (shape,))


def test_trainer_with_and_without_the_fused_loss_and_gather(dev, monkeypatch):
    """The public trainer with rgl_mse_step_f32 / rgl_gather_rows_f32 (default) and with torch's criterion / index_select
    (RGL_TRAINER_FUSED_LOSS=0 and fields the gather refuses): same parameters bit for bit -- the gradient arithmetic is torch's own --
    and reported losses within float32 summation noise.  A criterion other than nn.MSELoss() takes torch's path by itself."""
    from relationalgraphlearning_amd import trainer as tr
    H, n = 5, 300
    robot, humans = seeded_scenes(21, n, H)
    robot2, humans2 = seeded_scenes(22, n, H)
    rew = torch.rand(n, generator=torch.Generator().manual_seed(5))
    results = {}
    for tag in ("fused", "torch", "smooth_l1"):
        monkeypatch.setenv("RGL_TRAINER_FUSED_LOSS", "0" if tag == "torch" else "1")
        if tag == "torch":
            monkeypatch.setattr(tr, "_gather_fields", lambda fields, idx: [f.index_select(0, idx) for f in fields])
        pol = make_mprl_policy("trained", 1, device=dev)
        mem = rga.ReplayMemory(n)
        for i in range(n):
            mem.push((robot[i:i + 1].to(dev), humans[i].to(dev), rew[i:i + 1].to(dev), rew[i:i + 1].to(dev), robot2[i:i + 1].to(dev),
                      humans2[i].to(dev)))
        t = rga.MPRLTrainer(pol.value_estimator, pol.state_predictor, mem, dev, pol, _Writer(), 100, "Adam", H,
                            reduce_sp_update_frequency=False, freeze_state_predictor=False, detach_state_predictor=True, share_graph_model=False)
        if tag == "smooth_l1":
            t.criterion = torch.nn.SmoothL1Loss()
        t.set_learning_rate(1e-3)
        t.update_target_model(pol.value_estimator)
        torch.manual_seed(3)                                  # the batch order
        t.optimize_epoch(2)
        torch.manual_seed(4)
        losses = t.optimize_batch(2, 0)
        results[tag] = (torch.cat([p.detach().flatten() for p in list(pol.value_estimator.parameters()) + list(pol.state_predictor.parameters())]).cpu(),
                        losses)
    assert torch.equal(results["fused"][0], results["torch"][0]), float((results["fused"][0] - results["torch"][0]).abs().max())
    for a, b in zip(results["fused"][1], results["torch"][1]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (a, b)
    assert not torch.equal(results["fused"][0], results["smooth_l1"][0]) and bool(torch.isfinite(results["smooth_l1"][0]).all())
    report("MPRLTrainer with the fused loss / gather == with torch's criterion / index_select (parameters bit-identical)")


def test_captured_trainer_steps_are_repeatable(dev):
    """Imitation epochs followed by RL batches through the public trainer's captured steps, six times over in one process: bit-identical
    parameters every time and float32 noise away from the eager trainer.  (Round 5: the tile backward zeroed dH_L with hipMemsetAsync;
    replayed inside a captured step the memset NODE was not reliably ordered against the kernels around it -- NaN parameters on one
    box, a step without the heads' gradient now and then on another.  It is a kernel now; tools/micro/captured_step_repeatability.py
    is the longer form of this test.)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("captured_step_repeatability",
                                                  os.path.join(root, "tools", "micro", "captured_step_repeatability.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, H = 300, 5
    data = seeded_scenes(21, n, H) + seeded_scenes(22, n, H) + (torch.rand(n, generator=torch.Generator().manual_seed(5)),)
    eager = mod.run(dev, data, 2, 2, capture=False)
    outs = [mod.run(dev, data, 2, 2) for _ in range(6)]
    assert bool(torch.isfinite(outs[0]).all())
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), float((o - outs[0]).abs().max())
    worst = float((outs[0] - eager).abs().max())
    assert worst <= 1e-5, worst
    report("captured trainer steps (2 epochs + 2 batch calls) x 6: bit-identical parameters, %.1e from the eager trainer's" % worst)


def test_product_trainer_imitation_then_detached_rl_against_the_reference_fixture(dev):
    """Fixture training_queryenv.npz, case il_then_detach: the REFERENCE MPRLTrainer ran optimize_epoch(2) -- which trains the state
    predictor's graph model -- and then optimize_batch with detach_state_predictor (configs/icra_benchmark/mp_detach.py).  Upstream the
    detached graph model gets NO gradient in the RL phase and Adam skips it although its momentum is non-zero: its parameters stay
    bit for bit where imitation learning left them.  (A zero gradient is not the same: Adam would keep moving them -- what this
    package did until round 5.)  Same final parameters and losses through the public trainer's captured steps."""
    fx = gio.load("training_queryenv")
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=False, flavour="trained")
    _, ve, sp = build_modules(c, dev)
    items = [(torch.tensor(fx["tr.robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.humans"][i]).to(dev),
              torch.tensor(fx["tr.values"][i]).reshape(1).to(dev), torch.tensor(fx["tr.rewards"][i]).reshape(1).to(dev),
              torch.tensor(fx["tr.next_robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.next_humans"][i]).to(dev)) for i in range(48)]
    t = rga.MPRLTrainer(ve, sp, _ListDataset(items), dev, None, _Writer(), 16, "Adam", 5, reduce_sp_update_frequency=False,
                        freeze_state_predictor=False, detach_state_predictor=True, share_graph_model=False)
    t.set_learning_rate(1e-3)
    t.update_target_model(ve)
    t.data_loader = torch.utils.data.DataLoader(t.memory, 16, shuffle=False)          # the fixture's order
    t.optimize_epoch(2)
    after_il = {k: v.detach().clone() for k, v in sp.graph_model.state_dict().items()}
    for name, mod in (("graph_model2", sp.graph_model), ("motion_predictor", sp.human_motion_predictor)):
        for k, v in mod.state_dict().items():
            assert float(np.abs(v.cpu().numpy() - fx["tr.il_then_detach.after_il.%s.%s" % (name, k)]).max()) <= 2e-5, (name, k)
    av, as_ = t.optimize_batch(2, 0)
    assert t._capturable and len(t._steps) == 3                                    # il with / without the predictor update, rl
    for k, v in sp.graph_model.state_dict().items():
        assert torch.equal(v, after_il[k]), (k, float((v - after_il[k]).abs().max()))          # frozen, as upstream
    want_v, want_s = fx["tr.il_then_detach.losses"]
    assert abs(av - want_v) <= 1e-5 * max(1.0, abs(want_v)) and abs(as_ - want_s) <= 1e-5 * max(1.0, abs(want_s)), (av, as_)
    worst = 0.0
    for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                      ("motion_predictor", sp.human_motion_predictor)):
        for k, v in mod.state_dict().items():
            err = float(np.abs(v.cpu().numpy() - fx["tr.il_then_detach.%s.%s" % (name, k)]).max())
            worst = max(worst, err)
            assert err <= 2e-5, (name, k, err)
    report("product MPRLTrainer, optimize_epoch(2) then detached optimize_batch: the predictor's graph model stays where imitation "
           "learning left it (bit for bit); final parameters within %.1e of the reference trainer's, losses %.6f / %.6f" % (worst, av, as_))


@pytest.mark.parametrize("variant,detach", [("shared", False), ("shared", True), ("linear", False)])
def test_trainer_on_the_other_wirings_captured_vs_eager(variant, detach, dev):
    """model_predictive_rl.share_graph_model (one RGL under the value estimator AND the state predictor: both optimizers step its
    parameters, model_predictive_rl.py:84-90) and linear_state_predictor (nothing to train on the predictor side: trainer.py:47,
    `state_predictor.trainable`) through the public trainer: imitation epochs + RL batches from captured steps, three times over --
    bit-identical repetitions, float32 noise away from the eager trainer."""
    n, H = 300, 5
    robot, humans = seeded_scenes(31, n, H)
    robot2, humans2 = seeded_scenes(32, n, H)
    rew = torch.rand(n, generator=torch.Generator().manual_seed(6))

    def run(capture):
        pol = make_mprl_policy("trained", 1, variant=variant, device=dev, skip=False)
        if variant == "shared":
            assert pol.value_estimator.graph_model is pol.state_predictor.graph_model
        mem = rga.ReplayMemory(n)
        for i in range(n):
            mem.push((robot[i:i + 1].to(dev), humans[i].to(dev), rew[i:i + 1].to(dev) * 0.5, rew[i:i + 1].to(dev), robot2[i:i + 1].to(dev),
                      humans2[i].to(dev)))
        cls = rga.MPRLTrainer if capture else type("EagerTrainer", (rga.MPRLTrainer,), {"capture": False})
        t = cls(pol.value_estimator, pol.state_predictor, mem, dev, pol, _Writer(), 100, "Adam", H, reduce_sp_update_frequency=False,
                freeze_state_predictor=False, detach_state_predictor=detach, share_graph_model=(variant == "shared"))
        t.set_learning_rate(1e-3)
        assert (t.s_optimizer is None) == (variant == "linear")
        t.update_target_model(pol.value_estimator)
        torch.manual_seed(3)
        t.optimize_epoch(2)
        torch.manual_seed(4)
        losses = t.optimize_batch(2, 0)
        mods = [pol.value_estimator] + ([pol.state_predictor] if variant != "linear" else [])
        seen, flat = set(), []
        for m in mods:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    flat.append(p.detach().flatten())
        return torch.cat(flat).cpu(), losses
    eager, eager_losses = run(False)
    outs = [run(True) for _ in range(3)]
    assert bool(torch.isfinite(outs[0][0]).all())
    for o, l in outs[1:]:
        assert torch.equal(o, outs[0][0]) and l == outs[0][1]
    worst = float((outs[0][0] - eager).abs().max())
    assert worst <= 1e-5, worst
    for a, b in zip(outs[0][1], eager_losses):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (a, b)
    report("MPRLTrainer on the %s wiring (detach %s): captured steps x 3 bit-identical, %.1e from the eager trainer's parameters"
           % (variant, detach, worst))


@pytest.mark.parametrize("tag", ["sgd", "reduce", "freeze"])
def test_product_trainer_switches_against_the_reference_fixture(tag, dev):
    """Fixture training_queryenv.npz, cases sgd / reduce / freeze: the REFERENCE MPRLTrainer.optimize_batch with optimizer 'SGD'
    (momentum 0.9 on the value side, none on the predictor's: trainer.py:49-52; lr 1e-2), with reduce_sp_update_frequency (no
    predictor update on batch 0 of every five, :138-139) and with freeze_state_predictor (:136-137), three un-shuffled batches of 16.
    The product trainer -- eager steps for SGD, captured ones (two kinds under `reduce`) for Adam -- ends at the same parameters and
    reports the same losses."""
    fx = gio.load("training_queryenv")
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=False, flavour="trained")
    _, ve, sp = build_modules(c, dev)
    items = [(torch.tensor(fx["tr.robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.humans"][i]).to(dev),
              torch.zeros(1, device=dev), torch.tensor(fx["tr.rewards"][i]).reshape(1).to(dev),
              torch.tensor(fx["tr.next_robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.next_humans"][i]).to(dev)) for i in range(48)]
    t = rga.MPRLTrainer(ve, sp, _ListDataset(items), dev, None, _Writer(), 16, "SGD" if tag == "sgd" else "Adam", 5,
                        reduce_sp_update_frequency=(tag == "reduce"), freeze_state_predictor=(tag == "freeze"),
                        detach_state_predictor=False, share_graph_model=False)
    t.set_learning_rate(1e-2 if tag == "sgd" else 1e-3)
    t.update_target_model(ve)
    t.data_loader = torch.utils.data.DataLoader(t.memory, 16, shuffle=False)          # the fixture's order
    before = {k: v.detach().clone() for k, v in sp.state_dict().items()}
    av, as_ = t.optimize_batch(2, 0)
    assert len(t._steps) == {"sgd": 0, "reduce": 2, "freeze": 1}[tag]
    if tag == "freeze":
        assert as_ == 0.0 and all(torch.equal(v, before[k]) for k, v in sp.state_dict().items())
    want_v, want_s = fx["tr.%s.losses" % tag]
    assert abs(av - want_v) <= 1e-5 * max(1.0, abs(want_v)) and abs(as_ - want_s) <= 1e-5 * max(1.0, abs(want_s)), (av, as_)
    worst = 0.0
    for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                      ("motion_predictor", sp.human_motion_predictor)):
        for k, v in mod.state_dict().items():
            err = float(np.abs(v.cpu().numpy() - fx["tr.%s.%s.%s" % (tag, name, k)]).max())
            worst = max(worst, err)
            assert err <= 2e-5, (name, k, err)
    report("product MPRLTrainer.optimize_batch (%s): final parameters within %.1e of the reference trainer's, losses %.6f / %.6f"
           % (tag, worst, av, as_))


def test_product_vnrl_trainer_imitation_then_rl_against_the_reference_fixture(dev):
    """Fixture vnrl_trainer.npz, case il_then_rl: the REFERENCE VNRLTrainer ran optimize_epoch(2) and then optimize_batch(2)
    (trainer.py:199-250, train.py's order) on path G's shipped network.  The product trainer's captured steps (one for the imitation
    batches, one for the RL batches) end at the same parameters after each phase and report the same two losses; twice over, bit for
    bit."""
    fx = gio.load("vnrl_trainer")
    from relationalgraphlearning_amd.trainer import pad_batch

    def run():
        pol = make_gcn_policy(2, False, True, device=dev)
        items = [(torch.tensor(fx["vn.states"][i]).to(dev), torch.tensor(fx["vn.values"][i]).reshape(1).to(dev),
                  torch.tensor(fx["vn.rewards"][i]).reshape(1).to(dev), torch.tensor(fx["vn.next_states"][i]).to(dev)) for i in range(48)]
        t = rga.VNRLTrainer(pol.model, _ListDataset(items), dev, pol, 16, "Adam", _Writer())
        t.set_learning_rate(1e-3)
        t.update_target_model(pol.model)
        t.data_loader = torch.utils.data.DataLoader(t.memory, 16, shuffle=False, collate_fn=pad_batch)
        il = t.optimize_epoch(2)
        after_il = {k: v.detach().cpu().numpy().copy() for k, v in pol.model.state_dict().items()}
        rl = t.optimize_batch(2, 0)
        assert t._capturable and len(t._steps) == 2
        return il, rl, after_il, {k: v.detach().cpu().numpy().copy() for k, v in pol.model.state_dict().items()}
    il, rl, after_il, final = run()
    want_il, want_rl = fx["vn.il_then_rl.losses"]
    assert abs(il - want_il) <= 1e-5 * max(1.0, abs(want_il)) and abs(rl - want_rl) <= 1e-5 * max(1.0, abs(want_rl)), (il, rl)
    worst = 0.0
    for k in final:
        worst = max(worst, float(np.abs(after_il[k] - fx["vn.il_then_rl.after_il.model.%s" % k]).max()),
                    float(np.abs(final[k] - fx["vn.il_then_rl.model.%s" % k]).max()))
    assert worst <= 2e-5, worst
    il2, rl2, _, final2 = run()
    assert (il2, rl2) == (il, rl) and all(np.array_equal(final[k], final2[k]) for k in final)
    report("product VNRLTrainer, optimize_epoch(2) then optimize_batch(2): parameters within %.1e of the reference trainer's after "
           "each phase, losses %.6f / %.6f; a second run bit-identical" % (worst, il, rl))


def test_product_trainer_in_train_py_order_against_the_reference_fixture(dev):
    """Fixture training_queryenv.npz, case train_py_order -- what crowd_nav/train.py:134-176 does to a trainer: imitation learning at
    the IL rate, NEW optimizers at the RL rate (the captured steps of the first phase are dropped with the old ones), the target model
    refreshed, RL batches, the target refreshed IN PLACE under the captured RL step, more RL batches.  Same four losses and the same
    final parameters as the reference trainer."""
    fx = gio.load("training_queryenv")
    c = dict(L=2, sim="embedded_gaussian", layerwise=False, skip=False, flavour="trained")
    _, ve, sp = build_modules(c, dev)
    items = [(torch.tensor(fx["tr.robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.humans"][i]).to(dev),
              torch.tensor(fx["tr.values"][i]).reshape(1).to(dev), torch.tensor(fx["tr.rewards"][i]).reshape(1).to(dev),
              torch.tensor(fx["tr.next_robot"][i]).unsqueeze(0).to(dev), torch.tensor(fx["tr.next_humans"][i]).to(dev)) for i in range(48)]
    t = rga.MPRLTrainer(ve, sp, _ListDataset(items), dev, None, _Writer(), 16, "Adam", 5, reduce_sp_update_frequency=False,
                        freeze_state_predictor=False, detach_state_predictor=False, share_graph_model=False)
    t.data_loader = torch.utils.data.DataLoader(t.memory, 16, shuffle=False)
    t.set_learning_rate(1e-2)
    t.optimize_epoch(1)
    assert len(t._steps) == 2
    t.set_learning_rate(1e-3)
    assert len(t._steps) == 0                                                        # the optimizers are new
    t.update_target_model(ve)
    first = t.optimize_batch(2, 0)
    t.update_target_model(ve)
    assert len(t._steps) == 1                                                        # ... and stays: the target is refreshed in place
    second = t.optimize_batch(2, 1)
    got = np.array(list(first) + list(second))
    want = fx["tr.train_py_order.losses"]
    assert np.all(np.abs(got - want) <= 1e-5 * np.maximum(1.0, np.abs(want))), (got, want)
    worst = 0.0
    for name, mod in (("graph_model1", ve.graph_model), ("value_network", ve.value_network), ("graph_model2", sp.graph_model),
                      ("motion_predictor", sp.human_motion_predictor)):
        for k, v in mod.state_dict().items():
            err = float(np.abs(v.cpu().numpy() - fx["tr.train_py_order.%s.%s" % (name, k)]).max())
            worst = max(worst, err)
            assert err <= 5e-5, (name, k, err)
    report("product MPRLTrainer in train.py's order (IL at 1e-2, new optimizers at 1e-3, target refreshed twice): final parameters "
           "within %.1e of the reference trainer's, losses %s" % (worst, np.round(got, 6).tolist()))
