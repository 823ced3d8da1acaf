"""Batched simulator: host-side scene generation and the CPU oracle against the reference's fixtures (CPU tests), the
device step against the reference's recorded trajectories and the vectorised episode loop (GPU tests)."""
import numpy as np
import pytest
import torch

from oracle import sim_oracle as so
from relationalgraphlearning_amd.sim import BatchedCrowdSim, SimConfig, generate_scene, run_episodes
from tests import golden_io as gio


def sim_cases():
    return [str(c).split("|")[0] for c in gio.load("sim")["sim_cases"]]


def test_scene_generation_reproduces_reference_cases():
    sc = gio.load("scenes")
    cfg = SimConfig()
    for phase in ("test", "val"):
        for k in range(sc[phase + "_robot"].shape[0]):
            robot, humans, goals, vpref = generate_scene(cfg, phase, k)
            assert np.array_equal(robot, sc[phase + "_robot"][k])
            assert np.array_equal(humans, sc[phase + "_humans"][k])
            assert np.array_equal(goals, -humans[:, :2])                 # circle crossing: goal is the antipode
    # the first state of every recorded simulator trajectory is that case's scene
    sim = gio.load("sim")
    for line in sim["sim_cases"]:
        tag, case = str(line).split("|")
        robot, humans, goals, vpref = generate_scene(cfg, "test", int(case))
        assert np.array_equal(robot, sim["sim.%s.robot" % tag][0])
        assert np.array_equal(humans, sim["sim.%s.humans" % tag][0][:, :5])
        assert np.array_equal(goals, sim["sim.%s.humans" % tag][0][:, 5:7])
    sq = generate_scene(SimConfig(scenario="square_crossing", human_num=4), "test", 3)
    assert sq[1].shape == (4, 5) and np.all(np.abs(sq[1][:, :2]) <= 10.0) and np.all(np.abs(sq[2]) <= 10.0)


@pytest.mark.parametrize("tag", sim_cases())
def test_oracle_step_against_reference_trajectories(tag):
    sim = gio.load("sim")
    k = "sim.%s." % tag
    table = sim["sim.action_table"]
    robot = list(sim[k + "robot"][0])
    humans = [list(h) for h in sim[k + "humans"][0]]
    t = 0.0
    for i, ai in enumerate(sim[k + "actions"]):
        robot, humans, reward, done, info, dmin = so.step(robot, humans, table[ai], t)
        t += 0.25
        assert np.allclose(robot, sim[k + "robot"][i + 1], rtol=0, atol=1e-12)
        assert np.allclose(np.array(humans)[:, :4], sim[k + "humans"][i + 1][:, :4], rtol=0, atol=1e-9)
        assert abs(reward - sim[k + "reward"][i]) < 1e-12 and int(done) == sim[k + "done"][i] and info == sim[k + "info"][i]
        if info == so.INFO_DISCOMFORT:
            assert abs(dmin - sim[k + "dmin"][i]) < 1e-12
    assert done


@pytest.mark.gpu
def test_device_step_against_reference_trajectories():
    """All recorded trajectories at once, one environment each, every step compared with what the reference produced."""
    dev = torch.device("cuda:0")
    sim = gio.load("sim")
    tags = sim_cases()
    table = sim["sim.action_table"]
    n_steps = max(len(sim["sim.%s.actions" % t]) for t in tags)
    env = BatchedCrowdSim(dev)
    first = [sim["sim.%s.humans" % t][0] for t in tags]
    env.load(np.stack([sim["sim.%s.robot" % t][0] for t in tags]), np.stack([f[:, :5] for f in first]),
             np.stack([f[:, 5:7] for f in first]), np.stack([f[:, 7] for f in first]))
    finished = [False] * len(tags)
    for i in range(n_steps):
        acts = np.stack([table[sim["sim.%s.actions" % t][i]] if i < len(sim["sim.%s.actions" % t]) else table[0] for t in tags])
        obs, reward, done, info = env.step(acts)
        r, h, tm = env.robot.cpu().numpy(), env.humans.cpu().numpy(), env.time.cpu().numpy()
        reward, done, info, dmin = reward.cpu().numpy(), done.cpu().numpy(), info.cpu().numpy(), env.last_dmin.cpu().numpy()
        for e, t in enumerate(tags):
            k = "sim.%s." % t
            if i >= len(sim[k + "actions"]):
                assert finished[e] and info[e] == 5 and reward[e] == 0          # frozen after its episode ended
                continue
            assert np.allclose(r[e], sim[k + "robot"][i + 1], rtol=0, atol=1e-12), (t, i)
            assert np.allclose(h[e][:, :4], sim[k + "humans"][i + 1][:, :4], rtol=0, atol=1e-9), (t, i)
            assert abs(reward[e] - sim[k + "reward"][i]) < 1e-7 and int(done[e]) == sim[k + "done"][i], (t, i)
            assert info[e] == sim[k + "info"][i] and abs(tm[e] - sim[k + "time"][i + 1]) < 1e-12, (t, i)
            if info[e] == 1:
                assert abs(dmin[e] - sim[k + "dmin"][i]) < 1e-12
            finished[e] = bool(done[e])
        assert obs[0].dtype == torch.float32 and np.allclose(obs[0].cpu().numpy(), r.astype(np.float32))
    assert all(finished)
    # onestep_lookahead leaves the state untouched
    env.reset("test", [0, 1])
    before = env.robot.clone()
    _, rew, done, info = env.onestep_lookahead(np.tile(table[5], (2, 1)))
    assert torch.equal(env.robot, before) and float(env.time.sum()) == 0.0


@pytest.mark.gpu
def test_vectorised_episodes_with_the_hip_policy():
    """64 seeded test cases run in lock-step with the model-predictive policy deciding for all of them at once."""
    from tests.helpers import make_mprl_policy
    dev = torch.device("cuda:0")
    pol = make_mprl_policy("trained", 1, device=dev)
    env = BatchedCrowdSim(dev)
    stats = run_episodes(env, pol, "test", list(range(64)))
    assert stats["unfinished"] == 0
    assert abs(stats["success_rate"] + stats["collision_rate"] + stats["timeout_rate"] - 1.0) < 1e-9
    assert set(np.unique(stats["outcome"])) <= {2, 3, 4}
    again = run_episodes(env, pol, "test", list(range(64)))
    assert np.array_equal(stats["outcome"], again["outcome"]) and np.array_equal(stats["cumulative_reward"], again["cumulative_reward"])
    # batch composition does not matter: case 7 alone behaves as inside the batch
    solo = run_episodes(env, pol, "test", [7])
    assert solo["outcome"][0] == stats["outcome"][7] and solo["time"][0] == stats["time"][7]
