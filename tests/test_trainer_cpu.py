"""Host-side logic of the graph-replaying trainers (relationalgraphlearning_amd/trainer.py) that needs no GPU: the batch order
(the DataLoader's own, index for index), the stacked view of the replay memory, the reference's contract."""
import types

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

import relationalgraphlearning_amd as rga
from relationalgraphlearning_amd import trainer as tr
from tests.helpers import make_mprl_policy, make_gcn_policy


def _fill(memory, n, H=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    for i in range(n):
        memory.push((torch.randn(1, 9, generator=g), torch.randn(H, 5, generator=g), torch.tensor([float(i)]),
                     torch.randn(1, generator=g), torch.randn(1, 9, generator=g), torch.randn(H, 5, generator=g)))


@pytest.mark.parametrize("n,bs", [(23, 5), (10, 10), (7, 100), (1, 1)])
def test_index_batches_are_the_dataloaders_own(n, bs):
    """_ShuffledIndexBatches must yield the batches DataLoader(memory, bs, shuffle=True) yields -- same items, same order, same
    consumption of torch's global generator (two epochs in a row, then the generator states must agree)."""
    mem = rga.ReplayMemory(1000)
    _fill(mem, n)
    torch.manual_seed(1234)
    want = [[b[2].reshape(-1).tolist() for b in DataLoader(mem, bs, shuffle=True)] for _ in range(2)]
    state_a = torch.get_rng_state()
    torch.manual_seed(1234)
    got = [[[float(i) for i in idx.tolist()] for idx in tr._ShuffledIndexBatches(n, bs)] for _ in range(2)]
    assert got == want
    assert torch.equal(torch.get_rng_state(), state_a)


def test_replay_memory_mirror_follows_the_ring():
    """as_tensors(): what default_collate makes of the whole memory, kept current across pushes, wrap-around and clear()."""
    mem = rga.ReplayMemory(8)
    assert mem.as_tensors() is None
    _fill(mem, 5)
    for round_ in range(4):
        fields = mem.as_tensors()
        want = next(iter(DataLoader(mem, len(mem), shuffle=False)))
        assert len(fields) == 6 and all(torch.equal(f, w) for f, w in zip(fields, want))
        assert fields[0].shape == (len(mem), 1, 9) and fields[1].shape == (len(mem), 3, 5) and fields[2].shape == (len(mem), 1)
        _fill(mem, 3, seed=10 + round_)                      # 8, 11 (wraps), 14, 17 pushes in total
    mem.clear()
    assert mem.as_tensors() is None and len(mem) == 0
    _fill(mem, 2, H=4)                                       # another crowd size after clear(): a new mirror
    assert mem.as_tensors()[1].shape == (2, 4, 5)
    mem.push((torch.zeros(1, 9), torch.zeros(2, 5), torch.zeros(1), torch.zeros(1), torch.zeros(1, 9), torch.zeros(2, 5)))
    assert mem.as_tensors() is None                          # mixed crowd sizes: no stacked view (the DataLoader path serves them)


def test_pad_batch_orders_each_state_list_longest_first():
    """The path-G collate function (the role of crowd_nav/utils/trainer.py:253-272): both state lists longest first (equally long
    ones in batch order), zero padded, with their lengths; values / rewards in BATCH order, as upstream leaves them."""
    g = torch.Generator().manual_seed(3)
    lens, next_lens = [2, 4, 3, 4, 1], [3, 3, 5, 1, 2]
    batch = [(torch.randn(a, 13, generator=g), torch.tensor([float(i)]), torch.tensor([10.0 + i]), torch.randn(b, 13, generator=g))
             for i, (a, b) in enumerate(zip(lens, next_lens))]
    (states, ls), values, rewards, (nxt, nls) = tr.pad_batch(batch)
    assert ls.tolist() == [4, 4, 3, 2, 1] and nls.tolist() == [5, 3, 3, 2, 1] and ls.dtype == torch.int64
    assert states.shape == (5, 4, 13) and nxt.shape == (5, 5, 13)
    for row, src in zip(range(5), [1, 3, 2, 0, 4]):                       # stable: item 1 before item 3
        L = lens[src]
        assert torch.equal(states[row, :L], batch[src][0]) and not states[row, L:].any()
    for row, src in zip(range(5), [2, 0, 1, 4, 3]):
        L = next_lens[src]
        assert torch.equal(nxt[row, :L], batch[src][3]) and not nxt[row, L:].any()
    assert values.reshape(-1).tolist() == [0.0, 1.0, 2.0, 3.0, 4.0] and values.shape == (5, 1)
    assert rewards.reshape(-1).tolist() == [10.0, 11.0, 12.0, 13.0, 14.0]
    # against torch's own pack / unpack of the same longest-first lists (what upstream's function goes through)
    seqs = sorted([b[0] for b in batch], key=lambda t: -t.shape[0])
    ref, ref_l = torch.nn.utils.rnn.pad_packed_sequence(torch.nn.utils.rnn.pack_sequence(seqs), batch_first=True)
    assert torch.equal(ref, states) and torch.equal(ref_l, ls)


def test_replay_memory_capacity_fields_and_unstackable_memo():
    """stacked_capacity_fields(): the full-extent tensors as_tensors() views (what a captured step gathers from -- ADVICE r4);
    a memory of mixed crowd sizes remembers that it has no stacked view instead of re-walking itself on every call."""
    mem = rga.ReplayMemory(6)
    assert mem.stacked_capacity_fields() is None
    _fill(mem, 4)
    whole = mem.stacked_capacity_fields()
    assert [w.shape[0] for w in whole] == [6] * 6
    assert all(torch.equal(w[:4], f) for w, f in zip(whole, mem.as_tensors()))
    _fill(mem, 1, seed=4)
    again = mem.stacked_capacity_fields()
    assert all(a.data_ptr() == w.data_ptr() for a, w in zip(again, whole))       # same buffers while the memory grows
    assert torch.equal(again[2][:5].reshape(-1), torch.tensor([0.0, 1.0, 2.0, 3.0, 0.0]))
    mem.push((torch.zeros(1, 9), torch.zeros(2, 5), torch.zeros(1), torch.zeros(1), torch.zeros(1, 9), torch.zeros(2, 5)))
    assert mem.as_tensors() is None and mem._unstackable_at == 6 and mem._mirror is None
    assert mem.as_tensors() is None and mem._mirror is None                       # memoised: nothing re-allocated
    _fill(mem, 6, seed=9)                                                         # the ring wraps: the odd item is overwritten
    assert mem.as_tensors() is not None and mem.stacked_capacity_fields()[1].shape == (6, 3, 5)
    mem.clear()
    assert mem._unstackable_at == -1


def test_trainer_contract_on_cpu():
    """Constructor arguments, attributes and errors of crowd_nav/utils/trainer.py; on a CPU device nothing is capturable and the
    trainers refuse nothing they accept upstream (the forward itself needs the GPU: not run here)."""
    pol = make_mprl_policy("trained", 1)
    mem = rga.ReplayMemory(100)
    t = rga.MPRLTrainer(pol.value_estimator, pol.state_predictor, mem, torch.device("cpu"), pol, None, 100, "Adam", 5,
                        reduce_sp_update_frequency=False, freeze_state_predictor=False, detach_state_predictor=True,
                        share_graph_model=False)
    for name in ("value_estimator", "state_predictor", "device", "writer", "target_policy", "target_model", "criterion", "memory",
                 "data_loader", "batch_size", "optimizer_str", "reduce_sp_update_frequency", "state_predictor_update_interval",
                 "freeze_state_predictor", "detach_state_predictor", "share_graph_model", "v_optimizer", "s_optimizer", "gamma",
                 "time_step", "v_pref"):
        assert hasattr(t, name), name
    assert t.state_predictor_update_interval == 5 and (t.gamma, t.time_step, t.v_pref) == (0.9, 0.25, 1)
    with pytest.raises(ValueError, match="Learning rate is not set"):
        t.optimize_batch(1, 0)
    with pytest.raises(ValueError, match="Learning rate is not set"):
        t.optimize_epoch(1)
    t.set_learning_rate(1e-3)
    assert isinstance(t.v_optimizer, torch.optim.Adam) and isinstance(t.s_optimizer, torch.optim.Adam) and not t._capturable
    t.optimizer_str = "SGD"
    t.set_learning_rate(1e-2)
    assert isinstance(t.v_optimizer, torch.optim.SGD) and t.v_optimizer.defaults["momentum"] == 0.9
    t.optimizer_str = "RMSprop"
    with pytest.raises(NotImplementedError):
        t.set_learning_rate(1e-2)
    # the frozen copy: a deep copy first, then refreshed IN PLACE (captured steps keep reading the same storages)
    t.update_target_model(pol.value_estimator)
    first = t.target_model
    assert first is not pol.value_estimator
    ptrs = [p.data_ptr() for p in first.parameters()]
    with torch.no_grad():
        next(pol.value_estimator.parameters()).add_(1.0)
    t.update_target_model(pol.value_estimator)
    assert t.target_model is first and [p.data_ptr() for p in first.parameters()] == ptrs
    assert all(torch.equal(a, b) for a, b in zip(first.parameters(), pol.value_estimator.parameters()))
    g = make_gcn_policy()
    v = rga.VNRLTrainer(g.model, mem, torch.device("cpu"), g, 100, "Adam", None)
    with pytest.raises(ValueError, match="Learning rate is not set"):
        v.optimize_batch(1)
    v.set_learning_rate(1e-3)
    v.update_target_model(g.model)
    assert v.target_model is not g.model
    # registration: the names crowd_nav/train.py imports
    ns = types.ModuleType("crowd_nav_utils_trainer_stand_in")
    rga.register_trainers(ns)
    assert ns.MPRLTrainer is rga.MPRLTrainer and ns.VNRLTrainer is rga.VNRLTrainer and callable(ns.pad_batch)


def test_fused_adam_steps_are_announced_to_the_descriptor_caches():
    """torch's fused Adam kernel leaves autograd's version counters alone; the descriptor caches key on them.  The optimizer the
    trainers build for a captured step carries a post-hook that bumps the counter of every parameter the step updated (and only
    those), and an uncaptured trainer keeps upstream's plain Adam."""
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), torch.nn.Linear(4, 1))
    plain = tr._new_optimizer("Adam", net, 1e-3, capturable=False)
    assert isinstance(plain, torch.optim.Adam) and not plain.defaults.get("fused") and not plain.defaults["capturable"]
    assert not plain._optimizer_step_post_hooks
    frozen = net[2].bias
    net(torch.randn(5, 3)).sum().backward()
    frozen.grad = None                                            # "not part of this step"
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    before = [p._version for p in net.parameters()]
    tr._mark_parameters_changed(opt, (), {})
    after = [p._version for p in net.parameters()]
    assert [a - b for a, b in zip(after, before)] == [1, 1, 1, 0]
    with pytest.raises(NotImplementedError):
        tr._new_optimizer("RMSprop", net, 1e-3, capturable=False)


def test_loss_backward_torch_path_is_upstreams_three_lines():
    """_loss_backward away from the device (or with another criterion) is upstream's `loss = criterion(outputs, target);
    loss.backward(); losses += loss.item()` (crowd_nav/utils/trainer.py:130-137), with the value update's target formed as
    `rewards + gamma_bar * next values` (:128-129) when it is handed over in pieces."""
    class Host(tr._TrainerBase):
        def __init__(self, criterion):
            self.criterion = criterion
            self.device = torch.device("cpu")
            self._init_graphs()
    g = torch.Generator().manual_seed(2)
    net = torch.nn.Linear(4, 1)
    x, tgt = torch.randn(16, 4, generator=g), torch.randn(16, 1, generator=g)
    rew, nxt, gamma_bar = torch.rand(16, 1, generator=g), torch.randn(16, 1, generator=g), pow(0.9, 0.25)
    for crit in (torch.nn.MSELoss(), torch.nn.SmoothL1Loss()):
        for pieces in (False, True):
            want_t = rew + gamma_bar * nxt if pieces else tgt
            net.zero_grad()
            loss = crit(net(x), want_t)
            loss.backward()
            want_g = [p.grad.clone() for p in net.parameters()]
            h = Host(crit)
            h._loss_begin()
            net.zero_grad()
            h._loss_backward(1, net(x), None if pieces else tgt, (rew, nxt, gamma_bar) if pieces else None)
            h._loss_backward(1, net(x), None if pieces else tgt, (rew, nxt, gamma_bar) if pieces else None)     # accumulates
            for p, w in zip(net.parameters(), want_g):
                assert torch.equal(p.grad, 2 * w)
            v, s = h._loss_read()
            assert v == 0.0 and s == 2 * float(loss.detach())


def test_vnrl_trainer_contract_on_cpu():
    """Constructor arguments, attributes and errors of path G's trainer (crowd_nav/utils/trainer.py:164-197)."""
    pol = make_gcn_policy(2)
    mem = rga.ReplayMemory(100)
    t = rga.VNRLTrainer(pol.model, mem, torch.device("cpu"), pol, 100, "Adam", None)
    for name in ("model", "device", "policy", "target_model", "criterion", "memory", "data_loader", "batch_size", "optimizer_str",
                 "optimizer", "writer", "gamma", "time_step", "v_pref"):
        assert hasattr(t, name), name
    assert t.model is pol.model and t.policy is pol and t.memory is mem and t.batch_size == 100 and t.optimizer is None
    assert isinstance(t.criterion, torch.nn.MSELoss) and t.target_model is None and t.data_loader is None
    assert (t.gamma, t.time_step, t.v_pref) == (0.9, 0.25, 1)
    for call in (lambda: t.optimize_batch(1), lambda: t.optimize_epoch(1)):
        with pytest.raises(ValueError, match="Learning rate is not set"):
            call()
    t.set_learning_rate(1e-3)
    assert isinstance(t.optimizer, torch.optim.Adam) and not t._capturable and t._steps == {}
    t.optimizer_str = "SGD"
    t.set_learning_rate(1e-2)
    assert isinstance(t.optimizer, torch.optim.SGD) and t.optimizer.defaults["momentum"] == 0.9
    t.update_target_model(pol.model)
    assert t.target_model is not pol.model


# ---------------------------------------------------------------------------------------------------------------------------------
# the trainers' host logic against the REFERENCE trainers, on a CPU (fixture trainer_host.npz: tests/golden/make_golden.py drove
# crowd_nav/utils/trainer.py's MPRLTrainer / VNRLTrainer over tests/trainer_standins.py's small modules)
# ---------------------------------------------------------------------------------------------------------------------------------
from tests import golden_io as gio                                             # noqa: E402
from tests.trainer_standins import StandInValue, StandInPredictor, StandInPathG, seeded, flat_params   # noqa: E402


class _Scalars(object):
    def __init__(self):
        self.rows = []

    def add_scalar(self, tag, value, step):
        self.rows.append((tag, float(value), step))


@pytest.mark.parametrize("tag,shuffle,detach,reduce,opt,memory_kind", [
    ("ordered", False, True, False, "Adam", "list"),
    ("shuffled", True, False, True, "Adam", "list"),            # the product's own DataLoader, created lazily like upstream's
    ("shuffled", True, False, True, "Adam", "replay"),          # index-sampled from ReplayMemory.as_tensors(): the same batches
    ("shuffled_sgd", True, False, False, "SGD", "replay"),
])
def test_mprl_trainer_host_logic_against_the_reference_trainer(tag, shuffle, detach, reduce, opt, memory_kind):
    """train.py's order -- imitation epochs at one rate, new optimizers at another, target refreshed, RL batches, target refreshed, RL
    batches -- with a short last batch (70 transitions, batches of 16), the predictor updated on every third imitation batch, `reduce`
    / `detach` / SGD variants, un-shuffled and SHUFFLED batches (the loader's two draws from torch's global generator per pass): same
    final parameters and the same four reported losses as the reference MPRLTrainer on the same modules."""
    fx = gio.load("trainer_host")
    n = fx["th.robot"].shape[0]
    items = [tuple(torch.tensor(fx["th." + k][i]) for k in ("robot", "humans", "values", "rewards", "next_robot", "next_humans"))
             for i in range(n)]
    if memory_kind == "replay":
        memory = rga.ReplayMemory(1000)
        for it in items:
            memory.push(it)
    else:
        memory = items
    ve, sp = seeded(StandInValue, 11), seeded(StandInPredictor, 12)
    writer = _Scalars()
    t = rga.MPRLTrainer(ve, sp, memory, torch.device("cpu"), None, writer, 16, opt, 3, reduce, False, detach, False)
    if not shuffle:
        t.data_loader = DataLoader(memory, 16, shuffle=False)
    torch.manual_seed(5)
    t.set_learning_rate(1e-2)
    assert t.optimize_epoch(2) is None
    t.set_learning_rate(1e-3)
    t.update_target_model(ve)
    first = t.optimize_batch(2, 0)
    t.update_target_model(ve)
    second = t.optimize_batch(3, 1)
    assert not t._capturable and t._steps == {}
    got = np.array(list(first) + list(second))
    want = fx["th.mprl.%s.losses" % tag]
    assert np.all(np.abs(got - want) <= 1e-6 * np.maximum(1.0, np.abs(want))), (got, want)
    err = float(np.abs(flat_params(ve, sp) - fx["th.mprl.%s.params" % tag]).max())
    assert err <= 1e-6, err
    assert [r[0] for r in writer.rows] == ["IL/epoch_v_loss", "IL/epoch_s_loss"] * 2 + ["RL/average_v_loss", "RL/average_s_loss"] * 2
    assert [r[2] for r in writer.rows] == [0, 0, 1, 1, 0, 0, 1, 1]


@pytest.mark.parametrize("tag,shuffle", [("ordered", False), ("shuffled", True)])
def test_vnrl_trainer_host_logic_against_the_reference_trainer(tag, shuffle):
    """Path G's trainer the same way (pad_batch collate, imitation epochs then RL batches, shuffled loader included)."""
    fx = gio.load("trainer_host")
    n = fx["th.states"].shape[0]
    items = [tuple(torch.tensor(fx["th." + k][i]) for k in ("states", "values", "rewards", "next_states")) for i in range(n)]
    model = seeded(StandInPathG, 13)
    writer = _Scalars()
    t = rga.VNRLTrainer(model, items, torch.device("cpu"), None, 16, "Adam", writer)
    if not shuffle:
        t.data_loader = DataLoader(items, 16, shuffle=False, collate_fn=tr.pad_batch)
    torch.manual_seed(6)
    t.set_learning_rate(1e-2)
    il = t.optimize_epoch(2)
    t.set_learning_rate(1e-3)
    t.update_target_model(model)
    rl = t.optimize_batch(3, 0)
    want = fx["th.vnrl.%s.losses" % tag]
    assert abs(il - want[0]) <= 1e-6 * max(1.0, abs(want[0])) and abs(rl - want[1]) <= 1e-6 * max(1.0, abs(want[1])), (il, rl, want)
    err = float(np.abs(flat_params(model) - fx["th.vnrl.%s.params" % tag]).max())
    assert err <= 1e-6, err
    assert [r[0] for r in writer.rows] == ["IL/average_epoch_loss"] * 2
