"""Host-side logic of the graph-replaying trainers (relationalgraphlearning_amd/trainer.py) that needs no GPU: the batch order
(the DataLoader's own, index for index), the stacked view of the replay memory, the reference's contract."""
import types

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
