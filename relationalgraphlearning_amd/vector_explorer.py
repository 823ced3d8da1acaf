"""Explorer / replay-memory plumbing for many environments in lock-step.

`VectorExplorer.run_k_episodes` keeps the contract of crowd_nav/utils/explorer.py:21-111 -- same arguments, same five
statistics, same log lines (crowd_nav/utils/plot.py:50-66 parses them) and the same experience tuples pushed into the
replay memory (explorer.py:113-140) -- but runs the k episodes side by side on the device: one `policy.predict_batch`
and one `BatchedCrowdSim.step` per time step for all live environments.  `ReplayMemory` is the reference's ring buffer
(crowd_nav/utils/memory.py) so `MPRLTrainer` / `VNRLTrainer` consume it unchanged through a DataLoader.

Differences that follow from vectorisation (documented, not hidden): episodes of one call are the NEXT k seeded cases
of the phase (the reference draws them one after another from the same counter, so the set is identical; the order
inside the replay memory is episode-major here as well); epsilon-greedy exploration draws its random numbers per
time step for all environments at once, so the random stream differs from k sequential episodes.
"""
import logging

import numpy as np
import torch
from torch.utils.data import Dataset

from .actions import as_array
from .rollout import rotate

COLLISION, SUCCESS, TIMEOUT = 2, 3, 4          # BatchedCrowdSim info codes
CASE_SIZE = {"train": np.iinfo(np.uint32).max - 2000, "val": 100, "test": 500}     # crowd_sim.py:60-62 defaults


class ReplayMemory(Dataset):
    """Fixed-capacity ring of experience tuples; the oldest entry is overwritten once full (crowd_nav/utils/memory.py).
    Additive: `as_tensors()` -- the experience as one stacked tensor per tuple field, kept up to date incrementally -- which lets
    the trainers gather a batch with one index_select per field instead of collating 100 python tuples (trainer.py)."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.memory = []
        self.position = 0
        self._mirror = None           # per field: (capacity, *item shape) tensor on the items' device
        self._dirty = []              # positions written since the mirror was last brought up to date
        self._unstackable_at = -1     # len(memory) when the items were last found to differ in shape (no stacked view): the
                                      # check is not repeated -- O(n) host work and capacity-sized allocations -- until the ring
                                      # has been cleared or has wrapped far enough to have dropped the odd items

    def push(self, item):
        if self.position < len(self.memory):
            self.memory[self.position] = item
            self._dirty.append(self.position)
        else:
            self.memory.append(item)                 # (after clear() the write position is ahead of the list: upstream's ring)
            self._dirty.append(len(self.memory) - 1)
        self.position = (self.position + 1) % self.capacity

    def is_full(self):
        return len(self.memory) == self.capacity

    def clear(self):
        self.memory = []          # the write position is kept, as upstream does
        self._mirror, self._dirty, self._unstackable_at = None, [], -1

    def __getitem__(self, index):
        return self.memory[index]

    def __len__(self):
        return len(self.memory)

    def as_tensors(self):
        """[field 0 of every item stacked (n, ...), field 1 ..., ...] in item order -- what DataLoader's default collate would make
        of the whole memory -- or None when the items are not tuples of equally shaped tensors (path G's variable crowds).  Only the
        entries pushed since the last call are copied."""
        n = len(self.memory)
        if n == 0:
            return None
        first = self.memory[0]
        if not (isinstance(first, tuple) and all(torch.is_tensor(x) for x in first)):
            return None
        if self._unstackable_at >= 0:
            # mixed shapes were found with this many items: until the ring is full and has been overwritten once more (the odd
            # items may be gone then) the answer cannot change -- no re-allocation, no walk over the items per call
            if not (self.is_full() and len(self._dirty) >= self.capacity):
                if len(self._dirty) > self.capacity:
                    self._dirty = self._dirty[-self.capacity:]
                return None
            self._unstackable_at = -1
        if self._mirror is None or len(self._mirror) != len(first) or any(
                m.shape[1:] != x.shape or m.device != x.device or m.dtype != x.dtype for m, x in zip(self._mirror, first)):
            self._mirror = [torch.empty((self.capacity,) + tuple(x.shape), dtype=x.dtype, device=x.device) for x in first]
            self._dirty = list(range(n))
        dirty = sorted(set(i for i in self._dirty if i < n))
        if dirty:
            for f, m in enumerate(self._mirror):
                rows = [self.memory[i][f] for i in dirty]
                if any(r.shape != m.shape[1:] or r.device != m.device or r.dtype != m.dtype for r in rows):
                    self._mirror, self._dirty, self._unstackable_at = None, [], n
                    return None
                m.index_copy_(0, torch.tensor(dirty, dtype=torch.int64, device=m.device), torch.stack(rows))   # index on the field's device
        self._dirty = []
        return [m[:n] for m in self._mirror]

    def stacked_capacity_fields(self):
        """The tensors `as_tensors()` returns views of, at their full (capacity, ...) extent -- rows >= len(self) are unwritten.
        What a trainer's CAPTURED step gathers from: the extent never changes while the memory grows, so the recorded
        index_select kernels stay valid (indices are always < len(self)).  None when there is no stacked view."""
        if self.as_tensors() is None:
            return None
        return list(self._mirror)


def discounted_statistics(rewards, lengths, step_discount):
    """rewards (T,B) (zeros after an episode's end), lengths (B,) -> per episode (cumulative discounted reward,
    mean over its steps of the discounted return-to-go), as explorer.py:78-85 computes them.
    step_discount = gamma ** (time_step * v_pref)."""
    rewards = np.asarray(rewards, np.float64)
    T, B = rewards.shape
    togo = np.zeros((T + 1, B))
    for t in range(T - 1, -1, -1):                       # return-to-go G_t = r_t + d * G_{t+1}
        togo[t] = rewards[t] + step_discount * togo[t + 1]
    cumulative = togo[0].copy()
    valid = np.arange(T)[:, None] < np.asarray(lengths)[None, :]
    avg_return = np.where(valid, togo[:T], 0.0).sum(0) / np.maximum(np.asarray(lengths), 1)
    return cumulative, avg_return


def _mean(values):
    values = list(values)
    return sum(values) / len(values) if values else 0


class VectorExplorer(object):
    def __init__(self, sim, policy, device=None, writer=None, memory=None, gamma=None, target_policy=None,
                 case_size=None, max_batch=4096):
        self.sim = sim
        self.policy = policy                    # the acting policy (needs predict_batch)
        self.device = device or sim.device
        self.writer = writer
        self.memory = memory
        self.gamma = gamma
        self.target_policy = target_policy
        self.statistics = None
        self.case_size = dict(CASE_SIZE, **(case_size or {}))
        self.case_counter = {"train": 0, "val": 0, "test": 0}
        self.max_batch = int(max_batch)
        self.last_run = None                    # per-episode arrays of the most recent call

    # -- episodes ------------------------------------------------------------------------------------------------
    def _next_cases(self, phase, k):
        cases = [(self.case_counter[phase] + i) % self.case_size[phase] for i in range(k)]
        self.case_counter[phase] = (self.case_counter[phase] + k) % self.case_size[phase]
        return cases

    def _act(self, robot32, humans32, phase, n_actions):
        idx, _ = self.policy.predict_batch(robot32, humans32, roots_are_joint_states=True)
        idx = idx.long()
        eps = getattr(self.policy, "epsilon", None)
        if phase == "train" and eps:
            B = robot32.shape[0]
            explore = torch.as_tensor(np.random.random(B) < eps, device=idx.device)
            rand_idx = torch.as_tensor(np.random.randint(0, n_actions, B), device=idx.device)
            idx = torch.where(explore, rand_idx, idx)
        return idx

    def _run_chunk(self, phase, cases, keep_states):
        sim, policy = self.sim, self.policy
        robot32, humans32 = sim.reset(phase, cases)
        B = sim.B
        if policy.action_space is None:
            policy.build_action_space(sim.cfg.robot_v_pref)
        table = torch.tensor(as_array(policy.action_space), dtype=torch.float64, device=sim.device)
        max_steps = int(round(sim.cfg.time_limit / sim.cfg.time_step)) + 2
        outcome = torch.zeros(B, dtype=torch.int32, device=sim.device)
        rewards, infos, dmins, states, actions = [], [], [], [], []
        for _ in range(max_steps):
            if not bool((sim.done == 0).any()):
                break
            if keep_states:
                states.append((robot32.clone(), humans32.clone()))
            idx = self._act(robot32, humans32, phase, table.shape[0])
            (robot32, humans32), reward, _, info = sim.step(table[idx])
            actions.append(idx)
            rewards.append(reward)
            infos.append(info)
            dmins.append(sim.last_dmin)
            ended = (info >= COLLISION) & (info <= TIMEOUT)
            outcome = torch.where(ended, info, outcome)
        info_t = torch.stack(infos).cpu().numpy()                    # (T,B); 5 = finished earlier
        live = info_t != 5
        lengths = live.sum(0)
        reward_t = np.where(live, torch.stack(rewards).cpu().numpy().astype(np.float64), 0.0)
        return {"outcome": outcome.cpu().numpy(), "time": sim.time.cpu().numpy().copy(), "lengths": lengths,
                "rewards": reward_t, "info": info_t, "dmin": torch.stack(dmins).cpu().numpy(),
                "states": states, "actions": actions}

    def run_k_episodes(self, k, phase, update_memory=False, imitation_learning=False, episode=None, epoch=None,
                       print_failure=False):
        self.policy.set_phase(phase)
        cases = self._next_cases(phase, k)
        time_limit = self.sim.cfg.time_limit
        step_discount = pow(self.gamma if self.gamma is not None else 0.9,
                            self.sim.cfg.time_step * self.sim.cfg.robot_v_pref)
        success_times, collision_times, timeout_times = [], [], []
        collision_cases, timeout_cases, min_dist = [], [], []
        cumulative_rewards, average_returns = [], []
        discomfort = 0
        per_episode = {"case": [], "outcome": [], "time": [], "length": [], "actions": []}
        for lo in range(0, k, self.max_batch):
            chunk = cases[lo:lo + self.max_batch]
            run = self._run_chunk(phase, chunk, keep_states=update_memory)
            if (run["outcome"] == 0).any():
                raise ValueError('Invalid end signal from environment')
            cum, avg_ret = discounted_statistics(run["rewards"], run["lengths"], step_discount)
            acts = torch.stack(run["actions"]).cpu().numpy() if run["actions"] else np.zeros((0, len(chunk)), np.int64)
            for b in range(len(chunk)):
                i = lo + b
                code = int(run["outcome"][b])
                if code == SUCCESS:
                    success_times.append(float(run["time"][b]))
                elif code == COLLISION:
                    collision_cases.append(i)
                    collision_times.append(float(run["time"][b]))
                else:
                    timeout_cases.append(i)
                    timeout_times.append(time_limit)
                cumulative_rewards.append(float(cum[b]))
                average_returns.append(float(avg_ret[b]))
                per_episode["case"].append(chunk[b])
                per_episode["outcome"].append(code)
                per_episode["time"].append(float(run["time"][b]))
                per_episode["length"].append(int(run["lengths"][b]))
                per_episode["actions"].append([int(a) for a in acts[:int(run["lengths"][b]), b]])
            danger = run["info"] == 1
            discomfort += int(danger.sum())
            min_dist.extend(run["dmin"][danger].tolist())
            if update_memory:
                for b in range(len(chunk)):
                    if int(run["outcome"][b]) in (SUCCESS, COLLISION):       # positive or negative experience only
                        T = int(run["lengths"][b])
                        self.update_memory([(s[0][b:b + 1], s[1][b]) for s in run["states"][:T]],
                                           [a[b] for a in run["actions"][:T]],
                                           [float(r) for r in run["rewards"][:T, b]], imitation_learning)
        success, collision, timeout = len(success_times), len(collision_times), len(timeout_times)
        assert success + collision + timeout == k
        success_rate, collision_rate = success / k, collision / k
        avg_nav_time = sum(success_times) / len(success_times) if success_times else time_limit

        extra_info = '' if episode is None else 'in episode {} '.format(episode)
        extra_info = extra_info + '' if epoch is None else extra_info + ' in epoch {} '.format(epoch)
        logging.info('{:<5} {}has success rate: {:.2f}, collision rate: {:.2f}, nav time: {:.2f}, total reward: {:.4f},'
                     ' average return: {:.4f}'.format(phase.upper(), extra_info, success_rate, collision_rate,
                                                      avg_nav_time, _mean(cumulative_rewards), _mean(average_returns)))
        if phase in ['val', 'test']:
            total_time = sum(success_times + collision_times + timeout_times)
            logging.info('Frequency of being in danger: %.2f and average min separate distance in danger: %.2f',
                         discomfort / total_time, _mean(min_dist))
        if print_failure:
            logging.info('Collision cases: ' + ' '.join([str(x) for x in collision_cases]))
            logging.info('Timeout cases: ' + ' '.join([str(x) for x in timeout_cases]))

        self.last_run = dict(per_episode, cumulative_reward=cumulative_rewards, average_return=average_returns,
                             discomfort_steps=discomfort, min_dist=min_dist)
        self.statistics = (success_rate, collision_rate, avg_nav_time, _mean(cumulative_rewards), _mean(average_returns))
        return self.statistics

    # -- replay memory -------------------------------------------------------------------------------------------
    def _transform(self, state):
        """(robot (1,9), humans (H,5)) -> what `target_policy.transform(JointState)` returns for that state."""
        if self.target_policy.name == 'ModelPredictiveRL':
            return state
        robot, humans = state
        joint = torch.cat([robot.expand(humans.shape[0], 9), humans], dim=1).contiguous()
        return rotate(joint, self.target_policy.kinematics)

    def update_memory(self, states, actions, rewards, imitation_learning=False):
        """One finished episode: states[i] = (robot (1,9), humans (H,5)) fp32 device tensors (`policy.last_state`)."""
        if self.memory is None or self.gamma is None:
            raise ValueError('Memory or gamma value is not set!')
        step_discount = pow(self.gamma, self.sim.cfg.time_step * self.sim.cfg.robot_v_pref)
        n = len(states)
        togo = [0.0] * (n + 1)
        for t in range(n - 1, -1, -1):
            togo[t] = rewards[t] + step_discount * togo[t + 1]
        mprl = self.target_policy.name == 'ModelPredictiveRL'
        for i in range(n - 1):                                   # the last state has no successor: not stored
            if imitation_learning:
                value = togo[i]                                  # discounted return from step i
            else:
                value = 0                                        # RL: the trainer bootstraps from the target network
            state, next_state = self._transform(states[i]), self._transform(states[i + 1])
            value = torch.tensor([value], dtype=torch.float32, device=self.device)
            reward = torch.tensor([rewards[i]], dtype=torch.float32, device=self.device)
            if mprl:
                self.memory.push((state[0], state[1], value, reward, next_state[0], next_state[1]))
            else:
                self.memory.push((state, value, reward, next_state))

    def log(self, tag_prefix, global_step):
        """TensorBoard scalars under the reference's tag names (explorer.py:142-148)."""
        for tag, value in zip(("success_rate", "collision_rate", "time", "reward", "avg_return"), self.statistics):
            self.writer.add_scalar("%s/%s" % (tag_prefix, tag), value, global_step)
