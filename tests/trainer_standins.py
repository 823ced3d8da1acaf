"""Small CPU modules with the CALL CONTRACTS of the networks the trainers drive -- ValueEstimator (model((robot (B,1,9), humans
(B,H,5))) -> (B,1), value_estimator.py:11-20), StatePredictor (model(state, action, detach=False) -> (robot', humans' (B,H,5)),
`.trainable`, state_predictor.py:20-39), gcn.ValueNetwork (model((states (B,L,13), lengths)) -> (B,1), gcn.py:85-128) -- so that the
trainers' HOST logic (batching, step order, optimizer handling, target model, loss bookkeeping) can be held against the reference
trainer on a CPU, where the real networks (GPU kernels only) do not run.  No product imports: tests/golden/make_golden.py drives the
reference trainers with the same classes."""
import torch
import torch.nn as nn


class StandInValue(nn.Module):
    def __init__(self):
        super().__init__()
        self.robot, self.humans, self.head = nn.Linear(9, 8), nn.Linear(5, 8), nn.Linear(8, 1)

    def forward(self, state):
        robot, humans = state
        return self.head(torch.tanh(self.robot(robot[:, 0]) + self.humans(humans).mean(1)))


class StandInPredictor(nn.Module):
    trainable = True

    def __init__(self):
        super().__init__()
        self.enc, self.dec = nn.Linear(5, 8), nn.Linear(8, 5)

    def forward(self, state, action, detach=False):
        robot, humans = state
        emb = torch.tanh(self.enc(humans))
        if detach:
            emb = emb.detach()
        return None, self.dec(emb)


class StandInPathG(nn.Module):
    def __init__(self):
        super().__init__()
        self.enc, self.head = nn.Linear(13, 8), nn.Linear(8, 1)

    def forward(self, state_input):
        states = state_input[0] if isinstance(state_input, tuple) else state_input
        return self.head(torch.tanh(self.enc(states)).mean(1))


def seeded(cls, seed):
    torch.manual_seed(seed)
    return cls()


def flat_params(*modules):
    return torch.cat([p.detach().flatten() for m in modules for p in m.parameters()]).double().numpy()
