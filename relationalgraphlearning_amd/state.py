"""Agent state containers with the reference's field order, for standalone use (inside crowd_nav the simulator's own
classes are passed in; the policies only read attributes, so either works).

Mirrors crowd_sim/envs/utils/state.py:4-92: FullState = (px, py, vx, vy, radius, gx, gy, v_pref, theta),
ObservableState = (px, py, vx, vy, radius), JointState = robot FullState + list of human ObservableStates."""
import torch

FULL_FIELDS = ("px", "py", "vx", "vy", "radius", "gx", "gy", "v_pref", "theta")
OBSERVABLE_FIELDS = ("px", "py", "vx", "vy", "radius")


class _Fields(object):
    FIELDS = ()

    def __init__(self, *values):
        if len(values) != len(self.FIELDS):
            raise TypeError("%s takes %d values" % (type(self).__name__, len(self.FIELDS)))
        for name, v in zip(self.FIELDS, values):
            setattr(self, name, v)
        self.position = (self.px, self.py)
        self.velocity = (self.vx, self.vy)

    def to_tuple(self):
        return tuple(getattr(self, n) for n in self.FIELDS)

    def __add__(self, other):          # robot + human -> the 14-tuple [robot 9 | human 5] the pairwise path consumes
        return other + self.to_tuple()

    def __str__(self):
        return " ".join(str(x) for x in self.to_tuple())


class ObservableState(_Fields):
    FIELDS = OBSERVABLE_FIELDS


class FullState(_Fields):
    FIELDS = FULL_FIELDS

    def __init__(self, *values):
        super().__init__(*values)
        self.goal_position = (self.gx, self.gy)

    def get_observable_state(self):
        return ObservableState(self.px, self.py, self.vx, self.vy, self.radius)


class JointState(object):
    def __init__(self, robot_state, human_states):
        self.robot_state = robot_state
        self.human_states = list(human_states)

    def to_tensor(self, add_batch_size=False, device=None):
        robot = torch.tensor([self.robot_state.to_tuple()], dtype=torch.float32)
        humans = torch.tensor([h.to_tuple() for h in self.human_states], dtype=torch.float32).reshape(-1, 5)
        if add_batch_size:
            robot, humans = robot.unsqueeze(0), humans.unsqueeze(0)
        if device is not None:
            robot, humans = robot.to(device), humans.to(device)
        return robot, humans


def tensor_to_joint_state(state):
    """(robot (..,9), humans (..,H,5)) tensors of ONE scene -> JointState of float32 scalars."""
    robot, humans = state
    r = robot.detach().cpu().reshape(-1).numpy()
    h = humans.detach().cpu().reshape(-1, 5).numpy()
    return JointState(FullState(*[r[i] for i in range(9)]), [ObservableState(*[row[i] for i in range(5)]) for row in h])
