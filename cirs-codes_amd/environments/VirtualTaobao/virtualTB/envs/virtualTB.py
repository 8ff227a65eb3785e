"""VirtualTB (reference environments/VirtualTaobao/virtualTB/envs/virtualTB.py:11-146 with its three generators,
virtualTB/model/{UserModel,ActionModel,LeaveModel}.py): BASELINE configs[0] -- CPU plumbing, no GPU, exactly as in the reference
("VirtualTaobao, 4 parallel envs, ... on CPU via CIRS-RL-taobao.py").  Three small MLPs with the SHIPPED weights
(virtualTB/data/{generator,action,leave}_model.pt: data files of the reference, looked up in `data_dir`, $CIRS_VIRTUALTB_DATA,
next to this package, or under tests/golden/virtualtb) drive the simulation through torch's CPU generator:

  user        z ~ U(0,1)^128 -> generator MLP -> 11 soft-max groups (88 logits) -> one categorical draw per group -> 88-d one-hot
  leave page  categorical draw from the leave MLP (recorded; the CIRS exit rule below decides `done`)
  step        exit rule: leave iff some of the last min(t, N-1) actions lies within `leave_threshold` (L2) of the new one
              (SURVEY Q3), or t >= max_turn - 1; click count a ~ Cat(softmax(x[:11])), b ~ Cat(softmax(x[11:])) from the action
              MLP on [user (88), page (1), action (27)]; reward = a; a finished episode draws the next user immediately.

The ORDER of the generator calls is part of the behaviour (one torch.rand + eleven multinomials per user, one multinomial per
leave page, two per step): with torch.manual_seed(s) this class and the reference produce the same trajectories bit for bit
(tests/test_virtualtb_cpu.py)."""
import os
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

try:
    import gym
    from gym import spaces
except ImportError:
    from cirs_hip import gymlite as gym
    spaces = gym.spaces

_GROUPS = [(0, 8), (8, 16), (16, 27), (27, 38), (38, 49), (49, 60), (60, 62), (62, 64), (64, 67), (67, 85), (85, 88)]


def _mlp(sizes):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [nn.Linear(a, b), nn.LeakyReLU()]
    return nn.Sequential(*layers[:-1])


def _normal_init(seq):
    """N(0, sqrt(2 / (fan_in + fan_out))) weights, zero biases (virtualTB/utils.py:15-22 init_weight), in module order."""
    for m in seq:
        if isinstance(m, nn.Linear):
            fan_out, fan_in = m.weight.shape
            m.weight.data.normal_(0.0, float(np.sqrt(2.0 / (fan_in + fan_out))))
            m.bias.data.fill_(0.0)


def find_data_dir(data_dir=None):
    here = os.path.dirname(os.path.abspath(__file__))
    cands = [data_dir, os.environ.get("CIRS_VIRTUALTB_DATA"), os.path.join(here, "..", "data"),
             os.path.join(here, "..", "..", "..", "..", "..", "tests", "golden", "virtualtb")]
    for c in cands:
        if c and os.path.isfile(os.path.join(c, "generator_model.pt")):
            return c
    raise FileNotFoundError("VirtualTB simulator weights (generator_model.pt, action_model.pt, leave_model.pt) not found; "
                            "pass data_dir= or set CIRS_VIRTUALTB_DATA to the reference's environments/VirtualTaobao/virtualTB/data")


class VirtualTB(gym.Env):
    metadata = {"render.modes": ["human"]}
    simulated = False

    def __init__(self, num_leave_compute=5, leave_threshold=4.5, max_turn=100, data_dir=None):
        self.n_user_feature, self.n_item_feature, self.max_turn = 88, 27, max_turn
        self.obs_low = np.concatenate(([0] * self.n_user_feature, [0, 0, 0]))
        self.obs_high = np.concatenate(([1] * self.n_user_feature, [29, 9, 100]))
        self.observation_space = spaces.Box(low=self.obs_low, high=self.obs_high, dtype=np.int32)
        self.action_space = spaces.Box(low=-1, high=1, shape=(self.n_item_feature,), dtype=np.float32)
        d = find_data_dir(data_dir)
        # Construction order and the throw-away normal initialisation of the generator / leave MLPs are kept: the reference's
        # constructors draw them from torch's global generator BEFORE the shipped weights are loaded (UserModel.py:14,
        # LeaveModel.py:16 `apply(init_weight)`), so a run seeded with torch.manual_seed(s) before gym.make() only lines up with
        # the reference if this constructor consumes the generator the same way.
        self.generator = _mlp([128, 128, 88])                       # UserModel.generator_model
        _normal_init(self.generator)
        self.action_model = _mlp([88 + 1 + 27, 128, 256, 11 + 10])  # ActionModel.model
        self.leave_model = _mlp([88, 128, 256, 101])                # LeaveModel.model
        _normal_init(self.leave_model)
        for m, f in ((self.generator, "generator_model.pt"), (self.action_model, "action_model.pt"), (self.leave_model, "leave_model.pt")):
            m.load_state_dict(torch.load(os.path.join(d, f), map_location="cpu"))
        self.static = False
        self.num_leave_compute, self.leave_threshold = num_leave_compute, leave_threshold
        self.reset()

    def set_state_mode(self, is_static=False):
        self.static = is_static

    def seed(self, sd=0):
        torch.manual_seed(sd)

    # ---- generators -----------------------------------------------------------------------------------------------
    def _draw_user(self):
        z = torch.rand((1, 128))
        x = self.generator(z)
        probs = torch.cat([F.softmax(x[:, a:b], dim=1) for a, b in _GROUPS], dim=-1)
        one_hot = torch.cat([torch.zeros(1, b - a).scatter_(1, torch.multinomial(probs[:, a:b], 1), 1) for a, b in _GROUPS], dim=-1)
        self._leave_page = torch.multinomial(F.softmax(self.leave_model(one_hot), dim=1), 1)   # drawn, not used by the CIRS exit rule
        return one_hot.squeeze().detach().numpy()

    def _user_response(self, action):
        user = torch.FloatTensor(self.cur_user).unsqueeze(0)
        page = torch.FloatTensor([[self.total_turn]])
        x = self.action_model(torch.cat((user, page, torch.FloatTensor(action).unsqueeze(0)), dim=-1))
        a = torch.multinomial(F.softmax(x[:, :11], dim=1), 1)
        b = torch.multinomial(F.softmax(x[:, 11:], dim=1), 1)
        return torch.cat((a, b), dim=-1).detach().numpy()[0]

    # ---- gym protocol ---------------------------------------------------------------------------------------------
    @property
    def state(self):
        head = self.cur_user if (self.static or self.action is None) else self.action
        return np.concatenate((head, self.lst_action, np.array([self.total_turn])), axis=-1)

    def reset(self):
        self.cum_reward, self.total_turn = 0, 0
        self.cur_user = self._draw_user()
        self.lst_action = torch.FloatTensor([0, 0])
        self.rend_action = deepcopy(self.lst_action)
        self.action = None
        self.history_action, self.max_history = {}, 0
        return self.state

    def _determine_whether_to_leave(self, t, action):
        for t_l in range(t - 1, max(-1, t - self.num_leave_compute), -1):     # the last min(t, N - 1) actions
            if np.linalg.norm(np.asarray(action) - np.asarray(self.history_action[t_l])) <= self.leave_threshold:
                return True
        return False

    def step(self, action):
        self.action = action
        t = self.total_turn
        done = self._determine_whether_to_leave(t, action) or t >= self.max_turn - 1
        assert self.max_history == t
        self.history_action[t] = action
        self.max_history += 1
        self.lst_action = self._user_response(action)
        reward = int(self.lst_action[0])
        self.cum_reward += reward
        self.total_turn += 1
        self.rend_action = deepcopy(self.lst_action)
        if done:
            self.cur_user = self._draw_user()
            self.lst_action = torch.FloatTensor([0, 0])
        return self.state, reward, done, {"CTR": self.cum_reward / self.total_turn / 10}

    def render(self, mode="human", close=False):
        a, b = np.clip(self.rend_action, a_min=0, a_max=None)
        print("Current State:\n\t", self.state, "\nUser's action:\n\tclick:%2d, leave:%s, index:%2d" %
              (int(a), "True" if self.total_turn > (self.max_turn - 1) else "False", int(self.total_turn)), "\nTotal clicks:", self.cum_reward)
