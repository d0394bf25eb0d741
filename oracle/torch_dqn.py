"""torch-CPU fp32 restatement of one keras-rl double-DQN update -- the CPU baseline's learner.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests (cross-checked against the float64 oracle) and by the
`cpu_baseline` leg of bench.py, never by the product.  PARITY UNPINNED at source level like dqn_oracle.py (same published
keras-rl 0.4.x / Keras 2.2 arithmetic, call sites
/root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:61-90,119-130); this variant exists because the
reference's learner is TensorFlow fp32 on the host cores, and torch-CPU fp32 (MKL/oneDNN convolutions + autograd, all cores) is
the closest thing to it that runs on the GPU box (SURVEY.md 8d "CPU baseline beside it" (3)).
"""
import numpy as np
import torch
import torch.nn.functional as F


class TorchDQN:
    def __init__(self, spec, flat, lr=1e-4, gamma=0.99, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.spec, self.lr, self.gamma, self.b1, self.b2, self.eps = spec, lr, gamma, beta_1, beta_2, epsilon
        self.params = self._unflatten(flat)
        self.target = [p.detach().clone() for p in self.params]
        for p in self.params:
            p.requires_grad_(True)
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def _unflatten(self, flat):
        out = []
        for k, b in self.spec.split(np.asarray(flat, dtype=np.float32)):
            k = torch.from_numpy(np.ascontiguousarray(k))
            if k.dim() == 4:
                k = k.permute(3, 2, 0, 1).contiguous()           # Keras HWIO -> torch OIHW
            out += [k.clone(), torch.from_numpy(np.ascontiguousarray(b)).clone()]
        return out

    def flat(self, tensors=None):
        """Back to the flat Keras-ordered vector (float64 numpy)."""
        tensors = self.params if tensors is None else tensors
        out = []
        for i in range(0, len(tensors), 2):
            k, b = tensors[i].detach(), tensors[i + 1].detach()
            if k.dim() == 4:
                k = k.permute(2, 3, 1, 0)
            out += [k.reshape(-1).numpy().astype(np.float64), b.numpy().astype(np.float64)]
        return np.concatenate(out)

    def forward(self, params, obs, keep=None):
        """obs uint8 (B,C,H,W) tensor / array -> Q (B,A) fp32.  keep: (B,units) bool mask of the dropout layer (training) or None."""
        x = torch.as_tensor(obs).to(torch.float32)
        i = 0
        for kind, L in self.spec.layers:
            w, b = params[i], params[i + 1]
            i += 2
            if kind == "conv":
                x = F.relu(F.conv2d(x, w, b, stride=L["s"]))
            else:
                if x.dim() == 4:
                    x = x.flatten(1)                              # channels_first Flatten
                x = x @ w + b
                if L["relu"]:
                    x = F.relu(x)
                if keep is not None and L["dropout"] > 0.0:
                    x = torch.where(torch.as_tensor(keep), x / (1.0 - L["dropout"]), torch.zeros_like(x))
        if self.spec.dueling:
            x = x[:, 0:1] + x[:, 1:] - x[:, 1:].mean(dim=1, keepdim=True)
        return x

    def update(self, s0, action, reward, terminal, s1, keep):
        """One DQNAgent.backward training step: double-DQN target, 0.5 x^2 loss on the taken action, Keras Adam."""
        with torch.no_grad():
            a_star = self.forward(self.params, s1).argmax(dim=1)
            q_t = self.forward(self.target, s1).gather(1, a_star[:, None])[:, 0]
            y = torch.as_tensor(reward, dtype=torch.float32) + self.gamma * q_t * (1.0 - torch.as_tensor(terminal).to(torch.float32))
        q0 = self.forward(self.params, s0, keep=keep)
        qa = q0.gather(1, torch.as_tensor(action, dtype=torch.int64)[:, None])[:, 0]
        loss = 0.5 * ((qa - y) ** 2).mean()
        grads = torch.autograd.grad(loss, self.params)
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        with torch.no_grad():
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
                p.sub_(lr_t * m / (v.sqrt() + self.eps))
        return float(loss.detach()), float(q0.detach().max(dim=1).values.mean()), grads

    def sync_target(self):
        self.target = [p.detach().clone() for p in self.params]
