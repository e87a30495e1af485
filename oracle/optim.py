"""Adam and Polyak restated (oracle; test infrastructure only).

Third-party arithmetic: the reference calls `torch.optim.Adam` (un-vendored torch; call sites
agent/sac/sac_agent.py:71-81 and the per-agent optimizers listed in SURVEY.md row a13).  The update
rule below is the published Adam rule in the operation order of torch 2.10
`torch/optim/adam.py::_single_tensor_adam` (weight_decay=0, amsgrad=False, maximize=False,
capturable=False), which is what the reference executes on CPU (foreach=None -> single tensor).
"""
import math
import torch


class Adam:
    """Per-tensor Adam over a dict name -> tensor.  State is created lazily, only for tensors that
    received a gradient (torch skips `p.grad is None`: vlsac `l6`, diffsrsac's critic optimizer)."""

    def __init__(self, names, lr, betas=(0.9, 0.999), eps=1e-8):
        self.names = list(names)
        self.lr, self.b1, self.b2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        self.state = {}

    @torch.no_grad()
    def step(self, params, grads):
        for n in self.names:
            g = grads.get(n, None)
            if g is None:
                continue
            p = params[n]
            st = self.state.get(n)
            if st is None:
                st = self.state[n] = dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p))
            st['step'] += 1
            t = st['step']
            m, v = st['m'], st['v']
            m.lerp_(g, 1 - self.b1)                                   # m += (1-b1)(g-m)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)         # v = b2 v + (1-b2) g^2
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            step_size = self.lr / bc1
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-step_size)


@torch.no_grad()
def polyak(params, src_prefix, dst_prefix, tau):
    """target <- tau*src + (1-tau)*target per tensor (agent/sac/sac_agent.py:99-102)."""
    for n in list(params.keys()):
        if n.startswith(src_prefix + '.'):
            d = dst_prefix + n[len(src_prefix):]
            if d in params and not n.endswith('noise'):
                params[d].copy_(tau * params[n] + (1 - tau) * params[d])
