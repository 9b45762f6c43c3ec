"""torch.optim.Adam with the reference's settings (scene/gaussian_model.py:173: lr per group, eps = 1e-15, no weight
decay) whose step is ONE HIP kernel per parameter tensor (csrc/gsr_optim.hip).  State layout ("step",
"exp_avg", "exp_avg_sq") is torch's, so the reference-style optimizer-state surgery (prune / append / replace)
and state_dict round trips work unchanged.

Split groups: a group with `head_cols` = h and `tail` = "<name of another group>" holds tensors [rows, cols, ...]
whose columns [0, h) step with the group's own lr and columns [h, cols) with the lr of the named group, which
itself holds no tensor.  That is how this package's GaussianModel keeps the reference's groups "f_dc"
(lr = feature_lr) and "f_rest" (lr = feature_lr / 20) over ONE interleaved SH table; element for element the
update equals Adam on the two tensors separately.

native=False (and any host tensor) takes the same arithmetic as torch ops, in torch.optim.Adam's own order of
operations (_single_tensor_adam), so CPU runs of the model's host logic agree with torch.optim.Adam bit for bit.
"""
import math

import torch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, native=True):
        self.native = native
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def _tail_lr(self, group):
        for g in self.param_groups:
            if g.get("name") == group["tail"]:
                return float(g["lr"])
        raise KeyError(f"FusedAdam: group {group.get('name')!r} names the tail group {group['tail']!r}, which does not exist")

    @staticmethod
    def _torch_step(p, g, st, step_sizes, b1, b2, eps, step, head_cols):
        m, v = st["exp_avg"], st["exp_avg_sq"]
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        if head_cols is None:
            p.addcdiv_(m, denom, value=-step_sizes[0])
        else:
            p[:, :head_cols].addcdiv_(m[:, :head_cols], denom[:, :head_cols], value=-step_sizes[0])
            p[:, head_cols:].addcdiv_(m[:, head_cols:], denom[:, head_cols:], value=-step_sizes[1])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        batch = []          # native tensors of this step: ONE launch for all of them (groups that share betas and eps: the reference's do)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            head_cols = group.get("head_cols") if group.get("tail") else None
            lr_tail = self._tail_lr(group) if head_cols is not None else None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                step = int(st["step"])
                lr = float(group["lr"])
                if not (self.native and p.is_cuda):
                    if self.native:
                        raise RuntimeError("FusedAdam(native=True) (MI355X build) needs parameters on a HIP device")
                    bc1 = 1 - b1 ** step
                    self._torch_step(p, p.grad, st, (lr / bc1, None if lr_tail is None else lr_tail / bc1), b1, b2, group["eps"], step,
                                     head_cols)
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if head_cols is None or p.shape[1] <= head_cols:
                    item = (p, g, st["exp_avg"], st["exp_avg_sq"], lr, lr, step, 0, 0)
                else:
                    per_col = p.numel() // max(p.shape[0] * p.shape[1], 1)
                    item = (p, g, st["exp_avg"], st["exp_avg_sq"], lr, lr_tail, step, p.numel() // max(p.shape[0], 1), head_cols * per_col)
                batch.append(((b1, b2, group["eps"], p.device), item))
        if batch:
            from diff_gaussian_rasterization import _native as N
            while batch:
                key = batch[0][0]
                now = [it for k, it in batch if k == key][:N.ADAM_MAX_TENSORS]
                taken = {id(it[0]) for it in now}
                batch = [(k, it) for k, it in batch if id(it[0]) not in taken]
                with torch.cuda.device(key[3]):
                    N.adam_step_multi(now, key[0], key[1], key[2])
                for it in now:
                    torch.autograd.graph.increment_version(it[0])    # the kernel wrote through the raw pointer: tell autograd (and
                                                                     # the model's activation cache, which keys on versions)
        return loss
