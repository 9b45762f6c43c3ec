"""torch.optim.Adam with the reference's settings (scene/gaussian_model.py:173: lr per group, eps = 1e-15, no weight
decay) whose step is ONE HIP kernel per parameter tensor (csrc/gsr_optim.hip).  State layout ("step",
"exp_avg", "exp_avg_sq") is torch's, so the reference-style optimizer-state surgery (prune / append / replace)
and state_dict round trips work unchanged."""
import torch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        from diff_gaussian_rasterization import _native as N
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam (MI355X build) needs parameters on a HIP device")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                with torch.cuda.device(p.device):
                    N.adam_step(p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), b1, b2, group["eps"], int(st["step"]))
        return loss
