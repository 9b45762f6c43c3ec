"""The reference GaussianModel's activations (scene/gaussian_model.py:47-60, getters :101-125) as
ONE HIP launch forward and ONE backward (csrc/gsr_activations.hip) for all three tensors

    get_scaling = exp(_scaling)    get_rotation = normalize(_rotation)    get_opacity = sigmoid(_opacity)

behind the reference's three separate property getters.  The first getter a caller touches computes all three; the
others hand out the tensors of the same evaluation.  An evaluation is reused only while it is certainly current:
same three parameter objects, same autograd versions, same grad mode, and its backward has not run yet (a graph is
good for one backward).  `.data` edits do not bump a version: call ActivationCache.invalidate() after such an
edit (this package's optimizers bump the version themselves).

Host tensors (the CPU tests of the model's host logic) take the same formulas as torch ops.
"""
import torch


_TORCH = (torch.exp, torch.nn.functional.normalize, torch.sigmoid)


class _FusedActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, rotation, opacity, cache):
        from diff_gaussian_rasterization import _native
        s, q, o = _native.activations_forward(scaling.contiguous(), rotation.contiguous(), opacity.contiguous())
        ctx.save_for_backward(s, rotation, o)
        ctx.cache, ctx.epoch = cache, cache._epoch
        ctx.set_materialize_grads(False)             # an unused output arrives as None and costs nothing
        return s, q, o

    @staticmethod
    def backward(ctx, g_s, g_q, g_o):
        from diff_gaussian_rasterization import _native
        ctx.cache.invalidate(ctx.epoch)              # this graph has been used: the next getter call evaluates afresh
        s, rotation, o = ctx.saved_tensors
        need = ctx.needs_input_grad
        g = [None if (x is None or not need[i]) else x.contiguous() for i, x in enumerate((g_s, g_q, g_o))]
        if all(x is None for x in g):
            return None, None, None, None
        d_s, d_q, d_o = _native.activations_backward(s, rotation.contiguous(), o, g[0], g[1], g[2])
        return d_s, d_q, d_o, None


class ActivationCache:
    """One evaluation of the three activations, shared by the three getters."""

    def __init__(self):
        self._epoch = 0
        self.invalidate()

    def invalidate(self, epoch=None):
        if epoch is None or epoch == self._epoch:
            self._inputs = None     # strong references: an id() cannot be recycled while the entry lives
            self._versions = self._grad_mode = self._outputs = None

    def get(self, which, scaling, rotation, opacity):
        """which: 0 = scales, 1 = rotations, 2 = opacities."""
        inputs = (scaling, rotation, opacity)
        if not (scaling.is_cuda and all(t.dtype == torch.float32 and t.shape[0] == scaling.shape[0] for t in inputs)):
            return _TORCH[which](inputs[which])      # host tensors: the reference's own torch ops
        grad_mode = torch.is_grad_enabled()
        versions = tuple(t._version for t in inputs)
        if (self._outputs is not None and self._grad_mode == grad_mode and self._versions == versions
                and all(a is b for a, b in zip(self._inputs, inputs))):
            return self._outputs[which]
        self._epoch += 1
        if grad_mode and any(t.requires_grad for t in inputs):
            outs = _FusedActivations.apply(scaling, rotation, opacity, self)
        else:
            from diff_gaussian_rasterization import _native
            with torch.no_grad():
                outs = _native.activations_forward(scaling.contiguous(), rotation.contiguous(), opacity.contiguous())
        self._inputs, self._versions, self._grad_mode, self._outputs = inputs, versions, grad_mode, outs
        return outs[which]
