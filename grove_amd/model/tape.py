"""A minimal reverse-mode tape over the HIP kernels, used for the small trainable parts of the path
(mm_projector, text_hidden_fcs, prompt/mask decoder): forward code reads like the reference module,
every op pushes the closure that back-propagates through the matching backward kernels, weight
gradients are accumulated in fp32 straight into the caller's flat gradient buffer views.

The big towers (LLaMA, SAM) do not use the tape: their backward is written out explicitly so that
activation lifetime in HBM stays under control.
"""
import torch

from .. import ops

bf = torch.bfloat16


class Var:
    __slots__ = ("data", "grad", "needs_grad", "shared")

    def __init__(self, data, needs_grad=True):
        self.data = data
        self.grad = None
        self.needs_grad = needs_grad
        self.shared = False  # grad is a tensor somebody else holds too (see Tape.acc): never written in place


class Param:
    """A bf16 weight plus (optionally) the f32 view that receives its gradient."""
    __slots__ = ("w", "g")

    def __init__(self, w, g=None):
        self.w, self.g = w, g


class Tape:
    def __init__(self, enabled=True, side=None):
        """side (a torch.cuda.Stream or None): weight and bias gradients of `linear` are launched on that stream, gated on an event of
        the launching stream — they feed nothing but the optimizer, so they leave the dgrad chain's critical path (the caller joins the
        side stream before anything reads the gradient views)."""
        self.enabled = enabled
        self.side = side
        self.fns = []
        self.dgrad_weights = []  # weights whose transpose the backward's dgrads will read (batched into one launch by backward())
        self.wt = {}

    def push(self, fn):
        if self.enabled:
            self.fns.append(fn)

    def backward(self):
        # every W^T the dgrads below read, in ONE launch (ops.transpose2d_many) instead of one launch-bound transpose per linear on the serial
        # chain; made HERE, from the weights as they are now — nothing is kept across an optimizer step
        uniq = {}
        for w in self.dgrad_weights:
            uniq.setdefault(w.data_ptr(), w)
        if len(uniq) > 1:
            ws = list(uniq.values())
            self.wt = {w.data_ptr(): t for w, t in zip(ws, ops.transpose2d_many(ws))}
        for fn in reversed(self.fns):
            fn()
        self.fns = []
        self.dgrad_weights = []
        self.wt = {}

    # ---- helpers
    @staticmethod
    def acc(v, g, shared=False):
        """v.grad += g. The first gradient is taken by reference; `shared` says that the caller keeps using that tensor (it is another Var's
        gradient as well, or a weight gradient on the side stream still reads it), so the next accumulation makes a NEW tensor instead of
        adding in place — what a .clone() per fan-out bought before, without the copy launch."""
        if not v.needs_grad:
            return
        if v.grad is None:
            v.grad, v.shared = g, shared
        elif v.shared:
            v.grad, v.shared = ops.add(v.grad, g), False
        else:
            ops.add(v.grad, g, out=v.grad)

    # ---- ops
    def linear(self, x: Var, W: Param, b: Param = None, act=ops.ACT_NONE, residual: Var = None):
        """y = act(x W^T + b) (+ residual)."""
        need_pre = self.enabled and act != ops.ACT_NONE
        M = x.data.shape[0]
        pre = torch.empty((M, W.w.shape[0]), dtype=bf, device=x.data.device) if need_pre else None
        y = Var(ops.linear(x.data, W.w, b.w if b is not None else None, act=act, aux=pre,
                           residual=residual.data if residual is not None else None))
        if self.enabled and x.needs_grad and W.w.dim() == 2 and W.w.stride(1) == 1:
            self.dgrad_weights.append(W.w)

        def bwd():
            dy = y.grad
            if dy is None:
                return
            side = self.side if (W.g is not None or (b is not None and b.g is not None)) else None
            if residual is not None:
                # (the residual branch is handed dy itself, marked shared: dz below — and, on the side stream, the weight gradient — keep
                # reading it; with an activation dz is a fresh tensor and dy belongs to the residual branch alone)
                Tape.acc(residual, dy, shared=(act == ops.ACT_NONE))
            dz = ops.act_bwd(pre, dy, act) if act != ops.ACT_NONE else dy
            if residual is not None and act == ops.ACT_NONE:
                dz = dy  # shared with the residual branch: treated read-only below
            N, K = W.w.shape

            def param_grads():
                if W.g is not None:
                    ops.wgrad(dz, x.data, W.g, K=M)  # dW[N, K] += dz^T x on the K-major operands as they lie in HBM
                if b is not None and b.g is not None:
                    ops.colsum(dz, out=b.g, accumulate=True)
            if side is None:
                param_grads()
            else:
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    param_grads()
                dz.record_stream(side)
                x.data.record_stream(side)
            if x.needs_grad:
                assert N % 32 == 0, "tape.linear dgrad needs out_features % 32 == 0"
                wT = self.wt.get(W.w.data_ptr())  # [K, N]
                if wT is None:
                    wT = ops.transpose2d(W.w)
                Tape.acc(x, ops.linear(dz, wT))
        self.push(bwd)
        return y

    def layernorm(self, x: Var, W: Param, b: Param, eps, out_f32=False):
        y_data, mean, rstd = ops.layernorm(x.data, W.w, b.w, eps, save_stats=self.enabled,
                                           out_dtype=torch.float32 if out_f32 else bf)
        y = Var(y_data)

        def bwd():
            if y.grad is None:
                return
            dy = y.grad if y.grad.dtype == bf else ops.to_bf16(y.grad)
            dx = ops.layernorm_bwd(x.data, W.w, dy, mean, rstd, dweight=W.g, dbias=b.g)
            Tape.acc(x, dx)
        self.push(bwd)
        return y

    def add(self, a: Var, b_: Var):
        y = Var(ops.add(a.data, b_.data))

        def bwd():
            if y.grad is None:
                return
            both = a.needs_grad and b_.needs_grad
            Tape.acc(a, y.grad, shared=both)
            Tape.acc(b_, y.grad, shared=both)
        self.push(bwd)
        return y

    def add_const_rows(self, a: Var, const_rows, period):
        """y[r] = a[r] + const[r % period] (positional encodings: no gradient to the constant)."""
        y = Var(ops.add_bcast_rows(a.data, const_rows, period))

        def bwd():
            if y.grad is not None:
                Tape.acc(a, y.grad)
        self.push(bwd)
        return y

    def small_attn(self, q: Var, k: Var, v: Var, inst, heads, d, Lq, Lk):
        o = Var(ops.small_attn(q.data, k.data, v.data, inst, heads, d, Lq, Lk))

        def bwd():
            if o.grad is None:
                return
            dq, dk, dv = ops.small_attn_bwd(q.data, k.data, v.data, o.data, o.grad, inst, heads, d, Lq, Lk, bf16_grads=True)
            Tape.acc(q, dq)
            Tape.acc(k, dk)
            Tape.acc(v, dv)
        self.push(bwd)
        return o
