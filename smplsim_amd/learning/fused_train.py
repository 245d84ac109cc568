"""The PPO update's network passes on this library's own matrix-core GEMM (include/smplsim_mlp.h: ss_linear_bf16_train).

What it replaces: autograd over torch.nn.Linear + activation under bf16 autocast (hipBLASLt GEMMs plus separate bias, activation,
cast and transpose launches) in the reference's update_policy / update_value (agents/agent_ppo.py:20-83).  The loss, the optimiser,
the gradient clipping and the RunningNorm stay torch code and untouched; only `y = head(MLP(x))` and its backward are here:

  forward, per hidden layer   h, h^T, g = act(z), act(z)^T, act'(z)   with z = h_below W^T + b     ONE launch: the result, its transpose and
                                                                                                   the activation's derivative from one tile
  head                        y = h W^T + b in fp32 (the inference kernel: the action mean must not be rounded to bf16)
  backward, per layer         dW = dZ^T h_below     -> the same kernel on (dZ^T, h_below^T), contraction over the batch, split along K,
                                                      fp32 partial sums by hardware atomics
                              db = one more output column of the same product (a row of ones behind h_below^T)
                              dZ_below = (dZ W) * g_below   -> the same kernel on (dZ, W^T) with the multiply in its epilogue; writes dZ_below
                                                               and dZ_below^T

Every product is the K-contiguous `x W^T` form because each tensor that a later product contracts over its rows was written transposed by
the launch that produced it.  bf16 operands, fp32 accumulation: the precision class of the autocast path it replaces (weights, activations and
gradients of activations rounded to bf16; weight gradients and the head's output fp32).  No CPU path: the package has none.
"""
import torch

from .. import _cabi
from .._lib import lib
from ..batch import _check, _launch_stream, _ptr


def _pad(n, m):
    return (n + m - 1) // m * m


def _linear_train(x, w, bias, mul, y, yt, dact, M, N, K, ldy, ldyt, act, accumulate, stream):
    _check(lib().ss_linear_bf16_train(_ptr(x), _ptr(w), _ptr(bias), _ptr(mul), _ptr(y), _ptr(yt), _ptr(dact), M, N, K, ldy, ldyt, act, int(accumulate), stream))


class _FusedMLP(torch.autograd.Function):
    """y = Linear_{L+1}(act(Linear_L(... act(Linear_1(x))))) with x [M, D] fp32; params = W_1, b_1, ..., W_{L+1}, b_{L+1} (fp32, torch.nn.Linear layout)."""

    @staticmethod
    def forward(ctx, x, act, *params):
        dev, st = x.device, _launch_stream(x.device)
        ws, bs = params[0::2], params[1::2]
        nl = len(ws)
        M, D = x.shape
        Mp = _pad(M, 64)                                           # the batch is the K of the weight-gradient products
        kpad = [_pad(w.shape[1], 64) for w in ws]                 # a layer's input width as a K
        bf = dict(dtype=torch.bfloat16, device=dev)
        h = torch.zeros(Mp, kpad[0], **bf)
        h[:M, :D] = x

        def with_ones(n):                                          # a layer's transposed input + 64 rows, the first of them ones: as the "W" of the
            t = torch.zeros(n + 64, Mp, **bf)                      # weight-gradient product its extra output column is the bias gradient
            t[n, :M] = 1.0                                         # (sum of dZ over the batch: no separate reduction launches)
            return t

        t0 = with_ones(kpad[0])
        t0[:kpad[0]] = h.t()
        hts, gs, wbs = [t0], [], []
        for i in range(nl - 1):
            w = ws[i]
            N = w.shape[0]
            assert N % 64 == 0 and kpad[i + 1] == N, "hidden widths must be multiples of 64"
            wb = torch.zeros(N, kpad[i], **bf)
            wb[:, :w.shape[1]] = w
            y, yt, g = torch.empty(Mp, N, **bf), with_ones(N), torch.empty(Mp, N, **bf)
            _linear_train(h, wb, bs[i].detach().float().contiguous(), None, y, yt, g, Mp, N, kpad[i], N, Mp, act, False, st)
            wbs.append(wb); hts.append(yt); gs.append(g)
            h = y
        w = ws[-1]
        wb = torch.zeros(w.shape[0], kpad[-1], **bf)
        wb[:, :w.shape[1]] = w
        wbs.append(wb)
        out = torch.empty(Mp, w.shape[0], dtype=torch.float32, device=dev)
        _check(lib().ss_linear_bf16(_ptr(h), _ptr(wb), _ptr(bs[-1].detach().float().contiguous()), _ptr(out), Mp, w.shape[0], kpad[-1], w.shape[0],
                                    _cabi.ACTIVATIONS["none"], 1, st))
        ctx.act, ctx.M, ctx.Mp, ctx.kpad, ctx.dims = act, M, Mp, kpad, [(w_.shape[0], w_.shape[1]) for w_ in ws]
        ctx.hts, ctx.gs, ctx.wbs = hts, gs, wbs
        return out[:M]

    @staticmethod
    def backward(ctx, grad_out):
        dev, st = grad_out.device, _launch_stream(grad_out.device)
        M, Mp, kpad, dims = ctx.M, ctx.Mp, ctx.kpad, ctx.dims
        nl = len(dims)
        bf = dict(dtype=torch.bfloat16, device=dev)
        none = _cabi.ACTIVATIONS["none"]
        # the head's dZ: the caller's gradient, rounded to bf16, its width padded to a K
        nh = dims[-1][0]
        nhp = _pad(nh, 64)
        dz = torch.zeros(Mp, nhp, **bf)
        dz[:M, :nh] = grad_out
        dzt = dz.t().contiguous()
        grads = [None] * (2 * nl)
        for i in range(nl - 1, -1, -1):
            n_out, n_in = dims[i]
            n_outp = dz.shape[1]
            # dW [n_out, kpad_i] = dZ^T h_below: rows of dZ^T are output features, the contraction is the batch
            dw = torch.zeros(n_out, kpad[i] + 64, dtype=torch.float32, device=dev)
            _linear_train(dzt, ctx.hts[i], None, None, dw, None, None, n_out, kpad[i] + 64, Mp, kpad[i] + 64, 0, none, True, st)
            grads[2 * i] = dw[:, :n_in]
            grads[2 * i + 1] = dw[:, kpad[i]]                      # the ones row's column: sum of dZ over the batch
            if i > 0:
                # dZ_below = (dZ W) * act'(z_below): W^T [kpad_i, n_outp] as the kernel's "W", contraction over this layer's outputs
                wt = torch.zeros(kpad[i], n_outp, **bf)
                wt[:, :n_out] = ctx.wbs[i].t()
                nb = kpad[i]
                dzb, dzbt = torch.empty(Mp, nb, **bf), torch.empty(nb, Mp, **bf)
                _linear_train(dz, wt, None, ctx.gs[i - 1], dzb, dzbt, None, Mp, nb, n_outp, nb, Mp, none, False, st)
                dz, dzt = dzb, dzbt
        ctx.hts = ctx.gs = ctx.wbs = None
        return (None, None, *grads)


class FusedMLPTrain:
    """Callable over an existing stack of torch.nn.Linear layers (the hidden ones followed by `act`, then the head): differentiable with
    respect to the layers' parameters, not to the input (the update's inputs are rollout states)."""

    def __init__(self, hidden_layers, head, activation_name):
        if activation_name not in _cabi.ACTIVATIONS or activation_name == "none":
            raise ValueError(f"activation {activation_name!r} has no fused epilogue (silu, tanh, relu)")
        self.layers = list(hidden_layers) + [head]
        self.act = _cabi.ACTIVATIONS[activation_name]
        dev = self.layers[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("FusedMLPTrain needs the networks on a GPU (there is no CPU path)")

    def __call__(self, x):
        params = []
        for l in self.layers:
            params += [l.weight, l.bias]
        return _FusedMLP.apply(x.detach().float(), self.act, *params)
