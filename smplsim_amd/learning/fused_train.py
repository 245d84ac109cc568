"""The PPO update's network passes on this library's own matrix-core GEMMs (include/smplsim_mlp.h: ss_linear_bf16_train, ss_wgrad_bf16).

What it replaces: autograd over torch.nn.Linear + activation under bf16 autocast (hipBLASLt GEMMs plus separate bias, activation,
cast and transpose launches) in the reference's update_policy / update_value (agents/agent_ppo.py:20-83).  The loss, the optimiser,
the gradient clipping and the RunningNorm stay torch code and untouched; only `y = head(MLP(x))` and its backward are here:

  forward, per hidden layer   h = act(z), g = act'(z)   with z = h_below W^T + b     ONE launch: the result and the activation's derivative from one tile
  head                        y = h W^T + b in fp32 (the inference kernel: the action mean must not be rounded to bf16)
  backward, per layer         dW = dZ^T h_below     -> ss_wgrad_bf16: both operands as they lie ([batch, features] row-major), contraction over their ROWS
                                                      (fragments by ds_read_b64_tr_b16), split along the batch, fp32 partial sums by hardware atomics
                              dZ_below = (dZ W) * g_below   -> ss_linear_bf16_dx on (dZ, W^T): the multiply in its epilogue, and the column sums of
                                                               the fp32 result = db of the layer below from the same launch

No transposed copies of activations or gradients exist: the forward and dX products are bound by the bytes they WRITE (profiles/r06_gemm256.txt: 2.1 TB/s), and
the first version of this file wrote h^T and dZ^T next to h and dZ so that the weight gradient could be the K-contiguous `x W^T` kernel — 1.35 of the 3.4 GB a
pass wrote.  bf16 operands, fp32 accumulation: the precision class of the autocast path it replaces (weights, activations and gradients of activations rounded
to bf16; weight gradients and the head's output fp32).  No CPU path: the package has none.
"""
import torch

from .. import _cabi
from .._lib import lib
from ..batch import _check, _launch_stream, _ptr


def _pad(n, m):
    return (n + m - 1) // m * m


def _linear_train(x, w, bias, mul, y, yt, dact, M, N, K, ldy, ldyt, act, accumulate, stream):
    _check(lib().ss_linear_bf16_train(_ptr(x), _ptr(w), _ptr(bias), _ptr(mul), _ptr(y), _ptr(yt), _ptr(dact), M, N, K, ldy, ldyt, act, int(accumulate), stream))


class _Buffers:
    """Work tensors of one FusedMLPTrain, kept between calls (a pass allocated and zero-filled ~1 GB of activations and transposes: allocator
    round trips and memsets of tensors the kernels overwrite completely).  A forward pass that starts while the previous pass's backward has not run
    (two graphs alive) gets fresh tensors instead."""

    def __init__(self):
        self.t = {}
        self.busy = False

    def get(self, key, shape, dtype, device, fresh, init=None):
        if fresh:
            t = torch.zeros(shape, dtype=dtype, device=device)
            if init is not None:
                init(t)
            return t
        k = (key, tuple(shape), dtype)
        t = self.t.get(k)
        if t is None:
            t = self.t[k] = torch.zeros(shape, dtype=dtype, device=device)
            if init is not None:
                init(t)
        return t


class _FusedMLP(torch.autograd.Function):
    """y = Linear_{L+1}(act(Linear_L(... act(Linear_1(x))))) with x [M, D] fp32; params = W_1, b_1, ..., W_{L+1}, b_{L+1} (fp32, torch.nn.Linear layout)."""

    @staticmethod
    def forward(ctx, x, act, bufs, track, *params):
        dev, st = x.device, _launch_stream(x.device)
        ws, bs = params[0::2], params[1::2]
        nl = len(ws)
        M, D = x.shape
        Mp = _pad(M, 128)                                          # the batch is the K of the weight-gradient products; its pad rows are zero in every
                                                                   # dZ (zero rows of the head's gradient stay zero through (dZ W) * g), so what the
                                                                   # forward pass leaves in the pad rows of h (act(bias)) never reaches a gradient
        kpad = [_pad(w.shape[1], 128) if i == 0 else _pad(w.shape[1], 64) for i, w in enumerate(ws)]   # a layer's input width as a K (the first: a whole number of K-tile pairs for the 256-tile kernel)
        fresh = track and bufs.busy                                # (track: a backward pass will follow and release the tensors)
        if track and not fresh:
            bufs.busy = True
        ctx.bufs = bufs if track and not fresh else None
        bf = torch.bfloat16
        h = bufs.get("x", (Mp, kpad[0]), bf, dev, fresh)           # (pad rows and columns stay zero: only [:M, :D] is ever written)
        h[:M, :D] = x
        hs, gs, wbs = [h], [], []
        # the layers' weights in bf16, all by one multi-tensor copy (they change with every optimiser step; 7 launches a pass otherwise)
        wb_all = [bufs.get(("w", i), (ws[i].shape[0], kpad[i]), bf, dev, fresh) for i in range(nl)]
        torch._foreach_copy_([wb_all[i][:, :ws[i].shape[1]] for i in range(nl)], [w_.detach() for w_ in ws])
        for i in range(nl - 1):
            w = ws[i]
            N = w.shape[0]
            assert N % 64 == 0 and kpad[i + 1] == N, "hidden widths must be multiples of 64"
            wb = wb_all[i]
            y = bufs.get(("h", i), (Mp, N), bf, dev, fresh)
            g = bufs.get(("g", i), (Mp, N), bf, dev, fresh) if track else None   # no backward pass will follow (GAE's value pass): the result alone
            _linear_train(h, wb, bs[i].detach().float().contiguous(), None, y, None, g, Mp, N, kpad[i], N, 0, act, False, st)
            wbs.append(wb); hs.append(y); gs.append(g)
            h = y
        w = ws[-1]
        wb = wb_all[-1]
        wbs.append(wb)
        out = torch.empty(Mp, w.shape[0], dtype=torch.float32, device=dev)
        _check(lib().ss_linear_bf16(_ptr(h), _ptr(wb), _ptr(bs[-1].detach().float().contiguous()), _ptr(out), Mp, w.shape[0], kpad[-1], w.shape[0],
                                    _cabi.ACTIVATIONS["none"], 1, st))
        ctx.act, ctx.M, ctx.Mp, ctx.kpad, ctx.dims = act, M, Mp, kpad, [(w_.shape[0], w_.shape[1]) for w_ in ws]
        ctx.hs, ctx.gs, ctx.wbs = hs, gs, wbs
        return out[:M]

    @staticmethod
    def backward(ctx, grad_out):
        dev, st = grad_out.device, _launch_stream(grad_out.device)
        M, Mp, kpad, dims = ctx.M, ctx.Mp, ctx.kpad, ctx.dims
        nl = len(dims)
        bufs = ctx.bufs if ctx.bufs is not None else _Buffers()
        fresh = ctx.bufs is None
        bf = torch.bfloat16
        none = _cabi.ACTIVATIONS["none"]
        # the head's dZ: the caller's gradient, rounded to bf16, its width padded to a K
        nh = dims[-1][0]
        nhp = _pad(nh, 128)
        dz = bufs.get("dz_head", (Mp, nhp), bf, dev, fresh)
        dz[:M, :nh] = grad_out
        grads = [None] * (2 * nl)
        # the head's bias gradient (its dZ is the caller's tensor): dZ^T 1 by the weight-gradient kernel against a column of ones (torch's column reduction of a
        # [53 248, 69] tensor took 180 us, a matrix-vector product through rocBLAS 175)
        ones = bufs.get("ones", (Mp, 8), bf, dev, fresh, lambda t: t[:, 0].fill_(1.0))
        dbh = torch.zeros(_pad(nh, 8), 8, dtype=torch.float32, device=dev)
        _check(lib().ss_wgrad_bf16(_ptr(dz), _ptr(ones), _ptr(dbh), Mp, _pad(nh, 8), 8, nhp, 8, 8, st))
        db = dbh[:nh, 0]
        # every weight and bias gradient of the pass in ONE zero-filled tensor (the kernels accumulate into them: 13 memsets otherwise).  Not kept between
        # passes: the optimiser holds the views as .grad until the next backward
        no8s = [_pad(d[0], 8) for d in dims]
        sizes = [no8s[i] * kpad[i] for i in range(nl)] + [kpad[i] for i in range(1, nl)]
        offs = [0]
        for z in sizes:
            offs.append(offs[-1] + _pad(z, 64))
        flat = torch.zeros(offs[-1], dtype=torch.float32, device=dev)
        # W^T of every layer but the first (the "W" operand of the dX products), all by one multi-tensor copy
        wts = {i: bufs.get(("wt", i), (kpad[i], nhp if i == nl - 1 else dims[i][0]), bf, dev, fresh) for i in range(1, nl)}
        torch._foreach_copy_([wts[i][:, :dims[i][0]] for i in range(1, nl)], [ctx.wbs[i].t() for i in range(1, nl)])
        for i in range(nl - 1, -1, -1):
            n_out, n_in = dims[i]
            n_outp = dz.shape[1]
            # dW [n_out, kpad_i] = dZ^T h_below with both operands as they lie (ss_wgrad_bf16 contracts over their rows); db = the column sums of dZ
            no8 = no8s[i]
            dw = flat[offs[i]:offs[i] + no8 * kpad[i]].view(no8, kpad[i])
            _check(lib().ss_wgrad_bf16(_ptr(dz), _ptr(ctx.hs[i]), _ptr(dw), Mp, no8, kpad[i], n_outp, kpad[i], kpad[i], st))
            grads[2 * i] = dw[:n_out, :n_in]
            grads[2 * i + 1] = db if db is not None else dz[:, :n_out].sum(0, dtype=torch.float32)
            if i > 0:
                # dZ_below = (dZ W) * act'(z_below): W^T [kpad_i, n_outp] as the kernel's "W", contraction over this layer's outputs; the column sums of
                # the fp32 result (the bias gradient of the layer below) come out of the same launch where the 256 x 256 kernel serves the product
                wt = wts[i]
                nb = kpad[i]
                dzb = bufs.get(("dz", i), (Mp, nb), bf, dev, fresh)
                if Mp >= 2048 and nb >= 256 and n_outp % 128 == 0:
                    db = flat[offs[nl + i - 1]:offs[nl + i - 1] + nb]
                    _check(lib().ss_linear_bf16_dx(_ptr(dz), _ptr(wt), _ptr(ctx.gs[i - 1]), _ptr(dzb), _ptr(db), Mp, nb, n_outp, nb, st))
                else:
                    db = None
                    _linear_train(dz, wt, None, ctx.gs[i - 1], dzb, None, None, Mp, nb, n_outp, nb, 0, none, False, st)
                dz = dzb
        ctx.hs = ctx.gs = ctx.wbs = None
        if ctx.bufs is not None:
            ctx.bufs.busy = False
        return (None, None, None, None, *grads)


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
        self.bufs, self.bufs_nograd = _Buffers(), _Buffers()

    def __call__(self, x):
        params = []
        for l in self.layers:
            params += [l.weight, l.bias]
        track = torch.is_grad_enabled()                           # (inside Function.forward the grad mode is always off)
        return _FusedMLP.apply(x.detach().float(), self.act, self.bufs if track else self.bufs_nograd, track, *params)
