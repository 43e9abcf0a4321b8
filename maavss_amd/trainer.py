"""Training step of the hot path: loss, backward, gradient all-reduce, Adam.

Mirrors the inner loop of the reference trainer (train_avse_frames.py:150-181):
    yh_stft, yh_attn, latent = model(x_stft, x_attn)
    loss = (mse(yh_stft, y_stft) + loss_coeff * mse(yh_attn, y_attn)) / num_seq ; loss.backward()
    optimizer.step(); optimizer.zero_grad()
with every tensor operation in libmaavss_hip.so.  Two entry levels:

* `FusedAdam`   -- drop-in for `torch.optim.Adam(model.parameters(), lr=...)` (train_avse_frames.py:92): all
                   parameters / gradients / moments live in flat f32 buffers and one HIP kernel updates them.
* `TrainStep`   -- the whole step without autograd: engine forward, fused MSE pair + gradients, hand-written
                   backward straight into the flat gradient buffer, RCCL all-reduce of the gradients
                   (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI) overlapped with
                   the encoder backward, fused Adam.  This is what bench.py times.

Data parallelism: clips are independent, the batch dimension is sharded across ranks, weights
are replicated; the only collective is the gradient sum.  BatchNorm uses per-rank batch statistics (as
torch DDP does without SyncBatchNorm) -- documented in DESIGN.md.
"""
import torch

from . import ops

_FUSION_PREFIXES = ("lstm.", "fc1.", "fc2.", "a_fc1.", "v_fc1.")


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatParams:
    """Re-homes the parameters of a module in one flat f32 buffer (+ a flat gradient buffer).

    Layout: [fusion segment: lstm, fc1, fc2, heads | encoder segment: everything else], each tensor
    64-element aligned.  The fusion segment holds 98 % of the bytes and its gradients are complete first
    in the backward pass, so it is all-reduced while the conv backward still runs."""

    def __init__(self, model):
        named = [(n, p) for n, p in model.named_parameters() if not n.startswith("stft_autoencoder.")]
        fusion = [(n, p) for n, p in named if n.startswith(_FUSION_PREFIXES)]
        other = [(n, p) for n, p in named if not n.startswith(_FUSION_PREFIXES)]
        self.names, self.offsets, self.shapes = [], {}, {}
        off = 0
        for n, p in fusion + other:
            self.names.append(n)
            self.offsets[n] = off
            self.shapes[n] = tuple(p.shape)
            off = _align(off + p.numel())
            if n == fusion[-1][0]:
                self.fusion_end = off
        self.total = off
        dev = named[0][1].device
        self.params = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.param_views, self.grad_views = {}, {}
        for n, p in fusion + other:
            o, k = self.offsets[n], p.numel()
            pv = self.params[o:o + k].view(p.shape)
            pv.copy_(p.data)
            p.data = pv
            gv = self.grads[o:o + k].view(p.shape)
            p.grad = gv
            self.param_views[n], self.grad_views[n] = pv, gv


class FusedAdam:
    """torch.optim.Adam semantics (betas .9/.999, eps 1e-8, no weight decay, no amsgrad) in one HIP kernel."""

    def __init__(self, model_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.flat = model_or_flat if isinstance(model_or_flat, FlatParams) else FlatParams(model_or_flat)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = torch.zeros_like(self.flat.params)
        self.exp_avg_sq = torch.zeros_like(self.flat.params)
        self.step_count = 0

    def step(self, grad_scale=1.0):
        self.step_count += 1
        ops.adam_step(self.flat.params, self.flat.grads, self.exp_avg, self.exp_avg_sq, self.lr, self.step_count,
                      self.betas, self.eps, grad_scale)

    def zero_grad(self, set_to_none=False):
        self.flat.grads.zero_()        # memset; the flat views stay attached to p.grad

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.lr, "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class GradSync:
    """Sum-all-reduce of the flat gradient buffer across data-parallel ranks (K18).

    Works on any torch.distributed backend ("nccl" = RCCL on ROCm for the GPUs, "gloo" in CPU tests).
    `start_fusion()` launches the large segment asynchronously as soon as its gradients exist;
    `finish()` reduces the encoder segment and waits for both.  With world_size 1 it is a no-op."""

    def __init__(self, grads, fusion_end, process_group=None):
        import torch.distributed as dist
        self.dist = dist
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.group = process_group
        self.grads, self.fusion_end = grads, fusion_end
        self._pending = None

    def start_fusion(self):
        if self.enabled:
            self._pending = self.dist.all_reduce(self.grads[:self.fusion_end], op=self.dist.ReduceOp.SUM,
                                                 group=self.group, async_op=True)

    def finish(self):
        if not self.enabled:
            return
        if self._pending is None:
            self.start_fusion()
        tail = self.dist.all_reduce(self.grads[self.fusion_end:], op=self.dist.ReduceOp.SUM, group=self.group,
                                    async_op=True)
        self._pending.wait()
        tail.wait()
        self._pending = None


def shard_batch(global_batch, rank, world):
    """Contiguous shard [lo, hi) of the clip batch owned by `rank` (clips are independent units)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class TrainStep:
    """One optimizer step of the fusion network, autograd-free (see module docstring)."""

    def __init__(self, model, lr=1e-5, loss_coeff=0.001, num_seq=1, betas=(0.9, 0.999), eps=1e-8,
                 process_group=None):
        self.model = model
        self.flat = FlatParams(model)
        self.opt = FusedAdam(self.flat, lr, betas, eps)
        self.loss_coeff, self.num_seq = loss_coeff, num_seq
        self.sync = GradSync(self.flat.grads, self.flat.fusion_end, process_group)
        self.need = {n: bool(p.requires_grad) for n, p in model.named_parameters()}
        self.losses = None

    def __call__(self, x_a, x_v, y_a, y_v, optimizer_step=True):
        m = self.model
        (a, v, fused), sv = m._engine_forward(x_a, x_v, train=True)
        self.losses, d_a, d_v = ops.mse_pair(a, y_a.contiguous(), v, y_v.contiguous(), self.loss_coeff, self.num_seq)
        m._engine_backward(sv, d_a, d_v, None, self.need, grads=self.flat.grad_views, accumulate=False,
                           on_fusion_done=self.sync.start_fusion)
        self.sync.finish()
        if optimizer_step:
            self.opt.step(grad_scale=1.0 / self.sync.world)
        return self.losses
