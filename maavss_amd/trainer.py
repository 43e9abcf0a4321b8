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
are replicated (rank 0's parameters and BatchNorm buffers are broadcast when a TrainStep is built, as DDP
does); the only mandatory collective is the gradient sum.  BatchNorm uses per-rank batch statistics by default
(as torch DDP does without SyncBatchNorm); `TrainStep(sync_bn=True)` all-reduces the per-channel sums instead,
which reproduces the single-device reference on the concatenated batch -- documented in DESIGN.md.
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
        self.torch_order = [n for n, _ in named]     # index i of torch.optim.Adam(model.parameters()).state_dict()
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
        self.param_views, self.grad_views, self.tensors = {}, {}, {}
        for n, p in fusion + other:
            o, k = self.offsets[n], p.numel()
            pv = self.params[o:o + k].view(p.shape)
            pv.copy_(p.data)
            p.data = pv
            gv = self.grads[o:o + k].view(p.shape)
            p.grad = gv
            self.param_views[n], self.grad_views[n], self.tensors[n] = pv, gv, p
        # names whose gradient was produced since the last zero_grad(): Adam skips the others, like torch.optim.Adam skips
        # parameters whose .grad is None (e.g. stft_decoder.* under forward(), frozen sub-networks)
        self.touched = set()
        model._maavss_flat = self

    def mark(self, names):
        self.touched.update(names)

    def check_links(self):
        """The model's parameters must still be the flat views (model.to() / .float() / zero_grad(set_to_none=True) after
        construction re-home them).  A detached .grad is folded back in (None = no gradient this step, a foreign tensor
        is copied); a re-homed parameter cannot be repaired silently and raises."""
        from ._lib import MaavssError
        for n, p in self.tensors.items():
            if p.data_ptr() != self.param_views[n].data_ptr():
                raise MaavssError(f"parameter {n} no longer lives in the flat buffer (model.to()/.float() after the optimizer "
                                  f"was built?): build FusedAdam / TrainStep after moving the model")
            gv = self.grad_views[n]
            if p.grad is None:
                self.touched.discard(n)
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv
                self.touched.add(n)


class FusedAdam:
    """torch.optim.Adam semantics (betas .9/.999, eps 1e-8, no weight decay, no amsgrad) in HIP launches over the flat
    buffers.  Like torch, a parameter is only stepped when a backward pass produced a gradient for it since the last
    zero_grad() (frozen sub-networks and stft_decoder.* under forward() are left alone, keep their moments and their
    own step count); contiguous runs of stepped parameters share one launch -- one launch in the usual case."""

    def __init__(self, model_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.flat = model_or_flat if isinstance(model_or_flat, FlatParams) else FlatParams(model_or_flat)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.exp_avg = torch.zeros_like(self.flat.params)
        self.exp_avg_sq = torch.zeros_like(self.flat.params)
        self.steps = {n: 0 for n in self.flat.names}          # per-parameter step count (torch keeps one per parameter)

    @property
    def step_count(self):
        return max(self.steps.values())

    @step_count.setter
    def step_count(self, n):
        self.steps = {k: int(n) for k in self.steps}

    def _runs(self):
        """[(lo, hi, step, names)]: maximal runs of touched parameters with equal step counts, in buffer order.  Pure planning:
        the per-parameter step counts are committed by step() once the launches are issued."""
        f, runs = self.flat, []
        for n in f.names:
            if n not in f.touched or not f.tensors[n].requires_grad:
                continue
            lo, st = f.offsets[n], self.steps[n] + 1
            k = 1
            for d in f.shapes[n]:
                k *= d
            hi = _align(lo + k)
            if runs and runs[-1][1] == lo and runs[-1][2] == st:
                runs[-1][1] = hi
                runs[-1][3].append(n)
            else:
                runs.append([lo, hi, st, [n]])
        return [tuple(r) for r in runs]

    def _commit(self, runs):
        for _, _, st, names in runs:
            for n in names:
                self.steps[n] = st

    def step(self, grad_scale=1.0):
        self.flat.check_links()
        runs = self._runs()
        for lo, hi, st, _ in runs:
            hi = min(hi, self.flat.total)
            ops.adam_step(self.flat.params[lo:hi], self.flat.grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                          self.lr, st, self.betas, self.eps, grad_scale)
        self._commit(runs)

    def zero_grad(self, set_to_none=False):
        """memset of the flat gradient buffer; the views stay attached to p.grad whatever `set_to_none` says (gradients
        are accumulated in place).  Parameters count as gradient-less until the next backward produces one."""
        self.flat.check_links()
        self.flat.grads.zero_()
        self.flat.touched.clear()

    def state_dict(self):
        """torch.optim.Adam's layout (what utilities.save_checkpoint stores as 'optimizer_state_dict',
        utilities.py:168-175): per-parameter step / exp_avg / exp_avg_sq keyed by the position of the parameter in
        model.parameters(), plus one param_group.  Parameters that were never stepped have no entry, as in torch."""
        f = self.flat
        state = {}
        for i, n in enumerate(f.torch_order):
            if self.steps[n] == 0:
                continue
            o, k = f.offsets[n], 1
            for d in f.shapes[n]:
                k *= d
            state[i] = {"step": torch.tensor(float(self.steps[n])),
                        "exp_avg": self.exp_avg[o:o + k].view(f.shapes[n]).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + k].view(f.shapes[n]).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(f.torch_order)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the dict above and a reference checkpoint's torch.optim.Adam state (parameters the reference never
        stepped -- stft_decoder.* under forward() -- have no entry there: their moments and step count stay zero)."""
        f = self.flat
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(f.torch_order):
            raise ValueError(f"optimizer state has {sum(len(g['params']) for g in groups)} parameters in {len(groups)} group(s); "
                             f"this model has {len(f.torch_order)} in one")
        self.lr = float(groups[0]["lr"])
        self.betas = tuple(float(b) for b in groups[0]["betas"])
        self.eps = float(groups[0]["eps"])
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.steps = {n: 0 for n in f.names}
        for i, n in enumerate(f.torch_order):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            o, k = f.offsets[n], st["exp_avg"].numel()
            if tuple(st["exp_avg"].shape) != f.shapes[n]:
                raise ValueError(f"optimizer state of parameter {i} ({n}) has shape {tuple(st['exp_avg'].shape)}, expected {f.shapes[n]}")
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.steps[n] = int(float(st["step"]))


class GradSync:
    """Sum-all-reduce of the flat gradient buffer across data-parallel ranks (K18).

    Works on any torch.distributed backend ("nccl" = RCCL on ROCm for the GPUs, "gloo" in CPU tests); with world_size 1
    every method is a no-op.  The buffer is reduced in BUCKETS, in the order the backward pass completes them
    ("bucketed in reverse-layer order to overlap with conv3d backward"): with a FlatParams layout the fusion
    segment is one bucket per top-level module -- heads (a_fc1, v_fc1), fc2, fc1, lstm -- each launched asynchronously by
    `grad_ready(name)` the moment its last gradient has been enqueued, so the first all-reduce (the 26 M-float v_fc1 head)
    starts after the heads' weight gradients, not after the whole fusion segment; the encoder segment follows in `finish()`,
    which also waits for everything.  Only parameters that receive a gradient this step take part (`begin(expected)`, the
    same set on every rank): a bucket without any is skipped, stale regions inside a reduced bucket are zeroed first -- a
    frozen parameter's old gradient is never summed over the ranks step after step.
    `start_fusion()` (round-1/2 interface) launches whatever is left of the fusion segment in one go."""

    def __init__(self, grads, fusion_end, process_group=None, flat=None, wire_dtype=None):
        import torch.distributed as dist
        if wire_dtype not in (None, "f32", "bf16"):
            raise ValueError("wire_dtype must be None / 'f32' or 'bf16'")
        # "bf16": every bucket is rounded to bf16 for the collective (half the xGMI bytes; the sum over the ranks is then a bf16 sum)
        # and widened back into the f32 master buffer in finish().  Device tensors only: the conversion is a HIP kernel.
        self.wire_bf16 = wire_dtype == "bf16"
        self._wire = None
        self.dist = dist
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.group = process_group
        self.grads, self.fusion_end = grads, fusion_end
        # buckets: [lo, hi) of the flat buffer + the (name, lo, hi) of the parameters inside, fusion segment first
        self.buckets = []
        if flat is not None:
            for n in flat.names:
                lo = flat.offsets[n]
                k = 1
                for d in flat.shapes[n]:
                    k *= d
                hi = _align(lo + k)
                key = "encoders" if lo >= fusion_end else n.split(".")[0]
                if self.buckets and self.buckets[-1]["key"] == key:
                    self.buckets[-1]["hi"] = hi
                    self.buckets[-1]["params"].append((n, lo, hi))
                else:
                    self.buckets.append(dict(key=key, lo=lo, hi=hi, params=[(n, lo, hi)]))
            self.buckets[-1]["hi"] = min(self.buckets[-1]["hi"], grads.numel())
        else:
            self.buckets = [dict(key="fusion", lo=0, hi=fusion_end, params=[]),
                            dict(key="encoders", lo=fusion_end, hi=grads.numel(), params=[])]
        self._bucket_of = {n: i for i, b in enumerate(self.buckets) for n, _, _ in b["params"]}
        self._expected, self._ready, self._launched, self._pending = None, set(), set(), []
        self.launch_log = []          # keys of the buckets in launch order, per step (tests, tracing)

    def broadcast(self, tensors, src=0):
        """Make the replicas identical: rank `src`'s values everywhere (parameters, BatchNorm buffers), as DDP does at
        construction.  No-op with one rank."""
        if self.enabled:
            for t in tensors:
                self.dist.broadcast(t, src=src, group=self.group)

    def sum_(self, t):
        """in-place sum of a small tensor over the ranks (global-batch BatchNorm sums)"""
        if self.enabled:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def begin(self, expected=None):
        """Start of a backward pass whose gradients will be reduced: `expected` = names that will receive a gradient
        (None = every parameter).  Must be the same on every rank (it follows requires_grad, which is)."""
        self._expected = None if expected is None else set(expected)
        self._ready, self._launched, self._pending, self.launch_log = set(), set(), [], []

    def _wanted(self, b):
        return [p for p in b["params"] if self._expected is None or p[0] in self._expected]

    def _launch(self, i):
        if i in self._launched:
            return
        self._launched.add(i)
        b = self.buckets[i]
        if not self.enabled or b["hi"] <= b["lo"]:
            return
        if b["params"]:
            wanted = self._wanted(b)
            if not wanted:
                return                                   # nothing in this bucket received a gradient: not reduced at all
            if len(wanted) < len(b["params"]):
                keep = {p[0] for p in wanted}
                for n, lo, hi in b["params"]:
                    if n not in keep:
                        self.grads[lo:min(hi, b["hi"])].zero_()      # stale gradient of a frozen parameter
        self.launch_log.append(b["key"])
        if self.wire_bf16:
            if self._wire is None:
                self._wire = torch.empty(self.grads.numel(), device=self.grads.device, dtype=torch.bfloat16)
            lo, hi = b["lo"], (b["hi"] + 7) // 8 * 8
            hi = min(hi, self.grads.numel() // 8 * 8)
            ops.f32_to_bf16(self.grads[lo:hi], self._wire[lo:hi])
            work = self.dist.all_reduce(self._wire[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, lo, hi))
            return
        self._pending.append((self.dist.all_reduce(self.grads[b["lo"]:b["hi"]], op=self.dist.ReduceOp.SUM, group=self.group,
                                                   async_op=True), None, None))

    def grad_ready(self, name):
        """The gradient of `name` has been enqueued on the current stream; launches its bucket when that completes it."""
        i = self._bucket_of.get(name)
        if i is None or self.buckets[i]["key"] == "encoders":
            return
        self._ready.add(name)
        if all(p[0] in self._ready for p in self._wanted(self.buckets[i])):
            self._launch(i)

    def start_fusion(self):
        for i, b in enumerate(self.buckets):
            if b["key"] != "encoders":
                self._launch(i)

    def finish(self):
        self.start_fusion()
        for i, b in enumerate(self.buckets):
            self._launch(i)
        for w, lo, hi in self._pending:
            w.wait()
            if lo is not None:
                ops.bf16_to_f32(self._wire[lo:hi], self.grads[lo:hi])
        self._pending = []
        self._expected, self._ready, self._launched = None, set(), set()


def shard_batch(global_batch, rank, world):
    """Contiguous shard [lo, hi) of the clip batch owned by `rank` (clips are independent units)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class TrainStep:
    """One optimizer step of the fusion network, autograd-free (see module docstring)."""

    def __init__(self, model, lr=1e-5, loss_coeff=0.001, num_seq=1, betas=(0.9, 0.999), eps=1e-8,
                 process_group=None, sync_bn=False, grad_wire_dtype=None):
        self.model = model
        self.flat = FlatParams(model)
        self.opt = FusedAdam(self.flat, lr, betas, eps)
        self.loss_coeff, self.num_seq = loss_coeff, num_seq
        self.sync = GradSync(self.flat.grads, self.flat.fusion_end, process_group, flat=self.flat, wire_dtype=grad_wire_dtype)
        # replicas start identical: rank 0's weights (one flat buffer) and BatchNorm buffers, like DDP's constructor
        self.sync.broadcast([self.flat.params] + [b for _, b in model.named_buffers()])
        if sync_bn and self.sync.enabled:
            model.set_bn_sync(self.sync.sum_)
        self.losses = None

    def __call__(self, x_a, x_v, y_a, y_v, optimizer_step=True, accumulate=False, last=True):
        """One window: forward, loss / num_seq, backward.  `accumulate` adds into the flat gradient buffer instead of
        overwriting it; the gradient all-reduce and Adam run only when `last` (sliding_window_step drives both).
        requires_grad is read at every call (toggle_* between steps take effect, as with autograd)."""
        m = self.model
        need = {n: bool(p.requires_grad) for n, p in m.named_parameters()}
        if not accumulate:
            # gradients this step does not produce must not carry over (the engine overwrites the ones it does produce)
            self.flat.touched.clear()
        (a, v, fused), sv = m._engine_forward(x_a, x_v, train=True)
        self.outputs = (a, v, fused)
        self.losses, d_a, d_v = ops.mse_pair(a, y_a.contiguous(), v, y_v.contiguous(), self.loss_coeff, self.num_seq)
        if last:
            self.sync.begin(n for n in m._param_names if need.get(n, False))
        out = m._engine_backward(sv, d_a, d_v, None, need, grads=self.flat.grad_views, accumulate=accumulate,
                                 on_fusion_done=self.sync.start_fusion if last else None,
                                 on_grad=self.sync.grad_ready if last else None)
        self.flat.mark(out.keys())
        if last:
            self.sync.finish()
            if optimizer_step:
                self.opt.step(grad_scale=1.0 / self.sync.world)
        return self.losses

    def sliding_window_step(self, x_stft, y_stft, x_attn, y_attn, num_frames, hops_per_frame, collect=False):
        """The reference's optimizer step over `num_seq` overlapping windows (train_avse_frames.py:143-181):
        window j takes frames [j, j + num_frames) of x_attn [B,1,T,H,W] and the matching STFT frames of x_stft
        [B,2,T_a,F]; its targets are attention frame j + idx_mid of y_attn and STFT frames of the same video frame
        of y_stft, idx_mid = (num_seq - 1) // 2 (:105).  Every window contributes loss / num_seq to the gradients
        (accumulated in the flat buffer); ONE gradient all-reduce and ONE Adam step follow.  Returns the last
        window's (a_loss, v_loss, loss) like the reference logs (:183-188) and, with `collect`, the stitched
        outputs (output_stft [B,2,hpf*num_seq,F], output_attn [B,1,num_seq,H,W]) of its callback (:150-176)."""
        hpf, ns = hops_per_frame, self.num_seq
        mid = (ns - 1) // 2
        assert x_attn.shape[2] >= num_frames + ns - 1 and x_stft.shape[2] >= hpf * (num_frames + ns - 1)
        outs_a, outs_v = [], []
        for j in range(ns):
            xa = x_stft[:, :, hpf * j:hpf * (j + num_frames), :]
            ya = y_stft[:, :, hpf * (j + mid):hpf * (j + mid + 1), :]
            xv = x_attn[:, :, j:j + num_frames]
            yv = y_attn[:, :, j + mid]
            self(xa, xv, ya, yv, accumulate=j > 0, last=j == ns - 1)
            if collect:
                outs_a.append(self.outputs[0])
                outs_v.append(self.outputs[1])
        if collect:
            return self.losses, torch.cat(outs_a, dim=2), torch.stack(outs_v, dim=2)
        return self.losses
