"""CPU (gloo, world_size 2) tests of the data-parallel plumbing: batch sharding and the gradient all-reduce
(GradSync) that TrainStep runs over RCCL on the GPUs.  The collective code path is backend-agnostic, so gloo
exercises exactly what "nccl" (= RCCL) runs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maavss_amd.trainer import GradSync, shard_batch
    n, fusion_end = 1000, 640
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    sync = GradSync(g, fusion_end)
    assert sync.enabled and sync.world == world
    sync.start_fusion()              # large segment first (overlaps the encoder backward on the GPU)
    g[fusion_end:] += 0.0            # "encoder backward still running"
    sync.finish()
    want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(g, want), (rank, (g - want).abs().max())
    # a second step re-uses the object
    g.copy_(torch.ones(n) * (rank + 2))
    sync.finish()
    assert torch.equal(g, torch.ones(n) * sum(r + 2 for r in range(world)))
    lo, hi = shard_batch(37, rank, world)
    torch.save({"lo": lo, "hi": hi}, os.path.join(out_dir, f"shard_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_and_sharding_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    shards = [torch.load(os.path.join(tmp_path, f"shard_{r}.pt"), weights_only=True) for r in range(world)]
    assert shards[0]["lo"] == 0 and shards[-1]["hi"] == 37
    assert all(shards[r]["hi"] == shards[r + 1]["lo"] for r in range(world - 1))
    sizes = [s["hi"] - s["lo"] for s in shards]
    assert max(sizes) - min(sizes) <= 1


class _FakeFlat:
    """the part of trainer.FlatParams GradSync reads: names in buffer order, offsets, shapes"""
    def __init__(self):
        self.names = ["lstm.w_ih", "lstm.w_hh", "fc1.weight", "fc2.weight", "a_fc1.0.weight", "v_fc1.0.weight", "enc.0.weight", "enc.1.bias"]
        sizes = [64, 64, 256, 128, 64, 192, 100, 28]
        self.offsets, self.shapes, off = {}, {}, 0
        for n, k in zip(self.names, sizes):
            self.offsets[n], self.shapes[n] = off, (k,)
            off = (off + k + 63) // 64 * 64
        self.total, self.fusion_end = off, self.offsets["enc.0.weight"]


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maavss_amd.trainer import GradSync
    flat = _FakeFlat()
    g = torch.zeros(flat.total)
    sync = GradSync(g, flat.fusion_end, flat=flat)
    assert [b["key"] for b in sync.buckets] == ["lstm", "fc1", "fc2", "a_fc1", "v_fc1", "encoders"]
    tot = float(sum(r + 1 for r in range(world)))
    # step 1: everything receives a gradient, in backward order; each bucket goes out the moment it is complete
    g.fill_(rank + 1.0)
    sync.begin(flat.names)
    for n in ("a_fc1.0.weight", "v_fc1.0.weight", "fc2.weight", "fc1.weight", "lstm.w_ih"):
        sync.grad_ready(n)
    assert sync.launch_log == ["a_fc1", "v_fc1", "fc2", "fc1"]          # lstm waits for its second weight
    sync.grad_ready("lstm.w_hh")
    assert sync.launch_log[-1] == "lstm"
    sync.finish()
    assert sync.launch_log[-1] == "encoders" and torch.equal(g, torch.full_like(g, tot))
    # step 2: the encoders and fc1 are frozen.  Their stale gradients (here: 7) must neither be summed nor be touched where the
    # whole bucket is skipped; a frozen parameter INSIDE a reduced bucket (lstm.w_hh) is zeroed before the reduction
    g.fill_(rank + 1.0)
    lo, hi = flat.offsets["enc.0.weight"], flat.total
    g[lo:hi] = 7.0
    g[flat.offsets["fc1.weight"]:flat.offsets["fc2.weight"]] = 7.0
    g[flat.offsets["lstm.w_hh"]:flat.offsets["fc1.weight"]] = 7.0
    expected = ["lstm.w_ih", "fc2.weight", "a_fc1.0.weight", "v_fc1.0.weight"]
    sync.begin(expected)
    for n in ("a_fc1.0.weight", "v_fc1.0.weight", "fc2.weight", "lstm.w_ih"):
        sync.grad_ready(n)
    sync.finish()
    assert torch.equal(g[lo:hi], torch.full((hi - lo,), 7.0))                                  # skipped bucket: untouched
    assert torch.equal(g[flat.offsets["fc1.weight"]:flat.offsets["fc2.weight"]], torch.full((256,), 7.0))
    assert torch.equal(g[flat.offsets["lstm.w_hh"]:flat.offsets["fc1.weight"]], torch.zeros(64))   # zeroed, then summed: 0
    for n in expected:
        o = flat.offsets[n]
        assert torch.equal(g[o:o + flat.shapes[n][0]], torch.full((flat.shapes[n][0],), tot)), n
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_buckets_follow_the_backward_order_and_skip_frozen_parameters_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)


def test_shard_batch_covers_everything():
    from maavss_amd.trainer import shard_batch
    for world in (1, 2, 4, 8):
        for batch in (8, 32, 37, 256):
            spans = [shard_batch(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_gradsync_noop_single_process():
    from maavss_amd.trainer import GradSync
    g = torch.ones(10)
    s = GradSync(g, 5)
    assert not s.enabled and s.world == 1
    s.start_fusion()
    s.finish()
    assert torch.equal(g, torch.ones(10))


def _replica_worker(rank, world, port, out_dir):
    """TrainStep's constructor must leave every rank with rank 0's parameters and BatchNorm buffers (the
    replicas used to start from different seeds and were never synchronised)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import maavss_amd
    torch.manual_seed(100 + rank)                      # deliberately different initialisations
    model = maavss_amd.AV_Fusion_Model_Frames([2, 2, 64, 129], [2, 1, 8, 128, 128], 8)
    with torch.no_grad():
        for _, b in model.named_buffers():
            if b.dtype.is_floating_point:
                b.add_(float(rank))
    before = model.fc2.weight.detach().clone()
    step = maavss_amd.TrainStep(model, lr=1e-3, sync_bn=True)      # CPU tensors: construction only, no kernel runs
    assert step.sync.enabled and model._bn_sync is not None
    if rank > 0:
        assert not torch.equal(before, model.fc2.weight.detach()), "rank>0 kept its own initialisation"
    # the model's parameters are still views of the flat buffer after the broadcast
    for n, p in model.named_parameters():
        if n in step.flat.param_views:
            assert p.data_ptr() == step.flat.param_views[n].data_ptr(), n
    t = torch.zeros(3, dtype=torch.float64) + rank + 1
    model._bn_sync(t)                                   # the reduce function SyncBN hands to the engine
    assert torch.equal(t, torch.full((3,), float(sum(range(1, world + 1))), dtype=torch.float64))
    torch.save({"params": step.flat.params.clone(), "buffers": {k: v.clone() for k, v in model.named_buffers()}},
               os.path.join(out_dir, f"replica_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_trainstep_broadcasts_parameters_and_buffers_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_replica_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    reps = [torch.load(os.path.join(tmp_path, f"replica_{r}.pt"), weights_only=True) for r in range(world)]
    assert torch.equal(reps[0]["params"], reps[1]["params"])
    assert reps[0]["params"].abs().sum() > 0
    for k, v in reps[0]["buffers"].items():
        assert torch.equal(v, reps[1]["buffers"][k]), k

def _bf16_wire_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(1 << 16, generator=g) * (1 + 0.1 * rank)
    exact = x.double().clone()
    dist.all_reduce(exact)
    wire = x.bfloat16()                      # what GradSync(wire_dtype="bf16") puts on the wire (ops.f32_to_bf16: round to nearest even)
    dist.all_reduce(wire)                    # summed in bf16 by the collective
    rounded_in = x.bfloat16().double()       # the same inputs summed exactly: the part of the error that is the input rounding
    dist.all_reduce(rounded_in)
    torch.save({"err": ((wire.double() - exact).norm() / exact.norm()).item(), "err_in": ((rounded_in - exact).norm() / exact.norm()).item(),
                "sum": wire.float().sum().item()}, os.path.join(out_dir, f"wire_{rank}.pt"))
    dist.destroy_process_group()


def test_bf16_wire_error_figure_world8(tmp_path):
    """the bf16 wire format of the gradient all-reduce sums IN bf16 across the ranks; the only figure so far came from two
    ranks.  Eight ranks (gloo; RCCL's ring adds in another order, the magnitude is the same): relative L2 error of the reduced gradient against
    the exact sum, next to the part that is the rounding of the inputs alone.  Measured here: 4.0e-3 at 8 ranks (2.4e-3 at 2), input rounding
    1.7e-3 -- below the 16-bit backward's own distance to fp32 (<= 9.5e-3 per tensor, tests/test_parity_r4_gpu.py).  Every rank must hold the
    same bits afterwards (replicas stay identical)."""
    world, port = 8, _free_port()
    mp.spawn(_bf16_wire_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"wire_{r}.pt"), weights_only=True) for r in range(world)]
    print(f"[dist] bf16 wire, {world} ranks: relative L2 error {res[0]['err']:.3e} (input rounding alone {res[0]['err_in']:.3e})")
    assert all(r["sum"] == res[0]["sum"] for r in res)
    assert res[0]["err_in"] < 2.5e-3 and res[0]["err"] < 8e-3, res[0]
