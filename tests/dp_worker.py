"""Data-parallel rehearsal worker (launched by tests/test_dp_gpu.py through torch.distributed.run, one process per
rank, every rank on cuda:0, gloo backend -- RCCL refuses two ranks on one device; the collective code path of
trainer.GradSync / SyncBN is backend-agnostic).  Each rank takes its contiguous shard of the seeded global batch, runs
one TrainStep without the optimizer step and writes the all-reduced, 1/world-scaled gradients to a file.

TEST INFRASTRUCTURE: not part of the product path.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--sync-bn", type=int, default=0)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--adam", type=int, default=0)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--wire", default="f32")
    ap.add_argument("--freeze-enc-after", type=int, default=-1, help="toggle_enc_grads(False) before this step index")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import maavss_amd
    from oracle import avse_ref_cpu as orc
    t, w, fft, hpf = 8, 128, 256, 8
    t_a, n_bins = hpf * t, fft // 2 + 1
    b_local = args.batch // world
    shapes = ([b_local, 2, t_a, n_bins], [b_local, 1, t, w, w], hpf)
    model = maavss_amd.AV_Fusion_Model_Frames(*shapes, precise=True)
    twin = orc.AVFusionFramesRef(*shapes)
    # rank 0 holds the seeded weights, the other ranks start from garbage: the constructor's broadcast must repair that
    model.load_state_dict(orc.seeded_state_dict(twin, 31 if rank == 0 else 99), strict=True)
    model = model.to("cuda").train()
    step = maavss_amd.TrainStep(model, lr=1e-3, loss_coeff=0.001, num_seq=1, sync_bn=bool(args.sync_bn),
                                grad_wire_dtype=None if args.wire == "f32" else args.wire)
    x_a, x_v, y_a, y_v = orc.synthetic_batch(args.batch, t, w, t_a, n_bins, hpf, 32)
    lo, hi = maavss_amd.shard_batch(args.batch, rank, world)
    logs = []
    for i in range(args.steps):
        if i == args.freeze_enc_after:
            model.toggle_enc_grads(False)      # the encoders' last gradients stay in the flat buffer: they must not be re-reduced
        losses = step(x_a[lo:hi].cuda(), x_v[lo:hi].cuda(), y_a[lo:hi].cuda(), y_v[lo:hi].cuda(), optimizer_step=bool(args.adam))
        logs.append(list(step.sync.launch_log))
    torch.cuda.synchronize()
    grads = {n: (g / world).cpu() for n, g in step.flat.grad_views.items()}
    out = {"grads": grads, "losses": losses.cpu(), "a_out": step.outputs[0].cpu(),
           "bn": {k: v.cpu() for k, v in model.named_buffers() if not k.startswith("stft_")},
           "params_sum": float(step.flat.params.double().sum().item()), "params": step.flat.params.cpu(), "launch_logs": logs,
           "enc_grad_absmax": float(step.flat.grads[step.flat.fusion_end:].abs().max().item())}
    torch.save(out, os.path.join(args.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
