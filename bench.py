#!/usr/bin/env python
"""bench.py -- train clips/s of the MAAVSS hot path on MI355X (BASELINE.json metric, config[1]).

One "step" = one batch of synthetic clips through the whole hot path, all of it in libmaavss_hip.so:
  frames [B*T,3,W,W] --ViT-S/8 attention extraction (fwd only)--> attention frames [B,1,T,W,W]
  audio  [B,L]       --STFT + noise--> x_stft, y_stft [B,2,T_a,F]
  AV_Fusion_Model_Frames forward + backward (+ RCCL gradient all-reduce when N>1) + Adam.
Inputs are generated once and are resident in HBM before the timed region (SURVEY.md 8d recipe).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # one rank per GPU, weak scaling (B clips per GPU)
    python bench.py --gpus N --steps K --warmup W       # no launcher: bench.py starts the N ranks itself (child processes,
                                                        # before anything touches the GPU) and relays rank 0's JSON line

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed
over the timed region) and `cpu_baseline` (the CPU oracle = validated port of the reference path, timed on the
host cores in the same run, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0      # dense block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): the roof of `--attn-dtype fp8` attention
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (weak scaling)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--framesize", type=int, default=224)
    ap.add_argument("--fft_len", type=int, default=512)
    ap.add_argument("--hops_per_frame", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--precise", action="store_true", help="exact-f32 MFMA in the fusion network (parity mode)")
    ap.add_argument("--verbose", action="store_true", help="per-shape kernel table on stderr")
    ap.add_argument("--vit-chunk", type=int, default=0, help="frames per ViT launch group (0 = library default)")
    ap.add_argument("--sync-bn", action="store_true", help="global-batch BatchNorm statistics over the ranks (N > 1)")
    ap.add_argument("--attn-dtype", choices=["same", "fp8", "fp8-late"], default="same",
                    help="fp8: Q K^T / P V of the ViT blocks on block-scaled fp8 (e4m3) MFMA (BASELINE config 'fp8 MFMA attention'); fp8-late: "
                         "in blocks 8-10 only (end-to-end mask-MSE <= 1e-4, profiles/r4_fp8_operand_ablation.txt); the headline line uses 'same'")
    ap.add_argument("--grad-wire", choices=["f32", "bf16"], default="f32",
                    help="wire format of the gradient all-reduce at N > 1 (bf16: half the xGMI bytes, bf16 sum over the ranks)")
    ap.add_argument("--deterministic", action="store_true", help="no f32-atomic accumulation in the Linear kernels (bit-identical runs)")
    ap.add_argument("--pipeline", choices=["on", "off"], default="on",
                    help="on: attention-frame extraction + STFT of batch i+1 on a second HIP stream under the training step of batch i "
                         "(maavss_amd.ClipPipeline; the reference's data path has no dependency on the optimizer step); off: one stream")
    ap.add_argument("--vit-qkv-ln", choices=["pre", "post"], default="pre",
                    help="where norm1 is applied: on the way into the attn.qkv GEMM (pre, the default) or behind the product, on rounded raw rows with "
                         "gamma-folded weights (post: +0.7 ... 1.1 % clips/s same-box, another rounding realisation: end-to-end mask-MSE 4.0e-6 ... 9.2e-6 "
                         "over the gated cases against 3.8e-6 ... 8.4e-6 -- inside 1e-5 with less margin, hence not the default)")
    ap.add_argument("--vit-gelu", choices=["f32", "half"], default="f32",
                    help="half: mlp.fc1's GELU polynomial in packed IEEE half (faster; twice the rounding error of the hidden activation: a selectable mode)")
    ap.add_argument("--rank-echo", choices=["ok", "fail"], default=None, help=argparse.SUPPRESS)   # launcher self-test, no GPU work
    ap.add_argument("--vit-dtype", choices=["f16", "bf16"], default="f16",
                    help="16-bit storage / MFMA operand format of the ViT extractor (same MFMA rate; f16 meets the 1e-5 mask-MSE end to end)")
    return ap.parse_args()


def synthetic_inputs(torch, batch, frames, width, length, seed, device):
    """U[0,1) RGB, ImageNet-normalised; audio = 0.1*N(0,1) + 3 sinusoids, clipped."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(batch * frames, 3, width, width, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406])[None, :, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[None, :, None, None]
    x = (x - mean) / std
    t = torch.arange(length, dtype=torch.float32) / 16000.0
    a = 0.1 * torch.randn(batch, length, generator=g)
    for f in (220.0, 440.0, 880.0):
        ph = torch.rand(batch, 1, generator=g) * 2 * math.pi
        a = a + 0.2 * torch.sin(2 * math.pi * f * t[None, :] + ph)
    return x.to(device), a.clamp(-1, 1).to(device)


def cpu_baseline(args):
    """The CPU oracle (validated restatement of the reference path, oracle/) on the host cores: one clip
    end to end = ViT attention frames for its T frames + STFT + one AVSE train step (fwd+bwd+Adam).
    SURVEY.md 8(d): 2 warm-ups, then the median of >= 5 timed repetitions (bounded to ~25 s of CPU work)."""
    import statistics
    import torch
    from oracle import avse_ref_cpu as orc, stft_ref_cpu as sref, vit_ref_cpu as vref
    cores = host_cores()
    torch.set_num_threads(cores)
    t, w, hpf = args.frames, args.framesize, args.hops_per_frame
    hop, length, t_a = sref.calc_hop_size(t, hpf, 30, 16000)
    n_bins = args.fft_len // 2 + 1
    b = 1
    sd = vref.seeded_vit_state(3)
    model = orc.AVFusionFramesRef([b, 2, t_a, n_bins], [b, 1, t, w, w], hpf, spatial_match="adaptive")
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    model.train()
    frames = vref.synthetic_frames(t, w, 1)
    audio = sref.synthetic_audio(b, length, 2)
    mid = t // 2

    def clip():
        with torch.no_grad():
            att = vref.clip_normalise_ref(vref.inference_ref(sd, frames))[None]       # [1,1,T,W,W]
            y = sref.stft_ref(audio, args.fft_len, hop)
            x = y + 0.1 * torch.randn_like(y)
        opt.zero_grad()
        loss, *_ = orc.loss_ref(model, x, att, y[:, :, mid * hpf:(mid + 1) * hpf], att[:, :, mid], 0.001, 1)
        loss.backward()
        opt.step()

    warm, times, t_start = 2, [], time.perf_counter()
    for _ in range(warm):
        clip()
    while len(times) < 5 or (time.perf_counter() - t_start < 25.0 and len(times) < 9):
        t0 = time.perf_counter()
        clip()
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": 1.0 / med, "unit": "clips/s", "cores": cores, "cores_note": f"cgroup quota / affinity of this job; os.cpu_count()={os.cpu_count()}",
            "kind": "port",
            "batch": b,
            "sample": f"fp32 torch CPU oracle, BATCH 1 (BASELINE.md section 4 planned the B = 32 shapes; one clip at a time is what fits the ~25 s budget -- "
                      f"the ViT, 85 % of the work, runs its {t} frames as one batch either way), whole clips: ViT-S/8 attention frames of all {t} {w}x{w} frames + "
                      f"{args.fft_len}-pt STFT + noise + AVSE fwd/bwd/Adam; {warm} warm-ups, median of {len(times)} timed clips "
                      f"({med:.2f} s/clip, min {min(times):.2f}, max {max(times):.2f})"}


def lib_sha256():
    import hashlib
    path = os.path.join(ROOT, "maavss_amd", "lib", "libmaavss_hip.so")
    return hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.isfile(path) else None


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def launch_ranks(n, argv=None, env=None, timeout=None):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this script, one rank each (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them), relay rank 0's stdout (the ONE JSON line),
    return the worst exit status.  Runs before this process has imported torch or touched the GPU; children are spawned, never
    exec'ed into."""
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MAAVSS_BENCH_SELF_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    base.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    status = 0
    try:
        out0, _ = procs[0].communicate(timeout=timeout)
        for p in procs:
            status = max(status, abs(p.wait(timeout=timeout)))
    except subprocess.TimeoutExpired:
        status = 124
    finally:
        for p in procs:                      # a rank that died leaves the others in a collective: end exactly the PIDs started here
            if p.poll() is None:
                p.kill()
                p.wait()
                status = status or 1
    # rank 0's stdout carries the ONE JSON line; anything else a backend prints there (gloo announces its connections on stdout) goes to stderr
    for line in out0.decode().splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return status


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rank_echo:                      # launcher self-test (tests/test_bench_launcher_cpu.py): no torch, no GPU
        if args.rank_echo == "fail" and rank == world - 1:
            sys.exit(3)
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "master": os.environ.get("MASTER_ADDR"),
                          "port": int(os.environ.get("MASTER_PORT", "0")), "self_launched": os.environ.get("MAAVSS_BENCH_SELF_LAUNCHED") == "1"}))
        return
    import torch
    import torch.distributed as dist
    if args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or run plain `python bench.py --gpus {args.gpus}` (bench.py then starts the ranks itself)")
    # Rehearsal knobs for a box with fewer GPUs than ranks (never used by the driver): all ranks on device 0
    # and/or the gloo backend.  The collective code path (trainer.GradSync) is the same.
    if os.environ.get("MAAVSS_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("MAAVSS_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import maavss_amd
    from maavss_amd import _lib
    if args.deterministic:
        maavss_amd.set_deterministic(True)
    b, t, w, hpf = args.batch, args.frames, args.framesize, args.hops_per_frame
    hop, length, t_a = maavss_amd.calc_hop_size(t, hpf, 30, 16000)
    n_bins = args.fft_len // 2 + 1
    torch.manual_seed(1234)                  # replicas: identical initialisation on every rank (TrainStep also broadcasts rank 0's)
    frames, audio = synthetic_inputs(torch, b, t, w, length, 1234 + rank, dev)      # data: a different shard per rank

    va = maavss_amd.VideoAttention(path_to_weights="dino_deitsmall8_pretrain.pth", device=dev, act_dtype=args.vit_dtype,
                                   attn_dtype=None if args.attn_dtype == "same" else args.attn_dtype, gelu=args.vit_gelu, qkv_ln=args.vit_qkv_ln)   # random init: no network
    if args.vit_chunk:
        va.frames_per_launch = args.vit_chunk
    stft = maavss_amd.STFT(args.fft_len, hop, noise_std=0.1, device=dev)
    try:
        model = maavss_amd.AV_Fusion_Model_Frames([b, 2, t_a, n_bins], [b, 1, t, w, w], hpf, precise=args.precise)
        spatial = "exact"
    except ValueError:
        model = maavss_amd.AV_Fusion_Model_Frames([b, 2, t_a, n_bins], [b, 1, t, w, w], hpf, precise=args.precise,
                                                  spatial_match="adaptive")
        spatial = "adaptive"
    model = model.to(dev).train()
    step_fn = maavss_amd.TrainStep(model, lr=1e-5, loss_coeff=0.001, num_seq=1, sync_bn=args.sync_bn,
                                   grad_wire_dtype=None if args.grad_wire == "f32" else args.grad_wire)
    mid = t // 2
    attn = torch.empty(b * t, 1, w, w, device=dev, dtype=torch.float32)

    pipe = maavss_amd.ClipPipeline(va, stft, t) if args.pipeline == "on" else None

    def step(i, serial=False):
        """One batch through the whole hot path.  Pipelined form: every call enqueues exactly one extraction (batch i+1, side
        stream) and one training step (batch i, extracted by the previous call) -- the same work per step as the serial form."""
        if pipe is None or serial:
            va.attention_frames(frames, clip_frames=t, out=attn, finite_check="deferred")   # range guard of the half storage, no sync
            x_v = attn.view(b, 1, t, w, w)
            x_stft, y_stft = stft(audio, seed=i)
        else:
            if pipe.head == pipe.tail:
                pipe.submit(frames, audio, seed=i)          # first call only: fill the pipeline
            pipe.submit(frames, audio, seed=i + 1)
            x_v, x_stft, y_stft = pipe.get()
        y_a = y_stft[:, :, mid * hpf:(mid + 1) * hpf, :]
        y_v = x_v[:, :, mid]
        out = step_fn(x_stft, x_v, y_a, y_v)
        if pipe is not None and not serial:
            pipe.release()
        return out

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()             # all streams of the device, the pipeline's side stream included

    for i in range(args.warmup):
        step(i)
    sync_all()
    # HIP events around the entry points the roofline / stage rows are made of (the ViT GEMMs and attention, STFT, Adam); with
    # --verbose around every entry point (155 launches per step: the event pairs then cost ~0.9 ms of the step)
    staged = ("maavss_vit_attn", "maavss_vit_attn_mx", "maavss_vit_ws_gemm_ln_mx", "maavss_vit_panel_gemm", "maavss_vit_ws_gemm", "maavss_vit_ws_gemm_ln", "maavss_vit_ws_gemm_ln_post",
              "maavss_vit_gemm", "maavss_vit_gemm_stats", "maavss_stft_fwd", "maavss_adam_step")
    timer = _lib.KernelTimer(only=None if args.verbose else staged)
    _lib.set_timer(timer)
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = step(args.warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    _lib.set_timer(None)
    if world > 1:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = el.item()
    loss_val = float(losses[2].item())
    # N > 1: evidence that the collective saw N ranks and kept the replicas identical -- every rank's parameter checksum
    # (f64 sum and sum of squares of the flat buffer after the K timed Adam steps), MIN- and MAX-reduced over the ranks
    collective = None
    if world > 1:
        pf = step_fn.flat.params.double()
        cs = torch.stack([pf.sum(), (pf * pf).sum()])
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                       # every rank contributes 1: the sum IS the number of ranks that took part
        collective = {"backend": dist.get_backend(), "ranks_seen": int(dist.get_world_size()), "ranks_summed": int(round(ones.item())),
                      "replica_checksum_equal": bool(torch.equal(lo, hi)), "replica_checksum": [float(x) for x in cs.tolist()],
                      "gpus_visible": int(torch.cuda.device_count()),
                      "devices": "all ranks on cuda:0 (rehearsal)" if os.environ.get("MAAVSS_BENCH_SINGLE_DEVICE") == "1" else "one per rank",
                      "launcher": "bench.py (self-launched child processes)" if os.environ.get("MAAVSS_BENCH_SELF_LAUNCHED") == "1" else "external (torch.distributed.run)",
                      "grad_bytes_per_step": int(step_fn.flat.total * (2 if args.grad_wire == "bf16" else 4)),
                      "buckets_launched_last_step": list(step_fn.sync.launch_log)}
    if pipe is not None:
        pipe.drain()
    va.check_finite()                        # the deferred range flags of the last steps
    # Second pass of K steps on ONE stream.  Pipelined run: with the HIP events -- a kernel's duration is only its own when
    # it has the chip to itself, so the per-kernel roofline / stage rows come from this pass (the timed region above shares
    # the chip between the extractor's and the fusion network's kernels; its event times are reported next to them).
    # Serial run: without the events (transparency: what the events cost).
    timer_serial = None
    if pipe is not None:
        timer_serial = _lib.KernelTimer(only=None if args.verbose else staged)
        _lib.set_timer(timer_serial)
    t1 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + args.steps + i, serial=True)
    sync_all()
    elapsed_second = time.perf_counter() - t1
    _lib.set_timer(None)
    va.check_finite()

    if rank == 0:
        summ_timed = timer.summary()
        summ = timer_serial.summary() if timer_serial is not None else summ_timed
        # ---- roofline of the dominant kernel family (by accumulated HIP-event time)
        def flops_of(name, a):
            if name in ("maavss_vit_gemm", "maavss_vit_gemm_stats"):
                return 2.0 * a[8] * a[9] * a[10]                  # M, N, K
            if name == "maavss_vit_panel_gemm":
                return 2.0 * a[11] * a[12] * 384                  # M, N, K = 384 (LayerNorm flops not counted)
            if name == "maavss_vit_ws_gemm":
                return 2.0 * a[8] * a[9] * 384                    # M, N, K = 384 (LayerNorm output flops not counted)
            if name == "maavss_vit_ws_gemm_ln":
                return 2.0 * a[11] * a[12] * 384                  # M, N, K = 384 (LayerNorm flops not counted)
            if name == "maavss_vit_ws_gemm_ln_post":
                return 2.0 * a[10] * a[11] * 384
            if name == "maavss_vit_attn":
                return 4.0 * a[2] * a[4] * a[3] * a[3] * 64       # frames * heads * ntok^2 * 64 * (QK^T + PV)
            if name == "maavss_vit_attn_mx":
                return 4.0 * a[2] * a[4] * a[3] * a[3] * 64       # (ws, out, frames, ntok, heads, ...)
            if name == "maavss_vit_ws_gemm_ln_mx":
                return 2.0 * a[9] * 1152 * 384                    # M, N = 1152, K = 384
            return 0.0
        def bytes_of(name, a):
            """Algorithmic HBM bytes of one launch: every operand read once, every result written once."""
            if name in ("maavss_vit_gemm", "maavss_vit_gemm_stats"):
                m, n, k, epi = a[8], a[9], a[10], a[11]
                out = {2: 8.0, 3: 4.0}.get(epi, 2.0)                # f32 read-modify-write / f32 / bf16
                stats = 8.0 * m * (n // 128) if (name.endswith("_stats") and a[14]) else 0.0   # (mean, M2) per row and 128-column tile
                return 2.0 * m * k + 2.0 * n * k + out * m * n + stats
            if name == "maavss_vit_panel_gemm":
                m, n, epi = a[11], a[12], a[13]
                return m * 384 * (4.0 if a[0] else 2.0) + 2.0 * n * 384 + (8.0 if epi == 2 else 2.0) * m * n
            if name == "maavss_vit_ws_gemm":
                m, n, epi = a[8], a[9], a[10]                       # + the LayerNorm-ed 16-bit copy of the rows when asked for
                return 2.0 * m * 384 + 2.0 * n * 384 + (8.0 if epi == 2 else 2.0) * m * n + (2.0 * m * 384 if a[13] else 0.0)
            if name in ("maavss_vit_ws_gemm_ln", "maavss_vit_ws_gemm_ln_post"):
                m, n = (a[11], a[12]) if name.endswith("_ln") else (a[10], a[11])      # f32 rows + their 24-byte statistics in, 16-bit out
                return (4.0 * 384 + 24.0) * m + 2.0 * n * 384 + 2.0 * m * n
            if name == "maavss_vit_attn":
                return a[2] * a[3] * (1152 + 384) * 2.0             # qkv in, attention output out, bf16
            if name == "maavss_vit_attn_mx":
                return a[2] * a[3] * (1152 * 1.0 + 1152 / 32.0 + 384 * 2.0)      # fp8 images + scale bytes in, 16-bit out
            if name == "maavss_vit_ws_gemm_ln_mx":
                return (4.0 * 384 + 24.0) * a[9] + 2.0 * 1152 * 384 + a[9] * (1152 * 1.0 + 1152 / 32.0)
            return 0.0
        by_time = sorted(summ.items(), key=lambda kv: -kv[1]["ms"])
        dom_name, dom = by_time[0]
        roofline = None
        for name, d in by_time:
            fl = sum(flops_of(name, a) for a in d["args"])
            if fl > 0:
                by = sum(bytes_of(name, a) for a in d["args"])
                avg_ms = d["ms"] / d["calls"]
                tf = fl / d["calls"] / (avg_ms * 1e-3) / 1e12
                gbs = by / d["calls"] / (avg_ms * 1e-3) / 1e9
                # the binding roof is the one under which this launch would take longer (K = 384 layers: 230 FLOP per
                # algorithmic byte, below the chip's 2500 / 8 = 312 FLOP/B -> HBM; attention: 400 FLOP/B -> MFMA)
                mfma_peak = PEAK_FP8_TFLOPS if name == "maavss_vit_attn_mx" else PEAK_BF16_TFLOPS
                hbm_bound = by / (PEAK_HBM_GBS * 1e9) > fl / (mfma_peak * 1e12)
                roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
                            "achieved": round(gbs if hbm_bound else tf, 2), "peak": PEAK_HBM_GBS if hbm_bound else mfma_peak,
                            "unit": "GB/s" if hbm_bound else "TFLOP/s",
                            "frac": round(gbs / PEAK_HBM_GBS if hbm_bound else tf / mfma_peak, 4), "traffic": None,
                            "algorithmic_bytes_per_launch": round(by / d["calls"]), "algorithmic_flops_per_launch": round(fl / d["calls"]),
                            "mfma_tflops": round(tf, 2), "mfma_frac": round(tf / mfma_peak, 4),
                            "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
                            "launches": d["calls"], "avg_launch_us": round(avg_ms * 1e3, 2),
                            "share_of_event_timed_entry_points": round(d["ms"] / sum(x["ms"] for x in summ.values()), 3),
                            "share_of_step": round(d["ms"] / args.steps / (elapsed_second / args.steps * 1e3), 3) if timer_serial is not None
                            else round(d["ms"] / (elapsed * 1e3), 3)}
                if timer_serial is not None:
                    dt = summ_timed.get(name)
                    roofline["measured_over"] = (f"{args.steps} steps of this run on one stream (second pass, {elapsed_second / args.steps * 1e3:.2f} ms/step): "
                                                 "the kernel alone on the chip; `in_timed_region` = the same launches in the two-stream timed region, "
                                                 "where they share CUs and HBM with the fusion network's kernels")
                    if dt and dt["calls"]:
                        us = dt["ms"] / dt["calls"] * 1e3
                        gbs_t, tf_t = by / d["calls"] / (us * 1e-6) / 1e9, fl / d["calls"] / (us * 1e-6) / 1e12
                        roofline["in_timed_region"] = {"avg_launch_us": round(us, 2), "achieved": round(gbs_t if hbm_bound else tf_t, 2),
                                                       "frac": round(gbs_t / PEAK_HBM_GBS if hbm_bound else tf_t / mfma_peak, 4)}
                break
        # ---- the stages BASELINE.json's north star names explicitly: attention against the MFMA peak, the STFT path and
        # the other streaming kernels against HBM (algorithmic work / HIP-event time of the timed region)
        stages = {}
        def stage_row(name, flops=0.0, bytes_=0.0):
            d = summ.get(name)
            if not d or d["ms"] <= 0:
                return
            sec = d["ms"] * 1e-3
            row = {"ms_per_step": round(d["ms"] / args.steps, 3), "launches_per_step": d["calls"] // args.steps}
            if flops:
                # a kernel is priced against the peak of the instruction it issues: fp8 attention against 5 PF
                peak = PEAK_FP8_TFLOPS if name == "maavss_vit_attn_mx" else PEAK_BF16_TFLOPS
                row.update(tflops=round(flops / sec / 1e12, 1), mfma_frac=round(flops / sec / 1e12 / peak, 4), mfma_peak_tflops=peak)
                if peak != PEAK_BF16_TFLOPS:
                    row["mfma_frac_of_bf16_peak"] = round(flops / sec / 1e12 / PEAK_BF16_TFLOPS, 4)
            if bytes_:
                row.update(gbs=round(bytes_ / sec / 1e9, 1), hbm_frac=round(bytes_ / sec / 1e9 / PEAK_HBM_GBS, 4))
            stages[name.replace("maavss_", "")] = row
        for nm in ("maavss_vit_attn", "maavss_vit_attn_mx", "maavss_vit_ws_gemm_ln_mx", "maavss_vit_panel_gemm", "maavss_vit_ws_gemm", "maavss_vit_ws_gemm_ln", "maavss_vit_ws_gemm_ln_post",
                   "maavss_vit_gemm", "maavss_vit_gemm_stats"):
            if nm in summ:
                stage_row(nm, sum(flops_of(nm, a) for a in summ[nm]["args"]), sum(bytes_of(nm, a) for a in summ[nm]["args"]))
        if "maavss_stft_fwd" in summ:      # audio in (4 B/sample) + y and x out (2 planes x T_a x F x 4 B each)
            a0 = summ["maavss_stft_fwd"]["args"][0]
            per = a0[1] * (4.0 * a0[2] + 2 * 2.0 * a0[7] * a0[8] * 4.0)
            stage_row("maavss_stft_fwd", bytes_=per * summ["maavss_stft_fwd"]["calls"])
            if "stft_fwd" in stages:
                stages["stft_fwd"]["note"] = ("one launch per step over B*T_a frames: at B=32 that is ~2k waves = one wave round of the chip "
                                              "(launch/latency-bound); the kernel reaches 3.10 TB/s (0.39 of 8 TB/s) at B=256 and 2.9 TB/s at B=8192 "
                                              "(profiles/r4_stft_bench.json; round 3: 1.7 / 2.0): one wave per frame pair is bound by its Philox + FFT vector work, not by HBM")
        if "maavss_adam_step" in summ:     # p, g, m, v read + p, m, v written: 28 B per parameter
            stage_row("maavss_adam_step", bytes_=sum(28.0 * a[4] for a in summ["maavss_adam_step"]["args"]))
        # HBM bytes per launch of that kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs of this same command; FETCH doubled per the gfx950 guide) -- null if absent.
        pmc_path = os.path.join(ROOT, "profiles", "pmc_hbm_traffic_latest.json")
        if roofline is not None and os.path.isfile(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            # the counters describe the kernels of the build they were taken with: a profile of another library is stale
            if pmc.get("lib_sha256") != lib_sha256():
                roofline["traffic_note"] = ("profiles/pmc_hbm_traffic_latest.json was taken with another build of libmaavss_hip.so "
                                            f"(tag {pmc.get('tag')}); traffic left null")
                pmc = {"kernels": []}
            stem = roofline["kernel"].replace("maavss_", "") + "_kernel"
            rows = [r for r in pmc["kernels"] if stem in r["kernel"]]
            n = sum(r["launches"] for r in rows)
            if n:
                mb = sum(r["launches"] * (r["fetch_MB_per_launch_x2_corrected"] + r["write_MB_per_launch"]) for r in rows) / n
                roofline["traffic"] = round(mb * 1e6)
                roofline["traffic_unit"] = (f"bytes/launch (PMC FETCH_SIZE*2 + WRITE_SIZE, profiles/pmc_hbm_traffic_latest.json, "
                                            f"tag {pmc.get('tag')}, same libmaavss_hip.so)")
        breakdown = {k: round(v["ms"] / args.steps, 3) for k, v in by_time[:(None if args.verbose else 12)]}
        if args.verbose:
            shapes = {}
            rec_timer = timer_serial if timer_serial is not None else timer
            for name, a, e0, e1 in rec_timer.records:
                if name in ("maavss_vit_gemm", "maavss_vit_gemm_stats"):
                    key = f"vit_gemm epi{a[11]} M{a[8]} N{a[9]} K{a[10]}"
                    d = shapes.setdefault(key, [0, 0.0, 0.0])
                    d[0] += 1
                    d[1] += e0.elapsed_time(e1)
                    d[2] += 2.0 * a[8] * a[9] * a[10]
                if name == "maavss_gemm_f32":      # the Linear / LSTM-projection GEMMs: weight streaming at M = batch
                    # (A, lda, transA, B, ldb, transB, C, ldc, transC, M, N, K, alpha, beta, act, split_k, precise, stream)
                    key = f"gemm_f32 M{a[9]} N{a[10]} K{a[11]} ta{a[2]} tb{a[5]} tc{a[8]} act{a[14]} beta{a[13]} splitk{a[15]}"
                    d = shapes.setdefault(key, [0, 0.0, 0.0])
                    d[0] += 1
                    d[1] += e0.elapsed_time(e1)
                    d[2] += 2.0 * a[9] * a[10] * a[11]
            for key, (n, ms, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                print(f"[bench] {key}: {n} launches, {ms / n * 1e3:.1f} us avg, {fl / ms / 1e9:.0f} TFLOP/s", file=sys.stderr)
            print(f"[bench] sum of kernel time {sum(v['ms'] for v in summ.values()) / args.steps:.2f} ms/step, "
                  f"wall {(elapsed_second if timer_serial is not None else elapsed) / args.steps * 1e3:.2f} ms/step, {len(rec_timer.records) // args.steps} launches/step", file=sys.stderr)
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            metric = json.load(fh)["metric"]
        clips = b * world * args.steps
        out = {
            "metric": metric, "value": round(clips / elapsed, 3), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precise else args.vit_dtype, "data": "synthetic",
            ("ms_per_step_one_stream" if pipe is not None else "ms_per_step_without_kernel_events"): round(elapsed_second / args.steps * 1e3, 3),
            "config": {"workload": f"batch={b}/GPU, {t} frames {w}x{w}, {args.fft_len}-pt STFT, ViT-S/8 attention extraction "
                                   f"({args.vit_dtype} MFMA operands, f32 accumulate) + STFT + AV_Fusion_Model_Frames fwd+bwd (16-bit MFMA conv, f32 accumulate) + Adam",
                       "global_batch": b * world, "frames": t, "framesize": w, "fft_len": args.fft_len,
                       "parallelism": f"dp{world}", "streams": "2 (extraction of batch i+1 under the training step of batch i)" if pipe is not None else "1", "avse_spatial_match": spatial, "vit_weights": "random-init (no network)",
                       "vit": args.vit_dtype, "vit_gelu": args.vit_gelu, "vit_qkv_ln": args.vit_qkv_ln, "vit_attention": ("block-scaled fp8 (MX e4m3, e8m0 scale per 32; f32 accumulate -- wider than the config's bf16)" + (" in blocks 8-10, " + args.vit_dtype + " in blocks 0-7" if args.attn_dtype == "fp8-late" else "")) if args.attn_dtype != "same" else args.vit_dtype, "conv_fwd": "f32" if args.precise else "f16", "conv_bwd": "f32" if args.precise else "bf16",
                       "end_to_end_mask_mse_vs_fp32_reference_chain": ("3.0e-3 ... 5.7e-3 over four seed sets (tests/test_parity_r2_gpu.py[fp8], tests/test_parity_r3_gpu.py, shape P; e4m3 operands: a throughput mode; "
                                                                       "per-operand / per-block table: profiles/r4_fp8_operand_ablation.txt -- the error is made in blocks 0-5, no all-block fp8 point is within 1e-4)" if args.attn_dtype == "fp8"
                                                                    else "3.7e-5 ... 4.9e-5, gated <= 1e-4 (tests/test_parity_r4_gpu.py; profiles/r4_fp8_operand_ablation.txt: only late blocks stay below 1e-4, no operand subset over all blocks does).  Speed: 3 of 11 blocks in fp8 is within noise of the f16 step (808 vs 813 and 839 vs 834 clips/s on two boxes, profiles/r4_fp8late_bench.json, r4_g_fp8late_bench.json; all blocks: 821 / 855, +1 ... 2.5 %) -- no fp8 point is both inside 1e-4 and faster" if args.attn_dtype == "fp8-late"
                                                                    else ("3.0e-6 ... 6.5e-6 over five seed sets, shape P and the benched shape" if args.vit_dtype == "f16" else "2.3e-4 (shape P)") + " (tests/test_parity_r2_gpu.py, tests/test_parity_r3_gpu.py; target 1e-5)"),
                       "linear_lstm": "f32" + (" (deterministic: no atomic split-K)" if args.deterministic else " (split-K by f32 atomics)"), "batchnorm": "global-batch (sync)" if (args.sync_bn and world > 1) else "per-rank", "grad_all_reduce": f"{args.grad_wire} wire, per-module buckets in backward order",
                       "loss": loss_val},
            "roofline": roofline,
            "stages": stages,
            "kernel_ms_per_step": breakdown,
        }
        if timer_serial is not None:
            out["kernel_ms_per_step_in_timed_region"] = {k: round(v["ms"] / args.steps, 3) for k, v in sorted(summ_timed.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        if collective is not None:
            out["collective"] = collective
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
