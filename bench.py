#!/usr/bin/env python3
"""Training-throughput benchmark of the trimodal gesture GAN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1, either form: (a) plain `python bench.py --gpus N ...` -- this process then only LAUNCHES: it starts N fresh rank processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous) before touching any GPU itself, relays rank 0's JSON line and
    returns non-zero if any rank failed; (b) python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): config/multimodal_context.yml -- PoseGenerator + ConvDiscriminator, post-warm-up GAN
iteration (3 G forwards, 1 G backward, 3 D forwards, D backward passes, both Adam steps), batch 128 clips of 34 frames x 27
dims per GPU, 36 267 audio samples per clip, synthetic data (V = 20 000 words, 1 370 speakers), random-init weights, fp32.
A "step" is one full iteration; inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "gesture-generation-from-trimodal-context_amd"

V, S, T, D, A = 20000, 1371, 34, 27, 36267
# algorithmic work per clip of one post-warm-up iteration: 5 G-forward-equivalents + 9 D-forward-equivalents
FLOP_PER_CLIP = 2.735e9
PEAK_F32_MFMA = 157.3e12          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA = 2.5e15           # same table, dense bf16 matrix peak (no sparsity)


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv, worker=None, port=None, poll_s=0.2, env_extra=None):
    """One process per GPU (the role of nn.DataParallel's replicas, scripts/train.py:93-96), started by THIS process as fresh children:
    the launcher never initialises the GPU (no torch.cuda call happens before or after this function in the parent), so no process that
    owns a HIP context is ever replaced or forked.  Child r gets RANK = LOCAL_RANK = r, WORLD_SIZE = LOCAL_WORLD_SIZE = n and a
    127.0.0.1 rendezvous.  Rank 0's stdout is captured, every other stream passes through to stderr.  If a rank exits non-zero the
    others are terminated (exact PIDs) so a failed rank cannot leave its peers waiting in a collective.
    Returns (rc, last non-empty stdout line of rank 0 or None): rc is 0 only if every rank returned 0."""
    worker = worker or os.path.abspath(__file__)
    port = port or _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, worker] + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, text=True))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    kill_at = None
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench launcher: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for o in sorted(live):
                    procs[o].terminate()
                kill_at = time.time() + 10.0              # ranks that ignore SIGTERM inside a collective
        if live:
            if kill_at is not None and time.time() > kill_at:
                for o in sorted(live):
                    procs[o].kill()
                kill_at = time.time() + 3600.0
            time.sleep(poll_s)
    reader.join(timeout=5.0)
    text = [ln.rstrip("\n") for ln in lines if ln.strip()]
    for ln in text[:-1]:
        print(ln, file=sys.stderr, flush=True)            # anything rank 0 printed before its JSON line
    return rc, (text[-1] if text else None)


def host_cores(cap=64):
    """CPU threads this process may really use: the affinity mask, cut down to the cgroup's CPU quota where one is set (a GPU box hands a
    one-GPU job a share of the host -- an intra-op pool sized by the host's core count is throttled by the quota and runs many times
    slower than one sized by the share), capped at `cap` (ATen's intra-op pool stops scaling far below a full host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                             # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        pass
    try:                                             # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return max(1, min(n, cap))


def make_args():
    """config/multimodal_context.yml through the package's parse_args mirror (hidden 300, 4 layers, z_type speaker, batch 128 ...)."""
    return importlib.import_module(PKG + ".config").load_config("multimodal_context")


def synthetic_batch(batch, seed, device):
    """sparse-onset word ids (4..12 word onsets per clip, rest PAD=0), N(0, 0.1^2) audio and poses."""
    g = torch.Generator().manual_seed(seed)
    text = torch.zeros(batch, T, dtype=torch.int64)
    for b in range(batch):
        k = int(torch.randint(4, 13, (1,), generator=g))
        frames = torch.randperm(T, generator=g)[:k]
        text[b, frames] = torch.randint(4, V, (k,), generator=g)
    audio = (0.1 * torch.randn(batch, A, generator=g)).clamp_(-1, 1)
    vid = torch.randint(1, S, (batch,), generator=g)
    poses = 0.1 * torch.randn(batch, T, D, generator=g)
    return text.to(device), audio.to(device), poses.to(device), vid.to(device)


def build(pkg, device, seed=0):
    torch.manual_seed(seed)
    args = make_args()
    emb = (torch.randn(V, 300) / 300 ** 0.5).numpy()            # vocab.py:74-75 initialisation scale
    G = pkg.PoseGenerator(args, D, V, 300, emb, pkg.Vocab.speakers(S)).to(device)
    Dn = pkg.ConvDiscriminator(D).to(device)
    return args, G, Dn


_REAL_STDOUT = None          # a duplicate of fd 1 while fd 1 itself points at stderr (data-parallel runs: see quiet_stdout)


def quiet_stdout():
    """From here until finish(): everything written to fd 1 -- RCCL prints a version banner on C stdout when its first communicator comes up --
    goes to stderr, so that stdout carries the ONE JSON line and nothing else."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def finish(line, dist_on):
    """Tear the process group down, flush what native libraries buffered on C stdout, give stdout back, then print the ONE JSON line
    (rank 0) as the only line of stdout."""
    global _REAL_STDOUT
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if _REAL_STDOUT is not None:
        os.dup2(_REAL_STDOUT, 1)
        os.close(_REAL_STDOUT)
        _REAL_STDOUT = None
    if line is not None:
        print(line, flush=True)


def time_kernel(fn, iters=50, warm=5):
    """Average duration (s) of one launch sequence `fn`, HIP events on the stream the kernels are launched on
    (the ops launch on torch's current stream)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


PEAK_HBM = 8.0e12                  # /opt/skills/guides/MI355X_MICROARCH.md, "HBM3E peak BW" (spec; 6.29 TB/s measured for a float4 copy)


def decode_kernel_roofline(pkg, device, batch):
    """Dominant kernel of the eval forward (one per GRU layer, 4 per window): the persistent cluster GRU recurrence at B = utterances per GPU
    (no stacking, nothing saved).  Same accounting as the training line's roofline."""
    ops = pkg.ops
    H = 300
    gi = torch.randn(2, batch, T, 3 * H, device=device) * 0.1
    w = [torch.randn(3 * H, H, device=device) * 0.05 for _ in range(2)]
    b = [torch.randn(3 * H, device=device) * 0.05 for _ in range(2)]
    y = torch.empty(batch, T, 2 * H, device=device)
    dt = time_kernel(lambda: ops.gru_forward(gi, w, b, y, None), iters=20)
    ops.check_async_errors()
    flops = (T - 1) * 2 * batch * H * 3 * H * 2
    if ops.gru_vec_takes(batch, H, None, None):
        # a handful of utterances: the few-row kernel (csrc/gru_vec.hip) -- fp32 FMAs on resident fp32 weights, no matrix instruction at all; priced
        # against the fp32 VECTOR peak (256 CUs x 4 SIMDs x 16 lanes x 2 FLOP x 2.4 GHz = 78.6 TFLOP/s); what bounds it is the per-step hand-off
        peak_valu = 256 * 4 * 16 * 2 * 2.4e9
        return {"kernel": "gru_seq_fwd_vec_kernel<rows = %d> (eval, B = utterances <= 4)" % (1 if batch == 1 else 2 if batch == 2 else 4), "bound": "valu",
                "achieved": flops / dt / 1e12, "peak": peak_valu / 1e12, "unit": "TFLOP/s", "frac": flops / dt / peak_valu, "traffic": None,
                "launch_us": dt * 1e6, "flop_per_launch": flops, "us_per_step": dt * 1e6 / T,
                "note": "latency-bound: one memory round trip per step (fp32 exchange words polled against a sentinel), 20 of 256 CUs; fp32 arithmetic"}
    return {"kernel": "gru_seq_fwd_cluster_x3_kernel (eval, B = utterances)", "bound": "mfma", "achieved": flops / dt / 1e12, "peak": PEAK_F32_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": flops / dt / PEAK_F32_MFMA, "traffic": None, "launch_us": dt * 1e6, "flop_per_launch": flops,
            "us_per_step": dt * 1e6 / T, "note": "latency-bound per-step hand-off chain of the recurrence; fp32-accurate product on the bf16 matrix cores"}


def cpu_baseline_decode(batch, budget_s=15.0, max_windows=8):
    """The oracle's eval forward (CPU port of multimodal_context_net.py:110-160, ATen fp32 kernels) on `batch` windows at a time."""
    from oracle import ref_model as O
    O.FAST = True
    cores = host_cores()
    torch.set_num_threads(cores)
    gst = O.make_generator_state(0, V, S)
    g = torch.Generator().manual_seed(6)
    text = torch.zeros(batch, T, dtype=torch.int64)
    text[:, ::5] = torch.randint(4, V, (batch, len(range(0, T, 5))), generator=g)
    audio = 0.1 * torch.randn(batch, A, generator=g)
    pre = torch.zeros(batch, T, D + 1)
    vid = torch.randint(1, S, (batch,), generator=g)
    with torch.no_grad():
        O.generator_forward(gst, pre, text, audio, vid, training=False, rand=O.Rand(seed=1), fast_gru=True)
        t0 = time.perf_counter()
        n = 0
        while n < max_windows and (n == 0 or time.perf_counter() - t0 < budget_s):
            O.generator_forward(gst, pre, text, audio, vid, training=False, rand=O.Rand(seed=2 + n), fast_gru=True)
            n += 1
    dt = (time.perf_counter() - t0) / n
    O.FAST = False
    return {"value": batch * T / dt, "unit": "pose-frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} window steps of {batch} utterances (eval forward only) after 1 warm-up, fp32 ATen kernels, {dt:.2f} s per step"}


def cpu_baseline_ae(batch, budget_s=10.0, max_steps=20):
    """The oracle's autoencoder training iteration (CPU port of train_feature_extractor.py:54-97) at the same batch."""
    from oracle import ref_model as O
    cores = host_cores()
    torch.set_num_threads(cores)
    st, opt = O.make_autoencoder_state(2), {}
    poses = 0.1 * torch.randn(batch, T, D, generator=torch.Generator().manual_seed(7))
    O.ae_train_iter(st, opt, poses)
    t0 = time.perf_counter()
    n = 0
    while n < max_steps and (n == 0 or time.perf_counter() - t0 < budget_s):
        O.ae_train_iter(st, opt, poses)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": batch / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{n} training iterations at batch {batch} after 1 warm-up, fp32 ATen kernels, {dt * 1e3:.1f} ms per iteration"}


def dominant_kernel_roofline(pkg, device, batch):
    """The single kernel with the largest share of the iteration (profiles/r6_fin_by_shape.txt): the persistent cluster-synchronised GRU
    recurrence of the generator's stacked forward, gru_seq_fwd_cluster_x3_kernel<2, NS> at B = 3*batch, H = 300, T = 34 -- one launch per
    layer walks all 34 steps of both directions (csrc/gru_cluster_x3.hip).
    Algorithmic FLOPs per launch: (T-1) steps x 2 directions x B x 3H x H x 2 (the h_{t-1} @ W_hh^T products; gate maths excluded).
    Peak = the fp32 matrix peak: the kernel computes an fp32-accurate product.  Round 6: each product is THREE fp16 MFMAs on two-term split
    operands (h * 2^14 and per-row scaled W_hh as hi / lo fp16 planes; TG_GRU_H2=0: six bf16 MFMAs on three-term splits), i.e. 3 x the
    algorithmic FLOPs on the 16-bit matrix pipe: peak_f16x2 = dense fp16 peak / 3 is the ceiling of the arithmetic issued.  The kernel is
    bound by the per-step inter-workgroup hand-off latency, not by MFMA issue or HBM: the fraction says how far.  The timed call is
    tg_gru_forward_cluster on the launch stream (HIP events); rocprofv3's per-kernel average is the persistent kernel alone."""
    ops = pkg.ops
    Bs, H = 3 * batch, 300
    gi = torch.randn(2, Bs, T, 3 * H, device=device) * 0.1
    w = [torch.randn(3 * H, H, device=device) * 0.05 for _ in range(2)]
    b = [torch.randn(3 * H, device=device) * 0.05 for _ in range(2)]
    y = torch.empty(Bs, T, 2 * H, device=device)
    sv = torch.empty(2, Bs, T, 4 * H, device=device)
    dt = time_kernel(lambda: ops.gru_forward(gi, w, b, y, sv, save_rows=(batch, batch)), iters=20)      # as the trainer calls it (gates saved for call g2 only)
    ops.check_async_errors()
    flops = (T - 1) * 2 * Bs * H * 3 * H * 2
    # HBM-side bytes per launch: NOT measured in this run -- the figure of rocprofv3 PMC passes at exactly this shape and call on the round-6 build
    # (tools/r6_pmc.sh -> profiles/r6_pmc_gru_fwd.txt): 2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE.  Only valid for batch 128
    # (B_s = 384) and the fp16 x 2 kernel; `traffic_source` in the line says where it comes from.
    h2 = int(os.environ.get("TG_GRU_H2", "3")) & 1 == 1          # mask: 1 = forward recurrence, 2 = backward recurrence
    traffic = PMC_TRAFFIC_GRU_FWD if (batch == 128 and h2 and ops.get_math_mode() == "f32") else None
    if ops.get_math_mode() == "bf16":          # secondary tier: one bf16 MFMA per product -> price against the dense bf16 peak
        return {"kernel": "gru_seq_fwd_cluster_x3_kernel<2, NS = 1>", "bound": "mfma", "achieved": flops / dt / 1e12, "peak": PEAK_BF16_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": flops / dt / PEAK_BF16_MFMA, "traffic": None, "launch_us": dt * 1e6,
                "flop_per_launch": flops, "us_per_step": dt * 1e6 / T,
                "note": "plain bf16 operands, one MFMA per MAC, fp32 accumulate; peak = dense bf16 matrix peak; the kernel is bound by the "
                        "per-step hand-off chain, not by MFMA issue"}
    return {"kernel": "gru_seq_fwd_cluster_x3_kernel<2, NS = %d>" % (2 if h2 else 3), "bound": "mfma", "achieved": flops / dt / 1e12, "peak": PEAK_F32_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": flops / dt / PEAK_F32_MFMA, "traffic": traffic,
            "traffic_source": "profiles/r6_pmc_gru_fwd.txt (rocprofv3 --pmc passes of the same call on the round-6 build; not measured in this run)" if traffic else None,
            "launch_us": dt * 1e6, "flop_per_launch": flops, "us_per_step": dt * 1e6 / T, "mfma_per_mac": 3 if h2 else 6,
            "peak_f16x2": PEAK_BF16_MFMA / 3 / 1e12, "frac_f16x2": flops / dt / (PEAK_BF16_MFMA / 3),
            "peak_bf16x3": PEAK_BF16_MFMA / 6 / 1e12, "frac_bf16x3": flops / dt / (PEAK_BF16_MFMA / 6),
            "note": ("fp32-accurate product issued as 3 fp16 MFMAs per MAC on two-term split operands (round 6); peak = fp32 matrix peak; "
                     "peak_f16x2 = dense 16-bit MFMA peak / 3 is the ceiling of this arithmetic") if h2 else
                    ("fp32-accurate product issued as 6 bf16 MFMAs per MAC on split operands; peak = fp32 matrix peak "
                     "(peak_bf16x3 = dense bf16 MFMA peak / 6, the ceiling of this arithmetic)")}


PMC_TRAFFIC_GRU_FWD = 192.8e6      # bytes per launch: (2 x FETCH_SIZE 55 728 KB + WRITE_SIZE 76 850 KB) x 1024, profiles/r6_pmc_gru_fwd.txt (round-6 build, fp16 x 2: two exchange planes; round 4-5, bf16 x 3: 197.7 MB; gates saved for call g2 only)


def hbm_kernel_roofline(pkg, device):
    """The HBM-shaped kernel with the largest traffic of the iteration: the fused Adam step over the generator's flat parameter slab
    (13.2 M parameters at V = 20 000): reads p, g, m, v and writes p, m, v = 28 bytes per parameter, one launch."""
    ops = pkg.ops
    n = 13_204_940
    p, g, m, v = (torch.randn(n, device=device) * 0.01 for _ in range(4))
    v.abs_()
    step = torch.ones((), device=device, dtype=torch.int32)
    dt = time_kernel(lambda: ops.adam_step(p, g, m, v, 5e-4, 0.5, 0.999, 1e-8, step), iters=20)
    nbytes = 28 * n
    return {"kernel": "adam_kernel", "bound": "hbm", "achieved": nbytes / dt / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
            "frac": nbytes / dt / PEAK_HBM, "traffic": None, "launch_us": dt * 1e6, "bytes_per_launch": nbytes}


def cpu_baseline(batch, seed=1234, warm=2, max_steps=5, budget_s=60.0):
    """The oracle (a CPU port of the reference path, verified against the reference's own outputs) on the host cores, ATen native kernels
    (FAST), fp32, on the SAME inputs the GPU line is timed on (synthetic_batch(batch, seed)): SURVEY 8(d)'s 2 warm-up + 5 timed post-warm-up
    iterations (the timed count shrinks only if 5 would take more than budget_s)."""
    from oracle import ref_model as O
    O.FAST = True
    cores = host_cores()
    torch.set_num_threads(cores)
    gst, dst = O.make_generator_state(0, V, S), O.make_discriminator_state(1)
    text, audio, poses, vid = synthetic_batch(batch, seed, torch.device("cpu"))
    ga, da = {}, {}
    for w in range(warm):
        O.train_iter_gan(gst, dst, ga, da, 11, text, audio, poses, vid, O.Rand(seed=1 + w), fast_gru=True)
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
        O.train_iter_gan(gst, dst, ga, da, 11, text, audio, poses, vid, O.Rand(seed=10 + steps), fast_gru=True)
        steps += 1
    dt = (time.perf_counter() - t0) / steps
    O.FAST = False
    return {"value": batch / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{steps} post-warm-up GAN iterations at batch {batch} after {warm} warm-up, on the benchmark's own synthetic batch (seed {seed}), "
                      f"fp32 ATen kernels, {dt:.2f} s/iter"}


def measure_decode(pkg, args, G, device, batch, steps, warmup, *, graph=True, world=1, rank=0, cpu=True, cpu_budget_s=15.0):
    """Inference path (synthesize.py:82-160): every step is one 34-frame window for `batch` utterances in lock-step
    (seed hand-over + cross-fade on device, hipGraph replay).  Replicas only under --gpus N: no collective.  Returns the JSON line's dict."""
    syn = importlib.import_module(PKG + ".synthesize")
    args.motion_resampling_framerate = 15
    dec = syn.WindowDecoder(args, G, batch, device, graph=graph)
    g = torch.Generator().manual_seed(77 + rank)
    text = torch.zeros(batch, T, dtype=torch.int64)
    text[:, ::5] = torch.randint(4, V, (batch, len(range(0, T, 5))), generator=g)
    audio = (0.1 * torch.randn(batch, dec.audio_len, generator=g)).to(device)
    text, vid = text.to(device), torch.randint(1, S, (batch,), generator=g).to(device)
    dec.seed(None)
    dec.window(text, audio, vid, first=True)
    for _ in range(max(warmup, 1)):
        dec.window(text, audio, vid, first=False)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = dec.window(text, audio, vid, first=False)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    assert bool(torch.isfinite(out).all())
    if rank != 0:
        return None
    fps = world * batch * T * steps / dt
    extra = {"roofline": decode_kernel_roofline(pkg, device, batch)}
    if world == 1 and cpu:
        extra["cpu_baseline"] = cpu_baseline_decode(batch, budget_s=cpu_budget_s)
    return {
        **extra,
        "metric": "inference pose-frames/sec (batched 34-frame synthesis windows)", "value": fps, "unit": "pose-frames/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "latency_us_per_window": dt / steps * 1e6,          # batch 1 = the reference's own call (synthesize.py:131: one utterance)
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthesize.py window loop (BASELINE.json configs[3]): PoseGenerator eval forward + seed hand-over + cross-fade",
                   "utterances_per_gpu": batch, "frames": T, "hipgraph": graph, "parallelism": f"replicas x{world}"},
        "step_roofline": {"bound": "mfma", "achieved": fps / world / T * 0.5217e9 / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                          "frac": fps / world / T * 0.5217e9 / PEAK_F32_MFMA, "frac_f32": fps / world / T * 0.5217e9 / PEAK_F32_MFMA,
                          "frac_bf16x3": fps / world / T * 0.5217e9 / (PEAK_BF16_MFMA / 6), "note": "0.5217 GFLOP per window (SURVEY 8d)"}}


def decode_bench(pkg, a, args, G, device, world, rank):
    d = measure_decode(pkg, args, G, device, a.batch, a.steps, a.warmup, graph=not a.no_graph, world=world, rank=rank, cpu=not a.no_cpu_baseline)
    finish(json.dumps(d) if d is not None else None, world > 1)


def measure_ae(pkg, args, device, batch, steps, warmup, *, graph=True, world=1, rank=0, cpu=True, cpu_budget_s=10.0):
    """FGD autoencoder training (BASELINE.json configs[4], train_feature_extractor.py:54-97): one step = one Adam iteration
    of the pose-mode EmbeddingNet on `batch` synthetic clips.  1.04 M MAC/clip forward: launch/latency bound, clips/s only.
    Replicas only under --gpus N.  Returns the JSON line's dict."""
    fgd = importlib.import_module(PKG + ".fgd")
    torch.manual_seed(0)
    net = pkg.EmbeddingNet(args, D, T, None, None, None, mode="pose").to(device)
    net.train()
    tr = fgd.AutoencoderTrainer(net)
    g = torch.Generator().manual_seed(4321 + rank)
    poses = (0.1 * torch.randn(batch, T, D, generator=g)).to(device)
    cg = None
    if graph:
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                tr.train_iter(poses)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        # thread-local capture mode when a process group exists (N > 1 replicas): RCCL's helper threads must not invalidate the capture
        with torch.cuda.graph(cg, capture_error_mode="thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"):
            loss = tr.train_iter(poses)
    step = cg.replay if cg is not None else (lambda: tr.train_iter(poses))
    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    lv = float((loss if cg is not None else r).item())
    assert lv == lv and lv < 1e6
    if rank != 0:
        return None
    extra = {}
    if world == 1 and cpu:
        extra["cpu_baseline"] = cpu_baseline_ae(batch, budget_s=cpu_budget_s)
    fused = getattr(tr, "_plan", None) is not None
    return {
        **extra,
        # 1.04 M MAC per clip forward, twice that backward: the step is a chain of 18 dependent launches cut at the eight BatchNorms'
        # batch statistics / gradient sums (csrc/ae_step.hip), so the figure that matters is microseconds per launch, not the fraction
        "roofline": {"kernel": "ae_phase_kernel<1..17> + ae_tail_kernel (Adam inside)", "bound": "mfma", "achieved": 6 * 1.04e6 * batch / (dt / steps) / 1e12,
                     "peak": 157.3, "unit": "TFLOP/s", "frac": 6 * 1.04e6 * batch / (dt / steps) / 1e12 / 157.3, "traffic": None,
                     "note": "latency bound by construction: fp32 MFMA peak as the reference ceiling; see latency"},
        "latency": {"launches_per_step": 18 if fused else None, "us_per_step": dt / steps * 1e6,
                    "us_per_launch": dt / steps * 1e6 / 18 if fused else None, "fused": fused},
        "metric": "FGD autoencoder training clips/sec", "value": world * batch * steps / dt, "unit": "clips/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "gesture_autoencoder training iteration (BASELINE.json configs[4]): pose-mode EmbeddingNet fwd+bwd+Adam",
                   "batch_per_gpu": batch, "frames": T, "pose_dim": D, "hipgraph": graph, "parallelism": f"replicas x{world}"},
        "loss": lv}


def ae_bench(pkg, a, args, device, world, rank):
    d = measure_ae(pkg, args, device, a.batch, a.steps, a.warmup, graph=not a.no_graph, world=world, rank=rank, cpu=not a.no_cpu_baseline)
    finish(json.dumps(d) if d is not None else None, world > 1)


def step_roofline(clips_per_s_per_gpu, epoch, dtype):
    """The whole iteration against the matrix ceilings: the fp32 matrix peak (fp32 is the arithmetic delivered) and the ceilings of the arithmetic
    actually issued.  Round 6: the stacked forward's products, the forward recurrence, the GRU layers' input and weight gradients are three fp16
    MFMAs per MAC on two-term split operands (fp16 x 2: dense 16-bit peak / 3); the backward recurrence and the text encoder's backward products
    are still six bf16 MFMAs per MAC (bf16 x 3: peak / 6) -- both fractions are printed, the truth lies between them; the plain-bf16 tier (one
    MFMA per MAC) is priced against the dense bf16 peak."""
    flop = FLOP_PER_CLIP if epoch > 10 else 2.101e9
    ach = clips_per_s_per_gpu * flop
    d = {"bound": "mfma", "achieved": ach / 1e12, "unit": "TFLOP/s", "frac_f32": ach / PEAK_F32_MFMA, "frac_f16x2": ach / (PEAK_BF16_MFMA / 3),
         "frac_bf16x3": ach / (PEAK_BF16_MFMA / 6), "peak_f32": PEAK_F32_MFMA / 1e12, "peak_f16x2": PEAK_BF16_MFMA / 3 / 1e12,
         "peak_bf16x3": PEAK_BF16_MFMA / 6 / 1e12,
         "note": "whole iteration, algorithmic " + ("2.735" if epoch > 10 else "2.101") + " GFLOP/clip (SURVEY 8d), per GPU"}
    if dtype == "bf16":
        d.update(peak=PEAK_BF16_MFMA / 1e12, frac=ach / PEAK_BF16_MFMA, frac_bf16=ach / PEAK_BF16_MFMA)
    else:
        d.update(peak=PEAK_F32_MFMA / 1e12, frac=ach / PEAK_F32_MFMA)
    return d


def secondary_lines(pkg, a, device, cpu_train):
    """Short runs of the other BASELINE.json configurations, appended to the default single-GPU line so that the driver's one command times
    them too: configs[1] as written ("training bf16": the plain-bf16 operand tier), configs[3] (synthesis windows at 128
    utterances and at the reference's own single utterance), configs[4] (autoencoder training).  Each carries its own ms_per_step, roofline
    and cpu_baseline; everything here runs AFTER the headline's timed region."""
    out = {}
    t_all = time.perf_counter()

    def guarded(name, fn):
        """a secondary run must never cost the headline its line: its failure is reported in its own slot"""
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            try:
                torch.cuda.synchronize()
            except Exception:
                pass

    # ---- configs[1], bf16 operand tier: same iteration, one bf16 MFMA per product in every big product (fp32 master weights, accumulators,
    # BatchNorm, Adam); a fresh trainer because the math mode is baked into a captured graph
    def bf16_tier():
        pkg.ops.set_math_mode("bf16")
        try:
            args, G, Dn = build(pkg, device, seed=0)
            tr = pkg.GanTrainer(G, Dn, args)
            text, audio, poses, vid = synthetic_batch(a.batch, 1234, device)
            step = pkg.GraphedGanStep(tr, a.epoch, text, audio, poses, vid, warmup_iters=2)
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            n = 40
            t0 = time.perf_counter()
            for _ in range(n):
                losses = step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ld = losses.to_dict()
            assert all(v == v and abs(v) < 1e6 for v in ld.values()), ld
            cps = a.batch * n / dt
            return {"metric": "training clips/sec (34-frame, 27-dim pose), post-warm-up GAN iteration", "value": cps, "unit": "clips/s", "n_gpus": 1,
                    "steps": n, "warmup": 10, "ms_per_step": dt / n * 1e3, "dtype": "bf16",
                    "config": {"workload": "multimodal_context GAN training iteration, bf16 operand tier (BASELINE.json configs[1] as written)",
                               "batch_per_gpu": a.batch, "hipgraph": True},
                    "tolerance": "losses 2e-2, gradients 5e-2, FGD within 1 % of the fp32 tier (tests/test_engine_gpu.py bf16 tier tests)",
                    "step_roofline": step_roofline(cps, a.epoch, "bf16"), "roofline": dominant_kernel_roofline(pkg, device, a.batch),
                    "cpu_baseline": cpu_train, "losses": ld}
        finally:
            pkg.ops.set_math_mode("f32")
    guarded("train_bf16", bf16_tier)
    # ---- configs[3]: synthesis windows, 128 utterances in lock-step and the reference's single utterance
    def decode(batch, steps, budget):
        args, G, _ = build(pkg, device, seed=0)
        G.eval()
        return measure_decode(pkg, args, G, device, batch, steps, 10, cpu=cpu_train is not None, cpu_budget_s=budget)
    guarded("decode_b128", lambda: decode(128, 60, 3.0))
    guarded("decode_b1", lambda: decode(1, 100, 2.0))
    # ---- configs[4]: FGD autoencoder training
    guarded("ae_train", lambda: measure_ae(pkg, make_args(), device, 128, 200, 20, cpu=cpu_train is not None, cpu_budget_s=3.0))
    try:
        pkg.ops.check_async_errors()
    except Exception as e:
        out["async_error"] = str(e)
    # ---- configs[2]'s code path on this one rank: graph segments + RCCL collectives (a world of one: every exchange still launches), in a
    # child process because the queue configuration must be in the environment before the first HIP call.  Reported beside the plain rate so
    # that the N = 1 point of a scaling curve is known to start from HERE, not from the headline
    try:
        env = dict(os.environ)
        env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--force-ddp", "--steps", "60", "--warmup", "15", "--no-cpu-baseline",
                            "--batch", str(a.batch)], env=env, capture_output=True, text=True, timeout=180)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out["ddp_single_rank"] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "ddp": d.get("ddp"),
                                  "config": {"workload": "the data-parallel code path (BASELINE.json configs[2]) on one rank: bench.py --force-ddp"}}
    except Exception as e:                       # never fail the headline line over the secondary probe
        out["ddp_single_rank"] = {"error": f"{type(e).__name__}: {e}"}
    out["wall_s"] = time.perf_counter() - t_all
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="clips per GPU")
    ap.add_argument("--epoch", type=int, default=11, help="> loss_warmup (10) = full GAN iteration")
    ap.add_argument("--mode", choices=("train", "decode", "ae"), default="train",
                    help="train: GAN training iteration (headline metric); decode: BASELINE.json configs[3], batched 34-frame "
                         "synthesis windows with device-side seed hand-over / cross-fade, pose-frames/sec")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32: exact fp32 matrix cores (parity ~1e-6). bf16: forward / input-gradient GEMMs feed the matrix cores "
                         "with bf16 operands, fp32 accumulate, fp32 everywhere else (tolerance 2e-2 / 5e-2)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--host-input", action="store_true",
                    help="feed every iteration from host memory through data.DeviceBatchFeeder (PCIe-inclusive rate for DESIGN.md; "
                         "never the headline value)")
    ap.add_argument("--no-feed-overlap", action="store_true", help="--host-input: copy straight into the step's inputs on the compute stream")
    ap.add_argument("--host-records", action="store_true",
                    help="feed every iteration from RAW stored samples through data.DeviceRecordFeeder: host packs records, one H2D copy, the "
                         "per-sample assembly of SpeechMotionDataset.__getitem__ + collate runs on the device (never the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short runs of the other BASELINE configurations that the default single-GPU line appends (`secondary`)")
    ap.add_argument("--force-ddp", action="store_true", help="run the data-parallel code path (graph segments + RCCL) even with one rank")
    ap.add_argument("--deterministic", action="store_true",
                    help="tg_set_deterministic(1): fixed-order combines everywhere, bit-reproducible runs (cost line for DESIGN.md; never the headline value)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (no GPU call has been made in this process, none will be)
        rc, line = launch_ranks(a.gpus, sys.argv[1:])
        if line is not None:
            print(line, flush=True)
        sys.exit(rc if rc != 0 else (0 if line is not None else 1))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} in the environment (unset it to let bench.py start its own ranks, "
                 f"or launch with torch.distributed.run --nproc-per-node {a.gpus})")
    if world > 1 or a.force_ddp:
        # before the first HIP call: hardware queues for RCCL's stream beside the compute stream (ddp.configure_environment explains)
        importlib.import_module(PKG + ".ddp").configure_environment()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    pkg = importlib.import_module(PKG)
    pkg._lib.load()
    pkg.ops.set_math_mode(a.dtype)
    pkg.ops.set_deterministic(a.deterministic)

    grad_sync = None
    if world > 1 or a.force_ddp:
        import torch.distributed as dist
        if "GPU_MAX_HW_QUEUES" not in os.environ:
            print("bench.py: GPU_MAX_HW_QUEUES is unset -- ddp.configure_environment() must run before the first HIP call", file=sys.stderr)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        quiet_stdout()
        ddp = importlib.import_module(PKG + ".ddp")
        opts = ddp.process_group_options()
        if opts is not None:
            dist.init_process_group("nccl", device_id=device, pg_options=opts)
        else:
            dist.init_process_group("nccl", device_id=device)
        grad_sync = ddp.GradSync()

    args, G, Dn = build(pkg, device, seed=0)                 # same seed on every rank: identical replicas
    if a.mode == "decode":
        return decode_bench(pkg, a, args, G, device, world, rank)
    if a.mode == "ae":
        return ae_bench(pkg, a, args, device, world, rank)
    trainer = pkg.GanTrainer(G, Dn, args, grad_sync=grad_sync)
    if grad_sync is not None:
        ddp.broadcast_parameters([trainer.G.slab.ensure(), trainer.D.slab.ensure()])
    text, audio, poses, vid = synthetic_batch(a.batch, 1234 + rank, device)

    ddp_info = None
    if a.no_graph:
        step = lambda: trainer.train_iter(a.epoch, text, audio, poses, vid)
        if grad_sync is not None:
            ddp_info = {"ranks": grad_sync.world, "collectives": "eager", "rejected": []}
    elif grad_sync is not None:
        # graph segments with eager RCCL collectives between them (default; TG_DDP_CAPTURE=1 tries collectives captured inside the graph first):
        # the chosen form must reproduce an eager iteration on every rank, otherwise all ranks fall back in-process (train_gan.checked_ddp_step)
        tg = importlib.import_module(PKG + ".train_gan")
        step, ddp_info = tg.checked_ddp_step(trainer, a.epoch, text, audio, poses, vid, warmup_iters=2,
                                             log=(lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None)
    else:
        step = pkg.GraphedGanStep(trainer, a.epoch, text, audio, poses, vid, warmup_iters=2)
    feeder = None
    if a.host_input:
        assert not a.no_graph, "--host-input drives the captured step"
        data = importlib.import_module(PKG + ".data")
        feeder = data.DeviceBatchFeeder(*step.static, static_flat=step.static_flat, overlap=not a.no_feed_overlap)
        pool = [tuple(t.cpu() for t in synthetic_batch(a.batch, 4321 + 17 * i + rank, device)) for i in range(4)]
        pool = [(t, p_, au, v) for (t, au, p_, v) in pool]          # feeder.put(text, vec, audio, vid)
        feeder.put(*pool[0])
        plain_step = step

        def step(_k=[0]):
            feeder.ready()
            _k[0] += 1
            feeder.put(*pool[_k[0] % len(pool)])                    # next batch's host->device copy overlaps this iteration
            return plain_step()
    if a.host_records:
        assert not a.no_graph and not a.host_input, "--host-records drives the captured step"
        data = importlib.import_module(PKG + ".data")
        lang = pkg.Vocab("words")
        for i in range(V - 4):
            lang.index_word(f"w{i}")
        spk = pkg.Vocab.speakers(S)
        ds = data.SyntheticSpeechMotionDataset(4 * a.batch, lang, spk, seed=7 + rank)
        raw_pool = [[ds.raw(i) for i in range(k * a.batch, (k + 1) * a.batch)] for k in range(4)]
        rfeeder = data.DeviceRecordFeeder(*step.static, lang, spk)
        rfeeder.put(raw_pool[0])
        plain_step2 = step

        def step(_k=[0]):
            rfeeder.ready()
            _k[0] += 1
            rfeeder.put(raw_pool[_k[0] % len(raw_pool)])            # packing + H2D of the next batch overlap this iteration
            return plain_step2()
    for _ in range(a.warmup):
        losses = step()

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_dict = losses.to_dict()
    assert all(v == v and abs(v) < 1e6 for v in loss_dict.values()), loss_dict     # finite

    line = None
    if rank == 0:
        clips_per_s = world * a.batch * a.steps / dt
        out = {
            "metric": "training clips/sec (34-frame, 27-dim pose), " + ("post-warm-up GAN iteration" if a.epoch > 10 else "warm-up-phase iteration (epoch <= loss_warmup)")
                      + (" [host-fed, PCIe-inclusive]" if a.host_input else "") + (" [host-fed raw records, device-side batch assembly]" if a.host_records else "") + (" [deterministic mode]" if a.deterministic else ""),
            "value": clips_per_s, "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "multimodal_context GAN training iteration (BASELINE.json configs[1]), " + ("epoch > loss_warmup" if a.epoch > 10 else "epoch <= loss_warmup"),
                       "batch_per_gpu": a.batch, "global_batch": world * a.batch, "frames": T, "pose_dim": D,
                       "audio_samples": A, "n_words": V, "n_speakers": S - 1, "hipgraph": not a.no_graph,
                       "parallelism": f"dp{world}"},
            "step_roofline": step_roofline(clips_per_s / world, a.epoch, a.dtype),
            "losses": loss_dict,
        }
        if ddp_info is not None:
            out["ddp"] = {"ranks": ddp_info["ranks"], "collectives": ddp_info["collectives"],
                          "rejected": [f"{m}: {w}" for m, w in ddp_info["rejected"]]}
        out["roofline"] = dominant_kernel_roofline(pkg, device, a.batch)
        out["roofline_hbm"] = hbm_kernel_roofline(pkg, device)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.batch, seed=1234 + rank)
        plain = (world == 1 and grad_sync is None and a.dtype == "f32" and a.epoch > 10 and not (a.no_graph or a.host_input or a.host_records or a.deterministic))
        if plain and not a.no_secondary and not a.no_cpu_baseline:
            del step
            try:
                out["secondary"] = secondary_lines(pkg, a, device, out.get("cpu_baseline"))
            except Exception as e:               # (every item inside is guarded too: the headline line never depends on a secondary run)
                out["secondary"] = {"error": f"{type(e).__name__}: {e}"}
        line = json.dumps(out)
    finish(line, world > 1 or a.force_ddp)


if __name__ == "__main__":
    main()
