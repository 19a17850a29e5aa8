#!/usr/bin/env python3
"""Headline benchmark: train tokens/s (forward + backward + unit-norm projection + Adam) of the
Matryoshka transcoder at BASELINE.json's Gemma-2-2B shapes (A 2304, D 36864, k 96, B 4096,
cumulative prefixes [2304, 6912, 16128, 36864]) on synthetic Gaussian activations.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--workload gemma2-2b|gpt2]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  N > 1 shards the dictionary by features over the ranks (same
global batch: strong scaling) with RCCL collectives through torch.distributed.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (A, group sizes as BASELINE.json lists them, top_k, batch)
    "gemma2-2b": (2304, [2304, 4608, 9216, 20736], 96, 4096),
    "gpt2": (768, [768, 1536, 3072, 6912], 64, 1024),
    # the other BASELINE configs (quoted there on 8 GPUs; they fit one MI355X) -- swept by tools/sweep.py
    "gemma2-9b": (3584, [1792, 1792, 3584, 7168, 14336, 28672], 128, 8192),
    "gemma2-27b": (4608, [4608, 9216, 18432, 41472], 184, 16384),
    "gemma2-27b-2x": (4608, [9216, 18432, 36864, 82944], 184, 16384),
    # the README lists taken literally as cumulative prefix_sizes (what the reference code would build from them)
    "gpt2-literal": (768, [768, 768, 1536, 3840], 64, 1024),
    "gemma2-2b-literal": (2304, [2304, 2304, 4608, 11520], 96, 4096),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16
MFMA_F32_PEAK_TF = 157.3


def cumulative(groups):
    out, run = [], 0
    for g in groups:
        run += g
        out.append(run)
    return out


def make_cfg(workload, dtype, device, unit_norm=False):
    A, groups, k, B = WORKLOADS[workload]
    prefix = cumulative(groups)
    return dict(seed=49, prefix_sizes=prefix, act_size=A, dict_size=prefix[-1], top_k=k, top_k_aux=512,
                aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0, dtype=dtype, device=device,
                input_unit_norm=unit_norm, lr=3e-4, beta1=0.9, beta2=0.99, batch_size=B), B


def batch_pool(B, A, dtype, device, n=8):
    """n synthetic (source, target) batches, generated on the device they live on."""
    pool = []
    dev = torch.device(device)
    for i in range(n):
        g = torch.Generator(dev).manual_seed(1000 + i)
        xs = torch.randn(B, A, generator=g, device=dev).to(dtype)
        xt = torch.randn(B, A, generator=g, device=dev).to(dtype)
        pool.append((xs, xt))
    return pool


def cpu_baseline(workload, dtype, budget_rows=None, steps=3):
    """Reference algorithm on the host cores: oracle.DenseStep (same ATen call sequence as the reference's step;
    timed next to the imported reference itself in profiles/r02_cpu_reference.json), at the metric's own batch size
    (config #2 for 3 steps).  The sample is bounded: the step count shrinks if a step is slow."""
    from oracle import ref_cpu as R
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg, B = make_cfg(workload, dtype, "cpu")
    budget_rows = budget_rows or B
    A = cfg["act_size"]
    model = R.DenseStep(cfg)
    pool = batch_pool(budget_rows, A, dtype, "cpu", n=2)
    print(f"[bench] cpu_baseline: warm-up step on {cores} host threads ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    model.step(*pool[0])                      # warm-up
    warm = time.perf_counter() - t0
    steps = max(1, min(steps, int(25.0 / max(warm, 1e-3))))          # keep the timed sample near 10-30 s
    print(f"[bench] cpu_baseline: warm-up {warm:.1f} s, timing {steps} step(s)", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for i in range(steps):
        model.step(*pool[(i + 1) % 2])
    dt = (time.perf_counter() - t0) / steps
    return {"value": budget_rows / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps at B={budget_rows} rows of the {workload} shapes (D={cfg['prefix_sizes'][-1]}, "
                      f"k={cfg['top_k']}), {str(dtype).split('.')[-1]}, torch {torch.__version__} CPU, "
                      f"oracle.DenseStep (dense eager port of the reference step)"}


def hbm_calibration(device, gib=1.0, reps=10):
    """Streaming-copy rate of this GPU in this process: libmtc's own 16-byte-per-lane copy kernel (mtc_copy16) over a
    buffer far larger than the Infinity Cache, read + write bytes per second.  SURVEY 8d: the HBM-bound stages are shown
    against the 8 TB/s peak AND against this."""
    import ctypes as C
    from matryoshka_transcoder_amd import _lib
    lib = _lib.load()
    n = int(gib * (1 << 30))
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    src.zero_()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        _lib.check(lib.mtc_copy16(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, s), "mtc_copy16")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.mtc_copy16(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, s), "mtc_copy16")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"kernel": "mtc::copy16_kernel (16 B per lane, non-temporal)", "bytes_copied": n, "ms": ms,
            "GBps_read_plus_write": 2.0 * n / (ms * 1e-3) / 1e9}


def stage_times(tc, opt, pool, reps=5):
    """Per-stage device time of one step, HIP events on the launch stream (torch's current stream).  Stages are timed
    one after the other on ONE stream: the side-stream overlap of the production step (CSC build under the decode,
    b_dec update under the gradient kernels) is switched off here, so the stage times add up to slightly more than
    the measured step."""
    from matryoshka_transcoder_amd import engine
    hp = tc.hot_path()
    overlap_was, side_was = getattr(tc, "_overlap_ok", True), getattr(tc, "_side_stream", None)
    tc._overlap_ok, tc._side_stream = False, None
    names, evs = [], []
    orig = {}

    def wrap(name):
        fn = getattr(hp, name)
        orig[name] = fn

        def inner(*a, **k):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            names.append(name); evs.append((e0, e1))
            return r
        setattr(hp, name, inner)

    for n in ("normalize", "encode", "select_topk", "threshold_ema", "build_csc", "decode", "decode_bwd_vals", "reduce_scalars",
              "backward", "fused_backward_adam", "colsum", "unitnorm_adam", "adam_flat", "aux_dead_list", "aux_forward_dense",
              "aux_extra_rows"):
        wrap(n)
    # the feature-sharded classes sequence their select / decode + exchanges themselves (sharded.py): timed as a whole,
    # on the main stream (waits for the transport's stream included)
    model_wrapped = {}
    if hasattr(tc, "comm"):
        for n, label in (("_select", "select (sharded: 4 phases + 3 collectives)"),
                         ("_decode_and_losses", "decode + exchanges (sharded)")):
            fn = getattr(tc, n)
            model_wrapped[n] = fn

            def inner(*a, _fn=fn, _label=label, **k):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _fn(*a, **k)
                e1.record()
                names.append(_label); evs.append((e0, e1))
                return r
            object.__setattr__(tc, n, inner)
    acc = {}
    for r in range(reps):
        names.clear(); evs.clear()
        xs, xt = pool[r % len(pool)]
        opt.train_step(xs, xt)
        torch.cuda.synchronize()
        for n, (e0, e1) in zip(names, evs):
            acc.setdefault(n, []).append(e0.elapsed_time(e1))
    for n, fn in orig.items():
        setattr(hp, n, fn)
    for n in model_wrapped:
        object.__delattr__(tc, n)
    tc._overlap_ok, tc._side_stream = overlap_was, side_was
    out = {}
    # several calls of one stage inside a step (adam_flat x3) are summed per step
    for n, v in acc.items():
        per_step = len(v) // reps
        out[n] = sum(sorted(v)[: len(v)]) / reps if per_step else 0.0
    return out


def launch_ranks(n):
    """One child process per rank over torch.distributed.run (rendezvous on 127.0.0.1, a free port); rank 0's JSON
    line comes out on this process's stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def launch_test():
    """MTC_BENCH_LAUNCH_TEST=1 (tests/test_host_logic.py, no GPU): every rank joins a gloo group and rank 0 prints the
    one JSON line with the world size it saw -- proves the launcher, the rendezvous and the single-line contract."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"launch_test": True, "n_gpus": dist.get_world_size(), "ranks_seen": int(t.item())}))
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-steps", type=int, default=32,
                    help="untimed steps BEFORE the warm-up steps (recorded in the line as `settle_steps`): the chip's power "
                         "management needs ~30 ms of load to settle -- the encoder launch runs 545, 542, 636, 649, 613, 600, 557, "
                         "... 518 us over the first 20 steps after an idle period (rocprofv3 trace) -- and a 5-step warm-up ends "
                         "inside that transient; 0 switches it off")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="gemma2-2b", choices=list(WORKLOADS))
    ap.add_argument("--unit-norm", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="use the feature-sharded classes even on one GPU (overhead check)")
    ap.add_argument("--comm", default="torch", choices=["torch", "libmtc"],
                    help="transport of the sharded step's exchanges: torch.distributed (RCCL) or libmtc's own RCCL wrappers")
    ap.add_argument("--world1-exchanges", action="store_true",
                    help="--sharded at world 1: call the transport for the (identity) exchanges too instead of aliasing the buffers")
    ap.add_argument("--exchange-blocks", type=int, default=0,
                    help="row blocks the sharded step pipelines its critical exchange over (0 = default: 4 at world > 1, 1 at world 1)")
    ap.add_argument("--aux-active", action="store_true",
                    help="SURVEY 8d: the step with the auxiliary dead-feature loss active (every 7th feature forced dead: "
                         "tiny encoder column, counter preset); adds an `aux` block (its GEMMs on mtc_gemm_nt) to the line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start N ranks (one per GPU) through torch.distributed.run and pass its exit code on.
        # This parent has not touched the GPU (nothing above initialises HIP) and never does.
        sys.exit(launch_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch one rank per GPU "
              f"(python bench.py --gpus N does that itself)", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("MTC_BENCH_LAUNCH_TEST"):
        sys.exit(launch_test())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    # one rank per GPU; with fewer GPUs than ranks (rehearsal on a one-GPU box) the ranks share the cards
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    collective_backend = None
    if world > 1 or args.sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        # RCCL prints a version banner on stdout when its communicator comes up; the contract is ONE JSON line on
        # stdout, so fd 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > max(torch.cuda.device_count(), 1):
                # rehearsal only: RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so ranks that share a
                # card exchange through gloo (device tensors staged through the host, sharded._Comm)
                dist.init_process_group("gloo")
            else:
                # (a collective that never completes aborts the run after 5 minutes instead of the default 10)
                import datetime
                dist.init_process_group("nccl", device_id=torch.device(device), timeout=datetime.timedelta(seconds=300))
            dist.barrier()
            torch.cuda.synchronize()
            collective_backend = dist.get_backend()
            world = dist.get_world_size()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    cfg, B = make_cfg(args.workload, dtype, device, args.unit_norm)
    A, D, k = cfg["act_size"], cfg["prefix_sizes"][-1], cfg["top_k"]
    if world > 1 or args.sharded:
        from matryoshka_transcoder_amd.sharded import ShardedMatryoshkaTranscoder, ShardedFusedAdam
        if args.comm == "libmtc" and collective_backend == "nccl":
            cfg["comm"] = "libmtc"
        if args.exchange_blocks > 0:
            cfg["exchange_blocks"] = args.exchange_blocks
        cfg["world1_exchanges"] = bool(args.world1_exchanges)
        tc = ShardedMatryoshkaTranscoder(cfg)
        opt = ShardedFusedAdam(tc, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    else:
        from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
        from matryoshka_transcoder_amd.training import FusedAdam
        tc = MatryoshkaTranscoder(cfg)
        opt = FusedAdam(tc, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    n_forced = 0

    def force_dead():
        Dl = tc.hot_path().D                               # the local dictionary of a shard
        forced = torch.arange(5, Dl, 7, device=device)
        with torch.no_grad():
            tc.W_enc[:, forced] *= 0.01                    # does not win the global TopK until the aux loss has revived it
        cnt = tc.num_batches_not_active
        cnt[forced] = float(cfg["n_batches_to_dead"])
        tc.num_batches_not_active = cnt
        return int(forced.numel())

    if args.aux_active:
        n_forced = force_dead()
    # 96 distinct batches, walked without a restart between settle / warm-up / timed steps: training data does not recur within
    # a few steps, and a model that has just taken its first Adam steps on batch 0 meets that batch again with a visibly lower
    # TopK threshold -- the select's carried bound then fails and the step runs the full-size select (+0.26 ms).  With 8
    # batches that happened on steps 8 and 16 of a run; the line reports how the timed steps were served (`select_paths`)
    pool = batch_pool(B, A, dtype, device, n=96)

    def step(i):
        xs, xt = pool[i % len(pool)]
        # forward chain + {k-sparse gradient -> unit-norm projection -> Adam} per parameter row (training.py:172-179)
        return opt.train_step(xs, xt)

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    n_settle = max(args.settle_steps, 0)
    for i in range(n_settle):
        step(i)
    for i in range(args.warmup):
        step(n_settle + i)
    fence()
    # the dominant launch (the encoder) is timed LIVE: HIP events on its own stream around every 8th launch of the timed
    # region, read after the closing fence (no synchronisation inside the region; an event pair costs the stream ~6 us, so
    # every launch would take 0.8 % off the measured throughput, every 8th takes 0.1 %)
    enc_events, enc_calls = [], [0]
    hp_live = tc.hot_path()
    enc_orig = hp_live.encode
    if rank == 0 and world == 1 and not args.sharded:
        def enc_timed(*a, **k):
            enc_calls[0] += 1
            if enc_calls[0] % 8 != 1:
                return enc_orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = enc_orig(*a, **k)
            e1.record()
            enc_events.append((e0, e1))
            return r
        hp_live.encode = enc_timed
    n_dead_start = int((tc._cnt >= cfg["n_batches_to_dead"]).sum()) if args.aux_active else 0
    carry0 = hp_live.carry_state(device).cpu().tolist() if (world == 1 and not args.sharded) else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(n_settle + args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    hp_live.encode = enc_orig
    select_paths = None
    if carry0 is not None:
        c1 = hp_live.carry_state(device).cpu().tolist()
        # how the select served the TIMED steps: carried bound + window (one pass over the flagged granules), or full-size passes
        select_paths = {"carried_bound": int(c1[4] - carry0[4]), "full_passes": int(c1[5] - carry0[5])}
    enc_live_ms = sum(a.elapsed_time(b) for a, b in enc_events) / len(enc_events) if enc_events else None
    n_dead_end = int((tc._cnt >= cfg["n_batches_to_dead"]).sum()) if args.aux_active else 0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    value = B * args.steps / elapsed

    result = {
        "metric": "train tokens/sec (fwd+bwd+Adam), Gemma-2-2B d=36864 k=96" if args.workload == "gemma2-2b"
        else f"train tokens/sec (fwd+bwd+Adam), {args.workload} d={D} k={k}",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": n_settle,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.workload}: act {A}, dict {D}, cumulative prefixes {cfg['prefix_sizes']}, "
                               f"top_k {k}, batch {B}, input_unit_norm {args.unit_norm}"
                               + (f", auxiliary loss ACTIVE ({n_forced} features forced dead)" if args.aux_active else ""),
                   "global_batch": B, "parallelism": f"dictionary sharded x{world}" if (world > 1 or args.sharded) else "single GPU",
                   "collective_world": world if (world > 1 or args.sharded) else None,
                   "collective_backend": collective_backend,
                   "final_loss": float(out["loss"].detach().float()), "final_l0": float(out["l0_norm"].detach())},
    }
    if select_paths is not None:
        result["select_paths"] = select_paths
    if world > 1 or args.sharded:
        # After the timed region, on EVERY rank (collectives): a few instrumented steps -- HIP events around each exchange
        # on the stream it runs on, and around the stages of the sharded step on the main stream -- so that a scaling
        # curve can be read from its own lines.  RCCL knobs are passed through the environment and recorded.
        n_diag = 5
        tc.comm.timing = {}
        st = stage_times(tc, opt, pool, reps=n_diag)
        fence()
        timing, tc.comm.timing = tc.comm.timing, None
        exchanges = {}
        for tag, lst in timing.items():
            ms = [a.elapsed_time(b) for a, b, _ in lst]
            nbytes = lst[0][2]
            exchanges[tag] = {"calls_per_step": len(lst) / n_diag, "ms_per_call": sum(ms) / len(ms), "payload_bytes_per_call": nbytes,
                              "GBps": nbytes / (sum(ms) / len(ms) * 1e-3) / 1e9 if sum(ms) > 0 else None}
        result["stages"] = {n: {"ms": t} for n, t in st.items()}
        result["exchanges"] = exchanges
        result["config"]["exchange_blocks"] = tc.exchange_blocks(B)
        result["config"]["transport"] = type(tc.comm).__name__
        result["config"]["rccl_env"] = {k2: v for k2, v in os.environ.items() if k2.startswith(("NCCL_", "RCCL_"))}
    if rank == 0 and world == 1 and not args.sharded:
        if args.aux_active:
            force_dead()          # the timed steps revived most of the dead set: per-stage times are taken on a full one again
        st = stage_times(tc, opt, pool, reps=2 if args.aux_active else 5)
        s = 2 if dtype == torch.bfloat16 else 4
        nnz = k * B
        flop_enc = 2.0 * B * A * D
        # algorithmic work per launch: "hbm" = bytes that must cross HBM at least
        # once (weights / moments / activations streamed, outputs written); "gather" = bytes the k-sparse row
        # gathers request on top of that, served by L2 / Infinity Cache (reported, not priced against HBM)
        # Two prices per HBM-bound stage.  "compulsory": bytes that must cross HBM at least once whatever the
        # implementation (tables streamed once, outputs written).  "algorithmic" (SURVEY 8d): the k-sparse row gathers
        # counted per selected entry -- nnz rows of A elements per sweep -- plus the streams; the gathers are served by
        # L2 / Infinity Cache where the table fits, so this rate can exceed what HBM alone delivers.
        algo = {
            "encode": ("mfma", flop_enc, 0),
            "select_topk": ("hbm", B * D * s + nnz * (4 + s) + B * 4, 0),
            "decode": ("hbm", D * A * s + nnz * (4 + s) + 3 * B * A * s, nnz * A * s - D * A * s),      # one sweep: d_vals is deferred
            "fused_backward_adam": ("hbm", 2 * D * A * 6 * s + 2 * B * A * s + nnz * 20, 2 * nnz * A * s - 2 * B * A * s),
            "backward": ("hbm", 2 * D * A * s + 2 * B * A * s + nnz * 12, 2 * nnz * A * s - 2 * B * A * s),
        }
        calib = hbm_calibration(device)
        # stage -> (launches per step, kernel name in the committed PMC summary) for the stages priced below
        launches = {"encode": (1, "mtc::encode_bf16_mfma256p_kernel<false, true>"), "decode": (1, "mtc::decode_fwd_kernel<1, 5>"),
                    "fused_backward_adam": (2, "mtc::fused_grad_adam_kernel<1, 1, 5, 1, 1, 0>")}     # (+ fused_enc_sliced_kernel)
        kernels = {}
        for n, t_ms in st.items():
            entry = {"ms": t_ms, "launches": launches.get(n, (None, None))[0]}
            if n in algo:
                kind, q, gather = algo[n]
                if kind == "mfma":
                    peak = MFMA_BF16_PEAK_TF if dtype == torch.bfloat16 else MFMA_F32_PEAK_TF
                    ach = q / (t_ms * 1e-3) / 1e12
                    entry.update(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak)
                else:
                    ach = q / (t_ms * 1e-3) / 1e9
                    alg = (q + max(gather, 0)) / (t_ms * 1e-3) / 1e9
                    entry.update(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                                 compulsory_bytes=int(q), algorithmic_bytes=int(q + max(gather, 0)),
                                 frac_compulsory=ach / HBM_PEAK_GBS, algorithmic_GBps=alg, frac_algorithmic=alg / HBM_PEAK_GBS,
                                 frac_algorithmic_of_copy_rate=alg / calib["GBps_read_plus_write"])
            kernels[n] = entry
        # dominant KERNEL = longest single launch (the fused stage is two launches of one kernel template)
        per_launch = lambda n: kernels[n]["ms"] / (kernels[n]["launches"] or 1)
        dom = max((n for n in kernels if "bound" in kernels[n]), key=per_launch)
        if dom == "encode" and enc_live_ms:
            # the roofline block prices the average over the launches of the TIMED steps (the per-stage table below it comes
            # from a separate single-stream pass of a few steps and moves a few per cent with the chip's power state)
            ach = flop_enc / (enc_live_ms * 1e-3) / 1e12
            kernels["encode"].update(ms=enc_live_ms, achieved=ach, frac=ach / kernels["encode"]["peak"],
                                     ms_stage_pass=st["encode"], timed=f"HIP events around {len(enc_events)} launches of the timed steps (every 8th)")
        # HBM-side bytes per launch of that kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE,
        # profiles/README.md); only valid for the workload / dtype they were taken on
        traffic = None
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[23]*_pmc_traffic.json")))    # newest committed summary
        traffic_src = None
        if args.workload == "gemma2-2b" and args.dtype == "bf16" and not args.unit_norm and pmc_files:
            with open(pmc_files[-1]) as f:
                pmc = json.load(f)["kernels"]
            want = launches.get(dom, (None, ""))[1]
            hit = pmc.get(want) or next((v for k2, v in pmc.items() if want and k2.split("<")[0] == want.split("<")[0]), {})
            traffic = hit.get("hbm_bytes_per_launch")
            traffic_src = os.path.relpath(pmc_files[-1], ROOT)
        result["roofline"] = {"kernel": launches.get(dom, (None, dom))[1] or dom, "stage": dom,
                              **{k2: kernels[dom][k2] for k2 in ("bound", "achieved", "peak", "unit", "frac")},
                              "launch_ms": per_launch(dom), "traffic": traffic, "traffic_unit": "bytes/launch",
                              "traffic_source": traffic_src}
        result["stages"] = kernels
        result["hbm_calibration"] = calib
        if args.aux_active:
            # the branch's four products over the dead set (sae.py:336-341 and its autograd), all on mtc_gemm_nt:
            # forward B x A_t x n_pad, backward 3 products of the same volume; priced against the dense bf16 MFMA peak
            n_dead = int((tc._cnt >= cfg["n_batches_to_dead"]).sum())      # after the per-stage passes (the branch revives features)
            n_pad = tc.hot_path()._aux_bucket(n_dead, D)
            t_aux = (st.get("aux_forward_dense", 0.0) + st.get("aux_extra_rows", 0.0)) * 1e-3
            flop = 4 * 2.0 * B * n_pad * A
            peak = MFMA_BF16_PEAK_TF if dtype == torch.bfloat16 else MFMA_F32_PEAK_TF
            result["aux"] = {"forced_dead": n_forced, "n_dead_at_start_of_timed_steps": n_dead_start,
                             "n_dead_at_end_of_timed_steps": n_dead_end, "n_dead": n_dead, "n_pad": n_pad, "aux_loss": float(out["aux_loss"].float()),
                             "stage_ms": {k2: st.get(k2) for k2 in ("aux_dead_list", "aux_forward_dense", "aux_extra_rows")},
                             "roofline": {"bound": "mfma", "achieved": flop / max(t_aux, 1e-9) / 1e12, "peak": peak,
                                          "unit": "TFLOP/s", "frac": flop / max(t_aux, 1e-9) / 1e12 / peak,
                                          "note": "4 products of 2*B*n_pad*A FLOP each over the select / fill / transpose / "
                                                  "residual / mask launches of the branch as well (whole-branch time)"}}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload, dtype)
    if rank == 0:
        print(json.dumps(result))
    if world > 1 or args.sharded:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
