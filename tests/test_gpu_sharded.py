"""Feature-sharded step (2 ranks sharing the one GPU of the test box, gloo as the transport) against
the single-GPU path on the same weights and batches: losses, FVU, threshold, counters, and the
gathered weights after 3 optimizer steps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cfg(dt, unit, aux=False):
    # aux=True: features that stay silent for 2 batches count as dead, so the auxiliary loss (sae.py:318-347)
    # is active from the third step on
    return dict(seed=49, prefix_sizes=[256, 768, 1792, 4096], act_size=256, top_k=32, top_k_aux=64,
                aux_penalty=1 / 32, n_batches_to_dead=2 if aux else 20, l1_coeff=1e-3, dtype=dt, device="cuda:0",
                input_unit_norm=unit, lr=3e-4, beta1=0.9, beta2=0.99,
                aux_mode=aux if isinstance(aux, str) else "gemm")     # aux = "gemm" | "sparse": form of the aux branch


def _batches(dt, n=3, B=192):
    out = []
    for i in range(n):
        g = torch.Generator("cpu").manual_seed(1000 + i)
        out.append((torch.randn(B, 256, generator=g).to(dt), torch.randn(B, 256, generator=g).to(dt)))
    return out


def _selection_keys(model):
    """Selected (row, GLOBAL feature) pairs of the model's last forward as sorted int64 keys row * D + feature."""
    sel = model._last_res.sel
    rp = sel.row_ptr.cpu().long()
    nnz = int(rp[-1])
    cols = sel.col_idx[:nnz].cpu().long()
    rows = torch.repeat_interleave(torch.arange(rp.numel() - 1), rp[1:] - rp[:-1])
    layout = getattr(model, "layout", None)
    if layout is not None:
        cols = layout.global_index(layout.rank)[cols]
    return torch.sort(rows * model.total_dict_size + cols).values


def _train(model, opt, batches, fused=False, first_sel=None):
    log = []
    for xs, xt in batches:
        if fused:                      # what bench.py runs: FusedAdam.train_step, no materialised gradients
            out = opt.train_step(xs.cuda(), xt.cuda())
        else:
            out = model(xs.cuda(), xt.cuda())
            out["loss"].backward()
            opt.step(project=True)
            opt.zero_grad()
        log.append({k: float(v.detach().float()) for k, v in out.items() if v.dim() == 0})
        if first_sel is not None and not first_sel:
            first_sel.append(_selection_keys(model))
            first_sel.append(model._cnt.clone())
        # the reconstruction too (the sharded fused step gathers it by a collective of its own, beside the backward kernels)
        xh = out["transcoder_out"].detach().float()
        log[-1]["transcoder_out |mean|"] = float(xh.abs().mean())
        log[-1]["transcoder_out rms"] = float(xh.pow(2).mean().sqrt())
    return log


def _worker(rank, world, port, dt_name, unit, fused, aux, ret, backend="gloo", comm=None):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from matryoshka_transcoder_amd.sharded import ShardedFusedAdam, ShardedMatryoshkaTranscoder
        dt = getattr(torch, dt_name)
        cfg = _cfg(dt, unit, aux)
        if comm:
            cfg["comm"] = comm
        cfg["world1_exchanges"] = True      # a one-rank group still calls the transport (identity exchanges over RCCL)
        m = ShardedMatryoshkaTranscoder(cfg)
        opt = ShardedFusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
        first = []
        log = _train(m, opt, _batches(dt, n=6 if aux else 3), fused, first)
        # first step (identical weights on both sides): this rank's selection in global ids, gathered over the ranks
        keys = first[0].cuda()
        n_keys = m.comm.all_gather(torch.tensor([keys.numel()], device="cuda"))
        padded = torch.full((int(n_keys.max()),), -1, dtype=torch.int64, device="cuda")
        padded[:keys.numel()] = keys
        all_keys = m.comm.all_gather(padded).cpu()
        cnt1 = m._gather_features(first[1], 0).cpu()
        sd = {k: v.cpu() for k, v in m.state_dict().items()}
        cnt = m.gathered_dead_counters().cpu()
        m.eval()
        with torch.no_grad():
            xe = _batches(dt, n=1, B=40)[0][0].cuda()
            feats = m.encode(xe)
            recon = m.decode(feats)
        m.train()
        if rank == 0:
            ret["feats"], ret["recon"] = feats.cpu(), recon.cpu()
            ret["sel1"] = torch.sort(all_keys[all_keys >= 0]).values
            ret["cnt1"] = cnt1
            ret["log"] = log
            ret["sd"] = sd
            ret["cnt"] = cnt
        ret[rank] = "ok"
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()


@pytest.mark.parametrize("dt_name,unit,fused,aux", [("float32", False, False, False), ("bfloat16", True, False, False),
                                                    ("float32", False, True, False), ("bfloat16", True, True, False),
                                                    ("float32", False, True, "gemm"), ("float32", True, False, "gemm"),
                                                    ("bfloat16", True, True, "gemm"), ("float32", True, True, "sparse"),
                                                    ("bfloat16", True, True, "sparse")])
def test_sharded_matches_single_gpu(dt_name, unit, fused, aux, gpu_lib):
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    dt = getattr(torch, dt_name)
    cfg = _cfg(dt, unit, aux)
    ref = MatryoshkaTranscoder(cfg)          # same form of the aux branch (dense GEMMs or row gathers) on both sides
    ref_first = []
    ref_log = _train(ref, FusedAdam(ref, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"])), _batches(dt, n=6 if aux else 3),
                     first_sel=ref_first)
    if aux:
        assert ref_log[-1]["aux_loss"] > 0, "the test configuration must make the auxiliary loss active"
    ref_sd = {k: v.cpu() for k, v in ref.state_dict().items()}
    ref_cnt = ref.num_batches_not_active.cpu()
    ref.eval()
    with torch.no_grad():
        xe = _batches(dt, n=1, B=40)[0][0].cuda()
        ref_feats = ref.encode(xe)
        ref_recon = ref.decode(ref_feats).cpu()
        ref_feats = ref_feats.cpu()
    torch.cuda.synchronize()

    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dt_name, unit, fused, aux, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(170)
    for r in range(2):
        assert ret.get(r) == "ok", ret.get(r)
    f32 = dt == torch.float32
    rtol = 1e-5 if f32 else 2e-2
    # first step (same weights on both sides): the selection -- ties at tau included -- and the dead counters are the
    # single-GPU ones exactly: ties are granted in ascending global flat index whatever the world size
    assert torch.equal(ret["sel1"], ref_first[0].cpu()), "sharded selection differs from the single-GPU selection"
    assert torch.equal(ret["cnt1"], ref_first[1].cpu())
    for a, b in zip(ret["log"], ref_log):
        for k in b:
            assert abs(a[k] - b[k]) <= rtol * abs(b[k]) + 1e-7, (k, a[k], b[k])
    assert torch.equal(ret["cnt"], ref_cnt) or not f32
    # eval-mode encode() / decode() of the sharded model (collectives) against the single-GPU model: the
    # weights differ by the training tolerance above, so features are compared where both fire
    assert ret["feats"].shape == ref_feats.shape and ret["recon"].shape == ref_recon.shape
    both = (ret["feats"] > 0) & (ref_feats > 0)
    assert both.sum() >= 0.9 * (ref_feats > 0).sum()
    assert torch.allclose(ret["feats"][both].float(), ref_feats[both].float(), rtol=5e-2, atol=1e-3)
    err = (ret["recon"].float() - ref_recon.float()).abs().max()
    assert err <= (2e-3 if f32 else 1e-1) * max(1.0, float(ref_recon.float().abs().max())), float(err)
    for k, v in ref_sd.items():
        got = ret["sd"][k]
        assert got.shape == v.shape
        if f32:
            assert torch.allclose(got, v, rtol=0, atol=1e-2 * cfg["lr"] + 1e-7), k
        else:
            ulp = (v.float().abs() * 2 ** -7).clamp_min(2 ** -14)
            n_off = int(((got.float() - v.float()).abs() > ulp).sum())
            assert n_off <= 0.002 * v.numel() + 8, (k, n_off, v.numel())


@pytest.mark.parametrize("dt_name,unit,aux,comm", [("float32", False, False, None), ("bfloat16", True, "gemm", None),
                                                   ("bfloat16", True, "gemm", "libmtc"), ("float32", False, False, "libmtc")])
def test_sharded_over_rccl_world1_matches_single_gpu(dt_name, unit, aux, comm, gpu_lib):
    """The transport bench.py uses at N > 1: RCCL through torch.distributed on device tensors (sharded._Comm without
    host staging: reduce_scatter_tensor / all_gather_into_tensor / all_reduce).  RCCL refuses two ranks on one GPU, so
    on the one-GPU test box the group has ONE rank; the exchanges are then identities and the fused step must
    reproduce the single-GPU classes' losses.  comm = "libmtc": the same exchanges through libmtc's own RCCL wrappers
    (mtc_comm_*, cfg["comm"] = "libmtc") instead of torch.distributed."""
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    dt = getattr(torch, dt_name)
    cfg = _cfg(dt, unit, aux)
    ref = MatryoshkaTranscoder(cfg)
    n = 6 if aux else 3
    ref_log = _train(ref, FusedAdam(ref, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"])), _batches(dt, n=n), fused=True)
    ref_sd = {k: v.cpu() for k, v in ref.state_dict().items()}
    torch.cuda.synchronize()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), dt_name, unit, True, aux, ret, "nccl", comm))
    p.start()
    p.join(170)
    assert ret.get(0) == "ok", ret.get(0)
    f32 = dt == torch.float32
    rtol = 1e-5 if f32 else 2e-2
    for a, b in zip(ret["log"], ref_log):
        for k in b:
            assert abs(a[k] - b[k]) <= rtol * abs(b[k]) + 1e-7, (k, a[k], b[k])
    for k, v in ref_sd.items():
        got = ret["sd"][k]
        assert got.shape == v.shape
        if f32:
            assert torch.allclose(got, v, rtol=0, atol=1e-2 * cfg["lr"] + 1e-7), k


# ---------------------------------------------------------------------------------------------- config-#2 shapes
def _cfg2(dt, blocks):
    # BASELINE.json configs 2 / 3: Gemma-2-2B shapes (A 2304, D 36864, k 96, B 4096), dictionary sharded
    return dict(seed=49, prefix_sizes=[2304, 6912, 16128, 36864], act_size=2304, top_k=96, top_k_aux=512,
                aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0, dtype=dt, device="cuda:0",
                input_unit_norm=False, lr=3e-4, beta1=0.9, beta2=0.99, exchange_blocks=blocks,
                world1_exchanges=True)       # a one-rank group still calls the transport (identity exchanges over RCCL)


def _batches2(dt, n=3, B=4096, A=2304):
    out = []
    for i in range(n):
        g = torch.Generator("cpu").manual_seed(1000 + i)
        out.append((torch.randn(B, A, generator=g).to(dt), torch.randn(B, A, generator=g).to(dt)))
    return out


def _worker2(rank, world, port, backend, blocks, out_path, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from matryoshka_transcoder_amd.sharded import ShardedFusedAdam, ShardedMatryoshkaTranscoder
        dt = torch.bfloat16
        cfg = _cfg2(dt, blocks)
        m = ShardedMatryoshkaTranscoder(cfg)
        assert m.exchange_blocks(4096) == blocks
        opt = ShardedFusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
        first = []
        log = _train(m, opt, _batches2(dt), True, first)
        keys = first[0].cuda()
        n_keys = m.comm.all_gather(torch.tensor([keys.numel()], device="cuda"))
        padded = torch.full((int(n_keys.max()),), -1, dtype=torch.int64, device="cuda")
        padded[:keys.numel()] = keys
        all_keys = m.comm.all_gather(padded).cpu()
        cnt1 = m._gather_features(first[1], 0).cpu()
        sd = {k: v.cpu() for k, v in m.state_dict().items()}
        carry = m._sel_bufs[next(iter(m._sel_bufs))]["carry"].cpu().tolist()
        if rank == 0:
            torch.save({"sd": sd, "sel1": torch.sort(all_keys[all_keys >= 0]).values, "cnt1": cnt1, "log": log,
                        "carry": carry}, out_path)
        ret[rank] = "ok"
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()


@pytest.mark.parametrize("backend,world,blocks", [("nccl", 1, 4), ("gloo", 2, 2)])
def test_sharded_config2_shapes_match_single_gpu(backend, world, blocks, gpu_lib, tmp_path):
    """The sharded classes at the shapes of BASELINE configs 2 / 3 (bf16) against the single-GPU classes: RCCL itself with a
    one-rank group (the exchanges are identities, the batch still walks 4 row blocks), and two ranks sharing the card
    over gloo.  First step: selection and dead counters EXACTLY equal (ties in global flat-index order); three fused
    steps: losses within the bf16 tolerance, >= 99.9 % of the weights within one bf16 ulp."""
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    dt = torch.bfloat16
    cfg = _cfg2(dt, blocks)
    ref = MatryoshkaTranscoder(cfg)
    ref_first = []
    ref_log = _train(ref, FusedAdam(ref, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"])), _batches2(dt), True, ref_first)
    ref_sd = {k: v.cpu() for k, v in ref.state_dict().items()}
    torch.cuda.synchronize()
    del ref
    torch.cuda.empty_cache()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    out_path = str(tmp_path / "sharded.pt")
    procs = [ctx.Process(target=_worker2, args=(r, world, port, backend, blocks, out_path, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)
    got = torch.load(out_path)
    assert torch.equal(got["sel1"], ref_first[0].cpu()), "sharded selection differs from the single-GPU selection"
    assert torch.equal(got["cnt1"], ref_first[1].cpu())
    # steps 2 and 3 were served by the carried bound (counter of calls served by the compact window path)
    assert got["carry"][4] >= 2, got["carry"]
    for a, b in zip(got["log"], ref_log):
        for k in b:
            assert abs(a[k] - b[k]) <= 2e-2 * abs(b[k]) + 1e-7, (k, a[k], b[k])
    for k, v in ref_sd.items():
        w = got["sd"][k]
        assert w.shape == v.shape
        ulp = (v.float().abs() * 2 ** -7).clamp_min(2 ** -14)
        n_off = int(((w.float() - v.float()).abs() > ulp).sum())
        assert n_off <= 1e-3 * v.numel() + 8, (k, n_off, v.numel())
