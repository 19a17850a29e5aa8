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


def _train(model, opt, batches, fused=False):
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
        m = ShardedMatryoshkaTranscoder(cfg)
        opt = ShardedFusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
        log = _train(m, opt, _batches(dt, n=6 if aux else 3), fused)
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
    ref_log = _train(ref, FusedAdam(ref, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"])), _batches(dt, n=6 if aux else 3))
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
            assert n_off <= 0.03 * v.numel() + 8, (k, n_off, v.numel())


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
