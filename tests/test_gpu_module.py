"""GPU parity of the drop-in module (MatryoshkaTranscoder + FusedAdam / stock Adam) against the
fixtures the reference produced (tests/golden) and against the CPU oracle."""
import copy

import pytest
import torch

from conftest import SMALL_CASES, load_golden, tol
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol, atol, msg=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    ok = err <= atol + rtol * b.abs()
    assert bool(ok.all()), f"{msg}: {int((~ok).sum())} of {ok.numel()} off, max abs {err.max():.3e}"


def _model(fx, cfg, state_key="init"):
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    cfg = dict(cfg, device=DEV)
    m = MatryoshkaTranscoder(cfg)
    m.load_state_dict({k: v.to(DEV) for k, v in fx[state_key].items()})
    return m, cfg


NOAUX = [c for c in SMALL_CASES if "aux" not in c]


@pytest.mark.parametrize("name", NOAUX)
def test_init_and_state_dict_layout(name, gpu_lib):
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    fx, cfg = load_golden(name)
    m = MatryoshkaTranscoder(dict(cfg, device=DEV))
    sd = m.state_dict()
    assert list(sd.keys()) == list(fx["init"].keys()) == ["b_enc", "W_enc", "W_dec", "b_dec", "threshold"]
    for k, v in fx["init"].items():
        assert sd[k].shape == v.shape and sd[k].dtype == v.dtype
        assert torch.equal(sd[k].cpu(), v), k            # same CPU RNG stream as the reference
    assert [n for n, _ in m.named_parameters()] == ["b_enc", "W_enc", "W_dec", "b_dec"]
    assert m.W_enc.t().is_contiguous()                    # feature-major storage behind the reference shape


@pytest.mark.parametrize("name", SMALL_CASES)
@pytest.mark.parametrize("optim", ["fused", "fused_project", "stock"])
def test_three_steps_vs_reference_fixture(name, optim, gpu_lib):
    from matryoshka_transcoder_amd.training import FusedAdam, create_scheduler
    fx, cfg = load_golden(name)
    dt = cfg["dtype"]
    f32 = dt == torch.float32
    m, cfg = _model(fx, cfg)
    m.num_batches_not_active = fx["cnt0"].clone().to(DEV)
    if optim == "stock":
        opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    else:
        opt = FusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    sched = None
    if name == "fp32_sched":
        sched = create_scheduler(opt, dict(cfg, scheduler_type="warmup_decay", warmup_steps=2), 8)
    rtol, atol = tol(dt)
    for i, s in enumerate(fx["steps"]):
        out = m(s["x_source"].to(DEV), s["x_target"].to(DEV))
        assert list(out.keys()) == list(s["out"].keys())
        for k, ref in s["out"].items():
            assert out[k].dtype == ref.dtype, (k, out[k].dtype, ref.dtype)
            if k == "transcoder_out":
                continue
            _close(out[k], ref, rtol, atol, f"step {i} {k}")
        if f32:
            _close(out["transcoder_out"], s["out"]["transcoder_out"], 1e-4, 1e-5, "transcoder_out")
            assert torch.equal(m.num_batches_not_active.cpu(), s["cnt_after_fwd"])
        _close(m.threshold, s["threshold_after_fwd"], rtol, atol, "threshold")
        assert abs(opt.param_groups[0]["lr"] - s["lr"]) < 1e-12
        out["loss"].backward(retain_graph=True)
        if f32:
            for n, p in m.named_parameters():
                _close(p.grad, s["grads"][n], 1e-4, 1e-7, f"step {i} grad {n}")
        if optim == "fused_project":
            opt.step(project=True)
        else:
            m.make_decoder_weights_and_grad_unit_norm()
            if f32:
                _close(m.W_dec.grad, s["W_dec_grad_projected"], 1e-4, 1e-7, "projected grad")
            opt.step()
        opt.zero_grad()
        if sched is not None:
            sched.step()
        for n, p in m.named_parameters():
            ref = s["params_after"][n]
            if f32:
                _close(p, ref, 0, 1e-2 * cfg["lr"] + 1e-7, f"{n} after step {i}")
            else:
                ulp = (ref.float().abs() * 2 ** -7).clamp_min(2 ** -14)
                n_off = int(((p.detach().float().cpu() - ref.float()).abs() > ulp).sum())
                assert n_off <= 0.03 * ref.numel() + 4 * (i + 1), f"{n} step {i}: {n_off} of {ref.numel()} off > 1 ulp"


@pytest.mark.parametrize("name", ["fp32_plain", "fp32_unit", "bf16_plain", "bf16_unit", "fp32_cross"])
def test_encode_decode_vs_reference_fixture(name, gpu_lib):
    fx, cfg = load_golden(name)
    dt = cfg["dtype"]
    f32 = dt == torch.float32
    ed = fx["encdec"]
    state = {k: v for k, v in fx["steps"][-1]["params_after"].items()}
    for mode in ("train", "eval"):
        m, _ = _model({"s": state}, cfg, "s")
        m.train(mode == "train")
        x2 = ed["x2"].to(DEV)
        with torch.no_grad():
            e2 = m.encode(x2)
            d2 = m.decode(e2)
            thr = m.threshold.clone()
            e3 = m.encode(x2.reshape(2, x2.shape[0] // 2, -1))
        ref = ed[mode]
        assert e2.shape == ref["enc2"].shape and e3.shape == ref["enc3"].shape
        if f32:
            _close(e2, ref["enc2"], 1e-5, 1e-6, f"{mode} enc2")
            _close(e3, ref["enc3"], 1e-5, 1e-6, f"{mode} enc3")
            _close(d2, ref["dec2"], 1e-4, 1e-5, f"{mode} dec2")
        else:
            same = ((e2.cpu() != 0) == (ref["enc2"] != 0)).float().mean()
            assert same > 0.999
            _close(d2, ref["dec2"], 5e-2, 5e-2, f"{mode} dec2")
        _close(thr, ref["thr_after_enc2"], *tol(dt), f"{mode} threshold")
        if "dec3" in ref:
            with torch.no_grad():
                d3 = m.decode(e3.reshape(-1, e3.shape[-1]))
            if f32:
                _close(d3, ref["dec3"], 1e-4, 1e-5, f"{mode} dec3")


@pytest.mark.parametrize("name", ["fp32_plain", "fp32_unit", "bf16_plain"])
def test_eval_forward_vs_reference_fixture(name, gpu_lib):
    fx, cfg = load_golden(name)
    ev = fx["encdec"]["eval_forward"]
    m, _ = _model({"s": ev["state"]}, cfg, "s")
    m.num_batches_not_active = ev["cnt_before"].clone().to(DEV)
    m.eval()
    with torch.no_grad():
        out = m(ev["x_source"].to(DEV), ev["x_target"].to(DEV))
    rtol, atol = tol(cfg["dtype"])
    for k in ("loss", "l2_loss", "l1_loss", "l0_norm", "l1_norm", "fvu", "fvu_min", "fvu_max", "fvu_mean"):
        _close(out[k], ev["out"][k], rtol, atol, k)
    assert torch.equal(m.threshold.cpu(), ev["threshold"])
    if cfg["dtype"] == torch.float32:
        assert torch.equal(m.num_batches_not_active.cpu(), ev["cnt_after"])


def test_no_cpu_path(gpu_lib):
    from matryoshka_transcoder_amd import _lib
    fx, cfg = load_golden("fp32_plain")
    m, _ = _model(fx, cfg)
    with pytest.raises(_lib.MtcError):
        m(fx["x_source0"], fx["x_target0"])             # CPU tensors: refuse, never fall back


def test_full_size_properties_gemma2_2b(gpu_lib):
    """BASELINE config #2 at full size: size-independent invariants of the select / decode / step."""
    import bench
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    from matryoshka_transcoder_amd.training import FusedAdam
    cfg, B = bench.make_cfg("gemma2-2b", torch.bfloat16, DEV)
    m = MatryoshkaTranscoder(cfg)
    opt = FusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    pool = bench.batch_pool(B, cfg["act_size"], torch.bfloat16, DEV, n=2)
    D, k = cfg["prefix_sizes"][-1], cfg["top_k"]
    losses = []
    for i in range(4):
        out = m(*pool[i % 2])
        res = m._last_res
        rp = res.sel.row_ptr.cpu().long()
        nnz = int(rp[-1])
        assert nnz == k * B and float(out["l0_norm"]) == k          # exactly k*B selected
        cols = res.sel.col_idx.cpu()[:nnz].long()
        rows = torch.repeat_interleave(torch.arange(B), rp[1:] - rp[:-1])
        key = rows * D + cols
        assert bool((key[1:] > key[:-1]).all())                      # sorted, unique flat indices
        vals = res.sel.vals.cpu()[:nnz].float()
        tau = torch.tensor([int(res.sel.state.cpu()[4])], dtype=torch.int32).view(torch.float32)
        assert float(vals.min()) == float(tau) and bool((vals > 0).all())
        # the selection is exactly {v > tau} plus ties: check against a dense recount of one row block
        out["loss"].backward()
        gW = m.W_dec.grad
        fired = torch.zeros(D, dtype=torch.bool); fired[cols] = True
        assert bool((gW[~fired.to(DEV)] == 0).all())                 # silent features: zero gradient rows
        opt.step(project=True)
        opt.zero_grad()
        losses.append(float(out["loss"].detach().float()))
        assert float(out["fvu_min"]) <= float(out["fvu_mean"]) <= float(out["fvu_max"])
    norms = m.W_dec.detach().float().norm(dim=-1)
    assert float((norms - 1).abs().max()) < 2e-2                     # rows stay ~unit norm (post-Adam drift only)
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("name", ["fp32_plain", "bf16_unit", "fp32_cross", "fp32_aux", "bf16_aux"])
def test_fused_train_step_equals_autograd_path(name, gpu_lib):
    """FusedAdam.train_step (gradient rows consumed in registers) must give bit-identical weights, moments and
    outputs to forward -> backward -> step(project=True) on dense gradients."""
    from matryoshka_transcoder_amd.training import FusedAdam
    fx, cfg = load_golden(name)
    ma, cfg_d = _model(fx, cfg)
    mb, _ = _model(fx, cfg)
    for m in (ma, mb):
        m.num_batches_not_active = fx["cnt0"].clone().to(DEV)
    oa = FusedAdam(ma, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    ob = FusedAdam(mb, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    for s in fx["steps"]:
        xs, xt = s["x_source"].to(DEV), s["x_target"].to(DEV)
        out_a = ma(xs, xt)
        out_a["loss"].backward()
        oa.step(project=True)
        oa.zero_grad()
        out_b = ob.train_step(xs, xt)
        for k in out_a:
            assert out_b[k].dtype == out_a[k].dtype
            assert torch.equal(out_b[k].detach(), out_a[k].detach()), k
        for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            assert torch.equal(pa, pb), n
            assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]), n
            assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), n
        assert torch.equal(ma.threshold, mb.threshold)
        assert torch.equal(ma.num_batches_not_active, mb.num_batches_not_active)


@pytest.mark.parametrize("name", ["fp32_plain", "bf16_unit", "bf16_aux"])
def test_two_stream_step_is_bit_identical(name, gpu_lib):
    """FusedAdam.train_step issues the CSC build and the b_dec update on a side stream at large sizes; forced on here
    at fixture size, it must give the same bits as the single-stream step (ordering is by events only)."""
    from matryoshka_transcoder_amd.training import FusedAdam
    fx, cfg = load_golden(name)
    ma, _ = _model(fx, cfg)
    mb, _ = _model(fx, cfg)
    ma._OVERLAP_MIN_ELEMS = 0
    mb._overlap_ok = False
    for m in (ma, mb):
        m.num_batches_not_active = fx["cnt0"].clone().to(DEV)
    oa = FusedAdam(ma, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    ob = FusedAdam(mb, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    for rep in range(4):
        for s in fx["steps"]:
            xs, xt = s["x_source"].to(DEV), s["x_target"].to(DEV)
            out_a, out_b = oa.train_step(xs, xt), ob.train_step(xs, xt)
            for k in out_a:
                assert torch.equal(out_a[k], out_b[k]), k
    assert ma._side_stream is not None and getattr(mb, "_side_stream", None) is None
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), n
        assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), n
    assert torch.equal(ma.num_batches_not_active, mb.num_batches_not_active)


def test_two_stream_aux_step_is_bit_identical_at_config2_shapes(gpu_lib):
    """The same at the shapes where the second stream is actually used (config #2: B 4096, D 36864, A 2304, bf16) with the
    auxiliary loss active over ~5 000 dead features: CSC build in two halves with the dead list between them, operand
    transposes and the E_enc product on the side stream beside products that leave CUs idle, b_dec update beside the
    gradient launches -- against the one-stream step, eight steps, every output and every parameter bit for bit (a
    missing event shows up as a difference: the schedules overlap different launches)."""
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    A, D, B = 2304, 36864, 4096
    cfg = dict(seed=49, prefix_sizes=[2304, 6912, 16128, 36864], act_size=A, dict_size=D, top_k=96, top_k_aux=512,
               aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.bfloat16, device=DEV,
               input_unit_norm=False, lr=3e-4, beta1=0.9, beta2=0.99)
    ma = MatryoshkaTranscoder(cfg)
    mb = MatryoshkaTranscoder(cfg)
    mb.load_state_dict(ma.state_dict())
    mb._overlap_ok = False
    forced = torch.arange(5, D, 7, device=DEV)
    for m in (ma, mb):
        with torch.no_grad():
            m.W_enc[:, forced] *= 0.01
        cnt = m.num_batches_not_active
        cnt[forced] = 20.0
        m.num_batches_not_active = cnt
    oa = FusedAdam(ma, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    ob = FusedAdam(mb, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    g = torch.Generator(DEV).manual_seed(7)
    for step in range(8):
        xs = torch.randn(B, A, generator=g, device=DEV).to(torch.bfloat16)
        xt = torch.randn(B, A, generator=g, device=DEV).to(torch.bfloat16)
        out_a, out_b = oa.train_step(xs, xt), ob.train_step(xs, xt)
        for k in out_a:
            assert torch.equal(out_a[k], out_b[k]), (step, k)
        assert float(out_a["aux_loss"]) > 0
    torch.cuda.synchronize()
    assert ma._side_stream is not None and getattr(mb, "_side_stream", None) is None
    assert ma._last_res.aux is not None and ma._last_res.aux.get("acts_auxT") is not None      # the side-stream forms ran
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), n
        assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]), n
        assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), n
    assert torch.equal(ma.num_batches_not_active, mb.num_batches_not_active)
    assert torch.equal(ma.threshold, mb.threshold)


def test_device_guard_and_device_mismatch(gpu_lib, monkeypatch):
    """libmtc launches on HIP's current device with raw pointers: every public entry point makes the parameters' GPU
    current first (free when it already is) and refuses inputs that live elsewhere (a caller whose current
    device differs from cfg['device'] would otherwise launch on the wrong GPU)."""
    import contextlib
    from matryoshka_transcoder_amd import _lib
    fx, cfg = load_golden("fp32_plain")
    m, _ = _model(fx, cfg)
    assert isinstance(m._on_device(), contextlib.nullcontext)           # already current: no guard object, no cost
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)        # a caller sitting on another GPU
    guard = m._on_device()
    assert isinstance(guard, torch.cuda.device) and guard.idx == 0      # would switch to the parameters' GPU
    monkeypatch.undo()
    x = fx["steps"][0]["x_source"].to(DEV)

    class Elsewhere:                                                    # a tensor-like on another GPU
        device = torch.device("cuda:1"); is_cuda = True; dtype = x.dtype
    with pytest.raises(_lib.MtcError, match="cuda:1"):
        m._check_inputs(Elsewhere())
    with pytest.raises(_lib.MtcError):
        m._on_device(Elsewhere())
    out = m(x, fx["steps"][0]["x_target"].to(DEV))                      # and the guarded path still computes
    _close(out["l2_loss"], fx["steps"][0]["out"]["l2_loss"], 1e-4, 1e-6, "l2 through the guard")


def test_checkpoint_resume_is_bit_identical(gpu_lib, tmp_path):
    """results.py: the side file carries the optimizer state and the dead counters; a run resumed from
    checkpoints/step_N.{pt,extra.pt} continues exactly like the uninterrupted one (the reference saves weights only)."""
    from matryoshka_transcoder_amd.results import CleanResultsManager
    from matryoshka_transcoder_amd.training import FusedAdam
    fx, cfg = load_golden("bf16_unit")
    steps = fx["steps"]

    def fresh():
        m, c = _model(fx, cfg)
        return m, FusedAdam(m, lr=c["lr"], betas=(c["beta1"], c["beta2"]))

    def run(m, opt, batches):
        out = None
        for s in batches:
            out = opt.train_step(s["x_source"].to(DEV), s["x_target"].to(DEV))
        return out

    ma, oa = fresh()
    run(ma, oa, steps[:2])
    mgr = CleanResultsManager(str(tmp_path))
    exp = mgr.create_experiment_dir("m", 0, 3)
    mgr.save_checkpoint(exp, ma, dict(cfg), 2, {"loss": 0.0}, optimizer=oa)
    out_a = run(ma, oa, steps[2:] + steps[:2])
    mb, ob = fresh()
    extra = mgr.load_checkpoint(exp, mb, "step_2", optimizer=ob)
    assert extra["step"] == 2 and "optimizer" in extra
    out_b = run(mb, ob, steps[2:] + steps[:2])
    for k in out_a:
        assert torch.equal(out_a[k], out_b[k]), k
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), n
        assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), n
        assert oa.state[pa]["step"] == ob.state[pb]["step"]
    assert torch.equal(ma.num_batches_not_active, mb.num_batches_not_active)
    # the weights file alone is a reference-layout checkpoint
    sd = torch.load(exp / "checkpoints" / "step_2.pt", weights_only=True)
    assert list(sd.keys()) == ["b_enc", "W_enc", "W_dec", "b_dec", "threshold"] and sd["W_enc"].is_contiguous()
