"""GPU parity of the side paths: the device sample tables against the reference
collector's fixtures, and the SAEBench adapters' encode -> decode against the reference adapter's."""
import json
import os

import numpy as np
import pytest
import torch

from side_checks import ADAPTER_CASES, COLLECTOR_CASES, WordTokenizer, check_collector, load_side

pytestmark = pytest.mark.gpu


def _collector(fx):
    from matryoshka_transcoder_amd.activation_samples import ActivationSampleCollector
    return ActivationSampleCollector(max_samples_per_feature=fx["max_samples"], context_size=fx["context_size"],
                                     storage_threshold=fx["storage_threshold"])


@pytest.mark.parametrize("name", COLLECTOR_CASES)
def test_collector_tables_match_reference(gpu_lib, name, tmp_path):
    fx = load_side("collector", name)
    col = _collector(fx)
    tok = WordTokenizer()
    for r in fx["rounds"]:
        col.collect_batch_samples(r["acts"].cuda(), r["tokens"].cuda(), tok, feature_indices=fx["tracked"])
    check_collector(fx, lambda f: [s.to_dict() for s in col.get_top_samples(f)], col.get_feature_statistics,
                    col.get_all_statistics(), dict(col.feature_activation_counts.items()),
                    dict(col.feature_max_activations.items()))
    # every table row stays a permutation of its slots
    M = fx["max_samples"]
    assert torch.equal(torch.sort(col._slot, dim=1).values.cpu(),
                       torch.arange(M, dtype=torch.int32).repeat(col._slot.shape[0], 1))
    # files on disk: same names as the reference wrote for its top 6 features (ties in the maximum aside)
    col.save_samples(str(tmp_path), top_k_features=6, samples_per_feature=3)
    if name != "bf16_ties":
        assert sorted(os.listdir(tmp_path)) == fx["saved_files_top6"]
    summary = json.load(open(tmp_path / "collection_summary.json"))
    assert summary["total_samples_seen"] == fx["overall"]["total_samples_seen"]
    # save -> load into a fresh collector reproduces the saved samples
    col2 = _collector(fx)
    col2._tokenizer = tok
    col2.load_samples(str(tmp_path), num_features=len(fx["features"]))
    for fn in os.listdir(tmp_path):
        if fn.startswith("feature_"):
            data = json.load(open(tmp_path / fn))
            f = data["statistics"]["feature_idx"]
            assert [s.to_dict() for s in col2.get_top_samples(f)] == data["top_samples"]


def test_collector_equals_topk_at_scale(gpu_lib):
    """Size-independent property: after any number of batches a feature's table holds exactly the M largest
    values (> threshold) seen for it, and each stored position / token window is where that value occurred."""
    from matryoshka_transcoder_amd.activation_samples import ActivationSampleCollector
    g = torch.Generator("cpu").manual_seed(5)
    batch, seq, D, M = 32, 64, 512, 100
    col = ActivationSampleCollector(max_samples_per_feature=M, context_size=20, storage_threshold=0.1)
    alls, allt = [], []
    for r in range(3):
        a = torch.rand(batch * seq, D, generator=g)
        a = torch.where(torch.rand(batch * seq, D, generator=g) < 0.2, a, torch.zeros(()))
        a[:, 7] = torch.rand(batch * seq, generator=g) + 0.2           # a feature that fires on every position
        a[:, 9] = 0.0
        t = torch.randint(0, 250000, (batch, seq), generator=g)
        col.collect_batch_samples(a.cuda(), t.cuda(), None)
        alls.append(a)
        allt.append(t.reshape(-1))
    A, T = torch.cat(alls), torch.cat(allt)
    val = col._val.cpu()
    for f in range(D):
        c = A[:, f]
        c = c[c > 0.1]
        want = torch.sort(c, descending=True).values[:M]
        got = val[f][torch.isfinite(val[f])]
        assert torch.equal(got, want), f
    assert col.total_samples_seen == int((A > 0.1).sum())
    assert torch.equal(col._count.cpu(), (A > 0.1).sum(0))
    # stored window: the token at the stored position is the token of a row where the value occurred
    for f in (0, 7, 100, 511):
        for s in col.get_top_samples(f, k=5):
            rows = torch.nonzero(A[:, f] == s.activation_value).flatten()
            assert any(int(T[r]) == s.tokens[-1] and int(r) % seq == s.position for r in rows)
            assert len(s.context_tokens) == min(seq, s.position + 11) - max(0, s.position - 10)


def test_collector_rejects_bad_input(gpu_lib):
    from matryoshka_transcoder_amd import _lib
    from matryoshka_transcoder_amd.activation_samples import ActivationSampleCollector
    with pytest.raises(ValueError):
        ActivationSampleCollector(max_samples_per_feature=1000)
    col = ActivationSampleCollector(4, 4, 0.1)
    with pytest.raises(_lib.MtcError):
        col.collect_batch_samples(torch.zeros(8, 16), torch.zeros(2, 4, dtype=torch.long), None)      # CPU tensor
    with pytest.raises(ValueError):
        col.collect_batch_samples(torch.zeros(9, 16).cuda(), torch.zeros(2, 4, dtype=torch.long), None)
    assert col.get_top_samples(3) == [] and col.get_all_statistics()["total_samples_seen"] == 0


def _npz_adapter(fx, tmp_path):
    from matryoshka_transcoder_amd.sae_bench import NPZTranscoderSpec, build_npz_transcoder_adapter
    sp = fx["spec"]
    path = tmp_path / "ckpt.npz"
    np.savez(path, **{k: v.numpy() for k, v in fx["arrays"].items()})
    spec = NPZTranscoderSpec(path=str(path), model_name="gemma-2-2b", source_layer=sp["source_layer"],
                             source_hook_point="blocks.8.hook_resid_post", target_hook_point="blocks.9.hook_resid_pre",
                             d_in=sp["d_in"], d_out=sp["d_out"], d_sae=sp["d_sae"], target_layer=sp["target_layer"],
                             dtype=sp["dtype"], device="cuda", top_k=sp["top_k"], prefix_sizes=sp["prefix_sizes"])
    return build_npz_transcoder_adapter(spec)


@pytest.mark.parametrize("name", ADAPTER_CASES)
def test_npz_adapter_validate_reconstruction(gpu_lib, name, tmp_path):
    fx = load_side("adapter", name)
    ad = _npz_adapter(fx, tmp_path)
    dt = getattr(torch, fx["spec"]["dtype"])
    for k, v in ad.state_dict().items():
        if k == "W_dec":      # row normalisation runs on the device: last-bit differences of the norm reduction
            assert torch.allclose(v.cpu().float(), fx["state_after_load"][k].float(), rtol=2.0 ** -7 if dt != torch.float32 else 1e-6)
        else:
            assert torch.equal(v.cpu(), fx["state_after_load"][k]), k
    assert list(ad.prefix_sizes) == fx["prefix_sizes"]
    assert ad.config["top_k"] == fx["top_k"] and ad.config["top_k_aux"] == fx["top_k_aux"]
    for k, v in fx["cfg_fields"].items():
        if k != "device":
            assert getattr(ad.cfg, k) == v, k
    assert ad.cfg.device == "cuda"
    assert torch.allclose(ad.get_decoder_row_norms().float().cpu(), fx["row_norms"].float(), rtol=1e-6)
    xs, xt = fx["x_source"].cuda(), fx["x_target"].cuda()
    rel = 1e-4 if dt == torch.float32 else 2e-2                  # stated tolerance: fp32 1e-4, bf16 2e-2
    for mode in ("train", "eval"):
        w = fx[mode]
        ad.train(mode == "train")
        with torch.no_grad():
            ad.threshold.copy_(w["threshold_before"])
        rep = ad.validate_reconstruction(xs, xt)
        assert abs(rep.fvu - w["fvu"]) <= rel * abs(w["fvu"]), (mode, rep.fvu, w["fvu"])
        assert abs(rep.mse - w["mse"]) <= rel * abs(w["mse"]), (mode, rep.mse, w["mse"])
        assert abs(rep.mean_activation - w["mean_activation"]) <= rel * abs(w["mean_activation"])
        enc = ad.encode(xs).cpu()
        assert torch.allclose(ad.threshold.float().cpu(), w["threshold_after"].float(), rtol=rel, atol=1e-7)
        if dt == torch.float32:
            assert torch.equal(enc != 0, w["encoded_after"] != 0)
            assert torch.allclose(enc, w["encoded_after"], rtol=1e-4, atol=1e-6)
        else:
            assert int(((enc != 0) ^ (w["encoded_after"] != 0)).sum()) <= 4      # near-ties at the threshold


def test_build_matryoshka_adapter_from_checkpoint(gpu_lib, tmp_path):
    from matryoshka_transcoder_amd.sae_bench import build_matryoshka_adapter
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    cfg = dict(seed=3, prefix_sizes=[64, 256], dict_size=256, act_size=32, top_k=8, top_k_aux=16, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.float32, device="cuda", model_name="gpt2-small",
               layer=8, source_site="resid_post", seq_len=64)
    tc = MatryoshkaTranscoder(cfg)
    with torch.no_grad():
        tc.W_dec.mul_(1.7)
        tc.threshold.fill_(0.2)
    torch.save({k: v.detach().cpu().contiguous() for k, v in tc.state_dict().items()}, tmp_path / "final.pt")
    ad = build_matryoshka_adapter(cfg, state_dict_path=tmp_path / "final.pt")
    assert ad.cfg.hook_name == "blocks.8.hook_resid_post" and ad.cfg.d_sae == 256 and ad.cfg.context_size == 64
    assert torch.allclose(ad.get_decoder_row_norms(), torch.ones(256, device="cuda"), atol=1e-6)
    assert torch.equal(ad.W_enc.detach(), tc.W_enc.detach()) and float(ad.threshold) == float(tc.threshold)
    ad.eval()
    x = torch.randn(16, 32, device="cuda")
    rep = ad.validate_reconstruction(x, x)
    assert rep.fvu > 0 and rep.mean_activation >= 0


def test_training_loop_collects_samples(gpu_lib, tmp_path):
    """training.py:77-85,138-152,205-211 end to end on synthetic activations with token ids."""
    from matryoshka_transcoder_amd.training import train_transcoder
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    cfg = dict(seed=1, prefix_sizes=[32, 128], act_size=32, top_k=4, top_k_aux=8, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.float32, device="cuda", lr=1e-3, beta1=0.9, beta2=0.99,
               num_tokens=64 * 6, batch_size=64, model_name="toy", layer=0, results_root=str(tmp_path),
               perf_log_freq=1000, checkpoint_freq=1000, save_activation_samples=True, sample_collection_freq=2,
               max_samples_per_feature=8, sample_context_size=4, sample_activation_threshold=0.0,
               top_features_to_save=5, samples_per_feature_to_save=3)

    class Store:
        tokenizer = WordTokenizer()

        def __init__(self):
            self.g = torch.Generator("cpu").manual_seed(0)

        def next_batch(self):
            return (torch.randn(64, 32, generator=self.g).cuda(), torch.randn(64, 32, generator=self.g).cuda())

        def get_batch_with_tokens(self):
            return (torch.randn(4 * 16, 32, generator=self.g).cuda(),
                    torch.randint(0, 1000, (4, 16), generator=self.g).cuda())

    exp = train_transcoder(MatryoshkaTranscoder(cfg), Store(), None, cfg)
    files = sorted(os.listdir(exp / "activation_samples"))
    assert "collection_summary.json" in files and len([f for f in files if f.startswith("feature_")]) == 5
    data = json.load(open(exp / "activation_samples" / [f for f in files if f.startswith("feature_")][0]))
    assert 1 <= len(data["top_samples"]) <= 3 and data["top_samples"][0]["context_text"].startswith("t")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(37, 48, 264), (300, 64, 1024), (1024, 128, 4096)])
def test_fused_threshold_encode_equals_select_path(gpu_lib, dt, shape):
    """Eval-mode encode (sae.py:189-191,363-387): the threshold fused into the encoder epilogue gives bit for bit
    the dense tensor of encoder -> threshold select -> densify, on every encoder kernel (generic, 128^2 bf16 / fp32
    MFMA tiles; the 256^2 kernel shares the epilogue and is covered by the eval fixtures at full size in bench)."""
    from matryoshka_transcoder_amd.engine import HotPath
    B, A, D = shape
    g = torch.Generator("cpu").manual_seed(B)
    hp = HotPath(A, A, [D // 4, D], dt, False)
    d = hp.desc(B)
    x = torch.randn(B, A, generator=g).to(dt).cuda()
    W = (torch.randn(D, A, generator=g) / A ** 0.5).to(dt).cuda()
    b = (torch.randn(D, generator=g) * 0.1).to(dt).cuda()
    acts = hp.encode(d, x, W, b)
    for thr in (0.3, 0.0, -1.0, float("nan")):
        t = torch.tensor(thr, dtype=dt, device="cuda")
        fused = hp.encode_threshold(d, x, W, b, t)
        want = torch.where(acts > t, acts, torch.zeros_like(acts))
        assert torch.equal(fused, want), thr
        if thr == thr:
            sel = hp.select_threshold(d, acts, t)
            assert torch.equal(hp.densify(d, sel, D), fused)


def test_encode_at_activation_store_batch_size(gpu_lib):
    """The sample-collection / eval path encodes a whole model batch at once (training.py:141-145:
    model_batch_size x seq_len = 512 x 128 = 65 536 positions): B*D = 2.4e9 elements, beyond 32-bit indexing.
    Properties: rows are independent (a slice encoded alone gives the same rows, bit for bit); the training-mode
    encode keeps exactly k*N entries; the collector sees every position."""
    from matryoshka_transcoder_amd.activation_samples import ActivationSampleCollector
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    N, A, D, k = 65536, 256, 36864, 32
    cfg = dict(seed=2, prefix_sizes=[D // 4, D], act_size=A, top_k=k, top_k_aux=64, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.bfloat16, device="cuda")
    tc = MatryoshkaTranscoder(cfg)
    g = torch.Generator("cpu").manual_seed(9)
    x = torch.randn(N, A, generator=g).to(torch.bfloat16).cuda()
    with torch.no_grad():
        tc.eval()
        tc.threshold.fill_(0.35)
        full = tc.encode(x)
        assert full.shape == (N, D)
        for lo in (0, 30000, N - 512):
            assert torch.equal(tc.encode(x[lo:lo + 512].contiguous()), full[lo:lo + 512])
        nnz_eval = int((full != 0).sum())
        assert 0 < nnz_eval < N * D // 10
        col = ActivationSampleCollector(16, 8, 0.1)
        toks = torch.randint(0, 50000, (N // 128, 128), generator=g).cuda()
        col.collect_batch_samples(full.reshape(N // 128, 128, D), toks, None)
        assert col.total_samples_seen == nnz_eval
        f = int(torch.argmax(col._count))
        top = col.get_top_samples(f, k=1)[0]
        assert top.activation_value == float(full[:, f].max()) and 0 <= top.position < 128
        del full
        tc.train()
        feats = tc.encode(x)
        assert int((feats != 0).sum()) == k * N
        assert float(tc.threshold) > 0


def test_comm_wrappers_world1(gpu_lib):
    """mtc_comm_* (the C ABI's RCCL steps) on a one-rank communicator: all-reduce / reduce-scatter /
    all-gather through the library's run-time binding of the RCCL copy already in the process (PyTorch's)."""
    import ctypes as C
    from matryoshka_transcoder_amd import _lib
    DEV = "cuda:0"
    lib = gpu_lib
    uid = (C.c_char * 128)()
    _lib.check(lib.mtc_comm_unique_id(uid), "mtc_comm_unique_id")
    comm = C.c_void_p()
    _lib.check(lib.mtc_comm_init(C.byref(comm), 1, 0, uid), "mtc_comm_init")
    try:
        assert lib.mtc_comm_world(comm) == 1 and lib.mtc_comm_rank(comm) == 0
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        a = torch.arange(4096, dtype=torch.float32, device=DEV)
        out = torch.zeros_like(a)
        _lib.check(lib.mtc_comm_all_reduce(comm, C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr()), a.numel(), 0, 0, s), "all_reduce")
        h = torch.randint(0, 1000, (4096,), dtype=torch.int32, device=DEV)          # the select's histogram exchange
        h2 = torch.zeros_like(h)
        _lib.check(lib.mtc_comm_all_reduce(comm, C.c_void_p(h.data_ptr()), C.c_void_p(h2.data_ptr()), h.numel(), 2, 0, s), "all_reduce")
        b = torch.randn(64, 32, device=DEV).to(torch.bfloat16)
        rs = torch.zeros_like(b)
        _lib.check(lib.mtc_comm_reduce_scatter(comm, C.c_void_p(b.data_ptr()), C.c_void_p(rs.data_ptr()), b.numel(), 1, 0, s), "reduce_scatter")
        u = torch.randint(0, 255, (1000,), dtype=torch.uint8, device=DEV)           # the packed byte buffer of the finished rows
        ag = torch.zeros_like(u)
        _lib.check(lib.mtc_comm_all_gather(comm, C.c_void_p(u.data_ptr()), C.c_void_p(ag.data_ptr()), u.numel(), 3, s), "all_gather")
        torch.cuda.synchronize()
        assert torch.equal(out, a) and torch.equal(h2, h) and torch.equal(rs, b) and torch.equal(ag, u)
        assert lib.mtc_comm_all_reduce(comm, C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr()), 4, 99, 0, s) != 0
        assert b"dtype" in lib.mtc_last_error()
    finally:
        _lib.check(lib.mtc_comm_destroy(comm), "mtc_comm_destroy")
