"""Host logic of the side paths that needs no GPU: .npz loading / shape coercion of the SAEBench adapter
(eval/sae_bench/wrappers.py:22-43,159-242) and the collector's argument checks."""
import numpy as np
import pytest
import torch

from side_checks import ADAPTER_CASES, load_side


def _spec(fx, path, **kw):
    from matryoshka_transcoder_amd.sae_bench import NPZTranscoderSpec
    sp = fx["spec"]
    args = dict(path=str(path), model_name="gemma-2-2b", source_layer=sp["source_layer"],
                source_hook_point="blocks.8.hook_resid_post", target_hook_point="blocks.9.hook_resid_pre",
                d_in=sp["d_in"], d_out=sp["d_out"], d_sae=sp["d_sae"], target_layer=sp["target_layer"],
                dtype=sp["dtype"], device="cpu", top_k=sp["top_k"], prefix_sizes=sp["prefix_sizes"])
    args.update(kw)
    return NPZTranscoderSpec(**args)


@pytest.mark.parametrize("name", ADAPTER_CASES)
def test_npz_loading_matches_reference(name, tmp_path):
    from matryoshka_transcoder_amd.sae_bench import build_npz_transcoder_adapter
    fx = load_side("adapter", name)
    path = tmp_path / "ckpt.npz"
    np.savez(path, **{k: v.numpy() for k, v in fx["arrays"].items()})
    ad = build_npz_transcoder_adapter(_spec(fx, path))
    for k, v in ad.state_dict().items():
        assert torch.equal(v, fx["state_after_load"][k]), k
    assert ad.W_enc.t().is_contiguous()                       # the encoder keeps its feature-major storage
    assert list(ad.prefix_sizes) == fx["prefix_sizes"]
    assert ad.config["top_k"] == fx["top_k"] and ad.config["top_k_aux"] == fx["top_k_aux"]
    for k, v in fx["cfg_fields"].items():
        assert getattr(ad.cfg, k) == v, k
    assert torch.allclose(ad.get_decoder_row_norms().float(), fx["row_norms"].float(), rtol=1e-6)
    assert "hook_name" in ad.cfg.to_dict()
    # the product computes on the GPU only
    from matryoshka_transcoder_amd import _lib
    with pytest.raises(_lib.MtcError):
        ad.validate_reconstruction(fx["x_source"], fx["x_target"])


def test_npz_errors(tmp_path):
    from matryoshka_transcoder_amd.sae_bench import build_npz_transcoder_adapter
    fx = load_side("adapter", "fp32")
    with pytest.raises(FileNotFoundError):
        build_npz_transcoder_adapter(_spec(fx, tmp_path / "missing.npz"))
    arrays = {k: v.numpy() for k, v in fx["arrays"].items()}
    bad = dict(arrays, W_dec=arrays["W_dec"][:, :-1])
    np.savez(tmp_path / "bad.npz", **bad)
    with pytest.raises(ValueError, match="W_dec"):
        build_npz_transcoder_adapter(_spec(fx, tmp_path / "bad.npz"))
    del arrays["b_enc"]
    np.savez(tmp_path / "short.npz", **arrays)
    with pytest.raises(KeyError):
        build_npz_transcoder_adapter(_spec(fx, tmp_path / "short.npz"))
    # key map: checkpoint stores the encoder under another name
    arrays = {k: v.numpy() for k, v in fx["arrays"].items()}
    arrays["enc_weight"] = arrays.pop("W_enc")
    np.savez(tmp_path / "mapped.npz", **arrays)
    ad = build_npz_transcoder_adapter(_spec(fx, tmp_path / "mapped.npz",
                                            weight_key_map={"W_enc": "enc_weight"}))
    assert torch.equal(ad.W_enc.detach(), fx["state_after_load"]["W_enc"])


def test_top_k_and_prefix_fixing():
    from matryoshka_transcoder_amd.sae_bench import AutoInterpSAEConfig, MatryoshkaAutoInterpAdapter
    sae_cfg = AutoInterpSAEConfig.from_kwargs(model_name="m", hook_layer=1, hook_name="h", d_in=16, d_sae=64,
                                              dtype=torch.float32, device="cpu")
    base = dict(seed=0, act_size=16, dict_size=64, dtype=torch.float32, device="cpu", aux_penalty=0.0, l1_coeff=0.0)
    ad = MatryoshkaAutoInterpAdapter(dict(base, top_k=0, prefix_sizes=[16, 32]), sae_cfg=sae_cfg)
    assert ad.prefix_sizes == [16, 32, 64] and ad.config["top_k"] == 64 and ad.config["top_k_aux"] == 128
    assert ad.config["n_batches_to_dead"] == 20
    ad = MatryoshkaAutoInterpAdapter(dict(base, top_k=8, top_k_aux=4), sae_cfg=sae_cfg)
    assert ad.prefix_sizes == [64] and ad.config["top_k_aux"] == 8
    assert sae_cfg.dtype == "float32" and sae_cfg.device == "cpu"


def test_collector_argument_checks():
    from matryoshka_transcoder_amd.activation_samples import ActivationSample, ActivationSampleCollector
    with pytest.raises(ValueError):
        ActivationSampleCollector(max_samples_per_feature=0)
    col = ActivationSampleCollector(10, 20, 0.1)
    assert col.total_samples_seen == 0 and col.total_samples_stored == 0 and col.feature_samples == {}
    assert col.get_feature_statistics(2) == {"feature_idx": 2, "num_samples": 0, "max_activation": 0.0,
                                             "mean_activation": 0.0, "activation_count": 0}
    a = ActivationSample(1, 2.0, [3], "x", 0, [3, 4], "x y")
    b = ActivationSample(1, 3.0, [3], "x", 0, [3, 4], "x y")
    assert b < a and a.to_dict()["context_tokens"] == [3, 4]       # larger activation sorts first
