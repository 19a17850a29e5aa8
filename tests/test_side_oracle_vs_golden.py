"""Pins oracle/side_cpu.py to the fixtures the reference's own collector / SAEBench adapter produced
(tools/make_golden_side.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import side_cpu as S
from side_checks import ADAPTER_CASES, COLLECTOR_CASES, WordTokenizer, check_collector, load_side


@pytest.mark.parametrize("name", COLLECTOR_CASES)
def test_heap_collector_matches_reference(name):
    fx = load_side("collector", name)
    col = S.HeapCollector(fx["max_samples"], fx["context_size"], fx["storage_threshold"])
    for r in fx["rounds"]:
        col.collect(r["acts"], r["tokens"], fx["tracked"])
    tok = WordTokenizer()

    def top_of(f):
        return [dict(s, text=tok.decode(s["tokens"]), context_text=tok.decode(s["context_tokens"])) for s in col.top(f)]

    check_collector(fx, top_of, col.statistics, col.overall(), col.counts, col.maxima)


@pytest.mark.parametrize("name", ADAPTER_CASES)
def test_adapter_restatement_matches_reference(name):
    fx = load_side("adapter", name)
    sp = fx["spec"]
    dt = getattr(torch, sp["dtype"])
    arrays = {k: v.numpy() for k, v in fx["arrays"].items()}
    p = S.load_npz_params(arrays, sp["d_in"], sp["d_out"], sp["d_sae"], dt)
    for k in ("W_enc", "W_dec", "b_enc", "b_dec"):
        assert torch.equal(p[k], fx["state_after_load"][k]), k
    prefixes = S.fix_prefixes(sp["prefix_sizes"], sp["d_sae"])
    assert prefixes == fx["prefix_sizes"]
    top_k = S.fix_top_k(sp["top_k"], sp["d_sae"], min(512, sp["d_sae"]))
    assert top_k == fx["top_k"]
    cfg = {"top_k": top_k, "prefix_sizes": prefixes, "input_unit_norm": False}
    rel = 1e-5 if dt == torch.float32 else 2e-2
    for mode in ("train", "eval"):
        w = fx[mode]
        fvu, mse, mean_act, thr = S.validate_reconstruction(p, w["threshold_before"], fx["x_source"], fx["x_target"],
                                                            cfg, mode == "train")
        assert abs(fvu - w["fvu"]) <= rel * abs(w["fvu"]), (mode, fvu, w["fvu"])
        assert abs(mse - w["mse"]) <= rel * abs(w["mse"]), (mode, mse, w["mse"])
        assert abs(mean_act - w["mean_activation"]) <= rel * abs(w["mean_activation"])
        # the fixture's threshold_after saw two encodes (validate_reconstruction, then encode)
        enc, thr2, _, _ = S.encode_dense(p, thr, fx["x_source"], cfg, mode == "train")
        assert torch.allclose(thr2.float(), w["threshold_after"].float(), rtol=rel, atol=1e-7)
        if dt == torch.float32:
            assert torch.equal(enc != 0, w["encoded_after"] != 0)
            assert torch.allclose(enc, w["encoded_after"], rtol=1e-5, atol=1e-6)
        else:
            # bf16: index sets may differ only among values within one ulp of the selection threshold
            diff = (enc != 0) ^ (w["encoded_after"] != 0)
            assert int(diff.sum()) <= 4


def test_shape_coercion_errors():
    with pytest.raises(ValueError):
        S.coerce_matrix(np.zeros((3, 5)), 4, 5, "W")
    with pytest.raises(ValueError):
        S.coerce_vector(np.zeros((2, 3)), 6, "b")
