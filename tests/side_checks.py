"""Shared comparisons for the side-path tests (collector tables, SAEBench adapter)."""
import os

import torch

from conftest import GOLDEN

COLLECTOR_CASES = ["fp32", "fp32_flat_tracked", "bf16_ties"]
ADAPTER_CASES = ["fp32", "fp32_transposed", "bf16"]


class WordTokenizer:
    """The tokenizer stand-in the fixtures were made with (tools/make_golden_side.py)."""

    def decode(self, ids):
        return " ".join(f"t{int(i)}" for i in ids)


def load_side(kind, name):
    return torch.load(os.path.join(GOLDEN, f"side_{kind}_{name}.pt"), weights_only=True)


def check_collector(fx, top_of, stats_of, overall, counts, maxima, texts=True):
    """`top_of(f)` -> list of sample dicts (largest first), `stats_of(f)` -> statistics dict.

    Bar: per feature the stored VALUES equal the reference's exactly (bit-exact, same order); the stored samples
    (position, activating tokens, token window, texts) equal the reference's exactly for every sample strictly
    above the smallest stored value.  Samples AT the smallest stored value are interchangeable when several
    activations share it (the reference's heap evicts whichever of them sits at its root), so there only the
    count is compared."""
    n_feat = len(fx["features"])
    for f in range(n_feat):
        want = fx["features"][f]["samples"]
        got = top_of(f)
        assert [s["activation_value"] for s in got] == [s["activation_value"] for s in want], f
        if want:
            lo = want[-1]["activation_value"]
            keys = ("activation_value", "position", "tokens", "context_tokens") + (("text", "context_text") if texts else ())

            def ident(s):
                return tuple(tuple(s[k]) if isinstance(s[k], list) else s[k] for k in keys)
            assert sorted(ident(s) for s in got if s["activation_value"] > lo) == \
                sorted(ident(s) for s in want if s["activation_value"] > lo), f
        ws, gs = fx["features"][f]["statistics"], stats_of(f)
        assert set(ws) == set(gs), (f, ws, gs)
        for k, v in ws.items():
            if isinstance(v, float):
                assert abs(float(gs[k]) - v) <= 1e-6 * max(1.0, abs(v)), (f, k, gs[k], v)
            else:
                assert gs[k] == v, (f, k, gs[k], v)
    for k, v in fx["overall"].items():
        assert overall[k] == v, (k, overall[k], v)
    assert {int(k): int(v) for k, v in counts.items()} == fx["counts"]
    assert {int(k): float(v) for k, v in maxima.items()} == fx["max_activations"]
