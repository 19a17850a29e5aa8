"""CPU oracle for the side paths next to the hot path.  TEST INFRASTRUCTURE ONLY:
imported by ``tests/`` alone, never by the shipped package.

Restates, from scratch:

  * the top-activating-sample collector ... src/utils/activation_samples.py:83-249
  * dense encode / decode (a15) ........... src/models/sae.py:363-406
  * the SAEBench adapter helpers .......... eval/sae_bench/wrappers.py:18-43 (shape coercion, decoder-row
                                            normalisation), :100-109 (validate_reconstruction),
                                            :56-79 and :166-174 (prefix / top-k fixing)

Parity pin: ``tools/make_golden_side.py`` runs the reference's own classes in the build container and writes
``tests/golden/side_*.pt``; ``tests/test_side_oracle_vs_golden.py`` checks this file against them.
"""
from __future__ import annotations

import heapq
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ref_cpu as R

F32 = torch.float32


# --------------------------------------------------------------------------- collector (heap restatement)
class HeapCollector:
    """Per-feature bounded min-heaps of (value, arrival number, record).  A full heap takes a new sample only
    when it is strictly larger than the smallest one kept (activation_samples.py:171-183)."""

    def __init__(self, max_samples: int, context_size: int, storage_threshold: float):
        self.max_samples, self.context_size, self.storage_threshold = max_samples, context_size, storage_threshold
        self.heaps: Dict[int, list] = {}
        self.counts: Dict[int, int] = {}
        self.maxima: Dict[int, float] = {}
        self.seen = 0
        self.stored = 0
        self._arrival = 0

    def collect(self, acts: torch.Tensor, tokens: torch.Tensor, features: Optional[List[int]] = None):
        batch, seq = tokens.shape
        a = acts.reshape(batch, seq, -1)
        half = self.context_size // 2
        thr = torch.tensor(self.storage_threshold, dtype=a.dtype)      # the comparison runs in the tensor's dtype
        for f in (range(a.shape[-1]) if features is None else features):
            col = a[:, :, f]
            hit = torch.nonzero(col > thr)                              # ascending (sequence, position)
            if hit.numel() == 0:
                continue
            v = col[hit[:, 0], hit[:, 1]].to(F32)
            self.seen += len(v)
            self.counts[f] = self.counts.get(f, 0) + len(v)
            self.maxima[f] = max(self.maxima.get(f, 0.0), float(v.max()))
            heap = self.heaps.setdefault(f, [])
            for (b, s), val in zip(hit.tolist(), v.tolist()):
                rec = {"feature_idx": f, "activation_value": val, "position": s,
                       "tokens": tokens[b, max(0, s - 1):s + 1].tolist(),
                       "context_tokens": tokens[b, max(0, s - half):min(seq, s + half + 1)].tolist()}
                self._arrival += 1
                if len(heap) < self.max_samples:
                    heapq.heappush(heap, (val, self._arrival, rec))
                    self.stored += 1
                elif val > heap[0][0]:
                    # evict the smallest; among equal smallest values the NEWEST goes (-arrival would be needed
                    # for that in the key, so do it by hand)
                    lo = heap[0][0]
                    newest = max((e for e in heap if e[0] == lo), key=lambda e: e[1])
                    heap.remove(newest)
                    heapq.heapify(heap)
                    heapq.heappush(heap, (val, self._arrival, rec))

    def top(self, f: int, k: Optional[int] = None) -> List[dict]:
        """Largest first; equal values in arrival order."""
        recs = [e[2] for e in sorted(self.heaps.get(f, []), key=lambda e: (-e[0], e[1]))]
        return recs if k is None else recs[:k]

    def statistics(self, f: int) -> dict:
        a = [r["activation_value"] for r in self.top(f)]
        if not a:
            return {"feature_idx": f, "num_samples": 0, "max_activation": 0.0, "mean_activation": 0.0,
                    "activation_count": self.counts.get(f, 0)}
        return {"feature_idx": f, "num_samples": len(a), "max_activation": max(a), "mean_activation": float(np.mean(a)),
                "std_activation": float(np.std(a)), "min_activation": min(a), "activation_count": self.counts[f],
                "all_time_max": self.maxima[f]}

    def overall(self) -> dict:
        have = sorted(self.heaps)
        return {"total_samples_seen": self.seen, "total_samples_stored": self.stored, "num_features_tracked": len(have),
                "avg_samples_per_feature": self.stored / max(1, len(have)), "features_with_samples": have}


# --------------------------------------------------------------------------- a15 dense encode / decode
def encode_dense(params, threshold: torch.Tensor, x: torch.Tensor, cfg: dict, training: bool):
    """sae.py:363-387.  Returns (dense features [N, D], new threshold, mean, std)."""
    shape = x.shape
    xn, mean, std = R.preprocess(x, bool(cfg.get("input_unit_norm", False)))
    x2 = xn.reshape(-1, shape[-1])
    acts = R.encode_acts(x2, params["W_enc"], params["b_enc"])
    if training:
        tau, idx, _, _ = R.global_topk_select(acts, int(cfg["top_k"]) * x2.shape[0])
        dense = torch.zeros_like(acts).flatten()
        dense[idx] = acts.flatten()[idx]
        dense = dense.reshape(acts.shape)
        pos = dense[dense > 0]
        new_thr = R.ema_threshold(threshold, pos.min() if pos.numel() else None)
    else:
        dense = torch.where(acts > threshold, acts, torch.zeros_like(acts))
        new_thr = threshold
    if len(shape) == 3:
        dense = dense.reshape(shape[0], shape[1], -1)
    return dense, new_thr, mean, std


def decode_dense(params, dense: torch.Tensor, mean, std):
    """sae.py:389-406: dense features x W_dec + b_dec in the model dtype (fp32 accumulation, one rounding per
    materialised tensor), de-normalised with the statistics of the last encode."""
    dt = dense.dtype
    rec = (dense.to(F32) @ params["W_dec"].to(F32)).to(dt)
    rec = (rec.to(F32) + params["b_dec"].to(F32)).to(dt)
    return R.postprocess(rec, mean, std)


# --------------------------------------------------------------------------- SAEBench adapter helpers
def coerce_matrix(a: np.ndarray, rows: int, cols: int, name: str) -> np.ndarray:
    """wrappers.py:22-31: accept [rows, cols] or its transpose."""
    if a.shape == (rows, cols):
        return a
    if a.shape == (cols, rows):
        return a.T
    raise ValueError(f"Unexpected shape for {name}: {a.shape}")


def coerce_vector(a: np.ndarray, n: int, name: str) -> np.ndarray:
    """wrappers.py:34-43: accept [n], [n, 1] or [1, n]."""
    if a.shape in ((n,), (n, 1), (1, n)):
        return a.reshape(-1)
    raise ValueError(f"Unexpected shape for {name}: {a.shape}")


def fix_prefixes(prefix_sizes, dict_size: int) -> List[int]:
    """wrappers.py:64-68,168-170: default to one group; append the dictionary size when the list stops short."""
    p = list(prefix_sizes) if prefix_sizes else [dict_size]
    if p[-1] != dict_size:
        p.append(dict_size)
    return p


def fix_top_k(top_k, dict_size: int, default: int) -> int:
    k = default if top_k is None else int(top_k)
    return dict_size if (k <= 0 or k > dict_size) else k


def load_npz_params(arrays: Dict[str, np.ndarray], d_in: int, d_out: int, d_sae: int, dt) -> Dict[str, torch.Tensor]:
    """wrappers.py:209-242 + :207: fetch by the accepted key names, coerce shapes, cast, normalise decoder rows."""
    def fetch(*names):
        for n in names:
            if n in arrays:
                return np.asarray(arrays[n])
        raise KeyError(names)
    p = {"W_enc": coerce_matrix(fetch("W_enc", "encoder.W"), d_in, d_sae, "W_enc"),
         "W_dec": coerce_matrix(fetch("W_dec", "decoder.W"), d_sae, d_out, "W_dec"),
         "b_enc": coerce_vector(fetch("b_enc", "encoder.b"), d_sae, "b_enc"),
         "b_dec": coerce_vector(fetch("b_dec", "decoder.b"), d_out, "b_dec")}
    p = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dt) for k, v in p.items()}
    p["W_dec"] = p["W_dec"] / p["W_dec"].norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return p


def validate_reconstruction(params, threshold, xs, xt, cfg, training: bool):
    """wrappers.py:100-109.  Returns (fvu, mse, mean |feature|, new threshold)."""
    enc, thr, mean, std = encode_dense(params, threshold, xs, cfg, training)
    dec = decode_dense(params, enc, mean, std)
    mse = torch.nn.functional.mse_loss(dec, xt)
    d32, t32 = dec.to(F32), xt.to(F32)
    fvu = ((t32 - d32) ** 2).mean() / (t32.var() + 1e-8)           # config.py:205-234
    return float(fvu), float(mse), float(enc.abs().mean()), thr
