"""`ActivationSampleCollector` with the reference's interface (src/utils/activation_samples.py:21-370), kept
on the GPU.

The reference copies the dense `[batch, seq, D]` feature tensor to the host and walks it feature by feature in
Python, pushing one object per activation into a per-feature heap and decoding its text on the spot
(activation_samples.py:107-169).  Here the tables live in HBM and one libmtc launch per batch merges the batch
into them (`mtc_collect_top_samples`, csrc/collect.hip); nothing is copied to the host and no text is decoded
until samples are asked for (`get_top_samples`, `save_samples`, ...).  Per feature the table holds the
`max_samples_per_feature` largest activations seen, their sequence position and token window.

Differences a caller can see: `max_samples_per_feature` is limited to 256; among samples of exactly the same
activation value the older one is kept and listed first (the reference's heap leaves that order to its internal
layout); `feature_samples[f]` is a list of `(value, ActivationSample)` built on demand.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .engine import HotPath, _p, _stream, dtype_code


@dataclass
class ActivationSample:
    """One stored activation with its token window (activation_samples.py:21-46)."""
    feature_idx: int
    activation_value: float
    tokens: List[int]
    text: str
    position: int
    context_tokens: List[int]
    context_text: str

    def to_dict(self):
        return {"feature_idx": self.feature_idx, "activation_value": self.activation_value, "tokens": self.tokens,
                "text": self.text, "position": self.position, "context_tokens": self.context_tokens,
                "context_text": self.context_text}

    def __lt__(self, other):
        return self.activation_value > other.activation_value


class _PerFeature:
    """Read-only dict-like view over one per-feature device array (counts, maxima)."""

    def __init__(self, owner, attr, cast):
        self._o, self._a, self._cast = owner, attr, cast

    def _host(self):
        t = getattr(self._o, self._a)
        return None if t is None else t.cpu()

    def __getitem__(self, f):
        h = self._host()
        return self._cast(0) if h is None else self._cast(h[int(f)])

    def get(self, f, default=None):
        h = self._host()
        if h is None or not (0 <= int(f) < h.numel()) or self._o._count[int(f)] == 0:
            return default
        return self._cast(h[int(f)])

    def items(self):
        h = self._host()
        if h is None:
            return []
        cnt = self._o._count.cpu()
        return [(int(f), self._cast(h[f])) for f in torch.nonzero(cnt).flatten().tolist()]

    def keys(self):
        return [k for k, _ in self.items()]

    def __len__(self):
        return len(self.items())


class ActivationSampleCollector:
    def __init__(self, max_samples_per_feature: int = 100, context_size: int = 20, storage_threshold: float = 0.1):
        if not 1 <= int(max_samples_per_feature) <= 256:
            raise ValueError("max_samples_per_feature must be in 1..256 for the device tables "
                             f"(got {max_samples_per_feature})")
        self.max_samples = int(max_samples_per_feature)
        self.context_size = int(context_size)
        self.storage_threshold = storage_threshold
        self._cw = 2 * (self.context_size // 2) + 1
        self._tokenizer = None
        self._D = None
        self._val = self._slot = self._meta = self._ctx = self._count = self._max = None
        self.feature_activation_counts = _PerFeature(self, "_count", int)
        self.feature_max_activations = _PerFeature(self, "_max", float)

    # ------------------------------------------------------------------ device tables
    def _allocate(self, D: int, device):
        M = self.max_samples
        self._D = D
        self._val = torch.full((D, M), float("-inf"), dtype=torch.float32, device=device)
        self._slot = torch.arange(M, dtype=torch.int32, device=device).repeat(D, 1).contiguous()
        self._meta = torch.zeros(D, M, 4, dtype=torch.int32, device=device)
        self._ctx = torch.zeros(D, M, self._cw, dtype=torch.int32, device=device)
        self._count = torch.zeros(D, dtype=torch.int64, device=device)
        self._max = torch.zeros(D, dtype=torch.float32, device=device)     # defaultdict(float): starts at 0.0

    def collect_batch_samples(self, activations: torch.Tensor, tokens: torch.Tensor, tokenizer,
                              feature_indices: Optional[List[int]] = None):
        """activations: dense [batch, seq, D] or [batch*seq, D] feature tensor on the GPU (what `encode` returns);
        tokens: [batch, seq] token ids.  Merges every activation > storage_threshold into the tables."""
        if not activations.is_cuda:
            raise _lib.MtcError("ActivationSampleCollector keeps its tables on the GPU; got activations on "
                                f"{activations.device} and there is no CPU path")
        batch, seq = int(tokens.shape[0]), int(tokens.shape[1])
        D = int(activations.shape[-1])
        dense = activations.detach().reshape(-1, D).contiguous()
        if dense.shape[0] != batch * seq:
            raise ValueError(f"activations cover {dense.shape[0]} positions but tokens are {batch} x {seq}")
        if D % 8:
            raise _lib.MtcError("the number of features must be a multiple of 8")
        hp = HotPath(8, 8, [D], dense.dtype, False)
        d = hp.desc(batch * seq)
        # non-zero entries -> CSR -> CSC: entries arrive per feature in ascending (batch, position) order
        nan = torch.full((1,), float("nan"), dtype=dense.dtype, device=dense.device)
        sel = hp.select_threshold(d, dense, nan)
        hp.build_csc(d, sel, None)
        self.collect_selection(sel, dense.dtype, batch * seq, D, tokens, tokenizer, feature_indices)

    def collect_selection(self, sel, dtype, n_rows: int, D: int, tokens: torch.Tensor, tokenizer,
                          feature_indices: Optional[List[int]] = None):
        """Same from a sparse selection that already has its CSC view (e.g. the one the encoder's select made)."""
        dev = sel.col_ptr.device
        if self._D is None:
            self._allocate(D, dev)
        elif self._D != D:
            raise ValueError(f"collector tables were sized for {self._D} features, got {D}")
        self._tokenizer = tokenizer
        toks = tokens.detach().to(device=dev, dtype=torch.int64).contiguous()
        track = None
        if feature_indices is not None:
            track = torch.zeros(D, dtype=torch.uint8, device=dev)
            track[torch.as_tensor(list(feature_indices), dtype=torch.long, device=dev)] = 1
        # `acts > threshold` on a bf16 tensor compares against the threshold rounded to bf16
        thr = float(torch.tensor(self.storage_threshold, dtype=dtype).float())
        lib = _lib.load()
        _lib.check(lib.mtc_collect_top_samples(dtype_code(dtype), int(n_rows), int(D), int(toks.shape[1]), _p(sel.col_ptr),
                                               _p(sel.csc_row), _p(sel.csc_pos), _p(sel.vals), _p(toks), _p(track), thr,
                                               self.max_samples, self.context_size, _p(self._val), _p(self._slot),
                                               _p(self._meta), _p(self._ctx), _p(self._count), _p(self._max), _stream()),
                   "mtc_collect_top_samples")

    # ------------------------------------------------------------------ totals (device reductions on demand)
    @property
    def total_samples_seen(self) -> int:
        return 0 if self._count is None else int(self._count.sum().item())

    @property
    def total_samples_stored(self) -> int:
        """Pushes into heaps that were not yet full (activation_samples.py:175-178) = filled table entries."""
        return 0 if self._val is None else int(torch.isfinite(self._val).sum().item())

    @property
    def feature_samples(self) -> Dict[int, list]:
        if self._val is None:
            return {}
        have = torch.nonzero(torch.isfinite(self._val[:, 0])).flatten().tolist()
        return {f: [(s.activation_value, s) for s in self.get_top_samples(f)] for f in have}

    def _features_with_samples(self) -> List[int]:
        if self._val is None:
            return []
        return torch.nonzero(torch.isfinite(self._val[:, 0])).flatten().tolist()

    # ------------------------------------------------------------------ reading samples back
    def _decode(self, ids):
        try:
            return self._tokenizer.decode(ids)
        except Exception as e:  # the reference stores the error text instead of failing
            return f"<decode error: {e}>"

    def get_top_samples(self, feature_idx: int, k: Optional[int] = None) -> List[ActivationSample]:
        if self._val is None or not 0 <= int(feature_idx) < self._D:
            return []
        f = int(feature_idx)
        vals = self._val[f].cpu()
        n = int(torch.isfinite(vals).sum())
        if k is not None:
            n = min(n, k)
        if n == 0:
            return []
        slots = self._slot[f, :n].cpu().long()
        meta = self._meta[f].cpu()[slots]
        ctx = self._ctx[f].cpu()[slots]
        out = []
        for i in range(n):
            pos, length, prev = int(meta[i, 0]), int(meta[i, 1]), int(meta[i, 2])
            context = ctx[i, :length].tolist()
            at = pos - max(0, pos - self.context_size // 2)
            activating = ([prev] if prev >= 0 else []) + [context[at]]
            out.append(ActivationSample(f, float(vals[i]), activating, self._decode(activating), pos, context,
                                        self._decode(context)))
        return out

    def get_feature_statistics(self, feature_idx: int) -> Dict:
        samples = self.get_top_samples(feature_idx)
        count = 0 if self._count is None else int(self._count[int(feature_idx)])
        if not samples:
            return {"feature_idx": feature_idx, "num_samples": 0, "max_activation": 0.0, "mean_activation": 0.0,
                    "activation_count": count}
        a = [s.activation_value for s in samples]
        return {"feature_idx": feature_idx, "num_samples": len(samples), "max_activation": max(a),
                "mean_activation": np.mean(a), "std_activation": np.std(a), "min_activation": min(a),
                "activation_count": count, "all_time_max": float(self._max[int(feature_idx)])}

    def get_all_statistics(self) -> Dict:
        have = self._features_with_samples()
        stored = self.total_samples_stored
        return {"total_samples_seen": self.total_samples_seen, "total_samples_stored": stored,
                "num_features_tracked": len(have), "avg_samples_per_feature": stored / max(1, len(have)),
                "features_with_samples": sorted(have)}

    def _by_max_activation(self, n: Optional[int]) -> List[int]:
        have = self._features_with_samples()
        mx = self._max.cpu()
        # stable sort on the maximum, descending; equal maxima keep ascending feature order
        order = sorted(have, key=lambda f: float(mx[f]), reverse=True)
        return order if n is None else order[:n]

    def save_samples(self, save_dir: str, top_k_features: Optional[int] = None, samples_per_feature: int = 10):
        """Same files as the reference (activation_samples.py:251-314): feature_%05d_samples.json for the
        top_k_features features by maximum activation + collection_summary.json."""
        os.makedirs(save_dir, exist_ok=True)
        feats = self._by_max_activation(top_k_features) if top_k_features is not None else sorted(self._features_with_samples())
        for f in feats:
            data = {"statistics": _plain(self.get_feature_statistics(f)),
                    "top_samples": [s.to_dict() for s in self.get_top_samples(f, k=samples_per_feature)]}
            with open(os.path.join(save_dir, f"feature_{f:05d}_samples.json"), "w") as fh:
                json.dump(data, fh, indent=2)
        overall = self.get_all_statistics()
        mx = None if self._max is None else self._max.cpu()
        nfilled = None if self._val is None else torch.isfinite(self._val).sum(1).cpu()
        overall["top_features_by_activation"] = [
            {"feature_idx": f, "max_activation": float(mx[f]), "num_samples": int(nfilled[f])}
            for f in self._by_max_activation(50)]
        with open(os.path.join(save_dir, "collection_summary.json"), "w") as fh:
            json.dump(overall, fh, indent=2)
        print(f"\nSaved activation samples to {save_dir}")
        print(f"  Features saved: {len(feats)}")
        print(f"  Samples per feature: {samples_per_feature}")
        print(f"  Total samples stored: {overall['total_samples_stored']}")

    def load_samples(self, save_dir: str, device="cuda", num_features: Optional[int] = None):
        """Merge samples saved by `save_samples` (ours or the reference's) into the tables.  An offline path: the
        merge runs on the host copy of the tables and is written back once."""
        if not os.path.exists(save_dir):
            raise FileNotFoundError(f"Directory not found: {save_dir}")
        loaded = []
        for name in sorted(os.listdir(save_dir)):
            if name.startswith("feature_") and name.endswith("_samples.json"):
                with open(os.path.join(save_dir, name)) as fh:
                    data = json.load(fh)
                loaded.append((int(data["statistics"]["feature_idx"]), data["top_samples"]))
        if not loaded:
            return
        need = max(f for f, _ in loaded) + 1
        if self._D is None:
            D = int(num_features) if num_features is not None else (need + 7) // 8 * 8
            self._allocate(D, device)
        if need > self._D:
            raise ValueError(f"saved samples name feature {need - 1} but the tables hold {self._D} features")
        val, slot, meta, ctx = self._val.cpu(), self._slot.cpu(), self._meta.cpu(), self._ctx.cpu()
        M = self.max_samples
        for f, samples in loaded:
            for s in samples:
                v = float(s["activation_value"])
                if not v > float(val[f, M - 1]):          # full and not strictly greater: the stored one stays
                    continue
                free = int(slot[f, M - 1])                # slot of the entry pushed out (or an empty one)
                at = int((val[f] >= v).sum())             # after every entry of the same value
                val[f, at + 1:] = val[f, at:M - 1].clone()
                slot[f, at + 1:] = slot[f, at:M - 1].clone()
                val[f, at], slot[f, at] = v, free
                c = list(s["context_tokens"])[:self._cw]
                pos = int(s["position"])
                prev = int(s["tokens"][0]) if len(s["tokens"]) > 1 else -1
                meta[f, free] = torch.tensor([pos, len(c), prev, -1], dtype=torch.int32)
                ctx[f, free, :len(c)] = torch.tensor(c, dtype=torch.int32)
        dev = self._val.device
        self._val, self._slot, self._meta, self._ctx = val.to(dev), slot.to(dev), meta.to(dev), ctx.to(dev)
        print(f"Loaded samples from {save_dir}")
        print(f"  Features loaded: {len(self._features_with_samples())}")

    def print_feature_examples(self, feature_idx: int, k: int = 5):
        samples = self.get_top_samples(feature_idx, k=k)
        st = self.get_feature_statistics(feature_idx)
        print(f"\n{'=' * 80}\nFeature {feature_idx} - Top {k} Activating Examples\n{'=' * 80}")
        print(f"Max activation: {st['max_activation']:.4f}")
        print(f"Mean activation: {st['mean_activation']:.4f}")
        print(f"Total activations seen: {st['activation_count']}")
        print(f"Samples stored: {st['num_samples']}\n{'-' * 80}\n")
        if not samples:
            print("No samples found for this feature.")
            return
        for i, s in enumerate(samples, 1):
            print(f"Example {i} (activation: {s.activation_value:.4f}):")
            print(f"  Text: {s.text}\n  Context: ...{s.context_text}...\n  Position: {s.position}\n")

    def clear(self):
        if self._val is not None:
            self._allocate(self._D, self._val.device)


def _plain(d: Dict) -> Dict:
    return {k: (float(v) if isinstance(v, (np.floating,)) else v) for k, v in d.items()}
