"""`TranscoderActivationsStore` with the reference's interface
(src/models/transcoder_activation_store.py:8-214): `next_batch() -> (source, target)` on
cfg["device"] in cfg["dtype"], `get_batch_tokens`, `get_paired_activations`, `get_batch_with_tokens`,
`.tokenizer`.

What changed underneath: the reference wraps device tensors in a TensorDataset + DataLoader
(shuffle=True), i.e. per-sample host indexing and a host-side collate of B device rows per batch.
Here the buffer stays in HBM, the shuffle is one device `randperm` per refill and a batch is ONE libmtc
launch (`mtc_gather_row_pairs`, csrc/store.hip) that gathers the source and target rows of a permutation
slice together.  A CPU-resident store (cfg["device"] = "cpu": host-logic tests, data staging) keeps the
same logic with a torch row gather; the GPU path has no such fallback.
Three sources:
  * a language model exposing `run_with_cache(tokens, names_filter=..., stop_at_layer=...)` and a
    token iterator (the reference's path; needs transformer_lens + a dataset, neither ships here);
  * `SyntheticSource`  : Gaussian pairs of the configured widths (bench / tests);
  * `TensorPairSource` : pre-harvested (source, target) tensors or .pt files.
"""
from __future__ import annotations

import contextlib
from typing import Iterator, Optional, Tuple

import torch

from .config import create_transcoder_config  # noqa: F401  (re-exported like the reference module)


_NULL = contextlib.nullcontext()


class SyntheticSource:
    """Independent N(0,1) source / target rows, deterministic per chunk (seed 1000 + i like bench.py)."""

    def __init__(self, cfg: dict, rows_per_chunk: Optional[int] = None):
        self.a_s = cfg.get("source_act_size", cfg["act_size"])
        self.a_t = cfg.get("target_act_size", cfg["act_size"])
        self.rows = rows_per_chunk or cfg.get("model_batch_size", 512) * cfg.get("seq_len", 128)
        self.i = 0

    def next_chunk(self, device, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
        g = torch.Generator("cpu").manual_seed(1000 + self.i)
        self.i += 1
        xs = torch.randn(self.rows, self.a_s, generator=g).to(dtype)
        xt = torch.randn(self.rows, self.a_t, generator=g).to(dtype)
        return xs.to(device), xt.to(device)


class TensorPairSource:
    """Cycles over pre-harvested activation pairs: tensors, or paths of torch-saved (source, target) tuples."""

    def __init__(self, pairs):
        self.pairs = list(pairs)
        self.i = 0

    def next_chunk(self, device, dtype):
        item = self.pairs[self.i % len(self.pairs)]
        self.i += 1
        if isinstance(item, str):
            item = torch.load(item, weights_only=True)
        xs, xt = item
        return xs.to(device=device, dtype=dtype), xt.to(device=device, dtype=dtype)


class TranscoderActivationsStore:
    def __init__(self, model, cfg: dict, source=None, token_iterator: Optional[Iterator] = None):
        self.model = model
        self.config = cfg
        self.device = cfg["device"]
        self.source_hook_point = cfg.get("source_hook_point")
        self.target_hook_point = cfg.get("target_hook_point")
        self.model_batch_size = cfg.get("model_batch_size", 512)
        self.num_batches_in_buffer = cfg.get("num_batches_in_buffer", 10)
        n_ctx = getattr(getattr(model, "cfg", None), "n_ctx", cfg.get("seq_len", 128))
        self.context_size = min(cfg.get("seq_len", 128), n_ctx)
        self.stop_at_layer = cfg.get("target_layer", cfg.get("layer", 0)) + 1
        self.tokenizer = getattr(model, "tokenizer", None)
        self.token_iterator = token_iterator
        if source is None and model is None:
            source = SyntheticSource(cfg)
        self.source = source
        self._perm = None
        self._cursor = 0

    # ---- LM-hooked path (reference :67-155), duck-typed on the model object
    def get_batch_tokens(self):
        if self.token_iterator is None:
            raise RuntimeError("no token iterator: pass token_iterator=... (datasets are not reachable offline)")
        toks = []
        need = self.model_batch_size * self.context_size
        while len(toks) < need:
            toks.extend(next(self.token_iterator))
        t = torch.tensor(toks[:need], dtype=torch.long, device=self.device)
        return t.view(self.model_batch_size, self.context_size)

    def get_paired_activations(self, batch_tokens: torch.Tensor):
        names = [self.target_hook_point, self.source_hook_point]
        fallback, layer = None, None
        if self.source_hook_point and "hook_mlp_in" in self.source_hook_point:
            layer = int(self.source_hook_point.split(".")[1])
            fallback = f"blocks.{layer}.ln2.hook_normalized"
            names.append(fallback)
        with torch.no_grad():
            _, cache = self.model.run_with_cache(batch_tokens, names_filter=names, stop_at_layer=self.stop_at_layer)
        try:
            src = cache[self.source_hook_point]
        except KeyError:
            if fallback is None:
                raise
            ln2 = self.model.blocks[layer].ln2
            scale = next((getattr(ln2, a) for a in ("w", "weight", "scale", "g") if hasattr(ln2, a)), None)
            src = cache[fallback] if scale is None else cache[fallback] * scale
        tgt = cache[self.target_hook_point]
        dt = self.config.get("dtype")
        if dt is not None:
            src, tgt = src.to(dtype=dt), tgt.to(dtype=dt)
        return src, tgt

    def get_batch_with_tokens(self):
        toks = self.get_batch_tokens()
        src, _ = self.get_paired_activations(toks)
        return src, toks

    # ---- buffer (reference :157-207), on-device shuffle instead of a DataLoader
    def _fill_buffer(self):
        a_s = self.config.get("source_act_size", self.config["act_size"])
        a_t = self.config.get("target_act_size", self.config["act_size"])
        srcs, tgts = [], []
        for _ in range(self.num_batches_in_buffer):
            if self.source is not None:
                s, t = self.source.next_chunk(self.device, self.config["dtype"])
            else:
                s, t = self.get_paired_activations(self.get_batch_tokens())
            srcs.append(s.reshape(-1, a_s))
            tgts.append(t.reshape(-1, a_t))
        return torch.cat(srcs, 0), torch.cat(tgts, 0)

    def next_batch(self):
        B = self.config["batch_size"]
        if self._perm is None or self._cursor >= self._perm.numel():
            self.source_buffer, self.target_buffer = self._fill_buffer()
            self._perm = torch.randperm(self.source_buffer.shape[0], device=self.source_buffer.device)
            self._cursor = 0
        idx = self._perm[self._cursor:self._cursor + B]      # last batch may be short, like the DataLoader's
        self._cursor += B
        return self._gather_pairs(idx)

    def _gather_pairs(self, idx: torch.Tensor):
        src, tgt = self.source_buffer, self.target_buffer
        if not src.is_cuda:
            # host buffer (cfg["device"] = "cpu"): plain indexing.  Data plumbing only -- nothing of the hot path computes
            # here (a CPU batch handed to MatryoshkaTranscoder raises MtcError); the CPU test-suite checks the store's
            # bookkeeping (buffer refill, permutation, short last batch, aligned pairs) through this branch
            return src.index_select(0, idx), tgt.index_select(0, idx)
        import ctypes as C
        from . import _lib
        from .engine import _p, _stream, dtype_code
        if not (src.is_contiguous() and tgt.is_contiguous() and src.dtype == tgt.dtype and tgt.device == src.device):
            raise _lib.MtcError("activation buffers must be contiguous, of one dtype, on one GPU")
        n = idx.numel()
        out_s = torch.empty(n, src.shape[1], dtype=src.dtype, device=src.device)
        out_t = torch.empty(n, tgt.shape[1], dtype=tgt.dtype, device=tgt.device)
        idx = idx.contiguous()
        ctx = torch.cuda.device(src.device) if torch.cuda.current_device() != src.device.index else _NULL
        with ctx:
            _lib.check(_lib.load().mtc_gather_row_pairs(dtype_code(src.dtype), n, src.shape[1], tgt.shape[1], src.shape[0],
                                                        _p(src), _p(tgt), _p(idx), _p(out_s), _p(out_t), None, _stream()),
                       "mtc_gather_row_pairs")
        return out_s, out_t
