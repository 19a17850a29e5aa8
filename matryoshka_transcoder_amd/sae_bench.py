"""SAEBench / AutoInterp adapters over the HIP inference path (reference:
eval/sae_bench/wrappers.py:18-246, eval/sae_bench/config.py:14-134).

`MatryoshkaAutoInterpAdapter` is a `MatryoshkaTranscoder` carrying the `cfg` / `device` / `dtype` attributes
SAEBench reads; `encode` / `decode` are the libmtc path (MFMA encoder -> threshold or global-TopK select ->
sparse row-gather decode), so `validate_reconstruction` never forms a dense `[N, D] x [D, A]` product.
`NPZTranscoderAdapter` loads a `.npz` checkpoint, accepting transposed matrices and column / row vectors.

Deviation, on purpose: the reference's `NPZTranscoderAdapter` builds its config without `seed` and `act_size`
and so raises `KeyError` in `MatryoshkaTranscoder.__init__` (sae.py:89,111); here both are filled in (seed 0:
every weight is overwritten by the checkpoint anyway).
"""
from __future__ import annotations

import copy
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from .config import compute_fvu, get_hook_name
from .transcoder import MatryoshkaTranscoder


# ------------------------------------------------------------------ config records (eval/sae_bench/config.py)
@dataclass
class AutoInterpSAEConfig:
    """The subset of `sae_lens.SAE.cfg` that SAEBench's helpers read."""
    model_name: str
    hook_layer: int
    hook_name: str
    d_in: int
    d_sae: int
    context_size: Optional[int] = None
    hook_head_index: Optional[int] = None
    architecture: str = ""
    apply_b_dec_to_input: Optional[bool] = None
    finetuning_scaling_factor: Optional[float] = None
    activation_fn_str: str = "relu"
    activation_fn_kwargs: Mapping[str, Any] = field(default_factory=dict)
    prepend_bos: bool = True
    normalize_activations: str = "none"
    dtype: str = ""
    device: str = ""
    model_from_pretrained_kwargs: Mapping[str, Any] = field(default_factory=dict)
    dataset_path: str = ""
    dataset_trust_remote_code: bool = True
    seqpos_slice: Sequence[Optional[int]] = field(default_factory=lambda: (None,))
    training_tokens: int = -100_000
    sae_lens_training_version: Optional[str] = None
    neuronpedia_id: Optional[str] = None

    def to_dict(self) -> dict:
        out = dict(asdict(self))
        out.pop("metadata", None)
        return out

    @classmethod
    def from_kwargs(cls, *, model_name, hook_layer, hook_name, d_in, d_sae, dtype, device, **overrides):
        dev = str(device) if isinstance(device, str) else device.type
        dt = dtype if isinstance(dtype, torch.dtype) else torch.get_default_dtype()
        return cls(model_name=model_name, hook_layer=hook_layer, hook_name=hook_name, d_in=d_in, d_sae=d_sae,
                   dtype=str(dt).split(".")[-1], device=dev, **overrides)


@dataclass
class NPZTranscoderSpec:
    """Where a `.npz` transcoder checkpoint lives and what is in it."""
    path: str
    model_name: str
    source_layer: int
    source_hook_point: str
    target_hook_point: str
    d_in: int
    d_out: int
    d_sae: int
    target_layer: Optional[int] = None
    dtype: str = "bfloat16"
    device: str = "cuda"
    top_k: Optional[int] = None
    prefix_sizes: Optional[Sequence[int]] = None
    aux_penalty: Optional[float] = None
    notes: Optional[str] = None
    weight_key_map: Mapping[str, str] = field(
        default_factory=lambda: {"W_enc": "W_enc", "W_dec": "W_dec", "b_enc": "b_enc", "b_dec": "b_dec"})

    def torch_dtype(self):
        return getattr(torch, self.dtype)

    def torch_device(self):
        return torch.device(self.device)


@dataclass
class ReconstructionReport:
    fvu: float
    mse: float
    mean_activation: float


# ------------------------------------------------------------------ helpers (wrappers.py:18-43)
def _norm_decoder_rows(weights: torch.Tensor) -> torch.Tensor:
    return weights / weights.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def _coerce_matrix(array: np.ndarray, *, expected: Tuple[int, int], name: str) -> np.ndarray:
    if array.shape == tuple(expected):
        return array
    if array.shape == tuple(expected)[::-1]:
        return array.T
    raise ValueError(f"Unexpected shape for {name}: {array.shape} (expected {expected} or {expected[::-1]})")


def _coerce_vector(array: np.ndarray, *, expected: int, name: str) -> np.ndarray:
    if array.shape in ((expected,), (expected, 1), (1, expected)):
        return array.reshape(-1)
    raise ValueError(f"Unexpected shape for {name}: {array.shape} (expected {(expected,)})")


def _closed_prefixes(prefix_sizes, dict_size: int) -> list:
    p = list(prefix_sizes) if prefix_sizes else [dict_size]
    if p[-1] != dict_size:
        p.append(dict_size)
    return p


# ------------------------------------------------------------------ adapters
class MatryoshkaAutoInterpAdapter(MatryoshkaTranscoder):
    """`MatryoshkaTranscoder` + the attributes SAEBench expects on an SAE (wrappers.py:53-109)."""

    def __init__(self, cfg: dict, *, sae_cfg: AutoInterpSAEConfig) -> None:
        c = copy.deepcopy(dict(cfg))
        D = int(c["dict_size"])
        c["prefix_sizes"] = _closed_prefixes(c.get("prefix_sizes"), D)
        k = int(c.get("top_k", D))
        c["top_k"] = D if (k <= 0 or k > D) else k
        c["top_k_aux"] = max(int(c.get("top_k_aux", c["top_k"] * 2)), c["top_k"])
        c.setdefault("n_batches_to_dead", 20)
        super().__init__(c)
        self.cfg = sae_cfg
        self.device = torch.device(c["device"])
        self._sae_dtype = c["dtype"]
        self.to(device=self.device, dtype=c["dtype"])

    def to(self, *args, **kwargs):
        module = super().to(*args, **kwargs)
        if "device" in kwargs:
            self.device = torch.device(kwargs["device"])
        if "dtype" in kwargs:
            self._sae_dtype = kwargs["dtype"]
        return module

    def get_decoder_row_norms(self) -> torch.Tensor:
        return self.W_dec.detach().norm(dim=-1)

    @torch.no_grad()
    def validate_reconstruction(self, source_acts: torch.Tensor, target_acts: torch.Tensor) -> ReconstructionReport:
        """encode -> decode -> MSE (model dtype), FVU (fp32), mean |feature| (wrappers.py:100-109)."""
        encoded = self.encode(source_acts)
        decoded = self.decode(encoded)
        mse = torch.nn.functional.mse_loss(decoded, target_acts).item()
        fvu = compute_fvu(target_acts, decoded)
        return ReconstructionReport(fvu=float(fvu), mse=mse, mean_activation=encoded.abs().mean().item())


def build_matryoshka_adapter(cfg: Mapping[str, Any], *, state_dict_path=None, hook_layer: Optional[int] = None,
                             hook_point: Optional[str] = None) -> MatryoshkaAutoInterpAdapter:
    """wrappers.py:118-156: adapter from a training config (+ optionally a saved state_dict); decoder rows are
    re-normalised after loading."""
    c = copy.deepcopy(dict(cfg))
    if hook_layer is None:
        hook_layer = c.get("source_layer", c["layer"])
    if hook_point is None:
        hook_point = c.get("source_hook_point") or get_hook_name(c["source_site"], hook_layer, c["model_name"])
    sae_cfg = AutoInterpSAEConfig.from_kwargs(
        model_name=c["model_name"], hook_layer=hook_layer, hook_name=hook_point,
        d_in=c.get("source_act_size", c.get("act_size")), d_sae=c["dict_size"], dtype=c["dtype"],
        device=torch.device(c["device"]), context_size=c.get("seq_len"), dataset_path=c.get("dataset_path", ""))
    adapter = MatryoshkaAutoInterpAdapter(c, sae_cfg=sae_cfg)
    if state_dict_path is not None:
        state = torch.load(Path(state_dict_path), map_location=adapter.device, weights_only=True)
        adapter.load_state_dict(state, strict=False)
    with torch.no_grad():
        adapter.W_dec.data = _norm_decoder_rows(adapter.W_dec.data)
    return adapter


class NPZTranscoderAdapter(MatryoshkaAutoInterpAdapter):
    """Cross-layer transcoder from a `.npz` checkpoint (wrappers.py:159-242)."""

    def __init__(self, *, spec: NPZTranscoderSpec):
        dtype = getattr(torch, spec.dtype)
        D = int(spec.d_sae)
        k = spec.top_k if spec.top_k is not None else min(512, D)
        cfg = {
            "seed": 0, "act_size": spec.d_in,                 # absent in the reference's dict (see module docstring)
            "model_name": spec.model_name, "device": spec.device, "dtype": dtype, "dict_size": D,
            "source_act_size": spec.d_in, "target_act_size": spec.d_out,
            "top_k": D if (k <= 0 or k > D) else k, "prefix_sizes": _closed_prefixes(spec.prefix_sizes, D),
            "aux_penalty": spec.aux_penalty or 0.0, "layer": spec.source_layer, "source_layer": spec.source_layer,
            "target_layer": spec.target_layer or spec.source_layer,
            "source_site": spec.source_hook_point.split(".")[-1], "target_site": spec.target_hook_point.split(".")[-1],
            "seq_len": 128,
        }
        sae_cfg = AutoInterpSAEConfig.from_kwargs(model_name=spec.model_name, hook_layer=spec.source_layer,
                                                  hook_name=spec.source_hook_point, d_in=spec.d_in, d_sae=spec.d_sae,
                                                  dtype=dtype, device=torch.device(spec.device))
        super().__init__(cfg, sae_cfg=sae_cfg)
        self.spec = spec
        self.load_npz_weights(spec)
        with torch.no_grad():
            self.W_dec.data = _norm_decoder_rows(self.W_dec.data)

    def load_npz_weights(self, spec: NPZTranscoderSpec) -> None:
        path = Path(spec.path)
        if not path.exists():
            raise FileNotFoundError(f"Transcoder checkpoint not found: {path}")
        data = np.load(path)                                  # allow_pickle stays False

        def fetch(names: Iterable[str]) -> np.ndarray:
            names = list(names)
            for n in names:
                if n in data:
                    return data[n]
            raise KeyError(f"Unable to find any of {names} in {path.name}")

        km = spec.weight_key_map
        W_enc = _coerce_matrix(fetch([km.get("W_enc", "W_enc"), "W_enc", "encoder.W"]),
                               expected=(spec.d_in, spec.d_sae), name="W_enc")
        W_dec = _coerce_matrix(fetch([km.get("W_dec", "W_dec"), "W_dec", "decoder.W"]),
                               expected=(spec.d_sae, spec.d_out), name="W_dec")
        b_enc = _coerce_vector(fetch([km.get("b_enc", "b_enc"), "b_enc", "encoder.b"]), expected=spec.d_sae, name="b_enc")
        b_dec = _coerce_vector(fetch([km.get("b_dec", "b_dec"), "b_dec", "decoder.b"]), expected=spec.d_out, name="b_dec")
        dev, dt = self.device, self.W_dec.dtype
        with torch.no_grad():
            # W_enc keeps its feature-major storage: copy_ into the [A_s, D] view transposes on the fly
            self.W_enc.data.copy_(torch.from_numpy(np.ascontiguousarray(W_enc)).to(device=dev, dtype=dt))
            self.W_dec.data.copy_(torch.from_numpy(np.ascontiguousarray(W_dec)).to(device=dev, dtype=dt))
            self.b_enc.data.copy_(torch.from_numpy(np.ascontiguousarray(b_enc)).to(device=dev, dtype=dt))
            self.b_dec.data.copy_(torch.from_numpy(np.ascontiguousarray(b_dec)).to(device=dev, dtype=dt))


def build_npz_transcoder_adapter(spec: NPZTranscoderSpec) -> NPZTranscoderAdapter:
    return NPZTranscoderAdapter(spec=spec)
