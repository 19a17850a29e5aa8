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
    """[rows, cols] as given, or the transpose of a [cols, rows] array."""
    want = tuple(expected)
    if array.shape != want and array.shape != want[::-1]:
        raise ValueError(f"Unexpected shape for {name}: {array.shape} (expected {want} or {want[::-1]})")
    return array if array.shape == want else array.T


def _coerce_vector(array: np.ndarray, *, expected: int, name: str) -> np.ndarray:
    """[n], [n, 1] or [1, n] -> [n]."""
    if array.shape not in ((expected,), (expected, 1), (1, expected)):
        raise ValueError(f"Unexpected shape for {name}: {array.shape} (expected {(expected,)})")
    return array.reshape(-1)


def _closed_prefixes(prefix_sizes, dict_size: int) -> list:
    """One group by default; the dictionary size is appended when the list stops short of it."""
    out = list(prefix_sizes or [dict_size])
    return out if out[-1] == dict_size else out + [dict_size]


def _valid_top_k(top_k: int, dict_size: int) -> int:
    return top_k if 0 < top_k <= dict_size else dict_size


# ------------------------------------------------------------------ adapters
class MatryoshkaAutoInterpAdapter(MatryoshkaTranscoder):
    """`MatryoshkaTranscoder` + the attributes SAEBench expects on an SAE (wrappers.py:53-109)."""

    def __init__(self, cfg: dict, *, sae_cfg: AutoInterpSAEConfig) -> None:
        own = copy.deepcopy(dict(cfg))
        width = int(own["dict_size"])
        own["prefix_sizes"] = _closed_prefixes(own.get("prefix_sizes"), width)
        own["top_k"] = _valid_top_k(int(own.get("top_k", width)), width)
        own["top_k_aux"] = max(own["top_k"], int(own.get("top_k_aux", 2 * own["top_k"])))
        own.setdefault("n_batches_to_dead", 20)
        super().__init__(own)
        self.cfg = sae_cfg
        self.device = torch.device(own["device"])
        self._sae_dtype = own["dtype"]
        self.to(device=self.device, dtype=own["dtype"])

    def to(self, *args, **kwargs):
        moved = super().to(*args, **kwargs)
        if "dtype" in kwargs:
            self._sae_dtype = kwargs["dtype"]
        if "device" in kwargs:
            self.device = torch.device(kwargs["device"])
        return moved

    def get_decoder_row_norms(self) -> torch.Tensor:
        return self.W_dec.detach().norm(dim=-1)

    @torch.no_grad()
    def validate_reconstruction(self, source_acts: torch.Tensor, target_acts: torch.Tensor) -> ReconstructionReport:
        """encode -> decode -> MSE (model dtype), FVU (fp32), mean |feature| (wrappers.py:100-109)."""
        features = self.encode(source_acts)
        recon = self.decode(features)
        return ReconstructionReport(fvu=float(compute_fvu(target_acts, recon)),
                                    mse=torch.nn.functional.mse_loss(recon, target_acts).item(),
                                    mean_activation=features.abs().mean().item())

    def _renormalise_decoder(self):
        with torch.no_grad():
            self.W_dec.data = _norm_decoder_rows(self.W_dec.data)


def build_matryoshka_adapter(cfg: Mapping[str, Any], *, state_dict_path=None, hook_layer: Optional[int] = None,
                             hook_point: Optional[str] = None) -> MatryoshkaAutoInterpAdapter:
    """wrappers.py:118-156: adapter from a training config (+ optionally a saved state_dict); decoder rows are
    re-normalised after loading."""
    own = copy.deepcopy(dict(cfg))
    layer = own.get("source_layer", own["layer"]) if hook_layer is None else hook_layer
    point = hook_point or own.get("source_hook_point") or get_hook_name(own["source_site"], layer, own["model_name"])
    sae_cfg = AutoInterpSAEConfig.from_kwargs(
        model_name=own["model_name"], hook_layer=layer, hook_name=point, d_sae=own["dict_size"],
        d_in=own.get("source_act_size", own.get("act_size")), dtype=own["dtype"], device=torch.device(own["device"]),
        context_size=own.get("seq_len"), dataset_path=own.get("dataset_path", ""))
    adapter = MatryoshkaAutoInterpAdapter(own, sae_cfg=sae_cfg)
    if state_dict_path is not None:
        adapter.load_state_dict(torch.load(Path(state_dict_path), map_location=adapter.device, weights_only=True),
                                strict=False)
    adapter._renormalise_decoder()
    return adapter


class NPZTranscoderAdapter(MatryoshkaAutoInterpAdapter):
    """Cross-layer transcoder from a `.npz` checkpoint (wrappers.py:159-242)."""

    # parameter -> (other accepted key, matrix shape as spec attributes or vector length attribute)
    _ARRAYS = (("W_enc", "encoder.W", ("d_in", "d_sae")), ("W_dec", "decoder.W", ("d_sae", "d_out")),
               ("b_enc", "encoder.b", "d_sae"), ("b_dec", "decoder.b", "d_out"))

    def __init__(self, *, spec: NPZTranscoderSpec):
        dtype, width = getattr(torch, spec.dtype), int(spec.d_sae)
        site = lambda hook: hook.split(".")[-1]                      # noqa: E731
        cfg = dict(
            seed=0, act_size=spec.d_in,                              # absent in the reference's dict (module docstring)
            model_name=spec.model_name, dtype=dtype, device=spec.device, dict_size=width,
            source_act_size=spec.d_in, target_act_size=spec.d_out,
            prefix_sizes=_closed_prefixes(spec.prefix_sizes, width),
            top_k=_valid_top_k(min(512, width) if spec.top_k is None else spec.top_k, width),
            aux_penalty=spec.aux_penalty or 0.0, seq_len=128,
            layer=spec.source_layer, source_layer=spec.source_layer, target_layer=spec.target_layer or spec.source_layer,
            source_site=site(spec.source_hook_point), target_site=site(spec.target_hook_point))
        super().__init__(cfg, sae_cfg=AutoInterpSAEConfig.from_kwargs(
            model_name=spec.model_name, hook_layer=spec.source_layer, hook_name=spec.source_hook_point, d_in=spec.d_in,
            d_sae=spec.d_sae, dtype=dtype, device=torch.device(spec.device)))
        self.spec = spec
        self.load_npz_weights(spec)
        self._renormalise_decoder()

    def load_npz_weights(self, spec: NPZTranscoderSpec) -> None:
        ckpt = Path(spec.path)
        if not ckpt.exists():
            raise FileNotFoundError(f"Transcoder checkpoint not found: {ckpt}")
        stored = np.load(ckpt)                                        # allow_pickle stays False
        for param, alias, shape in self._ARRAYS:
            keys = [spec.weight_key_map.get(param, param), param, alias]
            hit = next((k for k in keys if k in stored), None)
            if hit is None:
                raise KeyError(f"Unable to find any of {keys} in {ckpt.name}")
            if isinstance(shape, tuple):
                arr = _coerce_matrix(stored[hit], expected=tuple(getattr(spec, a) for a in shape), name=param)
            else:
                arr = _coerce_vector(stored[hit], expected=getattr(spec, shape), name=param)
            target = getattr(self, param)
            with torch.no_grad():      # W_enc keeps its feature-major storage: copy_ into the [A_s, D] view transposes
                target.data.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(device=self.device, dtype=target.dtype))


def build_npz_transcoder_adapter(spec: NPZTranscoderSpec) -> NPZTranscoderAdapter:
    return NPZTranscoderAdapter(spec=spec)
