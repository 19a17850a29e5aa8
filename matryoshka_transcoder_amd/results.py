"""Checkpoint / results layout of the reference (src/utils/clean_results.py:26-97):

    results/{model}/layer{L}/{steps}/checkpoints/{step_N,final}.pt   <- model.state_dict()
                                    /config.json                      (dtype / device stringified)
                                    /metrics.json | metrics/step_N.json

`state_dict()` keys, shapes and dtypes are the reference's (b_enc, W_enc, W_dec, b_dec, threshold).
What the reference does not persist goes to a side file `checkpoints/{name}.extra.pt`, so that the reference layout
stays loadable by the reference: always `step` and the dead counters (`num_batches_not_active`), plus the optimizer's
`state_dict()` (Adam moments and step counts) when `save_checkpoint` is given the optimizer -- `train_transcoder`
passes it.  `load_extra` restores all of it (tensors only, `weights_only=True`): a resumed run continues bit for bit.
"""
from __future__ import annotations

import json
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Dict, Optional

import torch


def _jsonable(obj: Any) -> Any:
    """dtype / device objects become their string form; containers are walked."""
    if isinstance(obj, dict):
        return {key: _jsonable(val) for key, val in obj.items()}
    if isinstance(obj, (tuple, list)):
        return [_jsonable(val) for val in obj]
    return str(obj) if isinstance(obj, (torch.device, torch.dtype)) else obj


def _dump(path: Path, payload: Any) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(json.dumps(_jsonable(payload), indent=2), encoding="utf-8")


class CleanResultsManager:
    """Writes one run under `{base_dir}/{model}/layer{L}/{steps}/` (see the module docstring for the files)."""

    def __init__(self, base_dir: str = "results") -> None:
        self.base_dir = Path(base_dir)
        self.base_dir.mkdir(parents=True, exist_ok=True)

    # -- paths ---------------------------------------------------------------------------------------------
    def create_experiment_dir(self, model_name: str, layer: int, steps: int, description: str = "") -> Path:
        run = self.base_dir.joinpath(model_name, f"layer{layer}", f"{steps}")
        run.joinpath("checkpoints").mkdir(parents=True, exist_ok=True)
        return run

    # -- writers -------------------------------------------------------------------------------------------
    def save_checkpoint(self, experiment_dir: Path, model: torch.nn.Module, config: Dict[str, Any], step: int,
                        metrics: Optional[Dict[str, Any]] = None, is_final: bool = False,
                        extra: Optional[Dict[str, Any]] = None, optimizer: Optional[torch.optim.Optimizer] = None) -> None:
        run = Path(experiment_dir)
        stem = "final" if is_final else f"step_{step}"
        weights = run / "checkpoints" / f"{stem}.pt"
        weights.parent.mkdir(exist_ok=True)
        # contiguous row-major tensors: the file is laid out byte for byte like a reference checkpoint
        torch.save({name: t.detach().contiguous() for name, t in model.state_dict().items()}, weights)
        if extra is None and hasattr(model, "num_batches_not_active"):
            extra = {"step": step, "num_batches_not_active": model.num_batches_not_active.detach().clone()}
        if optimizer is not None:
            extra = dict(extra or {"step": step})
            extra["optimizer"] = optimizer.state_dict()
        if extra:
            torch.save(extra, weights.with_name(f"{stem}.extra.pt"))
        cfg_out = dict(config)
        cfg_out.pop("completed_at", None)
        if is_final:
            stamp = datetime.now(timezone.utc).replace(tzinfo=None).isoformat()
            cfg_out["completed_at"] = f"{stamp}Z"
        _dump(run / "config.json", cfg_out)
        if metrics:
            _dump(run / "metrics.json" if is_final else run / "metrics" / f"step_{step}.json", metrics)

    def save_activation_samples(self, experiment_dir: Path, sample_collector: Any, top_k_features: int = 100,
                                samples_per_feature: int = 10) -> None:
        """clean_results.py:88-97: {run}/activation_samples/feature_%05d_samples.json + collection_summary.json."""
        target = Path(experiment_dir, "activation_samples")
        target.mkdir(exist_ok=True)
        sample_collector.save_samples(str(target), top_k_features, samples_per_feature)

    # -- readers -------------------------------------------------------------------------------------------
    def load_checkpoint(self, experiment_dir: Path, model: torch.nn.Module, stem: str = "final",
                        optimizer: Optional[torch.optim.Optimizer] = None) -> Dict[str, Any]:
        """Weights from `checkpoints/{stem}.pt` (a reference checkpoint loads the same way) and, when the side file
        exists, the dead counters and the optimizer state.  Returns the side file's dict ({} without one)."""
        weights = Path(experiment_dir) / "checkpoints" / f"{stem}.pt"
        dev = next(model.parameters()).device
        model.load_state_dict(torch.load(weights, map_location=dev, weights_only=True))
        side = weights.with_name(f"{stem}.extra.pt")
        if not side.exists():
            return {}
        extra = torch.load(side, map_location=dev, weights_only=True)
        if "num_batches_not_active" in extra and hasattr(model, "num_batches_not_active"):
            model.num_batches_not_active = extra["num_batches_not_active"].to(dev)
        if optimizer is not None and "optimizer" in extra:
            optimizer.load_state_dict(extra["optimizer"])
        return extra

    def list_experiments(self, model_name: Optional[str] = None, layer: Optional[int] = None) -> Dict[str, Any]:
        found: Dict[str, Any] = {}
        for mdir in sorted(p for p in self.base_dir.iterdir() if p.is_dir()):
            if model_name and mdir.name != model_name:
                continue
            per_layer: Dict[int, Any] = {}
            for ldir in sorted(p for p in mdir.iterdir() if p.is_dir() and p.name.startswith("layer")):
                try:
                    lnum = int(ldir.name[len("layer"):])
                except ValueError:
                    continue
                if layer is not None and lnum != layer:
                    continue
                runs = [{"steps": r.name, "path": str(r),
                         "has_final_checkpoint": (r / "checkpoints" / "final.pt").exists(),
                         "has_metrics": (r / "metrics.json").exists()}
                        for r in sorted(p for p in ldir.iterdir() if p.is_dir())]
                if runs:
                    per_layer[lnum] = runs
            if per_layer:
                found[mdir.name] = per_layer
        return found
