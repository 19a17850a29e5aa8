"""Checkpoint / results layout of the reference (src/utils/clean_results.py:26-97):

    results/{model}/layer{L}/{steps}/checkpoints/{step_N,final}.pt   <- model.state_dict()
                                    /config.json                      (dtype / device stringified)
                                    /metrics.json | metrics/step_N.json

`state_dict()` keys, shapes and dtypes are the reference's (b_enc, W_enc, W_dec, b_dec, threshold).
What the reference does not persist (optimizer moments, step, dead counters) goes to a side file
`checkpoints/{name}.extra.pt` so that the reference layout stays loadable by the reference.
"""
from __future__ import annotations

import json
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, Dict, Optional

import torch


def _jsonable(obj: Any) -> Any:
    if isinstance(obj, (torch.dtype, torch.device)):
        return str(obj)
    if isinstance(obj, dict):
        return {k: _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    return obj


class CleanResultsManager:
    def __init__(self, base_dir: str = "results") -> None:
        self.base_dir = Path(base_dir)
        self.base_dir.mkdir(parents=True, exist_ok=True)

    def create_experiment_dir(self, model_name: str, layer: int, steps: int, description: str = "") -> Path:
        run_dir = self.base_dir / model_name / f"layer{layer}" / str(steps)
        (run_dir / "checkpoints").mkdir(parents=True, exist_ok=True)
        return run_dir

    def save_checkpoint(self, experiment_dir: Path, model: torch.nn.Module, config: Dict[str, Any], step: int,
                        metrics: Optional[Dict[str, Any]] = None, is_final: bool = False,
                        extra: Optional[Dict[str, Any]] = None) -> None:
        ckpt_dir = Path(experiment_dir) / "checkpoints"
        ckpt_dir.mkdir(exist_ok=True)
        name = "final" if is_final else f"step_{step}"
        # contiguous row-major tensors on save: byte-identical layout to a reference checkpoint
        state = {k: v.detach().contiguous() for k, v in model.state_dict().items()}
        torch.save(state, ckpt_dir / f"{name}.pt")
        if extra is None and hasattr(model, "num_batches_not_active"):
            extra = {"num_batches_not_active": model.num_batches_not_active.detach().clone(), "step": step}
        if extra:
            torch.save(extra, ckpt_dir / f"{name}.extra.pt")
        cfg = _jsonable(dict(config))
        if is_final:
            cfg["completed_at"] = datetime.now(timezone.utc).replace(tzinfo=None).isoformat() + "Z"
        else:
            cfg.pop("completed_at", None)
        with (Path(experiment_dir) / "config.json").open("w", encoding="utf-8") as f:
            json.dump(cfg, f, indent=2)
        if metrics:
            if is_final:
                path = Path(experiment_dir) / "metrics.json"
            else:
                (Path(experiment_dir) / "metrics").mkdir(exist_ok=True)
                path = Path(experiment_dir) / "metrics" / f"step_{step}.json"
            with path.open("w", encoding="utf-8") as f:
                json.dump(_jsonable(metrics), f, indent=2)

    def save_activation_samples(self, experiment_dir: Path, sample_collector: Any, top_k_features: int = 100,
                                samples_per_feature: int = 10) -> None:
        """clean_results.py:88-97: {run}/activation_samples/feature_%05d_samples.json + collection_summary.json."""
        samples_dir = Path(experiment_dir) / "activation_samples"
        samples_dir.mkdir(exist_ok=True)
        sample_collector.save_samples(str(samples_dir), top_k_features, samples_per_feature)

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
