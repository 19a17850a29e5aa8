"""Training loop, LR schedules and the fused optimizer (reference: src/training/training.py).

`train_transcoder(transcoder, activation_store, model, cfg)` keeps the reference's signature and
step order (forward -> backward -> decoder unit-norm projection -> Adam -> scheduler,
training.py:87-183) but never synchronises the device inside the loop: metrics stay on the GPU
and are read back only every `log_freq` steps and at checkpoints.
"""
from __future__ import annotations

import json
import math
import time
from typing import Optional

import torch

from . import _lib
from .activation_samples import ActivationSampleCollector
from .engine import adam_hyper
from .results import CleanResultsManager


# ------------------------------------------------------------------ schedules (training.py:12-47)
def linear_warmup(step, warmup_steps, base_lr=None):
    return min(1.0, step / warmup_steps)


def warmup_cosine_decay(step, warmup_steps, total_steps, base_lr=None):
    if step < warmup_steps:
        return step / warmup_steps            # factor 0 at step 0: the first update does not move weights
    progress = (step - warmup_steps) / (total_steps - warmup_steps)
    return 0.5 * (1 + math.cos(math.pi * progress))


def create_scheduler(optimizer, cfg, num_batches):
    """Same three schedules as the reference (`min_lr` is accepted and, as there, unused by
    `warmup_decay`).  Works with torch.optim.Adam and with FusedAdam (both are Optimizers)."""
    from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR
    kind = cfg.get("scheduler_type", "none")
    if kind == "warmup_only":
        w = cfg.get("warmup_steps", 1000)
        return LambdaLR(optimizer, lr_lambda=lambda s: linear_warmup(s, w))
    if kind == "decay_only":
        return CosineAnnealingLR(optimizer, T_max=num_batches, eta_min=cfg.get("min_lr", cfg["lr"] * 0.01))
    if kind == "warmup_decay":
        w = cfg.get("warmup_steps", 1000)
        return LambdaLR(optimizer, lr_lambda=lambda s: warmup_cosine_decay(s, w, num_batches))
    return None


# ------------------------------------------------------------------ fused optimizer (a12 + a13)
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (eps 1e-8, no weight decay / amsgrad; moments in the parameter
    dtype) executed by libmtc.  `step()` assumes the caller already ran
    `transcoder.make_decoder_weights_and_grad_unit_norm()` like the reference loop does;
    `step(project=True)` fuses that projection into the W_dec update (one pass over W_dec)."""

    def __init__(self, transcoder, lr=3e-4, betas=(0.9, 0.99), eps=1e-8):
        self.transcoder = transcoder
        super().__init__(list(transcoder.parameters()), dict(lr=lr, betas=betas, eps=eps))

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def train_step(self, x_source, x_target):
        """One whole training step (training.py:92,172-179) without materialising gradients: forward chain,
        then per parameter row {k-sparse gradient -> unit-norm projection -> Adam} in one kernel.  Returns the
        reference's output dict (no autograd graph).  While the auxiliary loss is active its gradient rows over the
        dead features (GEMM results) are added inside the same kernels; only the row-gather form of the aux branch
        (sharded model) still goes through dense gradients + `step(project=True)`."""
        tc = self.transcoder
        if not tc.training:
            raise _lib.MtcError("train_step() needs the transcoder in training mode")
        with tc._on_device(x_source):
            return self._train_step(x_source, x_target)

    def _train_step(self, x_source, x_target):
        tc = self.transcoder
        tc._defer_dvals = True          # d a_bj comes out of the decoder-gradient kernel below
        if getattr(tc, "_side_stream", None) is None and getattr(tc, "_overlap_ok", True) and x_source.is_cuda:
            tc._side_stream = torch.cuda.Stream(device=x_source.device)
        try:
            res = tc._run_forward(x_source, x_target, tc.W_enc, tc.b_enc, tc.W_dec, tc.b_dec, True)
        finally:
            tc._defer_dvals = False
        tc._last_res = res
        hp = tc.hot_path()
        two_streams = getattr(tc, "_two_streams", False)     # this step issued work on the side stream (see _run_forward)
        extra = None
        if res.aux is not None and res.aux.get("dense") and res.d_vals is None:
            # dense aux branch: its gradient rows over the dead set (three GEMMs) ride into the fused kernels
            a = res.aux
            extra = (a["dead"].row_map,) + tuple(hp.aux_extra_rows(a, res.x_norm, side=tc._side_stream if two_streams else None))
        elif res.aux is not None:
            tc._join_csc()
            if two_streams:
                torch.cuda.current_stream().wait_stream(tc._side_stream)
            dW_dec, dW_encT, db_enc, db_dec = tc._run_backward(res, None)
            tc.W_dec.grad, tc.W_enc.grad, tc.b_enc.grad, tc.b_dec.grad = dW_dec, dW_encT.t(), db_enc, db_dec
            self.step(project=True)
            self.zero_grad()
            return tc._output_dict(res, x_target)
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        hy = {}
        for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
            st = self._state(getattr(tc, name))
            st["step"] += 1
            hy[name] = adam_hyper(group["lr"], b1, b2, group["eps"], st["step"])
        sd, se, sb, sbd = (self.state[getattr(tc, n)] for n in ("W_dec", "W_enc", "b_enc", "b_dec"))
        W_encT = tc._w_encT(tc.W_enc)
        m_e, v_e = se["exp_avg"].t(), se["exp_avg_sq"].t()
        if not m_e.is_contiguous():
            se["exp_avg"], se["exp_avg_sq"] = m_e.contiguous().t(), v_e.contiguous().t()
            m_e, v_e = se["exp_avg"].t(), se["exp_avg_sq"].t()
        def update_b_dec():
            db_dec = hp.colsum(res.desc, res.g)
            hp.adam_flat(tc.b_dec.data, db_dec, sbd["exp_avg"], sbd["exp_avg_sq"], hy["b_dec"])

        e_b = None
        if two_streams:
            # the b_dec update only needs g: it runs on the side stream underneath the two gradient kernels
            main, side = torch.cuda.current_stream(), tc._side_stream
            e_g = torch.cuda.Event()
            e_g.record(main)
            with torch.cuda.stream(side):
                side.wait_event(e_g)
                update_b_dec()
                e_b = torch.cuda.Event()
                e_b.record(side)
        tc._join_csc()
        hp.fused_backward_adam(res.desc, res.sel, res.g, res.d_vals, res.x_norm, None, tc.W_dec.data,
                               (sd["exp_avg"], sd["exp_avg_sq"]), W_encT.data, (m_e, v_e), tc.b_enc.data,
                               (sb["exp_avg"], sb["exp_avg_sq"]), hy["W_dec"], hy["W_enc"], hy["b_enc"],
                               l1_term=float(tc.config["l1_coeff"]) / (res.desc.B * tc.total_dict_size), extra=extra,
                               enc_wait=res.aux.get("enc_rows_event") if extra is not None else None)
        if e_b is not None:
            torch.cuda.current_stream().wait_event(e_b)     # every later reader of b_dec / g / the scalars is ordered behind the side stream
        else:
            update_b_dec()
        return tc._output_dict(res, x_target)               # (flushes what the forward chain deferred behind the backward launches)

    @torch.no_grad()
    def step(self, closure=None, project: bool = False):
        if closure is not None:
            raise _lib.MtcError("FusedAdam does not take a closure")
        with self.transcoder._on_device(None):
            return self._step(project)

    def _step(self, project: bool):
        tc = self.transcoder
        hp = tc.hot_path()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._state(p)
                st["step"] += 1
                h = adam_hyper(group["lr"], b1, b2, group["eps"], st["step"])
                if p is tc.W_dec and project:
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    hp.unitnorm_adam(p.data, g, st["exp_avg"], st["exp_avg_sq"], h, write_grad=False)
                elif p is tc.W_enc:
                    # feature-major storage: run on the contiguous transposed views
                    tc._w_encT(p)
                    g = p.grad.t()
                    g = g if g.is_contiguous() else g.contiguous()
                    m, v = st["exp_avg"].t(), st["exp_avg_sq"].t()
                    if not m.is_contiguous():
                        st["exp_avg"] = m.contiguous().t(); st["exp_avg_sq"] = v.contiguous().t()
                        m, v = st["exp_avg"].t(), st["exp_avg_sq"].t()
                    hp.adam_flat(p.data.t(), g, m, v, h)
                else:
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    hp.adam_flat(p.data, g, st["exp_avg"], st["exp_avg_sq"], h)
        return None


# ------------------------------------------------------------------ the loop (training.py:49-213)
def train_transcoder(transcoder, activation_store, model, cfg):
    """Train one transcoder.  `activation_store` is anything with `next_batch() -> (source, target)` on
    cfg["device"] in cfg["dtype"]; `model` (the language model) is only handed through."""
    num_batches = int(cfg["num_tokens"] // cfg["batch_size"])
    optimizer = FusedAdam(transcoder, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    scheduler = create_scheduler(optimizer, cfg, num_batches)
    results = CleanResultsManager(cfg.get("results_root", "results"))
    exp_dir = results.create_experiment_dir(cfg.get("model_name", "model"), cfg.get("layer", 0), num_batches,
                                            cfg.get("experiment_description", ""))
    log_freq = int(cfg.get("log_freq", 100))
    log_path = exp_dir / "train_log.jsonl"
    n_dead_at = cfg.get("n_batches_to_dead", 20)
    out = None
    t0 = time.time()
    sample_collector = None
    if cfg.get("save_activation_samples", False):       # training.py:77-85
        sample_collector = ActivationSampleCollector(
            max_samples_per_feature=cfg.get("max_samples_per_feature", 100),
            context_size=cfg.get("sample_context_size", 20),
            storage_threshold=cfg.get("sample_activation_threshold", 0.1))

    def metrics_of(o):
        return {"loss": float(o["loss"].detach()), "l2_loss": float(o["l2_loss"].detach()),
                "l0_norm": float(o["l0_norm"].detach()), "fvu": float(o["fvu"].detach()),
                "dead_features": float(transcoder.num_batches_not_active.sum().item())}

    with open(log_path, "a", encoding="utf-8") as log:
        for i in range(num_batches):
            source_batch, target_batch = activation_store.next_batch()
            perf_now = i % cfg.get("perf_log_freq", 1000) == 0
            ckpt_now = i % cfg.get("checkpoint_freq", 10000) == 0
            sample_now = sample_collector is not None and i % cfg.get("sample_collection_freq", 1000) == 0
            if perf_now or ckpt_now or sample_now:
                # the reference logs performance / checkpoints between the forward and the update
                # (training.py:92-171): keep that order with the autograd path on those iterations
                out = transcoder(source_batch, target_batch)
                if perf_now:
                    log_transcoder_performance(log, i, model, activation_store, transcoder)
                if ckpt_now:
                    results.save_checkpoint(exp_dir, transcoder, cfg, i, metrics_of(out), is_final=False, optimizer=optimizer)
                if sample_now:
                    collect_activation_samples(sample_collector, transcoder, activation_store)
                out["loss"].backward()
                optimizer.step(project=True)      # unit-norm projection fused into the W_dec update
                optimizer.zero_grad()
            else:
                out = optimizer.train_step(source_batch, target_batch)     # same numbers, no materialised gradients
            if i % log_freq == 0:
                rec = metrics_of(out)
                rec.update(step=i, dead_features_count=int((transcoder.num_batches_not_active >= n_dead_at).sum()),
                           learning_rate=optimizer.param_groups[0]["lr"], elapsed_s=time.time() - t0)
                log.write(json.dumps(rec) + "\n")
            if scheduler is not None:
                scheduler.step()
    final_cfg = dict(cfg)
    final_cfg["completed_steps"] = num_batches
    if out is not None:
        results.save_checkpoint(exp_dir, transcoder, final_cfg, num_batches, metrics_of(out), is_final=True,
                                optimizer=optimizer)
    if sample_collector is not None:
        results.save_activation_samples(exp_dir, sample_collector, cfg.get("top_features_to_save", 100),
                                        cfg.get("samples_per_feature_to_save", 10))
    return exp_dir


def collect_activation_samples(sample_collector, transcoder, activation_store):
    """training.py:138-152: a batch with its tokens through `encode` (in the transcoder's current mode, so in
    training mode this is a global-TopK encode that moves the threshold EMA, as in the reference), then into the
    device tables.  Runs between the forward and the update of its iteration, like the reference."""
    source, tokens = activation_store.get_batch_with_tokens()
    with torch.no_grad():
        features = transcoder.encode(source)
    sample_collector.collect_batch_samples(features, tokens, activation_store.tokenizer)


def log_transcoder_performance(log, step, model, activation_store, transcoder):
    """training.py:216-245: an extra batch through the transcoder under no_grad *in its current mode*
    (in training mode this moves the threshold EMA and the dead counters, as in the reference)."""
    source_batch, target_batch = activation_store.next_batch()
    with torch.no_grad():
        o = transcoder(source_batch, target_batch)
        rec = o["transcoder_out"].float()
        tgt = target_batch.float()
        mse = (rec - tgt).pow(2).mean()
        mae = (rec - tgt).abs().mean()
        cos = torch.nn.functional.cosine_similarity(rec.flatten(), tgt.flatten(), dim=0)
    log.write(json.dumps({"step": step, "performance/mse": float(mse), "performance/mae": float(mae),
                          "performance/cosine_similarity": float(cos), "performance/fvu": float(o["fvu"])}) + "\n")
