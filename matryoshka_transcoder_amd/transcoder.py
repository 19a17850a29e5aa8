"""Drop-in `MatryoshkaTranscoder` whose math runs in libmtc's HIP kernels.

Mirrors the public surface of the reference class (src/models/sae.py:73-406): constructor
arguments, parameter names / registration order / shapes, the `threshold` buffer, the
`num_batches_not_active` attribute, the 11-key forward dict, `encode` / `decode`,
`make_decoder_weights_and_grad_unit_norm`, and all documented side effects.  What differs is
underneath: activations stay sparse (CSR rows / CSC features), the global TopK is a radix select,
the nested prefix reconstructions are one running sum, and the backward is k-sparse.

`W_enc` keeps the reference's shape `[A_s, D]` but its storage is feature-major (`[D, A_s]`
contiguous, exposed as a transposed view), which is what the MFMA encoder and the per-feature
gradient rows want; `state_dict()` round-trips with reference checkpoints unchanged.
"""
from __future__ import annotations

import contextlib
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .engine import HotPath, Selection, adam_hyper


def _feature_major(w: torch.Tensor) -> torch.Tensor:
    """[A_s, D] tensor whose memory is [D, A_s] row-major (w.t() is contiguous)."""
    return w.t().contiguous().t()


class _StepFn(torch.autograd.Function):
    """forward: the whole HIP forward chain; backward: the k-sparse backward (a11)."""

    @staticmethod
    def forward(ctx, module, x_source, x_target, W_enc, b_enc, W_dec, b_dec):
        need_grad = any(ctx.needs_input_grad[3:])
        module._check_inputs(x_source, x_target)
        with module._on_device():
            res = module._run_forward(x_source, x_target, W_enc, b_enc, W_dec, b_dec, need_grad)
        ctx.module = module
        ctx.res = res
        module._last_res = res
        ctx.mark_non_differentiable(res.x_hat, res.scalars)
        loss32 = res.scalars[5]
        if module._aux_is_on(res):
            loss = loss32.clone()
        elif module.dtype == torch.bfloat16:
            loss = res.scalars[12:16].view(torch.bfloat16)[4].clone()     # the aux-free loss in the model dtype (mtc_reduce_scalars)
        else:
            loss = loss32.to(module.dtype)
        return loss, res.x_hat, res.scalars

    @staticmethod
    def backward(ctx, g_loss, _g_xhat, _g_scalars):
        res = ctx.res
        if res.g is None:
            raise _lib.MtcError("backward called on a forward that did not record gradients")
        module = ctx.module
        with module._on_device():
            dW_dec, dW_encT, db_enc, db_dec = module._run_backward(res, g_loss)
        return None, None, None, dW_encT.t(), db_enc, dW_dec, db_dec


class MatryoshkaTranscoder(nn.Module):
    """Cross-layer transcoder with nested (Matryoshka) prefix groups and global batch-TopK sparsity."""

    def __init__(self, cfg: dict):
        super().__init__()
        self.config = cfg
        prefix_sizes = list(cfg["prefix_sizes"])
        for a, b in zip(prefix_sizes[:-1], prefix_sizes[1:]):
            if a > b:
                raise ValueError(
                    "prefix_sizes must be monotonically increasing (cumulative format). "
                    f"Got: {prefix_sizes}")
        self.prefix_sizes = prefix_sizes
        self.prefix_indices = [0] + prefix_sizes
        self.active_prefixes = len(prefix_sizes)
        D = prefix_sizes[-1]
        self.total_dict_size = D
        self.source_act_size = cfg.get("source_act_size", cfg.get("act_size"))
        self.target_act_size = cfg.get("target_act_size", cfg.get("act_size"))

        # same CPU RNG stream as the reference (sae.py:89,116-127): seed, W_enc draw, W_dec draw
        torch.manual_seed(cfg["seed"])
        w_enc = nn.init.kaiming_uniform_(torch.empty(self.source_act_size, D))
        w_dec = nn.init.kaiming_uniform_(torch.empty(D, self.target_act_size))
        w_dec = w_dec / w_dec.norm(dim=-1, keepdim=True)
        self.b_enc = nn.Parameter(torch.zeros(D))
        self.W_enc = nn.Parameter(_feature_major(w_enc))
        self.W_dec = nn.Parameter(w_dec)
        self.b_dec = nn.Parameter(torch.zeros(self.target_act_size))
        self._cnt = torch.zeros(D, device=cfg["device"])          # num_batches_not_active (fp32, not a buffer)
        self.register_buffer("threshold", torch.tensor(0.0))
        self.x_mean = None
        self.x_std = None
        self._hp: Optional[HotPath] = None
        self._last_res = None
        self._side_inputs = None
        self._deferred = None
        self._dead_check_in = 0          # forwards until a dead feature can exist (host-side bound)
        self._aux_bound = None           # upper bound of the next forward's dead count (from the last forward's dead list)
        self._fwd_serial = 0
        self.to(cfg["dtype"]).to(cfg["device"])

    # ------------------------------------------------------------------ plumbing
    def __setattr__(self, name, value):
        # per-step bookkeeping (`_last_res`, `_csc_event`, `x_mean`, ...) is plain state, never a Parameter / buffer /
        # sub-module: nn.Module.__setattr__ spends ~6 us per assignment checking for those, ~8 times per step
        if (name[0] == "_" or name in ("x_mean", "x_std")) and not isinstance(value, (nn.Parameter, nn.Module)):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    @property
    def num_batches_not_active(self) -> torch.Tensor:
        """Dead-feature counters (sae.py:134).  Handing the tensor out (or replacing it) voids the
        host-side "no feature can be dead yet" bound, because the caller may edit it in place."""
        self._dead_check_in = 0
        self._aux_bound = None
        return self._cnt

    @num_batches_not_active.setter
    def num_batches_not_active(self, value: torch.Tensor) -> None:
        self._dead_check_in = 0
        self._aux_bound = None
        self._cnt = value

    @property
    def dtype(self) -> torch.dtype:
        return self.W_dec.dtype

    def hot_path(self) -> HotPath:
        if self._hp is None or self._hp.dtype != self.dtype:
            self._hp = HotPath(self.source_act_size, self.target_act_size, self.prefix_sizes, self.dtype,
                               bool(self.config.get("input_unit_norm", False)))
        self._hp.unit_norm = bool(self.config.get("input_unit_norm", False))
        return self._hp

    def _w_encT(self, W_enc: torch.Tensor) -> torch.Tensor:
        wt = W_enc.t()
        if not wt.is_contiguous():       # e.g. weights assigned from a row-major tensor: re-lay once
            with torch.no_grad():
                self.W_enc.data = _feature_major(self.W_enc.data)
            wt = self.W_enc.t()
        return wt

    def _check_inputs(self, x_source, x_target=None):
        if not x_source.is_cuda:
            raise _lib.MtcError("matryoshka_transcoder_amd computes on the GPU only (HIP kernels); got a "
                                f"{x_source.device} input and there is no CPU fallback")
        if x_source.dtype != self.dtype or (x_target is not None and x_target.dtype != self.dtype):
            raise _lib.MtcError(f"inputs must have the model dtype {self.dtype}")
        dev = self.W_dec.device
        if x_source.device != dev or (x_target is not None and x_target.device != dev):
            raise _lib.MtcError(f"inputs live on {x_source.device}"
                                f"{'' if x_target is None else ' / ' + str(x_target.device)} but the parameters on {dev}: "
                                "libmtc launches on one GPU and takes raw pointers, move the inputs first")

    _NULL_CTX = contextlib.nullcontext()

    def _on_device(self, x=None):
        """Context that makes the parameters' GPU the current one (libmtc launches on HIP's current device and on
        torch's current stream OF that device, whatever the caller's current device is -- the reference honours
        cfg['device'] = 'cuda:1' without a set_device call because ATen ops carry device guards).  Free when the
        device already is current."""
        dev = self.W_dec.device
        if dev.type != "cuda":
            raise _lib.MtcError(f"the parameters live on {dev}: matryoshka_transcoder_amd computes on the GPU only")
        if x is not None and x.device != dev:
            raise _lib.MtcError(f"input on {x.device} but the parameters on {dev}")
        if torch.cuda.current_device() == dev.index:
            return self._NULL_CTX
        return torch.cuda.device(dev)

    # ------------------------------------------------------------------ reference helper methods
    def preprocess_input(self, x):
        """sae.py:139-147.  Returns (x_norm, mean, std) with [.., 1] statistics, or (x, None, None)."""
        if not self.config.get("input_unit_norm", False):
            return x, None, None
        self._check_inputs(x)
        hp = self.hot_path()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        with self._on_device():
            xn, mean, std = hp.normalize(hp.desc(x2.shape[0]), x2)
        shp = x.shape[:-1] + (1,)
        return xn.reshape(x.shape), mean.reshape(shp), std.reshape(shp)

    def postprocess_output(self, x_reconstruct, x_mean, x_std):
        """sae.py:149-153."""
        if self.config.get("input_unit_norm", False):
            x_reconstruct = x_reconstruct * x_std + x_mean
        return x_reconstruct

    @torch.no_grad()
    def make_decoder_weights_and_grad_unit_norm(self):
        """sae.py:155-163: normalise W_dec rows and remove the radial component of W_dec.grad."""
        if self.W_dec.grad is None:
            raise _lib.MtcError("W_dec.grad is None: call backward() first")
        if not self.W_dec.grad.is_contiguous():
            self.W_dec.grad = self.W_dec.grad.contiguous()
        with self._on_device():
            self.hot_path().unitnorm_project(self.W_dec.data, self.W_dec.grad)

    @torch.no_grad()
    def update_inactive_features(self, acts):
        """sae.py:165-169 for a dense [B, D] activation tensor (the fused forward does this from the
        CSC column pointers instead)."""
        col = acts.sum(0)
        self.num_batches_not_active = torch.where(col > 0, torch.zeros_like(self._cnt), self._cnt + (col == 0).float())

    @torch.no_grad()
    def update_threshold(self, acts_topk, lr=0.01):
        """sae.py:349-361 for a dense tensor (the fused forward uses tau from the select instead)."""
        pos = acts_topk > 0
        if bool(pos.any()):
            self.threshold.copy_((1 - lr) * self.threshold + lr * acts_topk[pos].min())

    # ------------------------------------------------------------------ the fused chain
    def _select_carry(self, hp, d, device):
        """The carried-bound state the training select of this model runs with (what the encoder marks candidates for)."""
        return hp.carry_state(device)

    def _select(self, hp, d, acts, training: bool) -> Selection:
        if training:
            # threshold EMA (a5) rides in the select's tail launch
            sel = hp.select_topk(d, acts, int(self.config["top_k"]) * d.B, carry=True, threshold=self.threshold)
        else:
            sel = hp.select_threshold(d, acts, self.threshold)
        return sel

    def _aux_may_be_active(self) -> bool:
        """Host-side bound: a counter grows by at most 1 per forward, so after reading the device
        maximum once we know how many forwards must pass before any feature can be dead; only then is
        the (synchronising) check repeated.  The reference synchronises on every call (sae.py:331)."""
        if self._dead_check_in > 0:
            return False
        if self._aux_bound is not None and self._aux_bound > 0:
            # the last forward's dead list says how many features CAN be dead now (its counters >= n - 1): no read-back
            return True
        n_dead_at = float(self.config["n_batches_to_dead"])
        mx = float(self._cnt.max().item())
        if mx >= n_dead_at:
            return True
        self._dead_check_in = max(int(math.ceil(n_dead_at - mx)) - 1, 0)
        return False

    _OVERLAP_MIN_ELEMS = 1 << 25       # below this a step is launch-bound and a second stream only adds host work

    def _join_csc(self):
        """Make the current stream wait for a CSC build that was issued on the side stream."""
        ev = getattr(self, "_csc_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._csc_event = None

    def _aux_prepare(self, hp, d, aux_active: bool, side=None, on_side: bool = False) -> bool:
        """Dense aux form: build the dead list (and read its length back) now, before the decode launch, so that the
        host has a long kernel to queue the aux launches behind.  An empty list switches the branch off.
        side: the stream the CSC build (which updates the counters) is being issued on -- with a bound carried from the
        previous forward nothing of the list is needed on the host, so the list is made there (on_side: the caller is
        already inside that stream's context, between the two parts of the build); the aux branch joins it
        (self._dead_event) when it first reads the list."""
        self._aux_dead = None
        self._dead_event = None
        bound, self._aux_bound = self._aux_bound, None
        if not aux_active or self.config.get("aux_mode", "gemm") != "gemm":
            return aux_active
        if side is not None and bound:
            def make():
                dl_ = hp.aux_dead_list(d, self._cnt, float(self.config["n_batches_to_dead"]))
                self._dead_event = torch.cuda.Event()
                self._dead_event.record(side)
                return dl_
            if on_side:
                dl = make()
            else:
                with torch.cuda.stream(side):
                    dl = make()
            dl.cap = bound
            self._aux_dead = dl
            return True
        dl = hp.aux_dead_list(d, self._cnt, float(self.config["n_batches_to_dead"]))
        # launch sizes: the bound the previous forward's list left behind, else (first forward of an active stretch,
        # counters edited from outside) this list's own length, read back once
        dl.cap = bound if bound is not None else dl.n()
        if dl.cap == 0:
            return False
        self._aux_dead = dl
        return True

    def _run_forward(self, x_source, x_target, W_enc, b_enc, W_dec, b_dec, need_grad: bool):
        self._check_inputs(x_source, x_target)
        hp = self.hot_path()
        cfg = self.config
        B = x_source.shape[0]
        d = hp.desc(B)
        x_source = x_source.contiguous()
        x_target = x_target.contiguous()
        if hp.unit_norm:
            xn, mean, std = hp.normalize(d, x_source)
            self.x_mean, self.x_std = mean.reshape(B, 1), std.reshape(B, 1)
        else:
            xn, mean, std = x_source, None, None
            self.x_mean = self.x_std = None
        acts = hp.encode(d, xn, self._w_encT(W_enc), b_enc.contiguous(),
                         mark_for=self._select_carry(hp, d, x_source.device) if self.training else None)
        sel = self._select(hp, d, acts, self.training)
        if self._cnt.device != x_source.device or self._cnt.dtype != torch.float32:
            self._cnt = self._cnt.to(device=x_source.device, dtype=torch.float32)
        # counters BEFORE aux (sae.py:233,242).  In the fused training step the CSC build (a chain of small latency-bound
        # kernels nobody needs before the gradient kernels) runs on a side stream underneath the decode launch.
        side = getattr(self, "_side_stream", None)
        self._csc_event = None
        self._two_streams = side is not None and getattr(self, "_defer_dvals", False) and B * hp.D >= self._OVERLAP_MIN_ELEMS
        self._dead_check_in -= 1
        # steady aux-active state on two streams: the previous list's bound sizes this forward's launches, so neither the
        # check nor the list needs the host -- both stay on the side stream, under the decode launch.  The list goes
        # BETWEEN the two parts of the CSC build (the counters are final after part 1): the chain is starved by the decode
        # launch and ends with it, and the aux branch, first thing after the decode, waits for the list, not for the index
        lagged = (self._two_streams and self._dead_check_in <= 0 and bool(self._aux_bound)
                  and cfg.get("aux_mode", "gemm") == "gemm")
        if self._two_streams:
            main = torch.cuda.current_stream()
            e_sel = torch.cuda.Event()
            e_sel.record(main)
            with torch.cuda.stream(side):                            # workspace / outputs come from the side stream's pool
                side.wait_event(e_sel)
                if lagged:
                    hp.build_csc(d, sel, self._cnt, part=1)
                    aux_active = self._aux_prepare(hp, d, True, side, on_side=True)
                    hp.build_csc(d, sel, self._cnt, part=2)
                else:
                    hp.build_csc(d, sel, self._cnt)
                self._csc_event = torch.cuda.Event()
                self._csc_event.record(side)
        else:
            hp.build_csc(d, sel, self._cnt)
        if not lagged:
            if self._dead_check_in <= 0:
                self._join_csc()                                     # the dead-feature check below reads the counters
            aux_active = self._aux_prepare(hp, d, self._aux_may_be_active(), None)
        x_hat, g, d_vals, row_stats, scalars, aux = self._decode_and_losses(
            hp, d, sel, acts, W_dec.contiguous(), b_dec.contiguous(), x_target, mean, std, need_grad, aux_active)
        from .engine import ForwardResult
        res = ForwardResult(B=B, desc=d, x_norm=xn, mean=mean, std=std, acts=None, sel=sel, x_hat=x_hat,
                            g=g if need_grad else None, d_vals=d_vals, row_stats=row_stats, scalars=scalars, aux=aux)
        # main-pool temporaries a side-stream launch reads (the loss reduction): alive until the next step has joined
        # the side stream, so the caching allocator cannot hand their blocks to a later main-stream allocation
        self._fwd_serial += 1
        res.extra["serial"] = self._fwd_serial
        res.extra["side_stream_inputs"] = self._side_inputs
        self._side_inputs = None
        return res

    def _decode_and_losses(self, hp, d, sel, acts, W_dec, b_dec, x_target, mean, std, need_grad, aux_active):
        """a6-a10: decode + losses (+ aux).  Overridden by the feature-sharded model."""
        cfg = self.config
        l1_term = float(cfg["l1_coeff"]) / (d.B * self.total_dict_size)
        # FusedAdam.train_step defers d a_bj to the decoder-gradient kernel (one gather of g serves both)
        defer = getattr(self, "_defer_dvals", False)
        x_hat, g, row_stats, _, d_vals = hp.decode(
            d, sel, W_dec, b_dec, x_target, mean, std, want_g=need_grad or aux_active,
            want_dvals=need_grad and not aux_active and not defer, l1_term=l1_term)
        aux = None
        aux_rows = None
        self._side_inputs = None
        if aux_active:
            ev = getattr(self, "_dead_event", None)
            if ev is not None:                                       # dead list made on the side stream (see _aux_prepare)
                torch.cuda.current_stream().wait_event(ev)
                self._dead_event = None
            aux, aux_rows = self._run_aux(hp, d, acts, sel, W_dec, x_target, x_hat, g, need_grad, l1_term)
            # g now carries the aux contribution; the fused step (defer) derives d a_bj from it inside the decoder-
            # gradient kernel when the aux branch is in its dense form, otherwise it is needed here
            if need_grad and not (defer and (aux is None or aux.get("dense"))):
                d_vals = hp.decode_bwd_vals(d, sel, W_dec, g, l1_term)
        if getattr(self, "_two_streams", False):
            # fused training step on two streams: the loss / FVU reduction (one small workgroup) leaves the critical
            # path too; FusedAdam.train_step joins the side stream before anything reads the scalars
            main, side = torch.cuda.current_stream(), self._side_stream
            e_rows = torch.cuda.Event()
            e_rows.record(main)
            self._side_inputs = (row_stats, aux_rows, sel.state)
            with torch.cuda.stream(side):
                side.wait_event(e_rows)
                scalars = hp.reduce_scalars(d, row_stats, sel, self.total_dict_size, float(cfg["l1_coeff"]),
                                            float(cfg["aux_penalty"]), aux_rows=aux_rows)
        else:
            scalars = hp.reduce_scalars(d, row_stats, sel, self.total_dict_size, float(cfg["l1_coeff"]),
                                        float(cfg["aux_penalty"]), aux_rows=aux_rows)
        return x_hat, g, d_vals, row_stats, scalars, aux

    def _run_aux(self, hp, d, acts, sel, W_dec, x_target, x_hat, g, need_grad, l1_term):
        cfg = self.config
        std = self.x_std.reshape(-1).contiguous() if hp.unit_norm else None
        if cfg.get("aux_mode", "gemm") == "gemm":
            # the reference's own dense form over the dead set, on MFMA (cheaper than k_aux row gathers per batch row)
            return hp.aux_forward_dense(d, acts, self._cnt, float(cfg["n_batches_to_dead"]), int(cfg["top_k_aux"]),
                                        float(cfg["aux_penalty"]), W_dec, x_target, x_hat, std, g, need_grad,
                                        dead=getattr(self, "_aux_dead", None),
                                        side=self._side_stream if getattr(self, "_two_streams", False) else None)
        asel, g_aux, aux_rows, d_aux = hp.aux_forward(
            d, acts, self._cnt, float(cfg["n_batches_to_dead"]), int(cfg["top_k_aux"]),
            float(cfg["aux_penalty"]), W_dec, x_target, x_hat, std, g, need_grad)
        if need_grad:
            hp.build_csc(d, asel, None)
        return dict(sel=asel, g_aux=g_aux, d_vals=d_aux), aux_rows

    def _run_aux_backward(self, hp, res, g_loss, grads):
        a = res.aux
        if a.get("dense"):
            return hp.aux_backward_dense(a, res.x_norm, g_loss, grads)
        return hp.backward(res.desc, a["sel"], a["g_aux"], a["d_vals"], res.x_norm, g_loss, grads=grads,
                           with_colsum=False)

    def _run_backward(self, res, g_loss):
        hp = self.hot_path()
        grads = hp.backward(res.desc, res.sel, res.g, res.d_vals, res.x_norm, g_loss)
        if res.aux is not None:
            grads = self._run_aux_backward(hp, res, g_loss, grads)
        return grads

    # ------------------------------------------------------------------ public API
    def compute_activations(self, x_source):
        """sae.py:171-193: returns dense (acts, acts_topk) like the reference."""
        self._check_inputs(x_source)
        hp = self.hot_path()
        d = hp.desc(x_source.shape[0])
        with self._on_device():
            acts = hp.encode(d, x_source.contiguous(), self._w_encT(self.W_enc), self.b_enc)
            sel = self._select(hp, d, acts, self.training)
            return acts, hp.densify(d, sel, hp.D)

    def forward(self, x_source, x_target):
        """sae.py:195-269.  Returns the reference's dict; `loss` carries the autograd graph."""
        loss, x_hat, s = _StepFn.apply(self, x_source, x_target, self.W_enc, self.b_enc, self.W_dec, self.b_dec)
        return self._output_dict(self._last_res, x_target, loss=loss)

    def _flush_deferred(self):
        """Work a forward chain postponed until the backward kernels were issued (the sharded model's exchange of the
        earlier prefixes' reconstructions: metrics only, kept off the critical path).  No-op otherwise."""
        fn = self._deferred
        if fn is not None:
            self._deferred = None
            fn()

    def _aux_is_on(self, res) -> bool:
        """Did the aux branch of this forward see a dead feature (sae.py:331)?  Decides the dtypes of `loss` / `aux_loss`
        (sae.py:342-347).  With the dense form the launches were sized from an upper bound, so the answer is the dead
        list's true length: it reached pinned host memory behind a copy that sits BEFORE the decode launch, so in the
        fused step this waits for an event long past, never for the stream.  The bound the same copy carries sizes the
        next forward's aux launches."""
        if res.aux is None:
            return False
        on = res.extra.get("aux_on")
        if on is None:
            on = True
            dead = res.aux.get("dead")
            if dead is not None and res.aux.get("local_counts_decide", True):
                on = dead.n() > 0
                if res.extra.get("serial") == self._fwd_serial:      # still the latest forward: its bound holds for the next
                    self._aux_bound = dead.next_bound()
                    self._dead_check_in = 0
            res.extra["aux_on"] = on
        return on

    def _output_dict(self, res, x_target, loss=None):
        """The reference's 11 keys with its dtypes (sae.py:257-269), from the device-resident scalars."""
        self._flush_deferred()
        s, dt = res.scalars, self.dtype
        aux_on = self._aux_is_on(res)
        # the kernel already rounded l2 / l1 / l1_norm / loss to the model dtype and, for bf16 models, left them as bf16
        # in slots 12..15 of the vector (mtc_reduce_scalars): the entries below are views, no conversion launch per step
        if dt == torch.bfloat16:
            pk = s[12:16].view(torch.bfloat16)              # {l2, l1, l1_norm, 0, aux-free loss, 0, 0, 0}
            l2, l1, l1n, zero, loss_dt = pk[0], pk[1], pk[2], pk[3], pk[4]
        else:
            l2, l1, l1n, zero, loss_dt = s[0], s[1], s[3], s[4], s[5]
        if loss is None:
            loss = s[5].clone() if aux_on else loss_dt
        # aux inactive: a 0-dim zero in the target dtype (slot 4 holds 0 then); active: fp32 (sae.py:342-347)
        if aux_on:
            aux_loss = s[4]
        elif x_target.dtype == dt:
            aux_loss = zero
        else:
            aux_loss = torch.zeros((), dtype=x_target.dtype, device=x_target.device)
        return {
            "transcoder_out": res.x_hat,
            "loss": loss,
            "l2_loss": l2,
            "l1_loss": l1,
            "l0_norm": s[2],
            "l1_norm": l1n,
            "aux_loss": aux_loss,
            "fvu": s[6],
            "fvu_min": s[7],
            "fvu_max": s[8],
            "fvu_mean": s[9],
        }

    def encode(self, x_source):
        """sae.py:363-387: dense sparse-feature tensor, 2-D or 3-D input; honours train/eval mode."""
        original_shape = x_source.shape
        x_source, x_mean, x_std = self.preprocess_input(x_source)
        self.x_mean, self.x_std = x_mean, x_std
        x2 = x_source.reshape(-1, x_source.shape[-1]).contiguous()
        self._check_inputs(x2)
        hp = self.hot_path()
        d = hp.desc(x2.shape[0])
        max_col = self.prefix_indices[self.active_prefixes]
        with self._on_device():
            if not self.training and max_col == hp.D:
                # eval mode: the threshold rides in the encoder's epilogue; the dense result is written once
                result = hp.encode_threshold(d, x2, self._w_encT(self.W_enc), self.b_enc, self.threshold)
            else:
                acts = hp.encode(d, x2, self._w_encT(self.W_enc), self.b_enc)
                sel = self._select(hp, d, acts, self.training)
                result = hp.densify(d, sel, max_col)
        if len(original_shape) == 3:
            result = result.reshape(original_shape[0], original_shape[1], -1)
        return result

    def decode(self, acts_topk):
        """sae.py:389-406: reconstruction from a dense [N, D] feature tensor, de-normalised with the
        statistics remembered from the last encode()."""
        if self.x_mean is None and self.config.get("input_unit_norm", False):
            # the reference raises AttributeError here too: its x_mean / x_std only exist after an encode() (sae.py:406)
            raise AttributeError("decode() needs the statistics of a previous encode()")
        self._check_inputs(acts_topk)
        hp = self.hot_path()
        max_idx = self.prefix_indices[self.active_prefixes]
        with self._on_device():
            if max_idx != hp.D:
                sub = HotPath(self.source_act_size, self.target_act_size, [max_idx], self.dtype, False)
                return self._decode_with(sub, acts_topk[:, :max_idx].contiguous(), self.W_dec[:max_idx])
            return self._decode_with(hp, acts_topk.contiguous(), self.W_dec)

    def _decode_with(self, hp, dense, W_dec):
        import ctypes as C
        from .engine import _p, _stream
        N = dense.shape[0]
        unit = bool(self.config.get("input_unit_norm", False))
        mean = std = None
        if unit:
            if self.x_mean is None or self.x_mean.numel() != N:
                raise RuntimeError(
                    f"decode(): statistics of the last encode() cover {0 if self.x_mean is None else self.x_mean.numel()} "
                    f"rows but {N} rows were given")
            mean, std = self.x_mean.reshape(N).contiguous(), self.x_std.reshape(N).contiguous()
        d = _lib.make_desc(N, hp.A_s, hp.A_t, hp.D, hp.prefix, hp.code, unit)
        nan = torch.full((1,), float("nan"), dtype=self.dtype, device=dense.device)     # "non-zero" select mode
        sel = hp.select_threshold(d, dense, nan)
        x_hat = torch.empty(N, hp.A_t, dtype=self.dtype, device=dense.device)
        _lib.check(hp.lib.mtc_decode_fwd(C.byref(d), _p(sel.row_ptr), _p(sel.col_idx), _p(sel.vals),
                                         _p(W_dec.contiguous()), _p(self.b_dec), None, _p(mean), _p(std), _p(x_hat), None,
                                         None, None, None, 0.0, _stream()), "mtc_decode_fwd")
        return x_hat
