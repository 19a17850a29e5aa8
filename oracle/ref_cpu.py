"""CPU oracle for the Matryoshka-transcoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped package (``matryoshka_transcoder_amd``) never imports anything
from ``oracle/`` and fails loudly when its HIP library is missing.

It restates, from scratch and in *sparse* functional form, the algorithm of the
reference (paths relative to the reference checkout):

  * parameter init ............ src/models/sae.py:84-137
  * input (de)normalisation ... src/models/sae.py:139-153
  * encoder + ReLU ............ src/models/sae.py:171-174
  * global batch-TopK ......... src/models/sae.py:176-188
  * eval-mode threshold ....... src/models/sae.py:189-191
  * threshold EMA ............. src/models/sae.py:349-361
  * nested prefix decode ...... src/models/sae.py:217-230
  * dead-feature counters ..... src/models/sae.py:165-169
  * losses .................... src/models/sae.py:235-245
  * aux dead-feature loss ..... src/models/sae.py:318-347
  * FVU ....................... src/utils/config.py:205-234, sae.py:248-268
  * unit-norm projection ...... src/models/sae.py:155-163
  * Adam ...................... src/training/training.py:60,178 (torch.optim.Adam
                                single-tensor update, PyTorch 2.10)
  * LR lambdas ................ src/training/training.py:12-25

Parity pin: ``tools/make_golden.py`` imports the reference class itself (it is
importable in the build container with an in-memory ``transformer_lens`` stub)
and writes ``tests/golden/*.pt``; ``tests/test_oracle_vs_golden.py`` checks this
file against those fixtures.  The reference ships no tests or vectors of its own.

Numerics model (shared with the HIP kernels, so GPU-vs-oracle can be tight):
all contractions accumulate in fp32; values are rounded to the model dtype
(``dt`` = bf16 or fp32) exactly where the reference rounds something that
feeds an *index decision* (GEMM result, then ``+ b_enc``), and once at each
tensor the reference materialises in ``dt`` (reconstructions, losses).

Tie policy for the global TopK (``torch.topk`` leaves it unspecified): every
value ``> tau`` is selected, then values ``== tau`` by ascending flat index
until ``k*B`` are taken; zeros are never selected (identical dense result).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

F32 = torch.float32


# --------------------------------------------------------------------------- config helpers
def cumulative_prefixes(group_sizes: List[int]) -> List[int]:
    """Group sizes (README/BASELINE lists) -> cumulative ``prefix_sizes`` the model consumes
    (sae.py:94-105: the last cumulative entry IS the dictionary size)."""
    out, run = [], 0
    for g in group_sizes:
        run += int(g)
        out.append(run)
    return out


def check_prefixes(prefix_sizes: List[int]) -> None:
    """sae.py:99-103."""
    for a, b in zip(prefix_sizes[:-1], prefix_sizes[1:]):
        if a > b:
            raise ValueError(
                "prefix_sizes must be monotonically increasing (cumulative format). "
                f"Got: {prefix_sizes}")


# --------------------------------------------------------------------------- init (sae.py:84-137)
def init_params(cfg: dict) -> Dict[str, torch.Tensor]:
    """Same CPU RNG stream as the reference: seed, W_enc draw, W_dec draw."""
    check_prefixes(cfg["prefix_sizes"])
    D = cfg["prefix_sizes"][-1]
    a_s = cfg.get("source_act_size", cfg.get("act_size"))
    a_t = cfg.get("target_act_size", cfg.get("act_size"))
    torch.manual_seed(cfg["seed"])
    # kaiming_uniform_(a=0, fan_in = size(1)): U(-sqrt(6/fan_in), +sqrt(6/fan_in))
    W_enc = torch.empty(a_s, D).uniform_(-math.sqrt(6.0 / D), math.sqrt(6.0 / D))
    W_dec = torch.empty(D, a_t).uniform_(-math.sqrt(6.0 / a_t), math.sqrt(6.0 / a_t))
    W_dec = W_dec / W_dec.norm(dim=-1, keepdim=True)
    dt = cfg["dtype"]
    return {
        "b_enc": torch.zeros(D).to(dt),
        "W_enc": W_enc.to(dt),
        "W_dec": W_dec.to(dt),
        "b_dec": torch.zeros(a_t).to(dt),
        "threshold": torch.tensor(0.0).to(dt),
    }


def init_state(cfg: dict) -> Dict[str, torch.Tensor]:
    """Non-checkpointed state: dead counters are fp32 whatever the model dtype (sae.py:134)."""
    return {"num_batches_not_active": torch.zeros(cfg["prefix_sizes"][-1], dtype=F32)}


# --------------------------------------------------------------------------- a2 (sae.py:139-153)
def preprocess(x: torch.Tensor, unit_norm: bool):
    """Row standardisation in the model dtype; each step rounds to ``dt`` as the reference's
    eager ops do."""
    if not unit_norm:
        return x, None, None
    mean = x.mean(dim=-1, keepdim=True)
    xc = x - mean
    std = xc.std(dim=-1, keepdim=True)
    return xc / (std + 1e-5), mean, std


def postprocess(recon: torch.Tensor, mean, std):
    if mean is None:
        return recon
    return recon * std + mean


# --------------------------------------------------------------------------- a3 encoder
def encode_acts(x: torch.Tensor, W_enc: torch.Tensor, b_enc: torch.Tensor) -> torch.Tensor:
    """relu(round(round(x.W_enc) + b_enc)) (sae.py:173-174); fp32 accumulation."""
    dt = x.dtype
    pre = (x.to(F32) @ W_enc.to(F32)).to(dt)
    pre = (pre.to(F32) + b_enc.to(F32)).to(dt)
    return torch.relu(pre)


# --------------------------------------------------------------------------- a3 global select
def global_topk_select(acts: torch.Tensor, k_total: int):
    """Global k_total-th-largest threshold select over the flattened [B, D] activations.

    Returns (tau, flat_idx ascending int64, n_gt, n_tie_taken).  ``tau`` is a 0-dim tensor in
    acts.dtype; ``tau == 0`` means fewer than k_total positives (all positives selected).
    """
    flat = acts.flatten()
    n = flat.numel()
    k_total = min(k_total, n)
    f = flat.to(F32)
    # k-th largest == (n-k+1)-th smallest
    tau = torch.kthvalue(f, n - k_total + 1).values
    if tau <= 0:
        idx = torch.nonzero(f > 0).flatten()
        return torch.zeros((), dtype=acts.dtype), idx, int(idx.numel()), 0
    gt = torch.nonzero(f > tau).flatten()
    eq = torch.nonzero(f == tau).flatten()
    take = k_total - gt.numel()
    idx = torch.sort(torch.cat([gt, eq[:take]])).values
    return tau.to(acts.dtype), idx, int(gt.numel()), int(take)


def to_csr(flat_idx: torch.Tensor, flat_vals: torch.Tensor, B: int, D: int):
    """Row-major CSR of the selected entries; columns ascending inside a row (= prefix-group
    order, because groups are contiguous feature ranges)."""
    rows = torch.div(flat_idx, D, rounding_mode="floor")
    cols = (flat_idx - rows * D).to(torch.int32)
    counts = torch.bincount(rows, minlength=B)
    row_ptr = torch.zeros(B + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return row_ptr, cols, flat_vals


# --------------------------------------------------------------------------- a5 (sae.py:349-361)
def ema_threshold(threshold: torch.Tensor, min_positive: Optional[torch.Tensor], lr: float = 0.01):
    """threshold <- (1-lr)*threshold + lr*min_positive, evaluated in the buffer dtype."""
    if min_positive is None:
        return threshold
    return (1 - lr) * threshold + lr * min_positive.to(threshold.dtype)


# --------------------------------------------------------------------------- a7 (sae.py:165-169)
def update_dead_counters(cnt: torch.Tensor, fired: torch.Tensor) -> torch.Tensor:
    """cnt += 1 where the feature did not fire in this batch; cnt = 0 where it did."""
    return torch.where(fired, torch.zeros_like(cnt), cnt + 1.0)


# --------------------------------------------------------------------------- a10 (config.py:205-234)
def fvu_from_sse(sse: torch.Tensor, n_elem: int, var_target: torch.Tensor) -> torch.Tensor:
    return (sse / n_elem) / (var_target + 1e-8)


# --------------------------------------------------------------------------- forward (sae.py:195-269)
def forward(params, state, x_source, x_target, cfg, training: bool = True, acts_override=None):
    """One forward of the transcoder in sparse form.  Mutates nothing; returns

      out    : the reference's 11-key dict (sae.py:257-269)
      new    : {"threshold", "num_batches_not_active"} after the call's side effects
      inter  : intermediates the kernels are checked against
    """
    dt = params["W_enc"].dtype
    B = x_source.shape[0]
    prefix = list(cfg["prefix_sizes"])
    D = prefix[-1]
    A_t = params["W_dec"].shape[1]
    unit = bool(cfg.get("input_unit_norm", False))
    G = len(prefix)

    xn, mean, std = preprocess(x_source, unit)
    acts = encode_acts(xn, params["W_enc"], params["b_enc"])                      # [B, D] dt
    if acts_override is not None:       # tests: decouple the select/decode checks from GEMM rounding
        acts = acts_override

    new_threshold = params["threshold"]
    if training:
        tau, flat_idx, n_gt, n_tie = global_topk_select(acts, cfg["top_k"] * B)
        vals = acts.flatten()[flat_idx]
        if flat_idx.numel() > 0:
            new_threshold = ema_threshold(params["threshold"], vals.min())
    else:
        flat_idx = torch.nonzero(acts.flatten() > params["threshold"]).flatten()
        vals = acts.flatten()[flat_idx]
        tau, n_gt, n_tie = params["threshold"], int(flat_idx.numel()), 0
    row_ptr, cols, vals = to_csr(flat_idx, vals, B, D)
    rows = torch.div(flat_idx, D, rounding_mode="floor")

    # nested prefix decode, incremental per group (sae.py:219-227)
    Wd = params["W_dec"].to(F32)
    contrib = vals.to(F32)[:, None] * Wd[cols.long()]                             # [nnz, A_t]
    recon32 = params["b_dec"].to(F32).expand(B, A_t).clone()
    recons, sse = [], []
    xt32 = x_target.to(F32)
    var_t = xt32.var()
    lo = 0
    for P in prefix:
        m = (cols >= lo) & (cols < P)
        recon32 = recon32.index_add(0, rows[m], contrib[m])
        r_dt = postprocess(recon32.to(dt), mean, std)                             # materialised in dt
        recons.append(r_dt)
        sse.append(((xt32 - r_dt.to(F32)) ** 2).sum())
        lo = P
    x_hat = recons[-1]
    n_elem = B * A_t

    # dead counters BEFORE the aux loss reads them (sae.py:233 then :242)
    fired = torch.zeros(D, dtype=torch.bool)
    fired[cols.long()] = True
    cnt = update_dead_counters(state["num_batches_not_active"], fired)

    # losses (sae.py:236-239); l2/l1 in the model dtype, l0 fp32
    l2 = (sse[-1] / n_elem).to(dt)
    abs_sum = vals.to(F32).abs().sum()
    l1 = (abs_sum / (B * D)).to(dt)
    l0 = torch.tensor(float(flat_idx.numel()) / B, dtype=F32)
    l1_norm = (abs_sum / B).to(dt)

    # aux dead-feature loss (sae.py:318-347)
    dead = cnt >= cfg["n_batches_to_dead"]
    n_dead = int(dead.sum())
    aux = None
    if n_dead > 0:
        dead_idx = torch.nonzero(dead).flatten()
        k_aux = min(cfg["top_k_aux"], n_dead)
        sub = acts[:, dead_idx]                                                   # [B, n_dead] dt
        # per-row top-k_aux; ties in ascending feature order (torch.topk leaves ties unspecified)
        tv, ti = torch.sort(sub.to(F32), dim=-1, descending=True, stable=True)
        tv, ti = tv[:, :k_aux], ti[:, :k_aux]
        keep = tv > 0                                                             # zeros add nothing
        a_rows = torch.arange(B)[:, None].expand(B, k_aux)[keep]
        a_cols = dead_idx[ti][keep]
        a_vals = tv[keep]
        recon_aux = torch.zeros(B, A_t, dtype=F32).index_add(
            0, a_rows, a_vals[:, None] * Wd[a_cols])
        recon_aux = recon_aux.to(dt).to(F32)                                      # matmul output in dt
        residual = xt32 - x_hat.to(F32)
        e = recon_aux - residual
        aux_loss = cfg["aux_penalty"] * (e ** 2).mean()                           # fp32
        aux = dict(rows=a_rows, cols=a_cols, vals=a_vals, e=e, recon_aux=recon_aux)
        loss = l2.to(F32) + cfg["l1_coeff"] * l1.to(F32) + aux_loss
    else:
        aux_loss = torch.tensor(0, dtype=dt)
        loss = l2 + cfg["l1_coeff"] * l1 + aux_loss

    fvus = [fvu_from_sse(s, n_elem, var_t) for s in sse]
    out = {
        "transcoder_out": x_hat, "loss": loss, "l2_loss": l2, "l1_loss": l1, "l0_norm": l0,
        "l1_norm": l1_norm, "aux_loss": aux_loss, "fvu": fvus[-1],
        "fvu_min": torch.stack(fvus).min(), "fvu_max": torch.stack(fvus).max(),
        "fvu_mean": torch.stack(fvus).mean(),
    }
    new = {"threshold": new_threshold, "num_batches_not_active": cnt}
    inter = dict(x_norm=xn, mean=mean, std=std, acts=acts, tau=tau, flat_idx=flat_idx,
                 n_gt=n_gt, n_tie=n_tie, row_ptr=row_ptr, cols=cols, vals=vals, rows=rows,
                 recons=recons, sse=torch.stack(sse), var_target=var_t, fired=fired, aux=aux,
                 n_dead=n_dead)
    return out, new, inter


# --------------------------------------------------------------------------- a11 analytic backward
def backward(params, x_source, x_target, cfg, inter, grad_out: float = 1.0):
    """Gradients of ``loss`` w.r.t. the four parameters, derived by hand (no autograd):

      g        = dL/dx_hat = 2/(B*A_t) * [(x_hat - x_t) + aux_penalty * e]     (aux term only if active)
      g_raw    = g * std_row                     (unit-norm de-normalisation, sae.py:151-152)
      db_dec   = sum_b g_raw
      dW_dec[j]= sum_b a_bj g_raw_b              (+ aux rows: a_aux_bj * g_aux_b)
      d a_bj   = g_raw_b . W_dec[j] + l1_coeff/(B*D)      on selected entries
      d pre    = d a (selected values are > 0, so the ReLU mask is 1)
      dW_enc   = x_norm^T d_pre ;  db_enc = sum_b d_pre
    """
    dt = params["W_enc"].dtype
    B = x_source.shape[0]
    D = cfg["prefix_sizes"][-1]
    A_t = params["W_dec"].shape[1]
    Wd = params["W_dec"].to(F32)
    xn = inter["x_norm"].to(F32)
    x_hat = inter["recons"][-1].to(F32)
    xt = x_target.to(F32)
    rows, cols, vals = inter["rows"], inter["cols"].long(), inter["vals"].to(F32)
    # dL/dx_hat exactly as autograd forms it on model-dtype tensors (checked against torch's bf16 autograd: every entry
    # equal; a single rounding of 2/N * (x_hat - x_t) * std reproduces only ~60 % of them):
    #   d = round(x_post - x_t); g_post = round(inv_n * 2 d), inv_n = 1/N as a model-dtype scalar (mean's backward);
    #   aux active: x_post has a second consumer, the fp32 residual: its gradient fl32(penalty / N) * 2 e arrives
    #   rounded (the .float() cast's backward) and the two are summed in the model dtype;
    #   de-normalisation backward: g_rec = round(g_post * std)
    rnd = lambda t: t.to(dt).to(F32)
    n_elem = B * A_t
    inv_n = rnd(torch.tensor(1.0 / n_elem, dtype=F32))
    g = rnd(inv_n * (2.0 * rnd(x_hat - xt)))
    aux = inter["aux"]
    g_aux = None
    if aux is not None:
        coef_aux = torch.tensor(cfg["aux_penalty"] / n_elem, dtype=F32)
        g_aux = rnd(coef_aux * (2.0 * aux["e"]))              # dL/d recon_aux as the model-dtype matmul output receives it
        g = rnd(g + g_aux)
    g = g * grad_out
    g_raw = rnd(g * inter["std"].to(F32)) if inter["std"] is not None else g

    db_dec = g_raw.sum(0)
    dW_dec = torch.zeros(D, A_t, dtype=F32).index_add(0, cols, vals[:, None] * g_raw[rows])
    # d loss / d acts_topk is a matmul OUTPUT in the reference's autograd (g @ W_dec^T: fp32 accumulation, one rounding to
    # the model dtype); the l1 term reaches it as a second gradient accumulated in that dtype (both no-ops for fp32)
    d_vals = (g_raw[rows] * Wd[cols]).sum(-1).to(dt).to(F32)
    if cfg["l1_coeff"]:
        d_vals = (d_vals + torch.tensor(grad_out * cfg["l1_coeff"] / (B * D)).to(dt).to(F32)).to(dt).to(F32)

    d_pre_rows, d_pre_cols, d_pre_vals = rows, cols, d_vals
    if aux is not None:
        g_aux = g_aux * grad_out
        ar, ac, av = aux["rows"], aux["cols"], aux["vals"]
        dW_dec = dW_dec.index_add(0, ac, av[:, None] * g_aux[ar])
        d_aux = (g_aux[ar] * Wd[ac]).sum(-1).to(dt).to(F32)
        d_pre_rows = torch.cat([rows, ar]); d_pre_cols = torch.cat([cols, ac])
        d_pre_vals = torch.cat([d_vals, d_aux])

    A_s = xn.shape[1]
    dW_encT = torch.zeros(D, A_s, dtype=F32).index_add(0, d_pre_cols, d_pre_vals[:, None] * xn[d_pre_rows])
    db_enc = torch.zeros(D, dtype=F32).index_add(0, d_pre_cols, d_pre_vals)
    return {"W_enc": dW_encT.t().contiguous(), "b_enc": db_enc, "W_dec": dW_dec, "b_dec": db_dec,
            "g": g_raw, "d_vals": d_vals}


# --------------------------------------------------------------------------- a12 (sae.py:155-163)
def unit_norm_project(W_dec: torch.Tensor, grad: torch.Tensor):
    """Returns (normalised weights, projected grad); evaluated in the parameter dtype."""
    Wn = W_dec / W_dec.norm(dim=-1, keepdim=True)
    proj = (grad * Wn).sum(-1, keepdim=True) * Wn
    return Wn, grad - proj


# --------------------------------------------------------------------------- a13 Adam
def adam_step(p, g, m, v, step: int, lr: float, beta1: float, beta2: float, eps: float = 1e-8):
    """torch.optim.Adam single-tensor update (no weight decay / amsgrad), evaluated op by op in
    the parameter dtype exactly as PyTorch's ``_single_tensor_adam`` does.  ``step`` is 1-based.
    Returns new (p, m, v)."""
    m = m.lerp(g, 1 - beta1)
    v = (v * beta2).addcmul(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p.addcdiv(m, denom, value=-(lr / bc1))
    return p, m, v


# --------------------------------------------------------------------------- schedules (training.py:12-25)
def lr_factor(scheduler_type: str, step: int, warmup_steps: int, total_steps: int) -> float:
    if scheduler_type == "warmup_only":
        return min(1.0, step / warmup_steps)
    if scheduler_type == "warmup_decay":
        if step < warmup_steps:
            return step / warmup_steps
        progress = (step - warmup_steps) / (total_steps - warmup_steps)
        return 0.5 * (1 + math.cos(math.pi * progress))
    return 1.0


# --------------------------------------------------------------------------- one whole train step
def train_step(params, state, opt, x_source, x_target, cfg, lr: Optional[float] = None, acts_override=None):
    """forward -> analytic backward -> unit-norm projection -> Adam (training.py:172-179).
    ``opt`` = {"step": int, "m": {name: tensor}, "v": {name: tensor}}; returns (out, inter, grads)
    and updates params/state/opt in place (dict entries rebound).  ``acts_override``: see forward()."""
    out, new, inter = forward(params, state, x_source, x_target, cfg, training=True, acts_override=acts_override)
    grads32 = backward(params, x_source, x_target, cfg, inter)
    dt = params["W_enc"].dtype
    grads = {k: grads32[k].to(dt) for k in ("b_enc", "W_enc", "W_dec", "b_dec")}
    Wn, gproj = unit_norm_project(params["W_dec"], grads["W_dec"])
    params["W_dec"] = Wn
    grads["W_dec"] = gproj
    opt["step"] += 1
    lr = cfg["lr"] if lr is None else lr
    for name in ("b_enc", "W_enc", "W_dec", "b_dec"):
        p, m, v = adam_step(params[name], grads[name], opt["m"][name], opt["v"][name],
                            opt["step"], lr, cfg["beta1"], cfg["beta2"])
        params[name], opt["m"][name], opt["v"][name] = p, m, v
    params["threshold"] = new["threshold"]
    state["num_batches_not_active"] = new["num_batches_not_active"]
    return out, inter, grads


def new_opt(params) -> dict:
    names = ("b_enc", "W_enc", "W_dec", "b_dec")
    return {"step": 0, "m": {n: torch.zeros_like(params[n]) for n in names},
            "v": {n: torch.zeros_like(params[n]) for n in names}}


# --------------------------------------------------------------------------- dense timing port
class DenseStep:
    """Dense eager-PyTorch port of the reference's step, for the ``cpu_baseline`` timing leg of
    bench.py ("kind": "port").  It issues the same ATen call sequence the reference does
    (matmul, relu, flattened topk, scatter, G dense prefix matmuls, autograd backward,
    projection, torch.optim.Adam) so its cost tracks the reference's CPU path
    (sae.py:171-269, training.py:172-179).  Numerically it is checked against the golden
    fixtures in tests/test_oracle_vs_golden.py.
    """

    def __init__(self, cfg: dict):
        p = init_params(cfg)
        self.cfg = cfg
        self.names = ("b_enc", "W_enc", "W_dec", "b_dec")
        self.p = {n: p[n].clone().requires_grad_(True) for n in self.names}
        self.threshold = p["threshold"]
        self.cnt = init_state(cfg)["num_batches_not_active"]
        self.opt = torch.optim.Adam([self.p[n] for n in self.names], lr=cfg["lr"],
                                    betas=(cfg["beta1"], cfg["beta2"]))

    def loss(self, x_source, x_target):
        cfg, p = self.cfg, self.p
        xs, mean, std = preprocess(x_source, bool(cfg.get("input_unit_norm", False)))
        acts = torch.relu(xs @ p["W_enc"] + p["b_enc"])
        B = xs.shape[0]
        top = torch.topk(acts.flatten(), cfg["top_k"] * B, dim=-1)
        sparse = torch.zeros_like(acts.flatten()).scatter(-1, top.indices, top.values).reshape(acts.shape)
        with torch.no_grad():
            pos = sparse > 0
            if pos.any():
                self.threshold = ema_threshold(self.threshold, sparse[pos].min())
        recons = [sparse[:, :P] @ p["W_dec"][:P] + p["b_dec"] for P in cfg["prefix_sizes"]]
        x_hat = postprocess(recons[-1], mean, std)
        with torch.no_grad():
            col = sparse.sum(0)
            self.cnt = torch.where(col > 0, torch.zeros_like(self.cnt), self.cnt + (col == 0).float())
        l2 = (x_hat - x_target).pow(2).mean()
        l1 = sparse.abs().mean()
        loss = l2 + cfg["l1_coeff"] * l1
        dead = self.cnt >= cfg["n_batches_to_dead"]
        if dead.sum() > 0:
            resid = x_target.float() - x_hat.float()
            sub = acts[:, dead]
            t = torch.topk(sub, min(cfg["top_k_aux"], int(dead.sum())), dim=-1)
            a_aux = torch.zeros_like(sub).scatter(-1, t.indices, t.values)
            loss = loss + cfg["aux_penalty"] * ((a_aux @ p["W_dec"][dead]).float() - resid).pow(2).mean()
        xt32 = x_target.float()
        var_t = xt32.var()
        fv = [(xt32 - postprocess(r, mean, std).float()).pow(2).mean() / (var_t + 1e-8) for r in recons]
        return loss, fv

    def step(self, x_source, x_target):
        loss, _ = self.loss(x_source, x_target)
        loss.backward()
        with torch.no_grad():
            W = self.p["W_dec"]
            Wn, gp = unit_norm_project(W.data, W.grad)
            W.grad.copy_(gp)
            W.data = Wn
        self.opt.step()
        self.opt.zero_grad()
        return loss.detach()
