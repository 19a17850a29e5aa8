"""Host-side sequencing of the HIP hot path (libmtc.so) on torch device tensors.

PyTorch is plumbing here: it owns device memory (caching allocator) and the stream; every
computation is a libmtc kernel launched on `torch.cuda.current_stream()`.  Nothing in this file
synchronises the device except where the reference itself needs a host value.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib
from ._lib import AdamHyper, check

_CODES = {torch.float32: _lib.MTC_F32, torch.bfloat16: _lib.MTC_BF16}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _CODES[dt]
    except KeyError:
        raise _lib.MtcError(f"unsupported model dtype {dt}: libmtc computes in float32 or bfloat16") from None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  The raw getter skips the Python Stream object
    (about 8 lookups per training step, several microseconds each: launch-bound configurations notice)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: torch.Tensor, name: str, dt=None, shape=None):
    if not t.is_cuda:
        raise _lib.MtcError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if not t.is_contiguous():
        raise _lib.MtcError(f"{name} must be contiguous")
    if dt is not None and t.dtype != dt:
        raise _lib.MtcError(f"{name} must be {dt}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.MtcError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")


@dataclass
class Selection:
    """Row-major (CSR) selected activations of one batch + the feature-major (CSC) view of them."""
    state: torch.Tensor            # int32[16]  (mtc_select_state)
    row_ptr: torch.Tensor          # int32[B+1]
    col_idx: torch.Tensor          # int32[cap]
    vals: torch.Tensor             # dt[cap]
    col_ptr: Optional[torch.Tensor] = None
    csc_row: Optional[torch.Tensor] = None
    csc_pos: Optional[torch.Tensor] = None
    _csc_ws: Optional[torch.Tensor] = None      # workspace of a CSC build between its two parts


@dataclass
class ForwardResult:
    B: int
    desc: object
    x_norm: torch.Tensor
    mean: Optional[torch.Tensor]
    std: Optional[torch.Tensor]
    acts: Optional[torch.Tensor]
    sel: Selection
    x_hat: torch.Tensor
    g: Optional[torch.Tensor]
    d_vals: Optional[torch.Tensor]
    row_stats: torch.Tensor
    scalars: torch.Tensor          # float32[16], see mtc_reduce_scalars
    recons: Optional[torch.Tensor] = None
    aux: Optional[dict] = None
    extra: dict = field(default_factory=dict)


class DeadList:
    """The dead-feature list of one forward (a9).  `counts` = device {n_dead, upper bound of the next forward's n_dead};
    `host` receives them through an asynchronous copy, `n()` / `next_bound()` wait for that copy only (not for the
    stream); `cap` = the host-side upper bound of n_dead the launches of this forward are sized with."""
    __slots__ = ("dead_idx", "counts", "row_map", "host", "event", "cap")

    def __init__(self):
        self.cap = None

    def n(self) -> int:
        self.event.synchronize()
        return int(self.host[0])

    def next_bound(self) -> int:
        self.event.synchronize()
        return int(self.host[1])


class HotPath:
    """One instance per transcoder; stateless apart from the static problem description."""

    def __init__(self, A_s: int, A_t: int, prefix: List[int], dtype: torch.dtype, unit_norm: bool):
        self.lib = _lib.load()
        self.A_s, self.A_t, self.prefix = int(A_s), int(A_t), [int(p) for p in prefix]
        self.D = self.prefix[-1]
        self.dtype = dtype
        self.code = dtype_code(dtype)
        self.unit_norm = bool(unit_norm)
        self._carry = {}                 # device -> 8 x uint32 carried lower bound of the TopK threshold
        self.use_marks = True            # encoder epilogue marks candidate granules for the select (tests switch it off to A/B)
        self._pinned = {}                # device -> ring of pinned int32[2] (dead-list counts on their way to the host)

    def desc(self, B: int):
        return _lib.make_desc(B, self.A_s, self.A_t, self.D, self.prefix, self.code, self.unit_norm)

    # ------------------------------------------------------------------ a2
    def normalize(self, d, x):
        B = d.B
        xn = torch.empty_like(x)
        mean = torch.empty(B, dtype=self.dtype, device=x.device)
        std = torch.empty(B, dtype=self.dtype, device=x.device)
        check(self.lib.mtc_row_normalize(C.byref(d), _p(x), _p(xn), _p(mean), _p(std), _stream()), "mtc_row_normalize")
        return xn, mean, std

    # ------------------------------------------------------------------ a3 dense part
    def encode(self, d, x_norm, W_encT, b_enc, mark_for=None):
        """a3 dense part.  mark_for = the carried-bound state of the select that will follow (training loop): the launch
        then also marks the granules that may hold TopK candidates (mtc_encode_fwd_marked) and the marks travel with the
        returned tensor (`acts._mtc_marks`) to select_topk / the sharded select; shapes that do not run the marking
        kernel return plain activations."""
        _need(W_encT, "W_encT", self.dtype, (self.D, self.A_s))
        _need(b_enc, "b_enc", self.dtype, (self.D,))
        acts = torch.empty(d.B, self.D, dtype=self.dtype, device=x_norm.device)
        nbytes = self.lib.mtc_encode_marks_bytes(C.byref(d)) if mark_for is not None and self.use_marks else 0
        if nbytes:
            marks = torch.empty(nbytes, dtype=torch.uint8, device=x_norm.device)
            check(self.lib.mtc_encode_fwd_marked(C.byref(d), _p(x_norm), _p(W_encT), _p(b_enc), _p(acts), _p(mark_for), _p(marks),
                                                 _stream()), "mtc_encode_fwd_marked")
            acts._mtc_marks = marks
            return acts
        check(self.lib.mtc_encode_fwd(C.byref(d), _p(x_norm), _p(W_encT), _p(b_enc), _p(acts), _stream()), "mtc_encode_fwd")
        return acts

    def encode_threshold(self, d, x_norm, W_encT, b_enc, threshold):
        """a4 + a15: eval-mode dense features in ONE launch (threshold applied in the encoder's epilogue)."""
        _need(W_encT, "W_encT", self.dtype, (self.D, self.A_s))
        _need(b_enc, "b_enc", self.dtype, (self.D,))
        thr = threshold.reshape(1).to(self.dtype).contiguous()
        feats = torch.empty(d.B, self.D, dtype=self.dtype, device=x_norm.device)
        check(self.lib.mtc_encode_threshold_infer(C.byref(d), _p(x_norm), _p(W_encT), _p(b_enc), _p(thr), _p(feats),
                                                  _stream()), "mtc_encode_threshold_infer")
        return feats

    # ------------------------------------------------------------------ a3 select / a4 threshold
    def carry_state(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._carry:
            self._carry[key] = torch.zeros(8, dtype=torch.int32, device=device)
        return self._carry[key]

    def select_topk(self, d, acts, k_total: int, carry: bool = False, threshold=None, ema_lr: float = 0.01) -> Selection:
        """Global batch-TopK.  carry=True (training loop): use / refresh the lower bound of tau carried from
        the previous call on this device (identical result, one pass over acts instead of three); `threshold` (the
        module's buffer) then also receives its EMA update (a5) inside the select's tail launch."""
        dev = acts.device
        st = torch.empty(16, dtype=torch.int32, device=dev)
        row_ptr = torch.empty(d.B + 1, dtype=torch.int32, device=dev)
        cap = max(int(k_total), 1)
        col_idx = torch.empty(cap, dtype=torch.int32, device=dev)
        vals = torch.empty(cap, dtype=self.dtype, device=dev)
        if carry:
            ws_bytes = self.lib.mtc_select_carry_workspace_bytes(C.byref(d))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            if threshold is not None:
                _need(threshold, "threshold", self.dtype)
            check(self.lib.mtc_select_global_topk_carry_marked(C.byref(d), _p(acts), int(k_total), _p(ws), ws_bytes,
                                                               _p(self.carry_state(dev)), _p(st), _p(row_ptr), _p(col_idx),
                                                               _p(vals), _p(threshold), float(ema_lr),
                                                               _p(getattr(acts, "_mtc_marks", None)), _stream()),
                  "mtc_select_global_topk_carry")
            return Selection(st, row_ptr, col_idx, vals)
        ws_bytes = self.lib.mtc_select_workspace_bytes(C.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(self.lib.mtc_select_global_topk(C.byref(d), _p(acts), int(k_total), None, _p(ws), ws_bytes, _p(st),
                                              _p(row_ptr), _p(col_idx), _p(vals), _stream()), "mtc_select_global_topk")
        return Selection(st, row_ptr, col_idx, vals)

    def select_threshold(self, d, acts, threshold) -> Selection:
        """Eval mode (sae.py:189-191).  The number of survivors is data dependent, so this path reads one
        integer back to size the output (inference only; the training path never synchronises)."""
        dev = acts.device
        lib = self.lib
        st = torch.empty(16, dtype=torch.int32, device=dev)
        row_ptr = torch.empty(d.B + 1, dtype=torch.int32, device=dev)
        row_gt = torch.empty(d.B, dtype=torch.int32, device=dev)
        row_eq = torch.empty(d.B, dtype=torch.int32, device=dev)
        tie_take = torch.empty(d.B, dtype=torch.int32, device=dev)
        thr = threshold.reshape(1).to(self.dtype).contiguous()
        s = _stream()
        check(lib.mtc_select_init(_p(st), 0, s), "mtc_select_init")
        check(lib.mtc_select_count(C.byref(d), _p(acts), _p(st), _p(thr), _p(row_gt), _p(row_eq), s), "mtc_select_count")
        check(lib.mtc_select_offsets(C.byref(d), _p(row_gt), _p(row_eq), None, _p(st), _p(row_ptr), _p(tie_take), s),
              "mtc_select_offsets")
        nnz = int(row_ptr[-1].item())
        col_idx = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        vals = torch.empty(max(nnz, 1), dtype=self.dtype, device=dev)
        check(lib.mtc_select_fill(C.byref(d), _p(acts), _p(st), _p(thr), _p(row_ptr), _p(tie_take), _p(col_idx),
                                  _p(vals), s), "mtc_select_fill")
        return Selection(st, row_ptr, col_idx, vals)

    def threshold_ema(self, d, sel: Selection, threshold: torch.Tensor, lr: float = 0.01):
        _need(threshold, "threshold", self.dtype)
        check(self.lib.mtc_threshold_ema(C.byref(d), _p(sel.state), _p(threshold), float(lr), _stream()), "mtc_threshold_ema")

    # ------------------------------------------------------------------ a7 + CSC
    def build_csc(self, d, sel: Selection, dead_cnt: Optional[torch.Tensor], part: int = 0):
        """a7 + the feature-major index.  part = 1 / 2: the two halves as separate enqueues (the counters are final after
        part 1: a caller may put the dead-feature list between them)."""
        dev = sel.row_ptr.device
        ws_bytes = self.lib.mtc_csc_workspace_bytes(C.byref(d))
        if part != 2:
            sel._csc_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            cap = sel.col_idx.numel()
            sel.col_ptr = torch.empty(self.D + 1, dtype=torch.int32, device=dev)
            sel.csc_row = torch.empty(cap, dtype=torch.int32, device=dev)
            sel.csc_pos = torch.empty(cap, dtype=torch.int32, device=dev)
        ws = sel._csc_ws
        if dead_cnt is not None:
            _need(dead_cnt, "num_batches_not_active", torch.float32, (self.D,))
        if part == 0:
            check(self.lib.mtc_build_csc(C.byref(d), _p(sel.row_ptr), _p(sel.col_idx), _p(ws), ws_bytes, _p(sel.col_ptr),
                                         _p(sel.csc_row), _p(sel.csc_pos), _p(dead_cnt), _stream()), "mtc_build_csc")
        else:
            check(self.lib.mtc_build_csc_part(C.byref(d), _p(sel.row_ptr), _p(sel.col_idx), _p(ws), ws_bytes, _p(sel.col_ptr),
                                              _p(sel.csc_row), _p(sel.csc_pos), _p(dead_cnt), int(part), _stream()),
                  "mtc_build_csc_part")
        if part != 1:
            sel._csc_ws = None

    # ------------------------------------------------------------------ a6 / a8 / a10
    def decode(self, d, sel: Selection, W_dec, b_dec, x_target, mean, std, *, want_g: bool, want_dvals: bool,
               l1_term: float = 0.0, want_recons: bool = False):
        _need(W_dec, "W_dec", self.dtype, (self.D, self.A_t))
        _need(b_dec, "b_dec", self.dtype, (self.A_t,))
        _need(x_target, "x_target", self.dtype, (d.B, self.A_t))
        dev = x_target.device
        x_hat = torch.empty(d.B, self.A_t, dtype=self.dtype, device=dev)
        g = torch.empty(d.B, self.A_t, dtype=self.dtype, device=dev) if want_g else None
        row_stats = torch.empty(d.G + 4, d.B, dtype=torch.float32, device=dev)
        recons = torch.empty(d.G, d.B, self.A_t, dtype=self.dtype, device=dev) if want_recons else None
        d_vals = torch.empty(sel.col_idx.numel(), dtype=torch.float32, device=dev) if want_dvals else None
        check(self.lib.mtc_decode_fwd(C.byref(d), _p(sel.row_ptr), _p(sel.col_idx), _p(sel.vals), _p(W_dec), _p(b_dec),
                                      _p(x_target), _p(mean), _p(std), _p(x_hat), _p(g), _p(row_stats), _p(recons),
                                      _p(d_vals), float(l1_term), _stream()), "mtc_decode_fwd")
        return x_hat, g, row_stats, recons, d_vals

    def decode_bwd_vals(self, d, sel: Selection, W_dec, g, l1_term: float):
        d_vals = torch.empty(sel.col_idx.numel(), dtype=torch.float32, device=g.device)
        check(self.lib.mtc_decode_bwd_vals(C.byref(d), _p(sel.row_ptr), _p(sel.col_idx), _p(W_dec), _p(g), float(l1_term),
                                           _p(d_vals), _stream()), "mtc_decode_bwd_vals")
        return d_vals

    def reduce_scalars(self, d, row_stats, sel: Selection, n_feat_total: int, l1_coeff: float, aux_penalty: float,
                       aux_rows=None, nnz_abs_override=None, out=None):
        scalars = out if out is not None else torch.empty(16, dtype=torch.float32, device=row_stats.device)
        check(self.lib.mtc_reduce_scalars(C.byref(d), _p(row_stats), _p(aux_rows), _p(sel.state), _p(nnz_abs_override),
                                          int(n_feat_total), float(l1_coeff), float(aux_penalty), _p(scalars), _stream()),
              "mtc_reduce_scalars")
        return scalars

    # ------------------------------------------------------------------ a11
    def backward(self, d, sel: Selection, g, d_vals, x_norm, scale: Optional[torch.Tensor], grads=None,
                 with_colsum: bool = True):
        """Dense gradients (model dtype) for W_dec [D,A_t], W_encT [D,A_s], b_enc [D], b_dec [A_t].
        `grads` given = accumulate into them (aux contributions)."""
        dev = g.device
        if grads is None:
            dW_dec = torch.empty(self.D, self.A_t, dtype=self.dtype, device=dev)
            dW_encT = torch.empty(self.D, self.A_s, dtype=self.dtype, device=dev)
            db_enc = torch.empty(self.D, dtype=self.dtype, device=dev)
            db_dec = torch.empty(self.A_t, dtype=self.dtype, device=dev)
            acc = 0
        else:
            dW_dec, dW_encT, db_enc, db_dec = grads
            acc = 1
        if scale is not None:
            scale = scale.reshape(1).to(torch.float32).contiguous()
        check(self.lib.mtc_backward_sparse(C.byref(d), _p(sel.col_ptr), _p(sel.csc_row), _p(sel.csc_pos), _p(sel.vals),
                                           _p(d_vals), _p(g), _p(x_norm), _p(scale), _p(dW_dec), _p(dW_encT), _p(db_enc),
                                           acc, _stream()), "mtc_backward_sparse")
        if not with_colsum:
            return dW_dec, dW_encT, db_enc, db_dec
        ws_bytes = self.lib.mtc_colsum_workspace_bytes(C.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(self.lib.mtc_colsum(C.byref(d), _p(g), _p(scale), _p(ws), ws_bytes, _p(db_dec), acc, _stream()), "mtc_colsum")
        return dW_dec, dW_encT, db_enc, db_dec

    # ------------------------------------------------------------------ a9
    def aux_forward(self, d, acts, dead_cnt, n_batches_to_dead: float, top_k_aux: int, aux_penalty: float, W_dec,
                    x_target, x_hat, std, g, need_grad: bool):
        """Sparse aux path; returns (aux Selection, g_aux, aux_rows[B], d_aux_vals).  g is updated in place."""
        dev = acts.device
        lib, s = self.lib, _stream()
        B = d.B
        dead_idx = torch.empty(self.D, dtype=torch.int32, device=dev)
        n_dead = torch.empty(1, dtype=torch.int32, device=dev)
        check(lib.mtc_aux_dead_list(C.byref(d), _p(dead_cnt), float(n_batches_to_dead), _p(dead_idx), _p(n_dead), None, None, s),
              "mtc_aux_dead_list")
        row_cnt = torch.empty(B, dtype=torch.int32, device=dev)
        row_tau = torch.empty(B, dtype=torch.int32, device=dev)
        row_quota = torch.empty(B, dtype=torch.int32, device=dev)
        check(lib.mtc_aux_select(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), int(top_k_aux), _p(row_cnt), _p(row_tau),
                                 _p(row_quota), s), "mtc_aux_select")
        st = torch.empty(16, dtype=torch.int32, device=dev)
        zeros = torch.zeros(B, dtype=torch.int32, device=dev)
        row_ptr = torch.empty(B + 1, dtype=torch.int32, device=dev)
        tie_take = torch.empty(B, dtype=torch.int32, device=dev)
        check(lib.mtc_select_init(_p(st), 0, s), "mtc_select_init")
        check(lib.mtc_select_offsets(C.byref(d), _p(row_cnt), _p(zeros), None, _p(st), _p(row_ptr), _p(tie_take), s),
              "mtc_select_offsets")
        cap = max(B * min(int(top_k_aux), self.D), 1)
        col_idx = torch.empty(cap, dtype=torch.int32, device=dev)
        vals = torch.empty(cap, dtype=self.dtype, device=dev)
        check(lib.mtc_aux_fill(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), _p(row_tau), _p(row_quota), _p(row_ptr),
                               _p(col_idx), _p(vals), s), "mtc_aux_fill")
        sel = Selection(st, row_ptr, col_idx, vals)
        g_aux = torch.empty(B, self.A_t, dtype=self.dtype, device=dev)
        aux_rows = torch.empty(B, dtype=torch.float32, device=dev)
        d_aux = torch.empty(cap, dtype=torch.float32, device=dev) if need_grad else None
        check(lib.mtc_aux_fwd(C.byref(d), _p(row_ptr), _p(col_idx), _p(vals), _p(W_dec), _p(x_target), _p(x_hat), _p(std),
                              float(aux_penalty), _p(g), _p(g_aux), _p(aux_rows), _p(d_aux), s), "mtc_aux_fwd")
        return sel, g_aux, aux_rows, d_aux

    def aux_dead_list(self, d, dead_cnt, n_batches_to_dead: float) -> "DeadList":
        """Dead list of this forward (ascending features with cnt >= n_batches_to_dead; the tail of `dead_idx` is
        gatherable), the feature -> slot map, and -- copied to pinned host memory behind the kernel -- its length and the
        upper bound of the NEXT forward's length.  Nothing is read back here: `DeadList.n()` waits for that copy only.
        Callers issue this right after the counters are updated and BEFORE the decode launch."""
        dev = dead_cnt.device
        dl = DeadList()
        dl.dead_idx = torch.empty(self.D, dtype=torch.int32, device=dev)
        dl.counts = torch.empty(2, dtype=torch.int32, device=dev)          # {n_dead, bound of the next forward's n_dead}
        dl.row_map = torch.empty(self.D, dtype=torch.int32, device=dev)
        check(self.lib.mtc_aux_dead_list(C.byref(d), _p(dead_cnt), float(n_batches_to_dead), _p(dl.dead_idx), _p(dl.counts),
                                         _p(dl.row_map), C.c_void_p(dl.counts.data_ptr() + 4), _stream()), "mtc_aux_dead_list")
        ring = self._pinned.setdefault(str(dev), [[torch.empty(2, dtype=torch.int32).pin_memory() for _ in range(4)], 0])
        dl.host = ring[0][ring[1] % 4]
        ring[1] += 1
        dl.host.copy_(dl.counts, non_blocking=True)
        dl.event = torch.cuda.Event()
        dl.event.record(torch.cuda.current_stream())
        return dl

    @staticmethod
    def _aux_bucket(n: int, D: int) -> int:
        """GEMM operand widths come in buckets (n_dead changes almost every step; every new size costs allocator growth);
        the padding columns are zeros."""
        step = 128 if n <= 2048 else 512 if n <= 8192 else 2048
        return max(min((max(n, 1) + step - 1) // step * step, D), 1)

    def gemm_nt(self, M, N, K, A, lda, Bt, ldb, out, ldc, a_rows=None, out_f32=False):
        check(self.lib.mtc_gemm_nt(self.code, int(M), int(N), int(K), _p(A), int(lda), _p(a_rows), _p(Bt), int(ldb), _p(out),
                                   int(ldc), int(bool(out_f32)), _stream()), "mtc_gemm_nt")
        return out

    def transpose_rows(self, src, R, Cc, R_pad=None, src_rows=None):
        """dst [Cc][R_pad] = src[row(r)][c] transposed (model dtype), zero columns beyond R."""
        R_pad = R if R_pad is None else R_pad
        dst = torch.empty(Cc, R_pad, dtype=self.dtype, device=src.device)
        check(self.lib.mtc_transpose_rows(self.code, int(R), int(Cc), int(R_pad), _p(src), int(src.stride(0)), _p(src_rows),
                                          _p(dst), int(R_pad), _stream()), "mtc_transpose_rows")
        return dst

    def aux_forward_dense(self, d, acts, dead_cnt, n_batches_to_dead: float, top_k_aux: int, aux_penalty: float, W_dec,
                          x_target, x_hat, std, g, need_grad: bool, dead=None, reduce_partial=None, global_active=None,
                          side=None):
        """a9 in the reference's own dense form over the compacted dead set (sae.py:332-341) on libmtc's MFMA tiles
        (mtc_gemm_nt; no BLAS library): per-row top-k_aux selection -> acts_aux [B, n_pad] -> x_hat_aux = acts_aux .
        W_dec[dead] -> residual / loss / g update fused elementwise.  `dead` = aux_dead_list() made by the caller, with
        `dead.cap` = a host-side upper bound of its length (sizes the launches; the true length stays on the device).
        `reduce_partial` (sharded dictionary): called on the fp32 partial reconstruction (all-reduce).  Returns
        (ctx dict, aux_rows[B]); g is updated in place.  `side` (fused two-stream step): a second stream for the two
        operand transposes of the backward, each of which depends on ONE forward result and is needed launches later --
        they run beside products that leave CUs idle instead of between them."""
        dev = acts.device
        lib, s = self.lib, _stream()
        B = d.B
        if dead is None:
            dead = self.aux_dead_list(d, dead_cnt, n_batches_to_dead)
        if dead.cap is None:
            dead.cap = dead.n()
        if dead.cap == 0 and reduce_partial is None:
            return None, None            # no dead feature after all (sae.py:331: the branch returns a zero)
        dead_idx, n_dead = dead.dead_idx, dead.counts
        row_cnt = torch.empty(B, dtype=torch.int32, device=dev)
        row_tau = torch.empty(B, dtype=torch.int32, device=dev)
        row_quota = torch.empty(B, dtype=torch.int32, device=dev)
        n_pad = self._aux_bucket(dead.cap, self.D)
        acts_aux = torch.empty(B, n_pad, dtype=self.dtype, device=dev)
        check(lib.mtc_aux_select_dense(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), int(top_k_aux), n_pad, _p(row_cnt),
                                       _p(row_tau), _p(row_quota), _p(acts_aux), s), "mtc_aux_select_dense")
        return self._aux_dense_tail(d, dead, acts_aux, n_pad, aux_penalty, W_dec, x_target, x_hat, std, g, need_grad,
                                    reduce_partial, n_dead if global_active is None else global_active, side=side)

    def _aux_dense_tail(self, d, dead, acts_aux, n_pad, aux_penalty, W_dec, x_target, x_hat, std, g, need_grad,
                        reduce_partial, active, side=None):
        dev, B = acts_aux.device, d.B
        main = torch.cuda.current_stream()
        ctx = dict(dense=True, dead=dead, n_pad=n_pad)
        # W_dec[dead]^T (rows >= n_dead meet zero activations only): gather + transpose in one pass, then the product
        # (it stays on this stream.  Measured elsewhere: beside the dense selection, which fills the chip, it took as long
        # as the selection; right behind the dead list on the side stream it pushed the second half of the CSC build
        # into the selection instead -- no gain either way)
        W_deadT = self.transpose_rows(W_dec, n_pad, self.A_t, src_rows=dead.dead_idx)             # [A_t][n_pad]
        if side is not None and need_grad:
            # acts_aux^T (the backward's operand) beside the forward product, which leaves CUs idle
            e_sel = torch.cuda.Event()
            e_sel.record(main)
            with torch.cuda.stream(side):
                side.wait_event(e_sel)
                ctx["acts_auxT"] = self.transpose_rows(acts_aux, B, n_pad)                         # [n_pad][B]
                ctx["acts_auxT_event"] = torch.cuda.Event()
                ctx["acts_auxT_event"].record(side)
        if reduce_partial is None:
            x_hat_aux = torch.empty(B, self.A_t, dtype=self.dtype, device=dev)
            self.gemm_nt(B, self.A_t, n_pad, acts_aux, n_pad, W_deadT, n_pad, x_hat_aux, self.A_t)   # fp32 accumulate, one rounding
        else:
            part = torch.empty(B, self.A_t, dtype=torch.float32, device=dev)
            self.gemm_nt(B, self.A_t, n_pad, acts_aux, n_pad, W_deadT, n_pad, part, self.A_t, out_f32=True)
            x_hat_aux = reduce_partial(part).to(self.dtype)
        g_aux = torch.empty(B, self.A_t, dtype=self.dtype, device=dev)
        aux_rows = torch.empty(B, dtype=torch.float32, device=dev)
        check(self.lib.mtc_aux_residual(C.byref(d), _p(x_hat_aux), _p(x_target), _p(x_hat), _p(std), float(aux_penalty), _p(g),
                                        _p(g_aux), _p(aux_rows), _p(active), _stream()), "mtc_aux_residual")
        if need_grad:
            ctx.update(acts_aux=acts_aux, g_aux=g_aux, W_dec=W_dec)
            if side is not None:
                # g_aux^T: needed by the E_dec product only, three launches from here
                e_res = torch.cuda.Event()
                e_res.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(e_res)
                    ctx["g_auxT"] = self.transpose_rows(g_aux, B, self.A_t)                        # [A_t][B]
                    ctx["g_auxT_event"] = torch.cuda.Event()
                    ctx["g_auxT_event"].record(side)
        return ctx, aux_rows

    def aux_extra_rows(self, ctx, x_norm, g_loss=None, side=None):
        """Gradient rows of the aux branch over the dead set (model dtype, one rounding per product like the reference's
        matmul backward), every product on mtc_gemm_nt in the feature-major layout the fused gradient kernels read:
        (E_dec [n_pad, A_t], E_enc [n_pad, A_s], e_b [n_pad]); rows of the padding are never referenced."""
        acts_aux, g_aux, dead, n_pad, W_dec = ctx["acts_aux"], ctx["g_aux"], ctx["dead"], ctx["n_pad"], ctx["W_dec"]
        B, dev = acts_aux.shape[0], acts_aux.device
        if g_loss is not None:
            g_aux = (g_aux.float() * g_loss.reshape(()).float()).to(self.dtype)
        ctx["enc_rows_event"] = None
        main = torch.cuda.current_stream()
        x_normT = None
        if side is not None:
            # the transpose of the step's input depends on nothing the branch computes: side stream, at once
            e_in = torch.cuda.Event()
            e_in.record(main)
            with torch.cuda.stream(side):
                side.wait_event(e_in)
                x_normT = self.transpose_rows(x_norm, B, self.A_s)                                 # [A_s][B]
        g_auxT = ctx.get("g_auxT") if g_loss is None else None       # made beside the forward (fused step: no loss scale)
        if g_auxT is None:
            g_auxT = self.transpose_rows(g_aux, B, self.A_t)                                       # [A_t][B]
        # d_acts^T [n_pad][B] = W_dec[dead] . g_aux^T, masked by the selection, and its row sums (b_enc elements)
        d_actsT = torch.empty(n_pad, B, dtype=self.dtype, device=dev)
        self.gemm_nt(n_pad, B, self.A_t, W_dec, self.A_t, g_aux, self.A_t, d_actsT, B, a_rows=dead.dead_idx)
        acts_auxT = ctx.get("acts_auxT")
        if acts_auxT is None:
            acts_auxT = self.transpose_rows(acts_aux, B, n_pad)                                    # [n_pad][B]
        else:
            # one wait: both transposes were issued on the same stream, g_aux^T last
            main.wait_event(ctx["g_auxT_event"] if ctx.get("g_auxT_event") is not None else ctx["acts_auxT_event"])
        e_b = torch.empty(n_pad, dtype=self.dtype, device=dev)
        check(self.lib.mtc_aux_mask_rowsum(self.code, n_pad, B, _p(acts_auxT), B, _p(d_actsT), B, _p(e_b), _stream()),
              "mtc_aux_mask_rowsum")
        E_dec = torch.empty(n_pad, self.A_t, dtype=self.dtype, device=dev)
        if side is None:
            self.gemm_nt(n_pad, self.A_t, B, acts_auxT, B, g_auxT, B, E_dec, self.A_t)
            x_normT = self.transpose_rows(x_norm, B, self.A_s)
            E_enc = torch.empty(n_pad, self.A_s, dtype=self.dtype, device=dev)
            self.gemm_nt(n_pad, self.A_s, B, d_actsT, B, x_normT, B, E_enc, self.A_s)
        else:
            # the two gradient products have the same shape and neither fills the chip (198 tiles of 256 x 256 on 256 CUs at
            # config #2): E_dec is launched first, on the main stream, the encoder rows follow on the side stream through a
            # cross-queue wait (~10 us later), take the CUs E_dec leaves idle and the ones it frees, and finish under the head
            # of the decoder-row launch, which does not need them (the encoder-row launch waits for enc_rows_event).
            # Operands the side stream reads / writes stay referenced in ctx until the next step (allocator safety).
            e_mask = torch.cuda.Event()
            e_mask.record(main)
            self.gemm_nt(n_pad, self.A_t, B, acts_auxT, B, g_auxT, B, E_dec, self.A_t)
            with torch.cuda.stream(side):
                side.wait_event(e_mask)
                E_enc = torch.empty(n_pad, self.A_s, dtype=self.dtype, device=dev)
                self.gemm_nt(n_pad, self.A_s, B, d_actsT, B, x_normT, B, E_enc, self.A_s)
                ctx["enc_rows_event"] = torch.cuda.Event()
                ctx["enc_rows_event"].record(side)
            ctx["enc_rows_keep"] = (d_actsT, x_normT, E_enc, x_norm)
        return E_dec, E_enc, e_b

    def aux_backward_dense(self, ctx, x_norm, g_loss, grads):
        """Gradients of the aux branch accumulated into materialised dense gradients (the autograd path): rows land at
        the dead features' positions (unique indices: deterministic)."""
        dW_dec, dW_encT, db_enc, db_dec = grads
        dead, n_pad = ctx["dead"], ctx["n_pad"]
        E_dec, E_enc, e_b = self.aux_extra_rows(ctx, x_norm, g_loss)
        s = _stream()
        for dst, src, A in ((dW_dec, E_dec, self.A_t), (dW_encT, E_enc, self.A_s), (db_enc, e_b, 1)):
            check(self.lib.mtc_rows_scatter_add(self.code, n_pad, A, _p(dead.dead_idx), _p(src), A, _p(dst), A, self.D,
                                                _p(dead.counts), s), "mtc_rows_scatter_add")
        return grads

    # ------------------------------------------------------------------ a12 / a13
    def unitnorm_project(self, W_dec, grad):
        _need(W_dec, "W_dec", self.dtype, (self.D, self.A_t))
        _need(grad, "W_dec.grad", self.dtype, (self.D, self.A_t))
        d = self.desc(1)
        check(self.lib.mtc_unitnorm_project(C.byref(d), _p(W_dec), _p(grad), _stream()), "mtc_unitnorm_project")

    def unitnorm_adam(self, W_dec, grad, m, v, hyper: AdamHyper, write_grad: bool = True):
        for t, n in ((W_dec, "W_dec"), (grad, "W_dec.grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            _need(t, n, self.dtype, (self.D, self.A_t))
        d = self.desc(1)
        check(self.lib.mtc_unitnorm_adam_rows(C.byref(d), _p(W_dec), _p(grad), _p(m), _p(v), C.byref(hyper),
                                              int(write_grad), _stream()), "mtc_unitnorm_adam_rows")

    def adam_flat(self, p, grad, m, v, hyper: AdamHyper):
        for t, n in ((p, "param"), (grad, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            _need(t, n, self.dtype)
        check(self.lib.mtc_adam_flat(self.code, p.numel(), _p(p), _p(grad), _p(m), _p(v), C.byref(hyper), _stream()),
              "mtc_adam_flat")

    def fused_backward_adam(self, d, sel: Selection, g, d_vals, x_norm, scale, W_dec, st_dec, W_encT, st_enc, b_enc,
                            st_benc, h_dec: AdamHyper, h_enc: AdamHyper, h_benc: AdamHyper, l1_term: float = 0.0, extra=None,
                            enc_wait=None):
        """a11 + a12 + a13 in two launches; st_* = (exp_avg, exp_avg_sq) in the storage layout of the parameter.
        d_vals=None: the decoder launch also produces d a_bj (the forward skipped its second sweep).
        extra = (row_map int32[D], E_dec, E_enc, e_b): dense extra gradient rows (aux branch) added per feature;
        enc_wait: event the current stream waits for between the two launches (E_enc made on another stream)."""
        dv_out = None
        if d_vals is None:
            d_vals = dv_out = torch.empty(max(sel.col_idx.numel(), 1), dtype=torch.float32, device=g.device)
        if scale is not None:
            scale = scale.reshape(1).to(torch.float32).contiguous()
        for t, n, shp in ((W_dec, "W_dec", (self.D, self.A_t)), (st_dec[0], "exp_avg(W_dec)", (self.D, self.A_t)),
                          (st_dec[1], "exp_avg_sq(W_dec)", (self.D, self.A_t)), (W_encT, "W_encT", (self.D, self.A_s)),
                          (st_enc[0], "exp_avg(W_enc)", (self.D, self.A_s)), (st_enc[1], "exp_avg_sq(W_enc)", (self.D, self.A_s)),
                          (b_enc, "b_enc", (self.D,)), (st_benc[0], "exp_avg(b_enc)", (self.D,)),
                          (st_benc[1], "exp_avg_sq(b_enc)", (self.D,))):
            _need(t, n, self.dtype, shp)
        s = _stream()
        if h_enc.step != h_benc.step or h_enc.lr != h_benc.lr:
            raise _lib.MtcError("W_enc and b_enc must share one Adam step count / lr for the fused update")
        row_map = E_dec = E_enc = e_b = None
        if extra is not None:
            row_map, E_dec, E_enc, e_b = extra
            _need(row_map, "extra row map", torch.int32, (self.D,))
            _need(E_dec, "extra decoder rows", self.dtype); _need(E_enc, "extra encoder rows", self.dtype)
            _need(e_b, "extra bias elements", self.dtype)
            if dv_out is None or E_dec.shape[1] != self.A_t or E_enc.shape[1] != self.A_s or e_b.numel() != E_enc.shape[0]:
                raise _lib.MtcError("extra gradient rows: pass d_vals=None and [n, A_t] / [n, A_s] / [n] operands")
        # d a_bj made by the decoder launch itself lives in CSC order (the order both launches walk; nothing else reads it)
        csc_order = int(dv_out is not None)
        check(self.lib.mtc_fused_backward_adam_dec_extra(
            C.byref(d), _p(sel.col_ptr), _p(sel.csc_row), _p(sel.csc_pos), _p(sel.vals), _p(g), _p(scale), _p(W_dec),
            _p(st_dec[0]), _p(st_dec[1]), C.byref(h_dec), _p(dv_out), float(l1_term), _p(row_map), _p(E_dec), csc_order, s),
            "mtc_fused_backward_adam_dec_extra")
        if enc_wait is not None:
            torch.cuda.current_stream().wait_event(enc_wait)
        check(self.lib.mtc_fused_backward_adam_enc_extra(
            C.byref(d), _p(sel.col_ptr), _p(sel.csc_row), _p(sel.csc_pos), _p(d_vals), _p(x_norm), _p(scale), _p(W_encT),
            _p(st_enc[0]), _p(st_enc[1]), _p(b_enc), _p(st_benc[0]), _p(st_benc[1]), C.byref(h_enc), _p(row_map),
            _p(E_enc), _p(e_b), csc_order, s), "mtc_fused_backward_adam_enc_extra")

    def colsum(self, d, g, scale=None):
        db_dec = torch.empty(self.A_t, dtype=self.dtype, device=g.device)
        ws_bytes = self.lib.mtc_colsum_workspace_bytes(C.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device)
        if scale is not None:
            scale = scale.reshape(1).to(torch.float32).contiguous()
        check(self.lib.mtc_colsum(C.byref(d), _p(g), _p(scale), _p(ws), ws_bytes, _p(db_dec), 0, _stream()), "mtc_colsum")
        return db_dec

    # ------------------------------------------------------------------ a15
    def densify(self, d, sel: Selection, max_col: int):
        dense = torch.empty(d.B, self.D, dtype=self.dtype, device=sel.row_ptr.device)
        check(self.lib.mtc_densify(C.byref(d), _p(sel.row_ptr), _p(sel.col_idx), _p(sel.vals), int(max_col), _p(dense),
                                   _stream()), "mtc_densify")
        return dense


def adam_hyper(lr, beta1, beta2, eps, step) -> AdamHyper:
    h = AdamHyper()
    h.lr, h.beta1, h.beta2, h.eps, h.step, h.pad = float(lr), float(beta1), float(beta2), float(eps), int(step), 0
    return h
