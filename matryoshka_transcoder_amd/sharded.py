"""Dictionary-sharded transcoder: one process per GPU, features split over the ranks
(BASELINE.json configs 3-5).  The reference has no multi-device code; this is the
MI355X layout of the same math, checked against the single-GPU path.

Layout.  Every prefix GROUP is cut into `world` equal contiguous slices and rank r owns slice r of
each group, so each rank holds 1/world of every nesting level (a plain contiguous split would put
all coarse prefixes on rank 0) and its local cumulative prefixes are the global ones / world.
Rank r owns W_enc[:, S_r], b_enc[S_r], W_dec[S_r, :], their Adam moments and the dead counters of
S_r; b_dec, the threshold and the batch are replicated.

Step (RCCL through torch.distributed; every exchange is a real data dependency of the algorithm):
  1. local encoder GEMM on the slice;
  2. global TopK threshold: per radix level, local 4096-bin histogram -> all-reduce(sum) (16 KiB)
     -> identical scan on every rank (bf16: one level + an 8-bin all-reduce); ties at tau are granted in (row, rank, feature) order from an
     all-gather of per-row tie counts ([world, B] ints);
  3. local sparse decode into bias-free per-prefix partial sums -> reduce-scatter over batch rows
     (last prefix fp32, earlier prefixes -- FVU only -- in the model dtype) -> each rank finishes
     x_hat / per-prefix SSE / g for B/world rows -> ONE all-gather of a packed byte buffer per rank
     (x_hat and g of its rows, the per-row statistics, its nnz and sum|a|);
  4. backward and Adam are local to the shard (no gradient all-reduce); db_dec is identical
     everywhere because g is.
  5. while dead features exist: the auxiliary loss over shards (`_run_aux`): all-gather of per-row
     candidates, all-gather of tie counts, fp32 all-reduce of the partial aux reconstruction.
`state_dict()` gathers to the reference's layout; `load_state_dict()` scatters from it.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

from . import _lib
from .engine import HotPath, Selection, _p, _stream, check
from .training import FusedAdam
from .transcoder import MatryoshkaTranscoder, _feature_major

import ctypes as C


class ShardLayout:
    """Pure index arithmetic of the feature partition (host side, no device work)."""

    def __init__(self, prefix_sizes: List[int], world: int, rank: int):
        groups = [b - a for a, b in zip([0] + list(prefix_sizes[:-1]), prefix_sizes)]
        for g in groups:
            if g % (8 * world):
                raise _lib.MtcError(f"every prefix group must be a multiple of 8*world={8 * world}; groups={groups}")
        self.world, self.rank = world, rank
        self.global_prefix = list(prefix_sizes)
        self.groups = groups
        self.local_groups = [g // world for g in groups]
        self.local_prefix, run = [], 0
        for g in self.local_groups:
            run += g
            self.local_prefix.append(run)
        self.D = prefix_sizes[-1]
        self.D_local = run

    def global_index(self, rank: int) -> torch.Tensor:
        """Global feature id of every local feature of `rank`, in local order."""
        parts, start = [], 0
        for g, lg in zip(self.groups, self.local_groups):
            parts.append(torch.arange(start + rank * lg, start + (rank + 1) * lg))
            start += g
        return torch.cat(parts)

    def owner_permutation(self) -> torch.Tensor:
        """perm such that cat([shard_0, ..., shard_{w-1}])[perm] is in global feature order."""
        order = torch.cat([self.global_index(r) for r in range(self.world)])
        return torch.argsort(order)


def tie_take_global_order(all_eq: torch.Tensor, quota: int, rank: int) -> torch.Tensor:
    """Host restatement of the tie grant of mtc_select_sharded_finish (tests; the step itself runs the kernel).
    all_eq [world, B, G] = per (row, prefix group) counts of activations equal to tau on every rank.  Ties are granted
    in ascending GLOBAL flat index -- row, then feature, and the global feature order inside a row is (group, rank,
    local feature) because rank r owns slice r of every group -- until `quota` are taken.  Returns, for `rank`, how
    many of its ties are taken in every row [B] (a prefix of the row's local ties: the grant is monotone in the order)."""
    W, B, G = all_eq.shape
    flat = all_eq.permute(1, 2, 0).reshape(-1).to(torch.int64)            # (row, group, rank) order
    before = torch.cumsum(flat, 0) - flat
    take = torch.minimum((quota - before).clamp_min(0), flat)
    return take.reshape(B, G, W)[:, :, rank].sum(1).to(torch.int32)


class _CommBase:
    """What the sharded step needs from a transport.  Collectives run on torch's CURRENT stream; `stream` is a second
    stream of this object on which the step issues the exchanges that run beside kernels of the main stream.  With
    `timing` set to a dict, every collective is bracketed by HIP events on the stream it runs on
    (name -> [(start, end, payload bytes)]; bench.py reads them after its fence)."""
    timing = None
    _cstream = None

    def comm_stream(self, device) -> "torch.cuda.Stream":
        if self._cstream is None:
            self._cstream = torch.cuda.Stream(device=device)
        return self._cstream

    def _timed(self, tag, nbytes):
        if self.timing is None or tag is None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.timing.setdefault(tag, []).append((e0, e1, int(nbytes)))
        return e1


class _Comm(_CommBase):
    """torch.distributed on device tensors.  backend "nccl" is RCCL on ROCm; with "gloo" (tests)
    device tensors are staged through the host."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) == "gloo"

    def all_reduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM, tag=None) -> torch.Tensor:
        e = self._timed(tag, t.numel() * t.element_size())
        if self.staged and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=self.group)
        if e is not None:
            e.record()
        return t

    def reduce_scatter_into(self, out: torch.Tensor, t: torch.Tensor, tag=None) -> torch.Tensor:
        """out (contiguous, numel = t.numel() / world) = this rank's block of the sum over ranks of t (contiguous)."""
        e = self._timed(tag, t.numel() * t.element_size())
        if self.staged:
            h = t.float().cpu() if t.is_cuda else t.float()
            dist.all_reduce(h, group=self.group)
            n = out.numel()
            out.copy_(h.reshape(-1)[self.rank * n:(self.rank + 1) * n].reshape(out.shape).to(out.dtype))
        else:
            dist.reduce_scatter_tensor(out, t, group=self.group)
        if e is not None:
            e.record()
        return out

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor, tag=None) -> torch.Tensor:
        """out (contiguous, numel = world * t.numel()) = the ranks' t (contiguous) one after the other."""
        e = self._timed(tag, out.numel() * out.element_size())
        if self.staged and t.is_cuda:
            h = t.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(parts, h, group=self.group)
            out.copy_(torch.stack(parts).reshape(out.shape))
        elif self.staged:
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t, group=self.group)
            out.copy_(torch.stack(parts).reshape(out.shape))
        else:
            dist.all_gather_into_tensor(out, t, group=self.group)
        if e is not None:
            e.record()
        return out

    def reduce_scatter(self, t: torch.Tensor, tag=None) -> torch.Tensor:
        """Sum over ranks of t [world * n, ...]; returns this rank's block [n, ...]."""
        n = t.shape[0] // self.world
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return self.reduce_scatter_into(out, t.contiguous(), tag)

    def all_gather(self, t: torch.Tensor, tag=None) -> torch.Tensor:
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        return self.all_gather_into(out, t, tag)


class _LibComm(_CommBase):
    """The same exchanges through libmtc's own RCCL wrappers (mtc_comm_*, include/mtc.h): ctypes calls on torch's current
    stream, no torch.distributed on the data path (it only hands the 128-byte unique id round once).  Opt-in
    (cfg["comm"] = "libmtc"): a collective then costs the host one ctypes call, which matters where the sharded step is
    bound by the host's issue rate.  Same interface as _Comm."""
    _DT = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2, torch.uint8: 3, torch.int64: 4}
    _OP = {dist.ReduceOp.SUM: 0, dist.ReduceOp.MAX: 1, dist.ReduceOp.MIN: 2}
    staged = False

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.lib = _lib.load()
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            check(self.lib.mtc_comm_unique_id(uid), "mtc_comm_unique_id")
        box = [uid.raw]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = C.c_void_p()
        check(self.lib.mtc_comm_init(C.byref(self._h), self.world, self.rank, C.create_string_buffer(box[0], 128)),
              "mtc_comm_init")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.lib.mtc_comm_destroy(h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def _code(self, t):
        try:
            return self._DT[t.dtype]
        except KeyError:
            raise _lib.MtcError(f"mtc_comm: unsupported dtype {t.dtype}") from None

    def all_reduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM, tag=None) -> torch.Tensor:
        if not t.is_contiguous():
            raise _lib.MtcError("mtc_comm: tensors must be contiguous")
        e = self._timed(tag, t.numel() * t.element_size())
        check(self.lib.mtc_comm_all_reduce(self._h, _p(t), _p(t), t.numel(), self._code(t), self._OP[op], _stream()),
              "mtc_comm_all_reduce")
        if e is not None:
            e.record()
        return t

    def reduce_scatter_into(self, out: torch.Tensor, t: torch.Tensor, tag=None) -> torch.Tensor:
        e = self._timed(tag, t.numel() * t.element_size())
        check(self.lib.mtc_comm_reduce_scatter(self._h, _p(t), _p(out), out.numel(), self._code(t), 0, _stream()),
              "mtc_comm_reduce_scatter")
        if e is not None:
            e.record()
        return out

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor, tag=None) -> torch.Tensor:
        e = self._timed(tag, out.numel() * out.element_size())
        check(self.lib.mtc_comm_all_gather(self._h, _p(t), _p(out), t.numel(), self._code(t), _stream()), "mtc_comm_all_gather")
        if e is not None:
            e.record()
        return out

    def reduce_scatter(self, t: torch.Tensor, tag=None) -> torch.Tensor:
        n = t.shape[0] // self.world
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return self.reduce_scatter_into(out, t.contiguous(), tag)

    def all_gather(self, t: torch.Tensor, tag=None) -> torch.Tensor:
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        return self.all_gather_into(out, t, tag)


class ShardedMatryoshkaTranscoder(MatryoshkaTranscoder):
    _overlap_ok = False        # the step has its own second stream (the transport's), see _decode_and_losses

    def __init__(self, cfg: dict, group=None):
        if not dist.is_initialized():
            raise _lib.MtcError("ShardedMatryoshkaTranscoder needs an initialised torch.distributed process group")
        self.comm_group = group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        layout = ShardLayout(list(cfg["prefix_sizes"]), world, rank)
        # full reference init on the host (same RNG stream on every rank), then keep only the shard
        full_cfg = dict(cfg, device="cpu")
        super().__init__(full_cfg)
        self.config = cfg
        self.layout = layout
        idx = layout.global_index(rank)
        with torch.no_grad():
            b_enc, W_enc, W_dec = self.b_enc.data[idx], self.W_enc.data[:, idx], self.W_dec.data[idx]
        self.b_enc = torch.nn.Parameter(b_enc.contiguous())
        self.W_enc = torch.nn.Parameter(_feature_major(W_enc))
        self.W_dec = torch.nn.Parameter(W_dec.contiguous())
        self._cnt = torch.zeros(layout.D_local)
        self.prefix_sizes = list(cfg["prefix_sizes"])            # public attributes stay global
        self.local_prefix_sizes = layout.local_prefix
        self._hp = None
        self.to(cfg["device"])
        self._cnt = self._cnt.to(cfg["device"])
        self.comm = _LibComm(group) if cfg.get("comm") == "libmtc" else _Comm(group)
        self._sel_bufs = {}              # B -> exchange buffers / workspace of the sharded select (reused every step)
        self._keep = None                # tensors the transport's stream still reads (freed when the next step starts)

    def hot_path(self) -> HotPath:
        if self._hp is None or self._hp.dtype != self.dtype:
            self._hp = HotPath(self.source_act_size, self.target_act_size, self.local_prefix_sizes, self.dtype,
                               bool(self.config.get("input_unit_norm", False)))
        return self._hp

    def exchange_blocks(self, B: int) -> int:
        """Row blocks the critical exchange is pipelined over (reduce-scatter of block i beside the decode of block i+1).
        cfg["exchange_blocks"]; default: blocks of >= 4096 rows, at most 4, when there is someone to exchange with -- a
        decode launch over fewer rows leaves CUs idle (1024-row blocks at config #2 measured +0.24 ms per step at world 1),
        so only the large batches (Gemma-2-9B / 27B shapes: 8192 / 16384 rows) are cut by default."""
        nb = int(self.config.get("exchange_blocks", min(4, max(B // 4096, 1)) if self.comm.world > 1 else 1))
        while nb > 1 and B % (nb * self.comm.world):
            nb -= 1
        return max(nb, 1)

    # ------------------------------------------------------------------ step 2: global select
    def _select_buffers(self, hp, d):
        key = (d.B, str(self.W_dec.device), self.dtype)
        bufs = self._sel_bufs.get(key)
        if bufs is None:
            dev, lib = self.W_dec.device, hp.lib
            i32 = dict(dtype=torch.int32, device=dev)
            ws_bytes = lib.mtc_select_sharded_workspace_bytes(C.byref(d))
            n_pack = d.B * d.G + 1
            bufs = dict(ws=torch.empty(ws_bytes, dtype=torch.uint8, device=dev), ws_bytes=ws_bytes,
                        xbuf=torch.zeros(lib.mtc_select_sharded_xbuf_words(), **i32),
                        hist_full=torch.zeros(lib.mtc_select_sharded_full_bins(), **i32),
                        pack=torch.zeros(n_pack, **i32), all_pack=torch.zeros(self.comm.world * n_pack, **i32),
                        carry=torch.zeros(8, **i32))
            self._sel_bufs[key] = bufs
        return bufs

    def _select_carry(self, hp, d, device):
        return self._select_buffers(hp, d)["carry"] if self.dtype == torch.bfloat16 else None

    def _select(self, hp, d, acts, training: bool) -> Selection:
        """sae.py:176-188 over the sharded dictionary: mtc_select_sharded_* (include/mtc.h) with three fixed-shape
        collectives; the carried bound of the single-GPU select is used here too (bf16), ties are granted in ascending
        global flat index, so the selection equals the single-GPU one whatever the world size."""
        if not training:
            return hp.select_threshold(d, acts, self.threshold)
        lib, s, dev, comm = hp.lib, _stream(), acts.device, self.comm
        B = d.B
        k_total = int(self.config["top_k"]) * B
        bf = self._select_buffers(hp, d)
        st = torch.empty(16, dtype=torch.int32, device=dev)
        bf16 = self.dtype == torch.bfloat16
        ws, xbuf, hist_full, pack, all_pack = bf["ws"], bf["xbuf"], bf["hist_full"], bf["pack"], bf["all_pack"]
        carry = bf["carry"] if bf16 else None
        check(lib.mtc_select_sharded_begin(C.byref(d), _p(acts), k_total, _p(ws), bf["ws_bytes"], _p(carry), _p(st), _p(xbuf),
                                           _p(hist_full) if bf16 else None, _p(pack), _p(getattr(acts, "_mtc_marks", None)), s),
              "mtc_select_sharded_begin")
        if bf16:
            comm.all_reduce(xbuf[:4104], tag="select window (all-reduce)")
            check(lib.mtc_select_sharded_window(C.byref(d), _p(acts), k_total, _p(carry), _p(st), _p(xbuf), _p(hist_full), s),
                  "mtc_select_sharded_window")
            comm.all_reduce(hist_full, tag="select fallback (all-reduce)")
        else:
            for level in range(lib.mtc_select_levels(C.byref(d))):
                hist = xbuf[level * 4096:(level + 1) * 4096]
                check(lib.mtc_select_hist(C.byref(d), _p(acts), level, _p(st), _p(hist), s), "mtc_select_hist")
                comm.all_reduce(hist, tag="select level (all-reduce)")
                check(lib.mtc_select_scan(C.byref(d), level, _p(hist), _p(st), s), "mtc_select_scan")
        check(lib.mtc_select_sharded_counts(C.byref(d), _p(acts), k_total, _p(ws), _p(carry), _p(st),
                                            _p(hist_full) if bf16 else None, _p(pack), s), "mtc_select_sharded_counts")
        comm.all_gather_into(all_pack, pack, tag="select ties (all-gather)")
        row_ptr = torch.empty(B + 1, dtype=torch.int32, device=dev)
        cap = max(k_total, 1)                       # a single rank can hold at most all k*B entries
        col_idx = torch.empty(cap, dtype=torch.int32, device=dev)
        vals = torch.empty(cap, dtype=self.dtype, device=dev)
        # threshold EMA (a5) from the global smallest selected value rides in the finish launch
        check(lib.mtc_select_sharded_finish(C.byref(d), _p(acts), _p(ws), _p(carry), _p(st), _p(all_pack), comm.world, comm.rank,
                                            _p(row_ptr), _p(col_idx), _p(vals), _p(self.threshold), 0.01, s),
              "mtc_select_sharded_finish")
        return Selection(st, row_ptr, col_idx, vals)

    # ------------------------------------------------------------------ step 3: decode + losses
    def _aux_prepare(self, hp, d, aux_active: bool, side=None, on_side: bool = False) -> bool:
        return aux_active            # the sharded aux branch runs in its row-gather form (no dead-list read-back)

    def _aux_may_be_active(self) -> bool:
        """Same host-side bound as the single-GPU model, on the GLOBAL maximum of the dead counters so that
        every rank takes the same branch (and issues the same collectives) on the same step."""
        if self._dead_check_in > 0:
            return False
        n_dead_at = float(self.config["n_batches_to_dead"])
        mx_t = self._cnt.max().reshape(1).clone()
        self.comm.all_reduce(mx_t, op=dist.ReduceOp.MAX)
        mx = float(mx_t.item())
        if mx >= n_dead_at:
            return True
        import math
        self._dead_check_in = max(int(math.ceil(n_dead_at - mx)) - 1, 0)
        return False

    def _run_aux(self, hp, d, acts, sel, W_dec, x_target, x_hat, g, need_grad, l1_term):
        """Auxiliary dead-feature loss (sae.py:318-347) over a feature-sharded dictionary, dense form (default): every
        rank holds its dead features' columns of acts_aux and rows of W_dec, so the aux reconstruction is a local GEMM
        + one fp32 all-reduce, and the three backward products are local GEMMs whose rows ride into the fused gradient
        kernels.  The per-row top-k_aux over ALL dead features: local top-k_aux values per row (mtc_aux_select +
        mtc_aux_row_candidates) -> all-gather -> mtc_aux_select on the gathered [B, world*k] matrix = global threshold
        and tie quota per row -> local counts at the threshold per prefix group (mtc_aux_count_at_groups), all-gathered so
        that ties are granted in ascending global feature order (group, rank, feature) -> mtc_aux_dense_fill.  Three small collectives + the all-reduce."""
        cfg, comm = self.config, self.comm
        if cfg.get("aux_mode", "gemm") != "gemm":
            return self._run_aux_sparse(hp, d, acts, sel, W_dec, x_target, x_hat, g, need_grad, l1_term)
        lib, s, dev = hp.lib, _stream(), acts.device
        B, A_t, Dl, W = d.B, hp.A_t, hp.D, comm.world
        k_aux = int(cfg["top_k_aux"])
        k_pad = (k_aux + 7) // 8 * 8
        dead = hp.aux_dead_list(d, self._cnt, float(cfg["n_batches_to_dead"]))
        dead_idx, n_dead = dead.dead_idx, dead.counts
        i32 = dict(dtype=torch.int32, device=dev)
        row_cnt, row_tau, row_quota = (torch.empty(B, **i32) for _ in range(3))
        check(lib.mtc_aux_select(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), k_aux, _p(row_cnt), _p(row_tau), _p(row_quota), s),
              "mtc_aux_select")
        cand = torch.empty(B, k_pad, dtype=self.dtype, device=dev)
        check(lib.mtc_aux_row_candidates(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), _p(row_tau), _p(row_quota), k_pad,
                                         _p(cand), s), "mtc_aux_row_candidates")
        allc = comm.all_gather(cand).permute(1, 0, 2).reshape(B, W * k_pad).contiguous()
        d2 = _lib.make_desc(B, hp.A_s, hp.A_t, W * k_pad, [W * k_pad], hp.code, False)
        idx2 = torch.arange(W * k_pad, **i32)
        n2 = torch.full((1,), W * k_pad, **i32)
        g_cnt, g_tau, g_quota = (torch.empty(B, **i32) for _ in range(3))
        check(lib.mtc_aux_select(C.byref(d2), _p(allc), _p(idx2), _p(n2), k_aux, _p(g_cnt), _p(g_tau), _p(g_quota), s),
              "mtc_aux_select")
        G = d.G
        gt_loc, eq_loc = torch.empty(B, **i32), torch.empty(B, G, **i32)
        check(lib.mtc_aux_count_at_groups(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), _p(g_tau), _p(gt_loc), _p(eq_loc), s),
              "mtc_aux_count_at_groups")
        all_eq = comm.all_gather(eq_loc)                                         # [world, B, G]
        # a row's ties in ascending GLOBAL feature order = (group, rank, local feature): what precedes this rank's ties
        # of group g = all ties of earlier groups + those of earlier ranks in g (the single-GPU order, whatever the world)
        grp_tot = all_eq.sum(0, dtype=torch.int32)
        before = torch.cumsum(grp_tot, 1, dtype=torch.int32) - grp_tot
        if comm.rank > 0:
            before = before + all_eq[:comm.rank].sum(0, dtype=torch.int32)
        take = torch.minimum((g_quota.unsqueeze(1) - before).clamp_min(0), eq_loc).sum(1, dtype=torch.int32).contiguous()
        # local part of acts_aux over the local dead set (bucketed width, zero padding), local product on libmtc's own
        # MFMA tiles with an fp32 result, fp32 all-reduce.  The local count sizes local launches only (the collectives
        # have fixed shapes), so it is the one read-back of this branch -- by now long on the host (copy issued above)
        dead.cap = dead.n()
        n_pad = hp._aux_bucket(dead.cap, Dl)
        acts_aux = torch.empty(B, n_pad, dtype=self.dtype, device=dev)
        check(lib.mtc_aux_dense_fill(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), _p(g_tau), _p(take), n_pad, _p(acts_aux), s),
              "mtc_aux_dense_fill")
        std = self.x_std.reshape(-1).contiguous() if hp.unit_norm else None
        ctx, aux_rows = hp._aux_dense_tail(d, dead, acts_aux, n_pad, float(cfg["aux_penalty"]), W_dec, x_target, x_hat, std, g,
                                           need_grad, comm.all_reduce, None)
        ctx["local_counts_decide"] = False          # "dead features exist" is global here (_aux_may_be_active), not this shard's count
        return ctx, aux_rows

    def _run_aux_sparse(self, hp, d, acts, sel, W_dec, x_target, x_hat, g, need_grad, l1_term):
        """Row-gather form of the same branch (`aux_mode="sparse"`; kept as the cross-check of the dense form).  The per-row
        top-k_aux over ALL dead features is found from an all-gather of every rank's local per-row
        candidates (k_aux values per row), ties at a row's threshold are granted in (rank, feature) order
        from an all-gather of the tie counts, the partial reconstructions are summed with an all-reduce
        (fp32), and the residual / loss / gradient part runs replicated.  The selection and the sums use
        the libmtc kernels; the candidate merge is torch.topk on [B, world * k_aux] -- this path runs only
        while dead features exist."""
        cfg, comm = self.config, self.comm
        lib, s, dev = hp.lib, _stream(), acts.device
        B, A_t, Dl = d.B, hp.A_t, hp.D
        k_aux = int(cfg["top_k_aux"])
        n_to_dead = float(cfg["n_batches_to_dead"])
        penalty = float(cfg["aux_penalty"])
        # local candidates: the k_aux largest activations of every row among the local dead columns
        dead_mask = self._cnt >= n_to_dead
        masked = torch.where(dead_mask.unsqueeze(0), acts, torch.zeros((), dtype=acts.dtype, device=dev))
        kk = min(k_aux, Dl)
        cand = torch.zeros(B, k_aux, dtype=acts.dtype, device=dev)
        cand[:, :kk] = torch.topk(masked, kk, dim=1).values
        allc = comm.all_gather(cand).permute(1, 0, 2).reshape(B, -1)            # [B, world * k_aux]
        top = torch.topk(allc.float(), k_aux, dim=1).values
        tau = top[:, -1].clamp_min(0.0)                                          # 0: fewer than k_aux positives -> take them all
        n_gt_glob = (allc.float() > tau.unsqueeze(1)).sum(1)
        quota = torch.where(tau > 0, k_aux - n_gt_glob, torch.zeros_like(n_gt_glob))
        mf = masked.float()
        gt_loc = (mf > tau.unsqueeze(1)).sum(1)
        eq_loc = ((mf == tau.unsqueeze(1)) & (tau.unsqueeze(1) > 0)).sum(1)
        all_eq = comm.all_gather(eq_loc.to(torch.int32))                         # [world, B]
        before = all_eq[:comm.rank].sum(0) if comm.rank > 0 else torch.zeros_like(eq_loc)
        take = torch.minimum((quota - before).clamp_min(0), eq_loc).to(torch.int32)
        row_cnt = (gt_loc + take).to(torch.int32).contiguous()
        row_tau = tau.contiguous().view(torch.int32)
        # CSR of the local aux entries (ascending local feature inside a row), same kernels as the single-GPU path
        dead_idx = torch.empty(Dl, dtype=torch.int32, device=dev)
        n_dead = torch.empty(1, dtype=torch.int32, device=dev)
        check(lib.mtc_aux_dead_list(C.byref(d), _p(self._cnt), n_to_dead, _p(dead_idx), _p(n_dead), None, None, s), "mtc_aux_dead_list")
        st = torch.empty(16, dtype=torch.int32, device=dev)
        zeros = torch.zeros(B, dtype=torch.int32, device=dev)
        row_ptr = torch.empty(B + 1, dtype=torch.int32, device=dev)
        tie_take = torch.empty(B, dtype=torch.int32, device=dev)
        check(lib.mtc_select_init(_p(st), 0, s), "mtc_select_init")
        check(lib.mtc_select_offsets(C.byref(d), _p(row_cnt), _p(zeros), None, _p(st), _p(row_ptr), _p(tie_take), s),
              "mtc_select_offsets")
        cap = max(B * min(k_aux, Dl), 1)
        col_idx = torch.empty(cap, dtype=torch.int32, device=dev)
        vals = torch.empty(cap, dtype=self.dtype, device=dev)
        check(lib.mtc_aux_fill(C.byref(d), _p(acts), _p(dead_idx), _p(n_dead), _p(row_tau), _p(take.contiguous()), _p(row_ptr),
                               _p(col_idx), _p(vals), s), "mtc_aux_fill")
        asel = Selection(st, row_ptr, col_idx, vals)
        # partial aux reconstruction of the local entries (bias-free, fp32) -> all-reduce
        d1 = _lib.make_desc(B, hp.A_s, hp.A_t, Dl, [Dl], hp.code, hp.unit_norm)
        part = torch.empty(B, A_t, dtype=torch.float32, device=dev)
        dummy = torch.empty(8, dtype=self.dtype, device=dev)
        check(lib.mtc_decode_partial(C.byref(d1), _p(row_ptr), _p(col_idx), _p(vals), _p(W_dec), B, _p(part), _p(dummy), None, s),
              "mtc_decode_partial")
        comm.all_reduce(part)
        # replicated: e = round(x_aux) - (x_t - x_hat); aux = penalty * mean(e^2); the gradients in autograd's own rounding
        # sequence on model-dtype tensors (see g_post_of in csrc/rowops.hpp): g_aux = round(fl32(penalty / N) * 2 e),
        # g_post = round(round(inv_n * 2 round(x_hat - x_t)) + g_aux), g = round(g_post * std_row)
        e = part.to(self.dtype).float() - (x_target.float() - x_hat.float())
        aux_rows = (e * e).sum(1).contiguous()
        n_elem = B * A_t
        g_aux = (torch.tensor(penalty / n_elem, dtype=torch.float32, device=dev) * (2.0 * e)).to(self.dtype).contiguous()
        inv_n = torch.tensor(1.0 / n_elem, dtype=torch.float32, device=dev).to(self.dtype)
        g_post = inv_n * (2.0 * (x_hat - x_target)) + g_aux                    # every op rounds to the model dtype
        g.copy_(g_post * self.x_std.reshape(-1, 1) if hp.unit_norm else g_post)
        d_aux = None
        if need_grad:
            d_aux = hp.decode_bwd_vals(d, asel, W_dec, g_aux, 0.0)
            hp.build_csc(d, asel, None)
        return dict(sel=asel, g_aux=g_aux, d_vals=d_aux), aux_rows

    def _decode_and_losses(self, hp, d, sel, acts, W_dec, b_dec, x_target, mean, std, need_grad, aux_active):
        """a6-a10 over the sharded dictionary.  The batch is walked in `nb` row blocks; per block: local bias-free partial
        sums (mtc_decode_partial) -> reduce-scatter of the last prefix (fp32) over the block's rows -> this rank finishes
        its rows (x_hat, g, SSE) -> all-gather of g straight into the block's rows of g.  The two collectives of block i
        run on the transport's stream beside the decode of block i+1 on the main stream; the earlier prefixes (FVU metrics
        only, sae.py:248-252: model dtype), x_hat and the per-row statistics follow in ONE reduce-scatter + ONE all-gather
        on that stream, beside the backward kernels when the fused step defers them."""
        cfg, comm = self.config, self.comm
        lib, dev = hp.lib, x_target.device
        B, A_t, G, W, rank = d.B, hp.A_t, d.G, comm.world, comm.rank
        if B % W:
            raise _lib.MtcError(f"the batch ({B} rows) must be a multiple of the world size ({W})")
        nb = self.exchange_blocks(B)
        Bb = B // nb                       # rows of a block
        rpb = Bb // W                      # rows of a block this rank finishes
        dt, es = self.dtype, x_target.element_size()
        main = torch.cuda.current_stream()
        cs = comm.comm_stream(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        p_last = torch.empty(B, A_t, **f32)                               # block i = rows [i Bb, (i+1) Bb), shard-major inside
        p_pre = torch.empty(W, nb, max(G - 1, 1), rpb, A_t, dtype=dt, device=dev)
        abs_rows = torch.empty(B, **f32)
        # world 1: every exchange is an identity (RCCL would copy each buffer onto a second one, 130 MB per step at config
        # #2 for nothing): the "received" buffers alias the "sent" ones and the calls are skipped, unless
        # cfg["world1_exchanges"] asks for them (tests: the transport's calls at the real shapes)
        ident = W == 1 and not cfg.get("world1_exchanges", False)
        g = torch.empty(B, A_t, dtype=dt, device=dev)
        my_last = p_last.view(nb, rpb, A_t) if ident else torch.empty(nb, rpb, A_t, **f32)
        g_l = g.view(nb, rpb, A_t) if ident else torch.empty(nb, rpb, A_t, dtype=dt, device=dev)
        x_hat = torch.empty(B, A_t, dtype=dt, device=dev)
        # tail record of this rank (one all-gather): x_hat of its rows | per-row statistics | {nnz, sum |a|} of its shard
        n_x, n_st = nb * rpb * A_t * es, nb * (G + 4) * rpb * 4
        tail = torch.empty(n_x + n_st + 8, dtype=torch.uint8, device=dev)
        x_hat_l = tail[:n_x].view(dt).view(nb, rpb, A_t)
        stats_l = tail[n_x:n_x + n_st].view(torch.float32).view(nb, G + 4, rpb)
        nnz_abs_l = tail[n_x + n_st:].view(torch.float32)
        all_tail = tail.view(1, -1) if ident else torch.empty(W, n_x + n_st + 8, dtype=torch.uint8, device=dev)
        stats_l.zero_()
        db = _lib.make_desc(Bb, hp.A_s, hp.A_t, hp.D, hp.prefix, hp.code, hp.unit_norm)
        dl = _lib.make_desc(rpb, hp.A_s, hp.A_t, hp.D, hp.prefix, hp.code, hp.unit_norm)
        pre_stride = nb * max(G - 1, 1) * rpb * A_t
        rp, ev = sel.row_ptr, torch.cuda.Event

        def my_rows(t, i):                 # rows of block i this rank finishes
            return None if t is None else t[i * Bb + rank * rpb: i * Bb + (rank + 1) * rpb]

        def finish_block(i):
            check(lib.mtc_decode_finish(C.byref(dl), B, _p(my_last[i]), None, _p(b_dec), _p(my_rows(x_target, i)),
                                        _p(my_rows(mean, i)), _p(my_rows(std, i)), _p(x_hat_l[i]), _p(g_l[i]), _p(stats_l[i]),
                                        None, _stream()), "mtc_decode_finish")

        e_ag = []
        e_prev_rs = None
        one_block = nb == 1                # nothing to overlap on the critical path: its two exchanges stay on the main stream
        for i in range(nb):
            check(lib.mtc_decode_partial_strided(C.byref(db), _p(rp[i * Bb:]), _p(sel.col_idx), _p(sel.vals), _p(W_dec), rpb,
                                                 pre_stride, _p(p_last[i * Bb:]), _p(p_pre[0, i]), _p(abs_rows[i * Bb:]),
                                                 _stream()), "mtc_decode_partial")
            if one_block:
                if not ident:
                    comm.reduce_scatter_into(my_last[0], p_last, tag="partial reconstruction (reduce-scatter, fp32)")
                break
            e_dp = ev()
            e_dp.record(main)
            if i > 0:                      # finish block i-1 behind the decode of block i (its reduce-scatter ran beside it)
                main.wait_event(e_prev_rs)
                finish_block(i - 1)
                e_f = ev()
                e_f.record(main)
            with torch.cuda.stream(cs):
                cs.wait_event(e_dp)
                if not ident:
                    comm.reduce_scatter_into(my_last[i], p_last[i * Bb:(i + 1) * Bb], tag="partial reconstruction (reduce-scatter, fp32)")
                e_prev_rs = ev()
                e_prev_rs.record(cs)
                if i > 0:
                    cs.wait_event(e_f)
                    if not ident:
                        comm.all_gather_into(g[(i - 1) * Bb: i * Bb], g_l[i - 1], tag="g (all-gather)")
                    e = ev()
                    e.record(cs)
                    e_ag.append(e)
        if not one_block:
            main.wait_event(e_prev_rs)
        finish_block(nb - 1)
        check(lib.mtc_shard_nnz_abs(C.byref(d), _p(abs_rows), _p(sel.state), _p(nnz_abs_l), _stream()), "mtc_shard_nnz_abs")
        if one_block and not ident:
            comm.all_gather_into(g, g_l[0], tag="g (all-gather)")
        e_f = ev()
        e_f.record(main)
        l1_term = float(cfg["l1_coeff"]) / (B * self.total_dict_size)
        deferred = bool(getattr(self, "_defer_dvals", False))
        # the tail's finish runs beside the optimizer kernels: it must see THIS step's b_dec (a few KB), not the updated one
        b_dec_fwd = b_dec.clone() if deferred and G > 1 else b_dec
        row_stats = torch.empty(G + 4, B, **f32)
        nnz_abs = torch.empty(2, **f32)
        with torch.cuda.stream(cs):
            cs.wait_event(e_f)
            if not one_block:
                if not ident:
                    comm.all_gather_into(g[(nb - 1) * Bb:], g_l[nb - 1], tag="g (all-gather)")
                e = ev()
                e.record(cs)
                e_ag.append(e)
            # ---- off the critical path: earlier prefixes, x_hat, statistics
            if G > 1:
                if ident:
                    my_pre = p_pre.view(nb, G - 1, rpb, A_t)
                else:
                    my_pre = torch.empty(nb, G - 1, rpb, A_t, dtype=dt, device=dev)
                    comm.reduce_scatter_into(my_pre, p_pre, tag="earlier prefixes (reduce-scatter, model dtype)")
                for i in range(nb):
                    check(lib.mtc_decode_finish(C.byref(dl), B, None, _p(my_pre[i]), _p(b_dec_fwd), _p(my_rows(x_target, i)),
                                                _p(my_rows(mean, i)), _p(my_rows(std, i)), None, None, _p(stats_l[i]), None,
                                                _stream()), "mtc_decode_finish")
            else:
                my_pre = None
            if not ident:
                comm.all_gather_into(all_tail, tail, tag="x_hat + statistics (all-gather)")
            # [W][nb][rpb][A_t] -> row order [nb][W][rpb][A_t]; statistics [W][nb][G+4][rpb] -> [G+4][nb][W][rpb]
            x_hat.view(nb, W, rpb, A_t).copy_(all_tail[:, :n_x].view(dt).view(W, nb, rpb, A_t).transpose(0, 1))
            row_stats.view(G + 4, nb, W, rpb).copy_(
                all_tail[:, n_x:n_x + n_st].view(torch.float32).view(W, nb, G + 4, rpb).permute(2, 1, 0, 3))
            torch.sum(all_tail[:, n_x + n_st:].view(torch.float32), 0, out=nnz_abs)
            e_tail = ev()
            e_tail.record(cs)
        for e in e_ag:
            main.wait_event(e)             # g is complete: the shard-local backward can start
        # operands the transport's stream reads stay referenced until the next forward (allocator safety across streams)
        self._keep = (p_last, p_pre, abs_rows, my_last, g_l, tail, all_tail, my_pre, b_dec_fwd, x_target, mean, std)
        aux, aux_rows = None, None
        if aux_active:
            main.wait_event(e_tail)        # the aux branch reads x_hat
            e_tail = None
            aux, aux_rows = self._run_aux(hp, d, acts, sel, W_dec, x_target, x_hat, g, need_grad, l1_term)
        # the fused step derives d a_bj inside the decoder-gradient kernel, also with the dense aux branch active
        defer = deferred and (aux is None or bool(aux.get("dense")))
        d_vals = hp.decode_bwd_vals(d, sel, W_dec, g, l1_term) if need_grad and not defer else None
        scalars = torch.empty(16, **f32)

        def finish_metrics():
            if e_tail is not None:
                torch.cuda.current_stream().wait_event(e_tail)
            hp.reduce_scalars(d, row_stats, sel, self.total_dict_size, float(cfg["l1_coeff"]), float(cfg["aux_penalty"]),
                              aux_rows=aux_rows, nnz_abs_override=nnz_abs, out=scalars)

        if deferred:
            self._deferred = finish_metrics          # FusedAdam.train_step: after the backward kernels are issued
        else:
            finish_metrics()
        return x_hat, g, d_vals, row_stats, scalars, aux

    # ------------------------------------------------------------------ checkpoints in the reference layout
    def _gather_features(self, local: torch.Tensor, dim: int) -> torch.Tensor:
        parts = self.comm.all_gather(local.movedim(dim, 0).contiguous())      # [world, D_local, ...]
        flat = parts.reshape((-1,) + tuple(parts.shape[2:]))
        return flat[self.layout.owner_permutation().to(flat.device)].movedim(0, dim)

    def state_dict(self, *args, **kwargs):
        """Collective: every rank returns the full reference-layout state dict."""
        local = super().state_dict(*args, **kwargs)
        out = type(local)()
        for k, v in local.items():
            name = k.split(".")[-1]
            if name in ("b_enc", "W_dec"):
                out[k] = self._gather_features(v.detach(), 0)
            elif name == "W_enc":
                out[k] = self._gather_features(v.detach(), 1).contiguous()
            else:
                out[k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        idx = self.layout.global_index(self.comm.rank)
        local = {}
        for k, v in state_dict.items():
            if k in ("b_enc", "W_dec"):
                local[k] = v[idx.to(v.device)]
            elif k == "W_enc":
                local[k] = v[:, idx.to(v.device)]
            else:
                local[k] = v
        return super().load_state_dict(local, strict=strict, **kw)

    def gathered_dead_counters(self) -> torch.Tensor:
        return self._gather_features(self._cnt, 0)

    def encode(self, x_source):
        """Dense [.., D] features in GLOBAL feature order (collective)."""
        shape = x_source.shape
        global_indices = self.prefix_indices
        self.prefix_indices = [0] + list(self.local_prefix_sizes)      # the base class cuts at the active prefix: local coordinates
        try:
            local = super().encode(x_source)
        finally:
            self.prefix_indices = global_indices
        full = self._gather_features(local.reshape(-1, local.shape[-1]), 1)
        return full.reshape(shape[:-1] + (self.total_dict_size,))

    def decode(self, acts_topk):
        """sae.py:389-406 for a dense [N, D] tensor in GLOBAL feature order (what encode() returns; collective):
        every rank decodes its own columns into a bias-free fp32 partial sum, the partials are all-reduced,
        bias and de-normalisation follow replicated."""
        if not hasattr(self, "x_mean"):
            raise AttributeError("decode() needs the statistics of a previous encode()")
        if acts_topk.dim() != 2 or acts_topk.shape[1] < self.prefix_indices[self.active_prefixes]:
            raise _lib.MtcError(f"decode() expects [N, >= {self.prefix_indices[self.active_prefixes]}] features in global order, "
                                f"got {tuple(acts_topk.shape)}")
        hp0 = self.hot_path()
        lib, s, dev = hp0.lib, _stream(), acts_topk.device
        N = acts_topk.shape[0]
        local_max = ([0] + list(self.local_prefix_sizes))[self.active_prefixes]
        idx = self.layout.global_index(self.comm.rank)[:local_max].to(dev)
        dense = acts_topk.index_select(1, idx).to(self.dtype).contiguous()
        sub = HotPath(self.source_act_size, self.target_act_size, [local_max], self.dtype, False)
        d = sub.desc(N)
        nan = torch.full((1,), float("nan"), dtype=self.dtype, device=dev)      # "non-zero" select mode
        sel = sub.select_threshold(d, dense, nan)
        part = torch.empty(N, sub.A_t, dtype=torch.float32, device=dev)
        dummy = torch.empty(8, dtype=self.dtype, device=dev)
        check(lib.mtc_decode_partial(C.byref(d), _p(sel.row_ptr), _p(sel.col_idx), _p(sel.vals),
                                     _p(self.W_dec.data[:local_max].contiguous()), N, _p(part), _p(dummy), None, s),
              "mtc_decode_partial")
        self.comm.all_reduce(part)
        out = (part + self.b_dec.data.float()).to(self.dtype)
        if bool(self.config.get("input_unit_norm", False)):
            if self.x_mean is None or self.x_mean.numel() != N:
                raise RuntimeError(f"decode(): statistics of the last encode() cover "
                                   f"{0 if self.x_mean is None else self.x_mean.numel()} rows but {N} rows were given")
            out = out * self.x_std.reshape(N, 1) + self.x_mean.reshape(N, 1)      # each op rounds to the model dtype
        return out


class ShardedFusedAdam(FusedAdam):
    """FusedAdam over the local shard; nothing to exchange (b_dec's gradient is identical on all ranks)."""
