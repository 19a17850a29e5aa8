"""World-size-2 gloo tests (CPU) of the host logic of the feature-sharded model: partition
arithmetic, tie ordering, construction / state_dict gather / scatter.  The kernels themselves are
covered by tests/test_gpu_sharded.py on the GPU box."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn_name](rank, world)
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def test_layout_partition():
    from matryoshka_transcoder_amd.sharded import ShardLayout
    prefix = [2304, 6912, 16128, 36864]
    for world in (2, 4, 8):
        seen = []
        for r in range(world):
            lay = ShardLayout(prefix, world, r)
            assert lay.local_prefix == [p // world for p in prefix]
            idx = lay.global_index(r)
            assert idx.numel() == lay.D_local
            # local group i only holds features of global group i (nesting is preserved per shard)
            lo = 0
            for i, lp in enumerate(lay.local_prefix):
                seg = idx[lo:lp]
                assert int(seg.min()) >= ([0] + prefix)[i] and int(seg.max()) < prefix[i]
                lo = lp
            seen.append(idx)
        allidx = torch.cat(seen)
        assert torch.equal(torch.sort(allidx).values, torch.arange(prefix[-1]))
        perm = ShardLayout(prefix, world, 0).owner_permutation()
        assert torch.equal(allidx[perm], torch.arange(prefix[-1]))
    from matryoshka_transcoder_amd import _lib
    with pytest.raises(_lib.MtcError):
        ShardLayout([16, 48, 112, 256], 4, 0)          # groups must split into multiples of 8 per rank


def test_tie_grant_is_global_flat_index_order():
    """Ties at tau are granted in ascending GLOBAL flat index (what the single-GPU select does), whatever the world
    size: brute force over the global feature order against the per-(row, group) arithmetic the kernel uses."""
    from matryoshka_transcoder_amd.sharded import ShardLayout, tie_take_global_order
    g = torch.Generator().manual_seed(5)
    prefix = [32, 96, 224, 512]
    B = 6
    tie_mask = torch.rand(B, prefix[-1], generator=g) < 0.08           # global [B, D]: entries equal to tau
    total = int(tie_mask.sum())
    for world in (1, 2, 4):
        lay0 = ShardLayout(prefix, world, 0)
        bounds = [0] + lay0.local_prefix
        all_eq = torch.zeros(world, B, len(prefix), dtype=torch.int64)
        local_masks = []
        for r in range(world):
            lm = tie_mask[:, lay0.global_index(r)]                     # this rank's columns, local order
            local_masks.append(lm)
            for gi in range(len(prefix)):
                all_eq[r, :, gi] = lm[:, bounds[gi]:bounds[gi + 1]].sum(1)
        for quota in (0, 1, total // 3, total - 1, total, total + 5):
            # brute force: the first `quota` ties in (row, global feature) order
            flat_pos = tie_mask.reshape(-1).nonzero().reshape(-1)[:quota]
            granted = torch.zeros(B * prefix[-1], dtype=torch.bool)
            granted[flat_pos] = True
            granted = granted.reshape(B, prefix[-1])
            for r in range(world):
                want_mask = granted[:, lay0.global_index(r)]
                take = tie_take_global_order(all_eq, quota, r)
                assert take.tolist() == want_mask.sum(1).tolist(), (world, quota, r)
                # and it is a PREFIX of the row's local ties (one number per row describes it)
                for b in range(B):
                    loc = local_masks[r][b].nonzero().reshape(-1)[: int(take[b])]
                    assert torch.equal(want_mask[b].nonzero().reshape(-1), loc), (world, quota, r, b)


def _state_dict_roundtrip(rank, world):
    from matryoshka_transcoder_amd.sharded import ShardedMatryoshkaTranscoder
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    fx, cfg = load_golden("fp32_plain")
    cfg = dict(cfg, prefix_sizes=[16, 48, 112, 256])
    m = ShardedMatryoshkaTranscoder(cfg)
    assert m.W_dec.shape == (128, 32) and m.W_enc.shape == (32, 128) and m.b_enc.shape == (128,)
    assert m.W_enc.t().is_contiguous()
    sd = m.state_dict()
    for k, v in fx["init"].items():                       # gathers back to the reference's init exactly
        assert sd[k].shape == v.shape and torch.equal(sd[k], v), k
    after = fx["steps"][-1]["params_after"]
    m.load_state_dict(after)
    sd2 = m.state_dict()
    for k, v in after.items():
        assert torch.equal(sd2[k], v), k
    full = MatryoshkaTranscoder(cfg)
    full.load_state_dict(sd2)                             # a single-GPU model loads the gathered dict
    # collectives used by the step
    t = torch.full((4,), float(rank + 1))
    assert m.comm.all_reduce(t).tolist() == [3.0] * 4
    g = m.comm.all_gather(torch.tensor([rank, rank + 10]))
    assert g.tolist() == [[0, 10], [1, 11]]


def test_state_dict_gather_scatter_world2():
    _run("_state_dict_roundtrip")
