"""f1: the activation store on the GPU -- transcoder_activation_store.py:157-207.  The shuffle is a device
permutation, a batch is one mtc_gather_row_pairs launch.  Properties the reference's DataLoader(shuffle=True) has and
this must keep: every buffered row appears exactly once per epoch, source / target rows stay paired, the last batch is
short, the next epoch reshuffles; plus the file-backed source and the token path that feeds the sample collector."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(**kw):
    cfg = dict(device=DEV, dtype=torch.bfloat16, act_size=64, source_act_size=64, target_act_size=40, batch_size=96,
               num_batches_in_buffer=3, model_batch_size=4, seq_len=25, layer=3)
    cfg.update(kw)
    return cfg


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_epoch_is_a_permutation_with_aligned_pairs_and_a_short_last_batch(dt, gpu_lib):
    from matryoshka_transcoder_amd.activation_store import TensorPairSource, TranscoderActivationsStore
    cfg = _cfg(dtype=dt)
    # bf16 holds integers exactly only up to 256: rows are tagged by (chunk, row) in their first two columns
    def tag(i):
        s = torch.zeros(100, 64); t = torch.zeros(100, 40)
        s[:, 0] = i; s[:, 1] = torch.arange(100.); t[:, 0] = -i; t[:, 1] = -torch.arange(100.)
        s[:, 2:] = torch.randn(100, 62, generator=torch.Generator().manual_seed(i)); t[:, 2:] = s[:, 2:40] * 2
        return s.to(dt), t.to(dt)
    chunks = [tag(i) for i in range(3)]
    store = TranscoderActivationsStore(None, cfg, source=TensorPairSource(chunks))
    seen, sizes, order = set(), [], []
    for _ in range(4):                                                           # 300 rows: 96 + 96 + 96 + a short 12
        s, t = store.next_batch()
        assert s.is_cuda and t.is_cuda and s.dtype == dt and s.shape[1] == 64 and t.shape[1] == 40
        sizes.append(s.shape[0])
        assert torch.equal(t[:, :2], -s[:, :2])                                  # the pair of each row came along
        assert torch.equal(t[:, 2:], (s[:, 2:40].float() * 2).to(dt))
        for c, r in s[:, :2].float().cpu().tolist():
            key = (int(c), int(r))
            assert key not in seen
            seen.add(key)
            order.append(key)
        full = torch.stack([chunks[int(c)][0][int(r)] for c, r in s[:, :2].float().cpu().tolist()])
        assert torch.equal(s.cpu(), full)                                        # whole rows, bit for bit
    assert sizes == [96, 96, 96, 12] and len(seen) == 300
    assert order != sorted(order)                                                # shuffled
    s2, _ = store.next_batch()                                                   # next epoch: refill + new permutation
    assert s2.shape[0] == 96
    second = [(int(c), int(r)) for c, r in s2[:, :2].float().cpu().tolist()]
    assert second != order[:96]


def test_gather_kernel_refuses_bad_indices_and_odd_rows(gpu_lib):
    from matryoshka_transcoder_amd import _lib
    from matryoshka_transcoder_amd.engine import _p, _stream
    src = torch.randn(10, 8, device=DEV); tgt = torch.randn(10, 4, device=DEV)
    perm = torch.tensor([3, 10, 0, -1], device=DEV)
    out_s = torch.full((4, 8), 7.0, device=DEV); out_t = torch.full((4, 4), 7.0, device=DEV)
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(gpu_lib.mtc_gather_row_pairs(_lib.MTC_F32, 4, 8, 4, 10, _p(src), _p(tgt), _p(perm), _p(out_s), _p(out_t), _p(bad),
                                            _stream()), "gather")
    assert int(bad) == 2 and torch.equal(out_s[0], src[3]) and torch.equal(out_t[2], tgt[0])
    assert not bool(out_s[1].any()) and not bool(out_t[3].any())                 # out-of-range rows: zeros, nothing read
    rc = gpu_lib.mtc_gather_row_pairs(_lib.MTC_BF16, 4, 12, 8, 10, _p(src), _p(tgt), _p(perm), _p(out_s), _p(out_t), None, _stream())
    assert rc != 0 and "16 bytes" in gpu_lib.mtc_last_error().decode()


def test_file_backed_source_and_training_on_store_batches(gpu_lib, tmp_path):
    """TensorPairSource over .pt files (loaded with weights_only=True) feeding train_transcoder's step."""
    from matryoshka_transcoder_amd.activation_store import TensorPairSource, TranscoderActivationsStore
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    paths = []
    for i in range(2):
        g = torch.Generator().manual_seed(50 + i)
        p = tmp_path / f"chunk{i}.pt"
        torch.save((torch.randn(160, 64, generator=g), torch.randn(160, 64, generator=g)), p)
        paths.append(str(p))
    cfg = _cfg(target_act_size=64, num_batches_in_buffer=2, batch_size=128, seed=3, prefix_sizes=[64, 256, 512], top_k=8,
               top_k_aux=16, aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0, lr=3e-4, beta1=0.9, beta2=0.99)
    store = TranscoderActivationsStore(None, cfg, source=TensorPairSource(paths))
    tc = MatryoshkaTranscoder(cfg)
    opt = FusedAdam(tc, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    losses = []
    for _ in range(5):                                                           # 320 rows: 128 + 128 + 64, refill, ...
        s, t = store.next_batch()
        assert s.dtype == torch.bfloat16 and s.device.type == "cuda"
        losses.append(float(opt.train_step(s, t)["loss"].float()))
    assert all(x == x for x in losses)


def test_tokens_path_feeds_the_sample_collector(gpu_lib):
    """get_batch_with_tokens -> transcoder.encode -> ActivationSampleCollector (training.py:138-152) on the GPU."""
    from matryoshka_transcoder_amd.activation_samples import ActivationSampleCollector
    from matryoshka_transcoder_amd.activation_store import TranscoderActivationsStore
    from matryoshka_transcoder_amd.training import collect_activation_samples
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder

    class FakeLM:
        class tokenizer:
            @staticmethod
            def decode(ids, **kw):
                return " ".join(f"t{int(i)}" for i in ids)
        class cfg: n_ctx = 25
        def run_with_cache(self, tokens, names_filter, stop_at_layer):
            g = torch.Generator(device=tokens.device).manual_seed(int(tokens.sum()) % 1000)
            acts = torch.randn(*tokens.shape, 64, generator=g, device=tokens.device)
            return None, {names_filter[1]: acts, names_filter[0]: -acts}

    cfg = _cfg(target_act_size=64, source_hook_point="blocks.3.hook_resid_pre", target_hook_point="blocks.3.hook_mlp_out",
               seed=3, prefix_sizes=[64, 256, 512], top_k=8, top_k_aux=16, aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0)
    toks = iter([[i % 500] * 25 for i in range(100_000)])
    store = TranscoderActivationsStore(FakeLM(), cfg, token_iterator=toks)
    s, t = store.next_batch()
    assert s.shape == (96, 64) and torch.equal(t, -s)                            # LM-hooked pairs through the same gather
    tc = MatryoshkaTranscoder(cfg)
    col = ActivationSampleCollector(max_samples_per_feature=4, context_size=6, storage_threshold=0.0)
    collect_activation_samples(col, tc, store)
    assert col.total_samples_seen == 8 * 4 * 25                                  # training-mode encode keeps k per position
