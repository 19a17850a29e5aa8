"""CPU-only tests of the host-side mirror of the reference interface: config defaults / hook names,
LR schedules, the activation store, and the reference import paths."""
import math
import os
import sys

import pytest
import torch

from conftest import ROOT


def test_default_cfg_and_names():
    from matryoshka_transcoder_amd import config as C
    cfg = C.get_default_cfg()
    assert cfg["seed"] == 49 and cfg["batch_size"] == 4096 and cfg["top_k"] == 32 and cfg["top_k_aux"] == 512
    assert cfg["aux_penalty"] == 1 / 32 and cfg["n_batches_to_dead"] == 20 and cfg["input_unit_norm"] is True
    assert cfg["hook_point"] == "blocks.8.hook_resid_pre"
    assert cfg["name"] == "gpt2-small_blocks.8.hook_resid_pre_12288_topk_32_0.0003"
    assert C.get_hook_name("mlp_in", 8, "gemma-2-2b") == "blocks.8.hook_mlp_in"
    assert C.get_hook_name("ln2_normalized", 8, "gemma-2-2b") == "blocks.8.ln2.hook_normalized"
    assert C.get_hook_name("mlp_out", 3, "gpt2") == "blocks.3.hook_mlp_out"
    assert C.get_model_config("gemma-2-9b")["act_size"] == 3584
    t = C.create_transcoder_config(dict(cfg, model_name="gemma-2-2b", dict_size=36864), 8, 9, "resid_post", "resid_pre")
    assert t["source_hook_point"] == "blocks.8.hook_resid_post" and t["target_hook_point"] == "blocks.9.hook_resid_pre"
    assert t["layer"] == 8 and t["source_act_size"] == t["target_act_size"] == 768
    assert C.group_sizes_to_prefix_sizes([2304, 4608, 9216, 20736]) == [2304, 6912, 16128, 36864]
    e = C.build_autointerp_eval_cfg(cfg, layer=5)
    assert e["source_hook_point"] == "blocks.5.hook_mlp_in" and e["num_tokens"] == 2_000_000
    x = torch.randn(50, 8); y = x + 0.1 * torch.randn(50, 8)
    assert abs(float(C.compute_fvu(x, y)) - float(((x - y) ** 2).mean() / (x.var() + 1e-8))) < 1e-7


def test_schedules():
    from matryoshka_transcoder_amd.training import create_scheduler, linear_warmup, warmup_cosine_decay
    assert warmup_cosine_decay(0, 10, 100) == 0.0                     # lr = 0 on step 0
    assert warmup_cosine_decay(5, 10, 100) == 0.5
    assert math.isclose(warmup_cosine_decay(55, 10, 100), 0.5)
    assert linear_warmup(0, 10) == 0.0 and linear_warmup(20, 10) == 1.0
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([p], lr=1e-3)
    s = create_scheduler(opt, {"scheduler_type": "warmup_decay", "warmup_steps": 4, "lr": 1e-3}, 20)
    assert opt.param_groups[0]["lr"] == 0.0
    opt.step(); s.step()
    assert math.isclose(opt.param_groups[0]["lr"], 0.25e-3)
    assert create_scheduler(opt, {"lr": 1e-3}, 20) is None
    s2 = create_scheduler(torch.optim.Adam([p], lr=1e-3), {"scheduler_type": "decay_only", "lr": 1e-3}, 20)
    assert s2.eta_min == 1e-5


def test_activation_store_sources():
    from matryoshka_transcoder_amd.activation_store import TensorPairSource, TranscoderActivationsStore
    cfg = dict(device="cpu", dtype=torch.float32, act_size=16, source_act_size=16, target_act_size=8, batch_size=32,
               num_batches_in_buffer=2, model_batch_size=4, seq_len=10, layer=3)
    store = TranscoderActivationsStore(None, cfg)
    seen = 0
    for _ in range(3):
        s, t = store.next_batch()
        assert s.shape[1] == 16 and t.shape[1] == 8 and s.shape[0] == t.shape[0] <= 32
        seen += s.shape[0]
    assert seen == 80                                                  # 2 chunks x 40 rows: 32 + 32 + a short 16
    src = torch.arange(64.).reshape(64, 1).repeat(1, 16)
    store = TranscoderActivationsStore(None, dict(cfg, target_act_size=16, num_batches_in_buffer=1),
                                       source=TensorPairSource([(src, src + 1)]))
    s, t = store.next_batch()
    assert torch.equal(t, s + 1)                                       # pairs stay aligned under the shuffle
    assert len(set(s[:, 0].tolist())) == 32

    class FakeLM:                                                      # run_with_cache duck type
        tokenizer = "tok"
        class cfg: n_ctx = 10
        def run_with_cache(self, tokens, names_filter, stop_at_layer):
            assert stop_at_layer == 4
            b, n = tokens.shape
            return None, {names_filter[1]: tokens.float()[..., None].repeat(1, 1, 16),
                          names_filter[0]: -tokens.float()[..., None].repeat(1, 1, 16)}
    cfg2 = dict(cfg, target_act_size=16, source_hook_point="blocks.3.hook_resid_pre", target_hook_point="blocks.3.hook_mlp_out")
    toks = iter([[i] * 10 for i in range(10_000)])
    store = TranscoderActivationsStore(FakeLM(), cfg2, token_iterator=toks)
    s, t = store.next_batch()
    assert torch.equal(t, -s) and store.tokenizer == "tok"
    a, tk = store.get_batch_with_tokens()
    assert a.shape == (4, 10, 16) and tk.shape == (4, 10)


def test_reference_import_paths():
    sys.path.insert(0, ROOT)
    from src.models.sae import MatryoshkaTranscoder as A
    from src.training.training import train_transcoder  # noqa: F401
    from src.models.transcoder_activation_store import TranscoderActivationsStore, create_transcoder_config  # noqa: F401
    from src.utils.config import get_default_cfg  # noqa: F401
    from src.utils.clean_results import CleanResultsManager  # noqa: F401
    from src.utils.activation_samples import ActivationSampleCollector  # noqa: F401
    from eval.sae_bench import build_matryoshka_adapter, build_npz_transcoder_adapter  # noqa: F401
    from eval.sae_bench.wrappers import MatryoshkaAutoInterpAdapter
    from eval.sae_bench.config import NPZTranscoderSpec  # noqa: F401
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder as B
    assert A is B and issubclass(MatryoshkaAutoInterpAdapter, B)


def test_bench_launches_one_rank_per_gpu_by_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts 2 ranks itself (it used to bench
    one GPU silently); rank 0 alone prints the JSON line, with the world size the ranks saw.  CPU / gloo rehearsal."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MTC_BENCH_LAUNCH_TEST"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec == {"launch_test": True, "n_gpus": 2, "ranks_seen": 2}
    # a world size that contradicts --gpus is refused instead of benchmarking something else
    env2 = dict(env, WORLD_SIZE="4", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr
