"""CPU-only: libmtc.so loads and exports exactly the symbols include/mtc.h declares; the ctypes
prototype table covers all of them; the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "mtc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from matryoshka_transcoder_amd import _lib
    assert os.path.exists(_lib.lib_path()), "build libmtc.so first (make)"
    lib = ctypes.CDLL(_lib.lib_path())
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mtc.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, set(_lib.PROTOTYPES) ^ set(names)
    assert _lib.load().mtc_version() == 1


def test_header_is_plain_c_and_links(tmp_path):
    """The boundary is a C ABI: include/mtc.h must compile as C99 (no C++ / HIP / torch types) and a C program must
    link against libmtc.so and call into it."""
    import shutil
    import subprocess
    from matryoshka_transcoder_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "host.c"
    src.write_text('#include <stdio.h>\n#include "mtc.h"\n'
                   'int main(void) {\n'
                   '    mtc_desc d = {0};\n'
                   '    int rc = mtc_row_normalize(&d, 0, 0, 0, 0, 0);   /* rejected on the host: B = 0 */\n'
                   '    printf("%d %d %s\\n", mtc_version(), rc, mtc_last_error());\n'
                   '    return 0;\n}\n')
    libdir = os.path.dirname(_lib.lib_path())
    exe = tmp_path / "host"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-L", libdir, "-l:libmtc.so", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split(" ", 2)
    assert out[0] == "1" and int(out[1]) != 0 and "must be positive" in out[2]


def test_host_side_validation_without_gpu():
    """Descriptor checks run on the host before any launch: errors, not crashes."""
    from matryoshka_transcoder_amd import _lib
    lib = _lib.load()
    d = _lib.make_desc(4, 30, 32, 256, [16, 48, 112, 256], _lib.MTC_F32, False)     # A_s not a multiple of 8
    assert lib.mtc_row_normalize(ctypes.byref(d), None, None, None, None, None) != 0
    assert b"multiples of 8" in lib.mtc_last_error()
    d = _lib.make_desc(4, 32, 32, 256, [16, 8, 112, 256], _lib.MTC_F32, False)       # non-monotone prefixes
    assert lib.mtc_encode_fwd(ctypes.byref(d), None, None, None, None, None) != 0
    assert b"non-decreasing" in lib.mtc_last_error()
    with pytest.raises(_lib.MtcError):
        _lib.make_desc(4, 32, 32, 256, list(range(8, 8 * 18, 8)), _lib.MTC_F32, False)


def test_no_cpu_fallback():
    from matryoshka_transcoder_amd import _lib
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    cfg = dict(seed=49, prefix_sizes=[16, 48, 112, 256], act_size=32, top_k=4, top_k_aux=8, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.float32, device="cpu", input_unit_norm=False)
    m = MatryoshkaTranscoder(cfg)          # construction / state_dict work anywhere
    assert list(m.state_dict()) == ["b_enc", "W_enc", "W_dec", "b_dec", "threshold"]
    with pytest.raises(_lib.MtcError):
        m(torch.randn(4, 32), torch.randn(4, 32))
    with pytest.raises(ValueError):
        MatryoshkaTranscoder(dict(cfg, prefix_sizes=[16, 8, 256]))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "matryoshka_transcoder_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src, fn


def test_results_layout(tmp_path):
    from matryoshka_transcoder_amd.results import CleanResultsManager
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    cfg = dict(seed=49, prefix_sizes=[16, 48, 112, 256], act_size=32, top_k=4, top_k_aux=8, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.bfloat16, device="cpu", input_unit_norm=False)
    m = MatryoshkaTranscoder(cfg)
    r = CleanResultsManager(str(tmp_path / "results"))
    d = r.create_experiment_dir("gpt2-small", 8, 100)
    r.save_checkpoint(d, m, cfg, 10, {"loss": 1.0}, is_final=False)
    r.save_checkpoint(d, m, cfg, 100, {"loss": 0.5}, is_final=True)
    assert (d / "checkpoints" / "step_10.pt").exists() and (d / "checkpoints" / "final.pt").exists()
    assert (d / "metrics" / "step_10.json").exists() and (d / "metrics.json").exists() and (d / "config.json").exists()
    sd = torch.load(d / "checkpoints" / "final.pt", weights_only=True)
    assert list(sd) == ["b_enc", "W_enc", "W_dec", "b_dec", "threshold"]
    assert sd["W_enc"].shape == (32, 256) and sd["W_enc"].is_contiguous() and sd["W_enc"].dtype == torch.bfloat16
    assert r.list_experiments()["gpt2-small"][8][0]["has_final_checkpoint"]
