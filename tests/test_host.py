"""CPU tests of the product's host side: the C-ABI library loads and exports every
symbol include/shdr.h declares, host-side validation fails loudly, the drop-in
modules expose the reference's call surface.  No kernel is launched here."""
import ctypes
import importlib
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import nets


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "shdr.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(shdr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(shdr):
    lib = shdr._lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libshdr.so does not export %s" % n
    assert set(names) == set(shdr._lib.SIGNATURES), "ctypes table and header disagree"
    assert b"gfx950" in lib.shdr_version()


def test_same_pad_c_matches_tf_rule(shdr):
    lib = shdr._lib.load()
    out, pad = ctypes.c_int(), ctypes.c_int()
    for in_size, k, s, exp in ((512, 7, 2, (256, 2)), (128, 1, 2, (64, 0)), (256, 3, 2, (128, 0)),
                               (64, 3, 1, (64, 1)), (64, 5, 1, (64, 2)), (7, 3, 2, (4, 1))):
        assert lib.shdr_same_pad(in_size, k, s, ctypes.byref(out), ctypes.byref(pad)) == 0
        assert (out.value, pad.value) == exp
        assert shdr._ops.same_pad(in_size, k, s) == exp
    assert lib.shdr_same_pad(0, 3, 1, None, None) < 0
    assert b"same_pad" in lib.shdr_last_error()


def test_host_validation_returns_error_codes(shdr):
    lib = shdr._lib.load()
    d = shdr._lib.ConvDesc()
    assert lib.shdr_conv2d_fwd_f32(ctypes.byref(d), None, None, None, None, None, None, None, None, None) == -5
    assert b"null" in lib.shdr_last_error()
    assert lib.shdr_soft_hist_fwd_f32(None, None, 4, 3, 8, None) == -5
    assert lib.shdr_increase_fwd_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 1, None) == -1
    assert lib.shdr_maxpool2_fwd_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 3, 4, 4, None) == -1
    assert lib.shdr_avgpool2_fwd_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 4, 4, 3, None) == -2
    assert lib.shdr_lin_frontend_fwd_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 8, 8, 64, None) == -1


def test_ops_refuse_cpu_tensors(shdr):
    x = torch.zeros(1, 8, 8, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        shdr._ops.avgpool2(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        shdr.dequantization_net.model(device=torch.device("cpu"))(torch.zeros(1, 16, 16, 3), training=False)


@pytest.mark.parametrize("mod,spec", [("dequantization_net", nets.deq_spec), ("linearization_net", nets.lin_spec),
                                      ("hallucination_net", nets.hal_spec), ("refinement_net", nets.ref_spec)])
def test_models_expose_keras_surface(shdr, mod, spec):
    m = getattr(shdr, mod).model(device=torch.device("cpu"))
    named = m.named_weights()
    assert [n for n, _, _ in named] == [n for n, _, _ in spec()]
    assert [tuple(t.shape) for _, t, _ in named] == [s for _, s, _ in spec()]
    tv = m.trainable_variables
    assert sum(t.numel() for t in tv) == nets.count_trainable(spec())
    assert all(t.requires_grad for t in tv) and not any(t.requires_grad for t in m.non_trainable_variables)
    assert isinstance(tv + tv, list)            # joint_training.py:185 concatenates with '+'
    # kernel before bias, gamma before beta
    names = [n for n, _, tr in named if tr]
    for i, n in enumerate(names):
        if n.endswith(".bias"):
            assert names[i - 1] == n[:-5] + ".kernel"
        if n.endswith(".beta"):
            assert names[i - 1] == n[:-5] + ".gamma"


def test_keras_default_initialisers(shdr):
    torch.manual_seed(0)
    m = shdr.hallucination_net.model(device=torch.device("cpu"))
    sd = m.state_dict()
    k = sd["d2.conv1.kernel"]
    lim = np.sqrt(6.0 / (9 * 64 + 9 * 128))
    assert float(k.abs().max()) <= lim and float(k.abs().max()) > 0.9 * lim
    assert float(sd["d2.conv1.bias"].abs().max()) == 0
    assert float((sd["norm1.gamma"] - 1).abs().max()) == 0 and float(sd["norm1.moving_variance"].min()) == 1
    assert not any(n.startswith("u5.conv2") for n in sd)   # never-built layer owns no weights


def test_load_numpy_roundtrip_and_padded_kernel(shdr):
    m = shdr.linearization_net.model(device=torch.device("cpu"))
    p = nets.init_params(nets.lin_spec(), 3)
    m.load_numpy(p)
    for n, t in m.state_dict().items():
        np.testing.assert_array_equal(t.numpy(), p[n])
    conv1 = m.crf_feature_net.conv1
    w96 = conv1.kernel_padded(96)
    assert tuple(w96.shape) == (7, 7, 96, 64)
    np.testing.assert_array_equal(w96[:, :, :93].numpy(), p["crf_feature_net.conv1.kernel"])
    assert float(w96[:, :, 93:].abs().max()) == 0
    with pytest.raises(KeyError):
        m.load_numpy({"bogus": np.zeros(1)})


def test_invemor_table_lookup_prefers_cwd(shdr, tmp_path, monkeypatch, emor_table):
    t = shdr.linearization_net.load_invemor_table(torch.device("cpu"))
    np.testing.assert_array_equal(t.numpy(), emor_table)
    # an invemor.txt in the CWD wins, as in the reference (linearization_net.py:219)
    lines = []
    for tag in ["B ="] + ["g0 ="] + ["hinv(%d)=" % (i + 1) for i in range(11)]:
        lines.append(tag)
        col = np.full(1024, 0.5 if tag == "g0 =" else 0.0)
        lines += ["   ".join("%e" % v for v in col[i:i + 4]) for i in range(0, 1024, 4)]
    (tmp_path / "invemor.txt").write_text("\n".join(lines))
    monkeypatch.chdir(tmp_path)
    t2 = shdr.linearization_net.load_invemor_table(torch.device("cpu"))
    assert float(t2[:, 0].min()) == 0.5 and float(t2[:, 1:].abs().max()) == 0


def test_dropin_module_names_importable_from_package_dir(shdr):
    """With the package directory on sys.path the reference's own import lines work unchanged."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import dequantization_net as deq, linearization_net as lin, "
            "hallucination_net as hal, refinement_net as ref, tf_utils; from vgg16 import Vgg16; "
            "print(deq.model.__name__, lin.model.__name__, hal.model.__name__, ref.model.__name__)"
            % os.path.join(ROOT, "singlehdr-tf2_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["model"] * 4


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "singlehdr-tf2_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_documented_ctypes_mirror_matches_the_header():
    """INTEGRATION.md shows the ctypes mirror of shdr_conv2d_desc a maintainer would write; it must list the fields of the
    struct in include/shdr.h (same names, same order) -- a short mirror makes the library read past the caller's struct."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "shdr.h")).read()
    body = hdr[hdr.index("typedef struct shdr_conv2d_desc {"):hdr.index("} shdr_conv2d_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = []
    for typ, names in re.findall(r"(int32_t|int64_t|float)\s+([^;]+);", body):
        c_fields += [(n.strip(), typ) for n in names.split(",")]
    lib = importlib.import_module("singlehdr-tf2_amd._lib")
    ctype = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float}
    assert [(n, ctype[t]) for n, t in c_fields] == list(lib.ConvDesc._fields_)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    snippet = doc[doc.index("class ConvDesc(ctypes.Structure):"):doc.index("lib.shdr_conv2d_fwd_f32.restype")]
    ns = {"ctypes": ctypes}
    exec(snippet, ns)
    assert [(n, t) for n, t in ns["ConvDesc"]._fields_] == list(lib.ConvDesc._fields_)


def test_bench_rank0_only_legs_hold_no_collective():
    """bench.py runs some legs on rank 0 only; a leg that steps a DP model there all-reduces alone and hangs the N > 1 runs
    (the instrumented fp16 fine-tuning step did): such legs must be restricted to world == 1."""
    src = open(os.path.join(ROOT, "bench.py")).read().splitlines()
    for i, line in enumerate(src):
        if "= fp16_roofline(K, fstep" in line:
            cond = " ".join(src[max(0, i - 3):i])
            assert "world == 1" in cond, "bench.py:%d steps the DP fine-tuning model on rank 0 only" % (i + 1)
            break
    else:
        raise AssertionError("fp16 roofline leg not found")
