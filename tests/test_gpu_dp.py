"""GPU test of the PRODUCT's data-parallel branches (pipeline.py: the `world > 1` paths of JointTrainStep, TrainStep and
FinetuneStep -- the scalar all-reduce of sum(loss_mask), the global batch factor of the broadcast loss, the TV rescale, the ONE
all_reduce(SUM) of the flat gradient, the fp16 skip after the collective).

Two fresh child processes (tests/dp_worker.py) share the one visible card over the `gloo` backend -- the process group is up
before either touches the GPU -- and run their shard of a seeded batch of 4 with a zero in `loss_mask`.  The parent computes
the single-process references with the same kernels.  Mirrors joint_training.py:179-186 (the reference itself is single-GPU;
SURVEY.md section 8e defines the sharded semantics)."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import dp_worker as W
from oracle import nets

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def ranks(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("dp")
    port = 29600 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), "2", str(port), str(tmp / ("r%d.npz" % r))],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return [dict(np.load(str(tmp / ("r%d.npz" % r)))) for r in range(2)]


def test_both_ranks_hold_the_same_reduced_gradient_and_update(ranks):
    r0, r1 = ranks
    for k in r0:
        if k.startswith(("joint_grad_", "hal_grad", "lin_grad", "ft_grad")):
            assert np.array_equal(r0[k], r1[k]), k                     # one all-reduce: bit-identical on every replica
    assert np.array_equal(r0["joint_params_after"], r1["joint_params_after"])    # ...hence identical Adam updates
    assert float(r0["joint_objective"]) != float(r1["joint_objective"])         # (the shards themselves differ)


def test_bucketed_overlapped_reduction_equals_the_blocking_one(ranks):
    """JointTrainStep(bucketed=True), the default the tests above ran: four async collectives (hal decoder half from a tape mark at the
    bottleneck, hal encoder half, lin, deq) launched from inside the backward pass on the stream their slice was written on.  The
    buckets partition the flat gradient; reducing a buffer bucket by bucket on side streams gives the bits of ONE all_reduce; the
    step's reduced gradients equal the blocking step's up to the atomics noise of two separate backward passes (5e-4, as above)."""
    for r in ranks:
        b = sorted(map(tuple, r["buckets"].tolist()))
        assert b[0][0] == 0 and b[-1][1] == int(r["flat_numel"]) and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1)) and len(b) == 4
        assert bool(r["bucketed_equals_blocking_collective"])
        for k in ("deq", "lin", "hal"):
            assert rel_l2(r["joint_grad_" + k], r["joint_blocking_grad_" + k]) <= 5e-4, k
    assert rel(ranks[0]["joint_grad_deq"], ranks[0]["joint_blocking_grad_deq"]) <= 1e-5


def test_joint_step_sharded_equals_full_batch(shdr, ranks, monkeypatch):
    P = shdr.pipeline
    d = W.make_data(4, 64)
    full = tuple(dev(d[k]) for k in ("ldr", "jpeg", "clipped", "hdr_t", "mask"))
    # (1) the single-process step on the WHOLE batch: the Dequantization-Net has no batch-coupled op, so its all-reduced
    #     gradient must equal the full-batch gradient -- factor B = 4 of the broadcast loss included
    models = W.build_models(shdr, nets)
    one = P.JointTrainStep(models["deq"], models["lin"], models["hal"], W.build_vgg(shdr, nets), lr=1e-3)
    one(full, dev(d["inv"]), apply=False)
    g_full = W.net_grads(one, models)
    assert rel(ranks[0]["joint_grad_deq"], g_full["deq"]) <= 1e-5
    # lin / hal normalise with per-replica batch statistics (recorded deviation): NOT the full-batch gradient
    assert rel_l2(ranks[0]["joint_grad_lin"], g_full["lin"]) > 1e-3
    assert rel_l2(ranks[0]["joint_grad_hal"], g_full["hal"]) > 1e-3
    # (2) ...and that is the ONLY difference: per-replica statistics == running the step on each shard alone, with the GLOBAL
    #     batch size and mask sum in the batch-coupled scalars (what the scalar all-reduce provides), and summing the gradients
    monkeypatch.setattr(P, "_dp_scalars", lambda step, mask: (4.0, torch.tensor(float(d["mask"].sum()), device="cuda")))
    acc = None
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        models = W.build_models(shdr, nets)
        emu = P.JointTrainStep(models["deq"], models["lin"], models["hal"], W.build_vgg(shdr, nets), lr=1e-3)
        emu.world = 2                                           # shard semantics of the TV weight; no process group: no collective
        emu(tuple(dev(d[k][sl]) for k in ("ldr", "jpeg", "clipped", "hdr_t", "mask")), dev(d["inv"][sl]), apply=False)
        g = W.net_grads(emu, models)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    for k in ("deq", "lin", "hal"):
        assert rel_l2(ranks[0]["joint_grad_" + k], acc[k]) <= 5e-4, k            # same kernels; atomics reorder the sums
    assert rel(ranks[0]["joint_grad_deq"], acc["deq"]) <= 1e-5


def test_per_network_steps_sharded(shdr, ranks, monkeypatch):
    P = shdr.pipeline
    d = W.make_data(4, 64)
    monkeypatch.setattr(P, "_dp_scalars", lambda step, mask: (4.0, torch.tensor(float(d["mask"].sum()), device="cuda")))
    acc_h = acc_l = None
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        models = W.build_models(shdr, nets)
        h = P.TrainStep("hal", models["hal"], W.build_vgg(shdr, nets))
        h.world = 2
        h((dev(d["hdr_t"][sl]), dev(d["clipped"][sl]), dev(d["mask"][sl])), apply=False)
        ln = P.TrainStep("lin", models["lin"])
        ln.world = 2
        ln((dev(d["ldr"][sl]), dev(d["clipped"][sl]), dev(d["mask"][sl]), dev(d["inv"][sl])), apply=False)
        gh, gl = h.params.grad.detach().cpu().numpy(), ln.params.grad.detach().cpu().numpy()
        acc_h, acc_l = (gh, gl) if acc_h is None else (acc_h + gh, acc_l + gl)
    # same kernels, same per-replica statistics: what is left is the order of the fp32 atomics in the weight-gradient kernels
    # (run-to-run reproducibility of these training-mode nets is ~1e-4 in relative L2, test_gpu_grad.py)
    assert rel_l2(ranks[0]["hal_grad"], acc_h) <= 5e-4
    assert rel_l2(ranks[0]["lin_grad"], acc_l) <= 5e-4


def test_finetune_step_sharded_and_fp16_skip_after_the_collective(shdr, ranks):
    P = shdr.pipeline
    d = W.make_data(4, 64)
    acc = None
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        models = W.build_models(shdr, nets, with_ref=True)
        f = P.FinetuneStep(models["deq"], models["lin"], models["hal"], models["ref"])
        f(dev(d["ldr"][sl]), dev(d["hdr"][sl]), apply=False)
        g = f.params.grad.detach().cpu().numpy()
        acc = g if acc is None else acc + g
    assert rel_l2(ranks[0]["ft_grad"], acc) <= 5e-4             # the un-reduced loss is a plain sum: SUM all-reduce is exact
    for r in range(2):
        assert int(ranks[r]["ft16_skipped"]) == 1 and bool(ranks[r]["ft16_params_unchanged"]), r
