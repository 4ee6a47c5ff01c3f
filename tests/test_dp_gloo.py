"""World-size-2 gloo test (CPU) of the data-parallel design of the joint step:
weights replicated, batch sharded, ONE all_reduce(SUM) of the flat gradient -- the batch-summed loss
makes SUM (not mean) the exact reduction -- and the scalar all-reduce that keeps the batch-global TV
term exact.  Gradients come from the float64 reference so that the test runs without a GPU; the
flat-buffer plumbing under test (pipeline.FlatParams) is the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import importlib
    import torch_ref as R
    from oracle import nets
    shdr = importlib.import_module("singlehdr-tf2_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    rng = np.random.default_rng(5)
    b, s = 4, 16
    x = np.round(rng.random((b, s, s, 3)) * 255) / 255
    tgt = np.round(rng.random((b, s, s, 3)) * 255) / 255
    mask = np.array([1.0, 0.0, 1.0, 1.0])
    P = nets.init_params(nets.deq_spec(), 3)

    def loss(tp, xs, ts, ms, tv_w):
        y = torch.clamp(R.deq_forward(tp, R.T(xs)), 0, 1)
        per = ((y - R.T(ts)) ** 2).mean(dim=(1, 2, 3)) * R.T(ms)
        return per.sum() + 0.1 * R.tv_loss(R.logc(y)) * tv_w     # batch-global TV times sum of the shard's weights

    # product plumbing: a model on CPU tensors, flattened exactly as JointTrainStep does
    m = shdr.dequantization_net.model(device=torch.device("cpu")).load_numpy(P)
    flat = shdr.pipeline.FlatParams([m])
    names = [n for n, _, tr in m.named_weights() if tr]
    sl = slice(rank * b // world, (rank + 1) * b // world)
    msum = torch.tensor(mask[sl].sum())
    dist.all_reduce(msum)                                           # scalar all-reduce of sum(loss_mask)
    tp = R.params_to_torch(P)
    loss(tp, x[sl], tgt[sl], mask[sl], float(msum) / world).backward()
    flat.zero_grad()
    with torch.no_grad():
        for v, n in zip(flat.variables, names):
            v.grad.copy_(tp[n].grad.float())
    dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM)                # the ONE gradient collective
    if rank == 0:
        tp1 = R.params_to_torch(P)
        # single-process semantics of the reference: tv is the mean over the WHOLE batch, multiplied by each mask
        y = torch.clamp(R.deq_forward(tp1, R.T(x)), 0, 1)
        full = (((y - R.T(tgt)) ** 2).mean(dim=(1, 2, 3)) * R.T(mask)).sum()
        # mean over the whole batch == mean of the equal-size shard means
        tv = sum(R.tv_loss(R.logc(y[i * b // world:(i + 1) * b // world])) for i in range(world)) / world
        (full + 0.1 * tv * mask.sum()).backward()
        err = max(float((v.grad - tp1[n].grad.float()).abs().max()) for v, n in zip(flat.variables, names))
        scale = max(float(tp1[n].grad.abs().max()) for n in names)
        q.put((err, scale, int(flat.num_params)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradient_sum_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    err, scale, n = q.get(timeout=600)
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    assert n == 1999779
    assert err <= 1e-5 * scale, (err, scale)
