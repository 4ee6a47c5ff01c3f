"""One rank of the data-parallel GPU test (tests/test_gpu_dp.py): started as a fresh child process, so the process group is
up before this process touches the GPU.  Runs the PRODUCT's data-parallel branches -- `JointTrainStep`, `TrainStep("hal")`
and `FinetuneStep` with `process_group=WORLD, world_size=2` -- on its shard of a seeded batch and writes the all-reduced flat
gradients (and the parameters after the Adam update) to an .npz file.

    python dp_worker.py <rank> <world> <port> <out.npz> [backend]
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def make_data(b, s, seed=17):
    """global batch of the joint step (SURVEY.md section 8d config 4 input law) with one masked sample"""
    rng = np.random.default_rng(seed)

    def q(shape):
        return np.round(rng.random(shape) * 255.0) / 255.0
    clipped = q((b, s, s, 3))
    clipped[0, :10, :10] = 1.0
    clipped[b - 1, 5:20, 7:30] = 1.0
    hdr_t = clipped * np.where(clipped >= 1.0, 1 + 3 * rng.random((b, s, s, 3)), 1.0)
    inv = np.cumsum(rng.random((b, 1024)), axis=1)
    inv = (inv - inv[:, :1]) / (inv[:, -1:] - inv[:, :1])
    mask = np.ones((b, 1, 1, 1))
    mask[1] = 0.0                                               # a zero in loss_mask (extreme-exposure sample)
    ldr, jpeg = q((b, s, s, 3)), q((b, s, s, 3))
    hdr = rng.random((b, s, s, 3)) * 1.5
    hdr = hdr / (1e-6 + hdr.mean(axis=(1, 2, 3), keepdims=True)) * 0.5
    return dict(ldr=ldr, jpeg=jpeg, clipped=clipped, hdr_t=hdr_t, mask=mask, inv=inv, hdr=hdr)


def build_models(shdr, nets_mod, with_ref=False):
    """identical weights on every rank: the oracle's seeded initialiser"""
    mods = dict(deq="dequantization_net", lin="linearization_net", hal="hallucination_net", ref="refinement_net")
    names = ("deq", "lin", "hal") + (("ref",) if with_ref else ())
    P = {k: nets_mod.init_params(getattr(nets_mod, k + "_spec")(), 300 + i) for i, k in enumerate(names)}
    return {k: getattr(shdr, mods[k]).model().load_numpy(P[k]) for k in names}


def build_vgg(shdr, nets_mod):
    V = nets_mod.init_params(nets_mod.vgg_spec(), 310)
    dd = {n: [V[n + ".kernel"], V[n + ".bias"]] for n in ("conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3")}
    return shdr.vgg16.Vgg16(data_dict=dd)


def net_grads(step, models):
    import torch
    return {k: torch.cat([t.grad.reshape(-1) for t in m.trainable_variables]).detach().cpu().numpy() for k, m in models.items()}


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend, rank=rank, world_size=world)       # before anything touches the GPU
    torch.cuda.set_device(0)                                             # both ranks share the one visible card
    from oracle import nets
    shdr = importlib.import_module("singlehdr-tf2_amd")
    P = shdr.pipeline

    def dev(x):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    b, s = 4, 64
    d = make_data(b, s)
    sl = slice(rank * b // world, (rank + 1) * b // world)
    res = {}

    # ---- joint step (joint_training.py:137-194) -----------------------------------------------------------------------------
    models = build_models(shdr, nets)
    step = P.JointTrainStep(models["deq"], models["lin"], models["hal"], build_vgg(shdr, nets), lr=1e-3,
                            process_group=dist.group.WORLD, world_size=world)
    ds = tuple(dev(d[k][sl]) for k in ("ldr", "jpeg", "clipped", "hdr_t", "mask"))
    outp = step(ds, dev(d["inv"][sl]), apply=False)
    for k, g in net_grads(step, models).items():
        res["joint_grad_" + k] = g
    res["joint_objective"] = np.asarray(float(outp["objective"]))
    step.optimizer.step()
    res["joint_params_after"] = step.params.flat.detach().cpu().numpy()
    # the default above is the BUCKETED, overlapped reduction (four async collectives launched from inside the backward pass); here the
    # one blocking all_reduce on fresh replicas of the same weights, and the two decompositions of the collective on one buffer
    models_b = build_models(shdr, nets)
    blocking = P.JointTrainStep(models_b["deq"], models_b["lin"], models_b["hal"], build_vgg(shdr, nets), lr=1e-3,
                                process_group=dist.group.WORLD, world_size=world, bucketed=False)
    blocking(ds, dev(d["inv"][sl]), apply=False)
    for k, g in net_grads(blocking, models_b).items():
        res["joint_blocking_grad_" + k] = g
    buckets = step.gradient_buckets()
    res["buckets"] = np.asarray(buckets)
    res["flat_numel"] = np.asarray(step.params.numel)
    gen = torch.Generator(device="cuda").manual_seed(100 + rank)
    whole = torch.randn(step.params.numel, device="cuda", generator=gen)
    pieces = whole.clone()
    dist.all_reduce(whole, op=dist.ReduceOp.SUM)
    side = [torch.cuda.Stream() for _ in buckets]
    works = []
    for st, (b0, b1) in zip(side, buckets):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            works.append(dist.all_reduce(pieces[b0:b1], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    for st in side:
        torch.cuda.current_stream().wait_stream(st)
    res["bucketed_equals_blocking_collective"] = np.asarray(bool(torch.equal(whole, pieces)))

    # ---- per-network step of train.py:203-244 (batch-global TV term) ---------------------------------------------------------
    models = build_models(shdr, nets)
    hstep = P.TrainStep("hal", models["hal"], build_vgg(shdr, nets), process_group=dist.group.WORLD, world_size=world)
    hstep((dev(d["hdr_t"][sl]), dev(d["clipped"][sl]), dev(d["mask"][sl])), apply=False)
    res["hal_grad"] = hstep.params.grad.detach().cpu().numpy()
    lstep = P.TrainStep("lin", models["lin"], process_group=dist.group.WORLD, world_size=world)
    lstep((dev(d["ldr"][sl]), dev(d["clipped"][sl]), dev(d["mask"][sl]), dev(d["inv"][sl])), apply=False)
    res["lin_grad"] = lstep.params.grad.detach().cpu().numpy()

    # ---- chained fine-tuning step (finetune_real_dataset.py:144-183), fp32 and the fp16 mode's skip-after-collective ---------
    models = build_models(shdr, nets, with_ref=True)
    fstep = P.FinetuneStep(models["deq"], models["lin"], models["hal"], models["ref"], process_group=dist.group.WORLD,
                           world_size=world)
    fstep(dev(d["ldr"][sl]), dev(d["hdr"][sl]), apply=False)
    res["ft_grad"] = fstep.params.grad.detach().cpu().numpy()
    # ONE rank overflows (an absurd loss scale makes its seed gradient infinite -- arithmetic only, no index is derived from
    # it): the non-finite gradient must be seen by BOTH ranks AFTER the all-reduce, and both must drop the step
    f16 = P.FinetuneStep(models["deq"], models["lin"], models["hal"], models["ref"], process_group=dist.group.WORLD,
                         world_size=world, precision="fp16", loss_scale=0.25 if rank == 0 else 1e38)
    before = f16.params.flat.detach().clone()
    f16(dev(d["ldr"][sl]), dev(d["hdr"][sl]))
    res["ft16_skipped"] = np.asarray(f16.skipped_steps)
    res["ft16_params_unchanged"] = np.asarray(bool(torch.equal(before, f16.params.flat)))

    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
