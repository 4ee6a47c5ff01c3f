"""Step closures of the hot path, restated on the HIP ops.

`inference` follows test_real_refinement.py:86-110 of the reference
(deq -> clip -> lin -> apply_rf -> alpha -> hal -> blend -> ref).
"""
import math

import torch

try:
    from . import _ops as K
    from . import tf_utils
except ImportError:
    import _ops as K
    import tf_utils

THRESHOLD = 0.12  # test_real_refinement.py:28


class Inference:
    """Callable equivalent of the reference's `inference(ldr)` tf.function."""

    def __init__(self, deq, lin, hal, ref=None, threshold=THRESHOLD, streams=1):
        self._deq, self._lin, self._hal, self._ref = deq, lin, hal, ref
        self.threshold = threshold
        # streams > 1: the batch is cut into that many slices, each run on its own HIP stream -- images are
        # independent, and one slice's small / low-occupancy kernels overlap with another's large ones
        self._streams = [torch.cuda.Stream() for _ in range(streams)] if streams > 1 else None

    def __call__(self, ldr, return_intermediates=False):
        with torch.no_grad():
            if self._streams is None or return_intermediates or ldr.shape[0] < len(self._streams):
                return self._run(ldr, return_intermediates)
            main = torch.cuda.current_stream()
            parts = torch.chunk(ldr, len(self._streams), dim=0)
            outs = []
            for st, part in zip(self._streams, parts):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    outs.append(self._run(part.contiguous(), False))
            for st in self._streams:
                main.wait_stream(st)
            for o in outs:
                o.record_stream(main)
            return torch.cat(outs, dim=0)

    def _run(self, ldr, return_intermediates):
        with K.range_scope():              # one zeroed slab of range slots per forward and stream (K: "range slots")
            return self._run_scoped(ldr, return_intermediates)

    def _run_scoped(self, ldr, return_intermediates):
        pred_deq = self._deq(ldr, training=False)
        C_pred = K.clip(pred_deq, 0.0, 1.0)
        pred_invcrf = self._lin(C_pred, training=False)
        B_pred = tf_utils.apply_rf(C_pred, pred_invcrf)
        bgr_hal_res = self._hal(B_pred, training=False)
        # alpha = clamp((max_c B - 1 + thr)/thr); A = B + alpha * rgb2bgr(hal)   (:98-105), one kernel
        A_pred = K.alpha_blend(B_pred, bgr_hal_res, self.threshold)
        out = A_pred
        if self._ref is not None:
            # tf.concat([A,B,C],-1) (:108), zero-padded to 12 channels for the MFMA tile
            out = self._ref(K.pack3([A_pred, B_pred, C_pred], 12), training=False)
        if return_intermediates:
            return dict(C_pred=C_pred, invcrf=pred_invcrf, B_pred=B_pred, hal=bgr_hal_res, A_pred=A_pred,
                        hdr=out if self._ref is not None else None)
        return out


# ---------------------------------------------------------------------------
# joint training step (joint_training.py:137-194)
# ---------------------------------------------------------------------------
class GraphedInference:
    """`Inference` captured once into a HIP graph (hipGraph via torch.cuda.CUDAGraph) and replayed: the
    ~100 kernel launches of one forward become ONE graph launch, which removes the launch gaps that
    dominate small-batch latency (the reference's tool runs one image at a time,
    test_real_refinement.py:119-155).  Shapes are static: one graph per input shape, kept in a cache.  The graph holds the
    derived constants of capture time (packed Winograd filters, padded filters, folded BatchNorm): call `reset()` after the
    weights change (checkpoint restore, training step).
    The returned tensor IS the graph's static output buffer: the next call with the same input shape overwrites it in place
    (`.clone()` it to keep a result across calls, or pass `copy_output=True`)."""

    def __init__(self, deq, lin, hal, ref=None, threshold=THRESHOLD, copy_output=False):
        self._copy_output = copy_output
        self._eager = Inference(deq, lin, hal, ref, threshold)
        self._graphs = {}

    def reset(self):
        """drop the captured graphs (weights changed)"""
        self._graphs = {}

    def __call__(self, ldr):
        key = tuple(ldr.shape)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = torch.empty_like(ldr)
            static_in.copy_(ldr)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):             # warm-up off the capture: kernel attributes, caches
                for _ in range(2):
                    self._eager(static_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._eager(static_in)
            entry = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(ldr)
        graph.replay()
        return static_out.clone() if self._copy_output else static_out


class FlatParams:
    """All trainable variables of several models as views of ONE flat fp32 buffer (plus flat grad / Adam
    moment buffers): one Adam kernel and one RCCL all-reduce per step instead of one per variable."""

    ALIGN = 64

    def __init__(self, models):
        self.variables = []
        for m in models:
            self.variables += m.trainable_variables            # joint_training.py:185 order
        # every variable starts on a 64-float (256-byte) boundary: the kernels read filters, biases and
        # BN vectors with 16-byte vector / LDS-DMA loads.  The gaps stay zero (zero gradient -> Adam no-op).
        self.offsets = []
        n = 0
        for v in self.variables:
            self.offsets.append(n)
            n += (v.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        dev = self.variables[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for v, o in zip(self.variables, self.offsets):
                k = v.numel()
                self.flat[o:o + k].copy_(v.detach().reshape(-1))
                v.data = self.flat[o:o + k].view(v.shape)       # the variable now aliases the flat buffer
                v.grad = self.grad[o:o + k].view(v.shape)       # autograd accumulates in place
                v._shdr_accum = True                            # ...and the backward kernels add into it directly (_autograd._acc)
        # Contract of `_shdr_accum` (_autograd._acc): the backward kernels ADD each parameter gradient into its slice of `self.grad`
        # and return None to autograd.  So (1) backward only with a zeroed flat gradient (`zero_grad()` at the top of every step),
        # (2) `torch.autograd.grad(...)` / a plain `.backward()` on these variables returns None for them and still adds into the flat
        # buffer, (3) whoever reads `self.grad` after `backward()` first joins every stream a forward op ran on (`_join_streams`).
        self.numel = n                                          # padded length of the flat buffers
        self.num_params = sum(v.numel() for v in self.variables)

    def zero_grad(self):
        self.grad.zero_()


class KerasAdam:
    """tf.keras.optimizers.Adam(lr): beta 0.9/0.999, epsilon 1e-7,
    theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)   (SURVEY.md section 8c item 10)."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        self.p, self.lr, self.b1, self.b2, self.eps, self.t = params, lr, beta1, beta2, eps, 0

    def step(self, grad_scale=1.0):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        K.adam_step(self.p.flat, self.p.grad, self.p.m, self.p.v, lr_t, self.b1, self.b2, self.eps, grad_scale)
        # the variables alias the flat buffer through `.data` and keep their OWN version counters: tell torch (and with it the
        # per-version caches of the layers) that every one of them has just been rewritten
        for v in self.p.variables:
            torch.autograd.graph.increment_version(v)


def _dp_scalars(step, mask):
    """(global batch size, global sum(loss_mask)) of a data-parallel step: the two batch-coupled scalars of the losses (the
    factor B of the broadcast loss, the crf weight, the batch-global TV mean).  One scalar all-reduce; equal shard sizes."""
    msum = mask.sum()
    if step.pg is not None and step.world > 1:
        import torch.distributed as dist
        dist.all_reduce(msum, group=step.pg)
    return float(mask.numel() * step.world), msum


def _broadcast_lin_loss(l2_term, crf_loss, mask):
    """(l2_term [b,1,1,1] + crf_loss [b,1]) * loss_mask [b,1,1,1] -> [b,1,b,1], the shapes (and therefore the broadcast) of
    joint_training.py:158-160 and train.py:189-191; inputs are the per-sample vectors [b]"""
    b = mask.numel()
    return (l2_term.reshape(b, 1, 1, 1) + crf_loss.reshape(b, 1)) * mask.reshape(b, 1, 1, 1)


class JointTrainStep:
    """The `train_step(ds, invcrf)` closure of joint_training.py:137-194.

    deq, lin and hal are fed ground-truth intermediates (jpeg, ldr, clipped_hdr_t).  The reference keeps `crf_loss` as
    [b,1] and adds it to the [b,1,1,1] L2 term (joint_training.py:158-160), so `loss_lin` and `total_loss` BROADCAST to
    [b,1,b,1] (element [i,0,j,0] = per-sample terms of i + mask_i * crf_j) and tape.gradient differentiates the sum of all
    b*b elements:   B * sum_i (loss_deq_i + 10 * mask_i * l2_i + loss_hal_i)  +  (sum_i mask_i) * sum_j crf_j,   B = batch size.
    That objective is reproduced exactly (`objective()`), including the factor B on every gradient (it matters through
    Adam's epsilon) and the crf gradient a masked sample still receives.
    Data parallel: weights replicated, batch sharded, ONE all_reduce(SUM) of the flat
    gradient over RCCL; BatchNorm statistics are per replica; B and sum(loss_mask) are the GLOBAL ones (a scalar
    all-reduce of sum(loss_mask)), which also keeps the batch-global TV mean exact."""

    LEARNING_RATE = 1e-5   # joint_training.py:20
    THRESHOLD = 0.12       # joint_training.py:140

    def __init__(self, deq, lin, hal, vgg, vgg2=None, lr=None, process_group=None, world_size=1, multi_stream=True, bucketed=True):
        """bucketed (data parallel only): the flat gradient is reduced in FOUR collectives instead of one, each launched as async work
        as soon as its slice is complete and overlapped with the backward ops that remain (SURVEY.md section 8e "bucket per net, hal
        first"): hal's decoder half (from a tape mark at the bottleneck), hal's encoder half, lin, deq.  Same SUM per element as the
        one blocking all_reduce (bucketed=False); no rank-local code path issues a collective."""
        self._deq, self._lin, self._hal, self._vgg, self._vgg2 = deq, lin, hal, vgg, vgg2 or vgg
        self.bucketed = bucketed
        self.params = FlatParams([deq, lin, hal])
        self.optimizer = KerasAdam(self.params, self.LEARNING_RATE if lr is None else lr)
        self.pg, self.world = process_group, world_size
        self.multi_stream = multi_stream
        self._streams = tuple(torch.cuda.Stream() for _ in range(3)) if multi_stream else None

    def losses(self, ds, invcrf):
        ldr, jpeg_img_float, clipped_hdr_t, hdr_t, loss_mask = ds
        mask = loss_mask.reshape(-1)
        thr = self.THRESHOLD
        # The three nets are fed ground-truth intermediates, so their forward AND backward passes are independent
        # (SURVEY.md section 3.2): each runs on its own HIP stream (torch replays an op's backward on its forward stream), which
        # lets the small Dequantization / Linearization kernels fill the CUs the big Hallucination kernels leave idle.
        main = torch.cuda.current_stream()
        streams = self._streams if self.multi_stream else (main, main, main)
        for st in streams:
            st.wait_stream(main)

        with torch.cuda.stream(streams[0]), K.range_scope():   # Dequantization (:150-153)
            pred_deq = self._deq(jpeg_img_float, training=True)
            C_pred = K.clip(pred_deq, 0.0, 1.0)
            loss_deq = K.diff_loss(C_pred, ldr, 0) * mask

        with torch.cuda.stream(streams[1]), K.range_scope():   # Linearization (:156-160)
            pred_invcrf = self._lin(ldr, training=True)
            B_pred = tf_utils.apply_rf(ldr, pred_invcrf)
            crf_loss = K.diff_loss(pred_invcrf, invcrf, 0)                         # [b] here, [b,1] in the reference
            l2_lin = K.diff_loss(B_pred, clipped_hdr_t, 0)
            loss_lin = _broadcast_lin_loss(10.0 * l2_lin, crf_loss, mask)            # [b,1,b,1]

        with torch.cuda.stream(streams[2]), K.range_scope():   # Hallucination (:163-182)
            alpha = K.alpha_mask(clipped_hdr_t, thr)
            bgr_pred_hal = self._hal(clipped_hdr_t, training=True)
            A_pred = K.blend_const(clipped_hdr_t, alpha, bgr_pred_hal, thr)      # clipped + alpha * bgr2rgb(hal)
            y_final_gamma, y_l1, y_tv = K.fork(K.logc(A_pred), 3)                  # perceptual, L1 and TV terms
            with torch.no_grad():
                hdr_t_gamma = K.logc(hdr_t)
                target_feats = self._vgg2(hdr_t_gamma)
            feats = self._vgg(y_final_gamma)
            perceptual_loss = sum(K.diff_loss(fa, fb, 1) for fa, fb in zip(feats, target_feats))
            l1loss_hal = K.diff_loss(y_l1, hdr_t_gamma, 1)
            tv_loss = K.tv_loss(y_tv)                                              # batch-global scalar [1]
            b_glob, msum = _dp_scalars(self, mask)
            # exact sharding of tv_loss * loss_mask: d/dtheta sums to (sum_all mask / G) * sum_r grad tv_r
            tv_w = mask if self.world == 1 else torch.ones_like(mask) * (msum / b_glob)
            loss_hal = (l1loss_hal + 0.001 * perceptual_loss) * mask + 0.1 * tv_loss * tv_w

        for st in streams:
            main.wait_stream(st)
        b = mask.numel()
        total_loss = (loss_deq + loss_hal).reshape(b, 1, 1, 1) + loss_lin          # [b,1,b,1] (:183)
        # the scalar tape.gradient differentiates (class docstring), with the GLOBAL batch size and mask sum under DP
        terms = dict(deq=b_glob * loss_deq.sum(), lin=b_glob * (10.0 * l2_lin * mask).sum() + msum * crf_loss.sum(),
                     hal=b_glob * loss_hal.sum())                   # the three nets share no variable: three independent tapes
        objective = terms["deq"] + terms["lin"] + terms["hal"]
        return dict(total=total_loss, objective=objective, objective_terms=terms, loss_deq=loss_deq, loss_lin=loss_lin, loss_hal=loss_hal,
                    crf_loss=crf_loss, C_pred=C_pred, B_pred=B_pred, A_pred=A_pred, alpha=alpha)

    def _join_streams(self):
        """The backward kernels add parameter gradients straight into the flat gradient buffer on the stream of their forward op
        (FlatParams: `_shdr_accum`) and hand autograd no tensor for them, so autograd's own producer -> AccumulateGrad stream
        hand-off does not cover those writes: the collective and the optimizer wait for the three side streams explicitly."""
        if self._streams is not None:
            main = torch.cuda.current_stream()
            for st in self._streams:
                main.wait_stream(st)

    def _net_slices(self):
        """{net: (begin, end)} of the flat buffers, and the offset inside hal at which the variables created after the encoder begin"""
        if getattr(self, "_slices", None) is None:
            bounds, i = {}, 0
            offs = self.params.offsets + [self.params.numel]
            for name, m in (("deq", self._deq), ("lin", self._lin), ("hal", self._hal)):
                n = len(m.trainable_variables)
                bounds[name] = (offs[i], offs[i + n])
                if name == "hal":
                    k = next(j for j, v in enumerate(m.trainable_variables) if v is self._hal.conv1.kernel)
                    self._hal_split = offs[i + k]
                i += n
            self._slices = bounds
        return self._slices

    def gradient_buckets(self):
        """[(begin, end)] of the flat gradient in launch order: hal decoder half, hal encoder half, lin, deq -- a partition of it"""
        sl = self._net_slices()
        return [(self._hal_split, sl["hal"][1]), (sl["hal"][0], self._hal_split), sl["lin"], sl["deq"]]

    def _decoder_grads_done(self):
        if self._on_decoder_grads is not None:
            self._on_decoder_grads()

    def _bucketed_backward(self, out):
        """backward + gradient collectives, overlapped: hal first (the largest bucket and the longest backward), its decoder half
        reduced while its encoder half is still being computed; then lin, then deq.  Each collective is enqueued behind the stream
        its slice was written on and nothing else."""
        import torch.distributed as dist
        sl = self._net_slices()
        streams = self._streams if self._streams is not None else (torch.cuda.current_stream(),) * 3
        works = []

        def reduce(begin, end, stream):
            with torch.cuda.stream(stream):
                works.append(dist.all_reduce(self.params.grad[begin:end], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._on_decoder_grads = lambda: reduce(self._hal_split, sl["hal"][1], torch.cuda.current_stream())
        try:
            terms = out["objective_terms"]
            terms["hal"].backward()
            reduce(sl["hal"][0], self._hal_split, streams[2])
            terms["lin"].backward()
            reduce(*sl["lin"], streams[1])
            terms["deq"].backward()
            reduce(*sl["deq"], streams[0])
        finally:
            self._on_decoder_grads = None
        self._join_streams()
        for w in works:
            w.wait()

    def __call__(self, ds, invcrf, apply=True):
        self.params.zero_grad()
        dp = self.pg is not None and self.world > 1
        if dp and self.bucketed:
            self._net_slices()
            self._on_decoder_grads = None
            self._hal._decoder_grads_done = self._decoder_grads_done      # the tape mark is recorded in the forward pass
        out = self.losses(ds, invcrf)
        self._hal._decoder_grads_done = None
        if dp and self.bucketed:
            with K.range_scope():                         # the backward pass's range slots (one zeroed slab per stream)
                self._bucketed_backward(out)
        else:
            with K.range_scope():
                out["objective"].backward()               # == total_loss.sum() on one GPU: the sum over the [b,1,b,1] tensor
            self._join_streams()
            if dp:
                import torch.distributed as dist
                dist.all_reduce(self.params.grad, op=dist.ReduceOp.SUM, group=self.pg)   # the ONE gradient collective
        if apply:
            self.optimizer.step()
        return out


class TrainStep:
    """The per-network `*_train_step(ds)` closures of train.py:164-244 (`--deq` / `--lin` / `--hal`): one network, its own
    loss, Adam 1e-4 on its own variables (train.py:19).  Same kernels and tape as the joint step; the differences are the
    Linearization loss weights (L2 + 0.1 * crf here, 10 * L2 + crf in joint_training.py:160) and the learning rate.

        step = TrainStep("deq", deq_net);            pred,               = step((ldr, jpeg_img_float, loss_mask))
        step = TrainStep("lin", lin_net);            b_pred, crf_mean    = step((ldr, clipped_hdr_t, loss_mask, invcrf))
        step = TrainStep("hal", hal_net, vgg, vgg2); pred, y_final, alpha = step((hdr_t, clipped_hdr_t, loss_mask))
    """

    LEARNING_RATE = 1e-4   # train.py:19
    THRESHOLD = 0.12       # train.py:209

    def __init__(self, which, net, vgg=None, vgg2=None, lr=None, process_group=None, world_size=1):
        if which not in ("deq", "lin", "hal"):
            raise ValueError("TrainStep: which must be 'deq', 'lin' or 'hal'")
        if which == "hal" and vgg is None:
            raise ValueError("TrainStep('hal'): the perceptual loss needs a Vgg16")
        self.which, self.net, self._vgg, self._vgg2 = which, net, vgg, vgg2 or vgg
        self.params = FlatParams([net])
        self.optimizer = KerasAdam(self.params, self.LEARNING_RATE if lr is None else lr)
        self.pg, self.world = process_group, world_size
        self.last_loss = None
        self._objective = None

    def forward(self, ds):
        """(per-sample loss [b], the list the reference's step returns)"""
        if self.which == "deq":                      # train.py:165-177
            ldr, jpeg_img_float, loss_mask = ds
            pred = K.clip(self.net(jpeg_img_float, training=True), 0.0, 1.0)
            return K.diff_loss(pred, ldr, 0) * loss_mask.reshape(-1), [pred]
        if self.which == "lin":                      # train.py:183-197
            ldr, clipped_hdr_t, loss_mask, invcrf = ds
            pred_invcrf = self.net(ldr, training=True)
            pred_lin_ldr = tf_utils.apply_rf(ldr, pred_invcrf)
            crf_loss = K.diff_loss(pred_invcrf, invcrf, 0)
            l2 = K.diff_loss(pred_lin_ldr, clipped_hdr_t, 0)
            mask = loss_mask.reshape(-1)
            # train.py:189-191: `loss [b,1,1,1] + 0.1 * crf_loss [b,1]` broadcasts to [b,1,b,1] like the joint step's loss_lin;
            # the differentiated sum is  B * sum_i mask_i * l2_i + 0.1 * (sum_i mask_i) * sum_j crf_j  (global B, mask sum under DP)
            b_glob, msum = _dp_scalars(self, mask)
            self._objective = b_glob * (l2 * mask).sum() + 0.1 * msum * crf_loss.sum()
            return _broadcast_lin_loss(l2, 0.1 * crf_loss, mask), [pred_lin_ldr, crf_loss.detach().mean()]
        hdr_t, clipped_hdr_t, loss_mask = ds         # train.py:203-244
        mask = loss_mask.reshape(-1)
        alpha = K.alpha_mask(clipped_hdr_t, self.THRESHOLD)
        bgr_pred = self.net(clipped_hdr_t, training=True)
        y_final = K.blend_const(clipped_hdr_t, alpha, bgr_pred, self.THRESHOLD)        # clipped + alpha * bgr2rgb(pred)
        y_final_gamma = K.logc(y_final)
        with torch.no_grad():
            hdr_t_gamma = K.logc(hdr_t)
            target_feats = self._vgg2(hdr_t_gamma)
        feats = self._vgg(y_final_gamma)
        perceptual_loss = sum(K.diff_loss(fa, fb, 1) for fa, fb in zip(feats, target_feats))
        loss = K.diff_loss(y_final_gamma, hdr_t_gamma, 1)
        tv_loss = K.tv_loss(y_final_gamma)
        tv_w = mask
        if self.world > 1:                            # exact sharding of the batch-global TV mean, as in JointTrainStep
            b_glob, msum = _dp_scalars(self, mask)
            tv_w = torch.ones_like(mask) * (msum / b_glob)
        hal_loss = (loss + 0.001 * perceptual_loss) * mask + 0.1 * tv_loss * tv_w
        with torch.no_grad():
            pred_rgb = tf_utils.bgr2rgb(bgr_pred.detach())
        return hal_loss, [pred_rgb, y_final, alpha]

    def __call__(self, ds, apply=True):
        self.params.zero_grad()
        self._objective = None
        with K.range_scope():
            loss, outputs = self.forward(ds)
            # tape.gradient of a non-scalar loss = gradient of its sum (for `lin` the sum over the broadcast [b,1,b,1] tensor)
            (loss.sum() if self._objective is None else self._objective).backward()
        if self.pg is not None and self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.params.grad, op=dist.ReduceOp.SUM, group=self.pg)
        if apply:
            self.optimizer.step()
        self.last_loss = loss.detach()
        return outputs


class FinetuneStep:
    """The chained `train_step(ldr, hdr)` of finetune_real_dataset.py:144-183 (with the reference's
    `_hal(pred, ...)` typo read as `_hal(B_pred, ...)`): deq -> clip -> lin -> apply_rf
    -> alpha -> hal -> blend -> ref -> mean-normalise -> log-compress -> |.|, the UN-reduced loss is
    differentiated (= gradient of its sum), Adam 1e-5 on the variables of all four nets."""

    LEARNING_RATE = 1e-5   # finetune_real_dataset.py:24
    THRESHOLD = 0.12       # finetune_real_dataset.py:26

    def __init__(self, deq, lin, hal, ref, lr=None, process_group=None, world_size=1, precision="fp32",
                 loss_scale=1.0):
        """precision: "fp32" (parity path); "fp16" = BASELINE configs[4], the NATIVE fp16 conv path: fp16 feature maps in HBM
        for every conv forward / dgrad / wgrad, BatchNorm, pooling and resize pass, fp32 master weights, fp32 parameter gradients
        and fp32 accumulation (_ops.PRECISION); "fp16op" / "bf16" = the round-1 operand-rounding modes (fp32 tensors in HBM).  loss_scale: static scale applied to the seed gradient
        and divided out of the flat gradient before the collective.  The loss here is a SUM over pixels, so output
        gradients are O(1)..O(1e3) and need no up-scaling; 256 overflows fp16 at 512x512 tiles (measured)."""
        self._deq, self._lin, self._hal, self._ref = deq, lin, hal, ref
        self.params = FlatParams([deq, lin, hal, ref])
        self.optimizer = KerasAdam(self.params, self.LEARNING_RATE if lr is None else lr)
        self.pg, self.world = process_group, world_size
        self.precision, self.loss_scale = precision, float(loss_scale)
        self.skipped_steps = 0

    def forward(self, ldr, hdr):
        pred_deq = self._deq(ldr, training=True)
        # the chained predictions fan out (C: lin, apply_rf, ref; B: hal, blend, ref; A: ref): K.fork sums their gradients in libshdr
        C_pred, c_rf, c_ref = K.fork(K.clip(pred_deq, 0.0, 1.0), 3)
        pred_invcrf = self._lin(C_pred, training=True)
        B_pred, b_blend, b_ref = K.fork(tf_utils.apply_rf(c_rf, pred_invcrf), 3)
        bgr_hal_res = self._hal(B_pred, training=True)
        A_pred = K.alpha_blend(b_blend, bgr_hal_res, self.THRESHOLD)     # alpha is a function of B_pred here
        with torch.no_grad():
            hdr_gamma = K.logc(hdr)
        if self.precision == "fp16":                 # fp16 feature maps: [A, B, C, 0...] on two 16-byte channel groups
            refinement_output = self._ref(K.pack3([A_pred, b_ref, c_ref], 16, K.HALF), training=True)
        else:
            refinement_output = self._ref(K.pack3([A_pred, b_ref, c_ref], 12), training=True)
        refinement_output = K.mean_norm(refinement_output, 1e-6, 0.5)
        refinement_output_gamma = K.logc(refinement_output)
        n_per = refinement_output_gamma[0].numel()
        loss_sum = K.diff_loss(refinement_output_gamma, hdr_gamma, 1) * float(n_per)   # per-sample SUM of |.|
        return dict(loss_sum=loss_sum, C_pred=C_pred, B_pred=B_pred, A_pred=A_pred, refinement_output=refinement_output)

    def __call__(self, ldr, hdr, apply=True):
        self.params.zero_grad()
        with K.precision(self.precision), K.range_scope():
            out = self.forward(ldr, hdr)
            (out["loss_sum"].sum() * self.loss_scale).backward()
        if self.loss_scale != 1.0:
            self.params.grad.mul_(1.0 / self.loss_scale)
        if self.pg is not None and self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.params.grad, op=dist.ReduceOp.SUM, group=self.pg)
        if self.precision in ("fp16", "fp16op") and not bool(torch.isfinite(self.params.grad.sum())):
            # an fp16 operand overflowed (|value| > 65504).  Checked after the collective, so every rank sees it: drop
            # the step like a dynamic loss scaler would.  The batch-summed loss makes output gradients GROW with the
            # tile size, so the remedy is loss_scale < 1.
            self.skipped_steps += 1
            return out
        if apply:
            self.optimizer.step()
        return out
