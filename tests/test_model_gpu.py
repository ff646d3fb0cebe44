"""GPU parity of the whole step: the same tiny student-teacher (and teacher) step run (a) on the MI355X
through the HIP ops and (b) on CPU tensors with the native ops routed to the oracle, same weights, same
inputs, same injected noise, deterministic sampling (batch sizes large enough that the fg/bg samplers
take everything).  Tolerance 1e-3 relative (north_star) on every loss; gradients compared element-wise (relative L2
distance per parameter tensor <= 5e-3)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(name):
    from tests.tiny_model import build_tiny

    return build_tiny(name)


def _run(model, e_vocab, e_seen, images, targets, device, ctx):
    model = model.to(device)
    model.set_class_embeddings(e_seen.to(device))
    if hasattr(model, "set_caption_vocab"):
        model.set_caption_vocab(e_vocab.to(device))
    tg = [t.to(device) for t in targets]
    g = torch.Generator().manual_seed(3)
    eps = torch.randn(1, 4096, 2, 14, 14, generator=g)  # same noise pool on both sides
    is_student = hasattr(model, "roi_heads_student")
    for p in model.parameters():
        p.grad = None
    with ctx:
        losses = model(images.to(device), tg, eps=eps) if is_student else model(images.to(device), tg)
        sum(losses.values()).backward()
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return {k: float(v.detach()) for k, v in losses.items()}, grads


@pytest.mark.parametrize("name", ["student_teacher_mask_rcnn_uncertainty", "zeroshot_mask"])
def test_step_matches_oracle_backed_cpu_step(name):
    import contextlib

    from tests.oracle_backend import oracle_ops

    model, e_vocab, e_seen, images, targets = _build(name)
    import copy

    cpu_model = copy.deepcopy(model)
    if hasattr(model, "iter"):
        cpu_model.iter = model.iter
    l_gpu, g_gpu = _run(model, e_vocab, e_seen, images, targets, "cuda", contextlib.nullcontext())
    l_cpu, g_cpu = _run(cpu_model, e_vocab, e_seen, images, targets, "cpu", oracle_ops())
    assert set(l_gpu) == set(l_cpu)
    for k in l_cpu:
        assert abs(l_gpu[k] - l_cpu[k]) <= 1e-3 * max(abs(l_cpu[k]), 1e-3), (k, l_gpu[k], l_cpu[k])
    # gradients element-wise: relative L2 distance per parameter tensor (a permuted or sign-flipped gradient has the
    # right norm and a distance of O(1)).  5e-3: a ReLU pre-activation within rounding of zero may flip its gate
    # between the two arithmetics, which moves isolated entries.
    checked = 0
    for n, v in g_cpu.items():
        nv = v.norm().item()
        if nv > 1e-6 and n in g_gpu:
            d = (g_gpu[n] - v).norm().item()
            assert d <= 5e-3 * nv + 1e-7, (n, d, nv)
            checked += 1
    assert checked >= 10


def test_pipelined_trainer_matches_plain_steps():
    """engine.trainer.PipelinedTrainer (frozen half of step k+1 issued from a worker thread on a side stream while the
    student half of step k runs) produces the losses of plain sequential train_step, step by step: the frozen half
    does not depend on the student's weights, and neither half draws from the other's random stream."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer

    model, e_vocab, e_seen, images, targets = _build("student_teacher_mask_rcnn_uncertainty")
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/student_teacher_mask_rcnn_uncertainty.yaml"))
    cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4])
    cfg.freeze()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]

    def run(kind):
        m = copy.deepcopy(model).cuda()
        m.iter = model.iter
        m.set_class_embeddings(e_seen.cuda())
        m.set_caption_vocab(e_vocab.cuda())
        m.train()
        opt = solver.make_optimizer(cfg, m)
        red = comm.BucketedGradReducer(m)
        pipe = trainer.PipelinedTrainer(m, opt, red, threaded=(kind == "threaded"))
        pipe.enabled = kind != "plain"
        out = []
        for i in range(4):
            torch.manual_seed(100 + i)
            out.append({k: float(v) for k, v in pipe.step(images, tg, (images, tg)).items()})
        pipe.drain()
        red.remove()
        return out

    plain = run("plain")
    for kind in ("serial_side_stream", "threaded"):
        got = run(kind)
        for a, b in zip(got, plain):
            for k in b:
                # the head GEMMs run on the split GEMM (fixed-order K slices): the step repeats to fp32 round-off
                assert abs(a[k] - b[k]) <= 1e-6 * max(abs(b[k]), 1e-3), (kind, k, a[k], b[k])


def test_pipelined_trainer_with_accumulation_and_clipping_matches_plain_steps():
    """SOLVER.GRADIENT_ACCUMULATION_STEPS 2 + SOLVER.CLIP_GRAD_NORM_AT through the two-stream trainer (``StepPolicy``): the
    optimizer steps every second iteration on the accumulated, clipped gradients -- losses step by step and the weights after
    four iterations (two optimizer steps) equal those of the plain sequential loop."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer

    model, e_vocab, e_seen, images, targets = _build("student_teacher_mask_rcnn_uncertainty")
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/student_teacher_mask_rcnn_uncertainty.yaml"))
    cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4, "SOLVER.GRADIENT_ACCUMULATION_STEPS", 2, "SOLVER.CLIP_GRAD_NORM_AT", 1.0])
    cfg.freeze()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]

    def run(pipelined):
        m = copy.deepcopy(model).cuda()
        m.iter = model.iter
        m.set_class_embeddings(e_seen.cuda())
        m.set_caption_vocab(e_vocab.cuda())
        m.train()
        opt = solver.make_optimizer(cfg, m)
        sched = solver.make_lr_scheduler(cfg, opt)
        red = comm.BucketedGradReducer(m)
        pipe = trainer.PipelinedTrainer(m, opt, red, sched, policy=trainer.StepPolicy.from_cfg(cfg))
        pipe.enabled = pipelined
        m.prepare_model()   # the teacher -> student copy of iteration 0 happens here, not inside the first timed comparison
        out, stepped = [], []
        for i in range(4):
            torch.manual_seed(100 + i)
            w = m.roi_heads_student["box"].predictor.bbox_pred.weight
            before = w.detach().clone()
            out.append({k: float(v) for k, v in pipe.step(images, tg, (images, tg)).items()})
            stepped.append(not torch.equal(before, w.detach()))
        pipe.drain()
        red.remove()
        return out, stepped, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, opt.param_groups[0]["lr"]

    plain, stepped_p, w_plain, lr_plain = run(False)
    got, stepped_g, w_got, lr_got = run(True)
    assert stepped_p == stepped_g == [False, True, False, True]   # the weights move on every second iteration only
    assert lr_plain == lr_got
    for a, b in zip(got, plain):
        for k in b:
            assert abs(a[k] - b[k]) <= 1e-6 * max(abs(b[k]), 1e-3), (k, a[k], b[k])
    for n in w_plain:
        assert torch.allclose(w_got[n], w_plain[n], rtol=1e-5, atol=1e-7), n


def test_pipelined_trainer_runs_the_teacher_steps_frozen_trunk_prefix_ahead():
    """Teacher training (zeroshot_mask.yaml, FREEZE_CONV_BODY_AT 2): stem + layer1 of batch k+1 run on the side stream beside
    the backward of batch k (``GeneralizedRCNN.forward_frozen`` / ``forward_student``).  The split chain hands on what the
    un-split one does, so the losses are those of plain ``train_step`` -- step by step, weights updated in between."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer

    model, e_vocab, e_seen, images, targets = _build("zeroshot_mask")
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/zeroshot_mask.yaml"))
    cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4])
    cfg.freeze()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]
    batches = [(images, tg), (images.flip(-1).contiguous(), tg), (images * 0.5, tg), (images, tg)]

    def run(kind):
        m = copy.deepcopy(model).cuda()
        m.set_class_embeddings(e_seen.cuda())
        m.train()
        opt = solver.make_optimizer(cfg, m)
        red = comm.BucketedGradReducer(m)
        pipe = trainer.PipelinedTrainer(m, opt, red, threaded=(kind == "threaded"))
        assert pipe.enabled
        pipe.enabled = kind != "plain"
        if kind != "plain":
            fz = m.forward_frozen(*batches[0])
            assert fz["prefix"] is not None and fz["prefix"][2] == 3  # the three blocks of layer1
        out = []
        for i, (im, t) in enumerate(batches):
            torch.manual_seed(100 + i)
            nxt = batches[i + 1] if i + 1 < len(batches) else None
            out.append({k: float(v) for k, v in pipe.step(im, t, nxt).items()})
        pipe.drain()
        red.remove()
        return out

    plain = run("plain")
    for kind in ("serial_side_stream", "threaded"):
        got = run(kind)
        for i, (a, b) in enumerate(zip(got, plain)):
            assert set(a) == set(b)
            for k in b:
                # Step 0 (same weights on both sides) is the statement about the split: equal to fp32 round-off.  Later steps
                # compare trajectories of two DIFFERENT launch sequences (the prefix product runs as its own launches), whose
                # round-off differs; a stale or foreign prefix would be an O(1) difference.  (Two runs of the SAME sequence
                # are bit-identical since round 6: test_training_steps_are_bit_reproducible.)
                tol = 1e-6 if i == 0 else 1e-3
                assert abs(a[k] - b[k]) <= tol * max(abs(b[k]), 1e-3), (kind, i, k, a[k], b[k])


@pytest.mark.parametrize("name", ["zeroshot_mask", "student_teacher_mask_rcnn_uncertainty"])
def test_training_steps_are_bit_reproducible(name):
    """Two runs of three optimisation steps (forward, backward, bucketed reduce, fused SGD) from the same weights, seeds and
    batches give IDENTICAL losses, gradients and parameters -- bit for bit, both shipped configurations.  The reference cannot
    (its RoIAlign backward adds with fp32 atomics, ROIAlign_cuda.cu:246-249); here no kernel of the step accumulates in an
    order that depends on scheduling: the RoIAlign backward owns its planes, weight-gradient slabs and the fp32 head GEMM's
    K slices (round 6: they were fp32 atomics until then -- the one source of run-to-run noise, 1e-5 .. 2e-4 on the losses)
    are summed in a fixed order, the pooled-row atomics of the res5 epilogue add at most two addends onto zero."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer

    model, e_vocab, e_seen, images, targets = _build(name)
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, f"configs/coco_cap_det/{name}.yaml"))
    cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4])
    cfg.freeze()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]
    batches = [(images, tg), (images.flip(-1).contiguous(), tg), (images * 0.5, tg)]

    def run(pipelined):
        m = copy.deepcopy(model).cuda()
        m.set_class_embeddings(e_seen.cuda())
        if hasattr(m, "set_caption_vocab"):
            m.set_caption_vocab(e_vocab.cuda())
        m.train()
        opt = solver.make_optimizer(cfg, m)
        red = comm.BucketedGradReducer(m)
        pipe = trainer.PipelinedTrainer(m, opt, red)
        pipe.enabled = pipe.enabled and pipelined
        losses, grads = [], None
        for i, (im, t) in enumerate(batches):
            torch.manual_seed(100 + i)
            nxt = batches[i + 1] if i + 1 < len(batches) else None
            losses.append({k: float(v.detach()) for k, v in pipe.step(im, t, nxt).items()})
            if i == 0:
                grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        pipe.drain()
        red.remove()
        return losses, grads, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}

    for pipelined in (False, True):
        l0, g0, w0 = run(pipelined)
        l1, g1, w1 = run(pipelined)
        assert l0 == l1, (pipelined, l0, l1)
        assert len(g0) >= 10 and set(g0) == set(g1)
        assert [n for n in g0 if not torch.equal(g0[n], g1[n])] == [], pipelined
        assert [n for n in w0 if not torch.equal(w0[n], w1[n])] == [], pipelined
        assert any(float(w0[n].sub(p.detach().cuda()).abs().max()) > 0 for n, p in model.named_parameters() if n in w0)  # it trained


@pytest.mark.parametrize("name", ["zeroshot_mask", "student_teacher_mask_rcnn_uncertainty"])
def test_weights_prepared_behind_the_optimizer_step_equal_inline_preparation(name):
    """The trainers prepare the GEMM operands of every trainable bottleneck in one launch right behind ``optimizer.step()``
    (``prepare_weights_ahead`` / ``WeightPrepPlan``); the next forward's blocks find them instead of running 3-4 preparation
    launches each.  Same bytes: losses, gradients and parameters over four iterations are IDENTICAL to the run whose blocks
    prepare their own weights, the plan did serve the blocks (13 bottlenecks of the teacher's trunk + res5 and its RPN 3x3, 3 in the student's head), and
    a weight edited between two steps is noticed (that block prepares its own operands, the result is that of the plain run)."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import pair_bottleneck

    model, e_vocab, e_seen, images, targets = _build(name)
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, f"configs/coco_cap_det/{name}.yaml"))
    cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4])
    cfg.freeze()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]
    batches = [(images, tg), (images.flip(-1).contiguous(), tg), (images * 0.5, tg), (images, tg)]

    def run(ahead):
        m = copy.deepcopy(model).cuda()
        m.set_class_embeddings(e_seen.cuda())
        if hasattr(m, "set_caption_vocab"):
            m.set_caption_vocab(e_vocab.cuda())
        m.train()
        opt = solver.make_optimizer(cfg, m)
        red = comm.BucketedGradReducer(m)
        policy = trainer.StepPolicy()
        assert policy.prepare_weights_ahead
        policy.prepare_weights_ahead = ahead
        pipe = trainer.PipelinedTrainer(m, opt, red, policy=policy)
        served = []
        lookup = pair_bottleneck.WeightPrepPlan.lookup

        def counting(self, key, scales):
            out = lookup(self, key, scales)
            served.append(out is not None)
            return out

        pair_bottleneck.WeightPrepPlan.lookup = counting
        try:
            losses, grads = [], []
            for i, (im, t) in enumerate(batches):
                torch.manual_seed(100 + i)
                if i == 2:  # an edit behind the trainer's back, between two steps
                    victim = [p for n, p in m.named_parameters() if p.requires_grad and n.endswith("conv2.weight")][0]
                    with torch.no_grad():
                        victim.mul_(1.0 + 2 ** -10)
                nxt = batches[i + 1] if i + 1 < len(batches) else None
                losses.append({k: float(v.detach()) for k, v in pipe.step(im, t, nxt).items()})
                grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            pair_bottleneck.WeightPrepPlan.lookup = lookup
        pipe.drain()
        red.remove()
        return losses, grads, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, served

    l0, g0, w0, s0 = run(False)
    l1, g1, w1, s1 = run(True)
    blocks = 14 if name == "zeroshot_mask" else 3  # teacher: 10 trunk + 3 res5 bottlenecks + the RPN head's 3x3
    assert s0 == []                                           # no plan: nothing to look up
    assert len(s1) == 3 * blocks and s1.count(False) == 1      # steps 2-4 look up; the edited block of step 3 is refused once
    assert l0 == l1
    for a, b in zip(g0, g1):
        assert set(a) == set(b) and [n for n in a if not torch.equal(a[n], b[n])] == []
    assert [n for n in w0 if not torch.equal(w0[n], w1[n])] == []


def test_trunk_blocks_backward_in_one_call_equals_the_launch_by_launch_backward():
    """Teacher step: the trunk's identity bottlenecks (8 of its 10 blocks) run their backward through ONE native call each, the
    weight gradients on a second stream (``layers.pair_bottleneck.ONE_CALL_BACKWARD_ROWS``).  Same launches underneath: losses,
    gradients and parameters over three iterations are IDENTICAL to the launch-by-launch backward, and the call was used."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd import _C
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
    from cvpr22_cross_modal_pseudo_labeling_amd.layers import pair_bottleneck

    model, e_vocab, e_seen, images, targets = _build("zeroshot_mask")
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/zeroshot_mask.yaml"))
    cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4])
    cfg.freeze()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]
    batches = [(images, tg), (images.flip(-1).contiguous(), tg), (images * 0.5, tg)]
    rows_default = pair_bottleneck.ONE_CALL_BACKWARD_ROWS
    assert rows_default >= 33400
    calls = []
    orig = _C.bottleneck_identity_backward

    def counting(*a, **k):
        calls.append(a[0].shape[0])
        return orig(*a, **k)

    def run(rows):
        pair_bottleneck.ONE_CALL_BACKWARD_ROWS = rows
        m = copy.deepcopy(model).cuda()
        m.set_class_embeddings(e_seen.cuda())
        m.train()
        opt = solver.make_optimizer(cfg, m)
        red = comm.BucketedGradReducer(m)
        pipe = trainer.PipelinedTrainer(m, opt, red)
        losses, grads = [], []
        for i, (im, t) in enumerate(batches):
            torch.manual_seed(100 + i)
            nxt = batches[i + 1] if i + 1 < len(batches) else None
            losses.append({k: float(v.detach()) for k, v in pipe.step(im, t, nxt).items()})
            grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        pipe.drain()
        red.remove()
        return losses, grads, {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}

    _C.bottleneck_identity_backward = counting
    try:
        l0, g0, w0 = run(0)
        assert calls == []
        l1, g1, w1 = run(rows_default)
    finally:
        _C.bottleneck_identity_backward = orig
        pair_bottleneck.ONE_CALL_BACKWARD_ROWS = rows_default
    assert len(calls) >= 3 * 6 and len(calls) % 3 == 0            # the identity blocks of layer2 / layer3 (and of the tiny res5 head), every step
    assert l0 == l1
    for a, b in zip(g0, g1):
        assert set(a) == set(b) and [n for n in a if not torch.equal(a[n], b[n])] == []
    assert [n for n in w0 if not torch.equal(w0[n], w1[n])] == []


def test_rpn_branch_on_a_second_stream_equals_the_one_stream_order():
    """Teacher step.  (a) ``RPNModule.forward`` issues the RPN loss on a second stream beside the proposal selection;
    (b) ``RPNModule.forward_ahead`` (what the detector calls) also runs the branch's whole BACKWARD ahead on that stream and
    joins its gradients to the graph.  Same kernels on the same operands, only stream and order of issue differ: losses and
    every gradient are IDENTICAL to the one-stream order, bit for bit, three steps in a row -- also under the 1 / 2 of a
    gradient accumulation; two loss terms weighted differently cannot be served by the run-ahead pass and give NaN, loudly."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.engine import trainer

    model, e_vocab, e_seen, images, targets = _build("zeroshot_mask")
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]

    def run(beside, ahead, scale=1.0, skew=False):
        m = copy.deepcopy(model).cuda()
        m.set_class_embeddings(e_seen.cuda())
        m.train()
        assert m.rpn.loss_beside_selection and m.rpn.backward_ahead
        m.rpn.loss_beside_selection, m.rpn.backward_ahead = beside, ahead
        out = []
        for i in range(3):
            torch.manual_seed(7 + i)
            m.zero_grad(set_to_none=True)
            loss_dict = m(images, tg)
            total = trainer.total_loss(loss_dict) * scale
            if skew:
                total = total + loss_dict["loss_objectness"]
            total.backward()
            torch.cuda.synchronize()
            out.append(({k: float(v.detach()) for k, v in loss_dict.items()},
                        {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
        return out

    for scale in (1.0, 0.5):
        one = run(False, False, scale)
        for beside, ahead in ((True, False), (True, True)):
            two = run(beside, ahead, scale)
            for (la, ga), (lb, gb) in zip(one, two):
                assert la == lb and "loss_objectness" in la and "loss_rpn_box_reg" in la
                assert set(ga) == set(gb) and any(n.startswith("rpn.head") for n in ga) and any(n.startswith("backbone") for n in ga)
                assert [n for n in ga if not torch.equal(ga[n], gb[n])] == [], (scale, beside, ahead)
    _, g = run(True, True, skew=True)[0]
    assert all(torch.isnan(g[n]).all() for n in g if n.startswith("rpn.head"))
    _, g = run(True, False, skew=True)[0]   # ... which the in-graph backward serves
    assert all(torch.isfinite(g[n]).all() for n in g)


def test_teacher_heads_as_one_batched_branch_match_the_plain_heads():
    """Teacher step: ``GeneralizedRCNN.forward`` runs its heads through ``CombinedROIHeads.forward_branches`` with ONE branch
    (the last res5 block hands the positives' maps to the mask head and takes their gradient as dense maps).  Same sampling,
    same losses; the gradients agree with ``CombinedROIHeads.forward``'s to round-off (the mask gradient joins the pooled
    one inside the gate kernel instead of through a scatter + add)."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.engine import trainer

    model, e_vocab, e_seen, images, targets = _build("zeroshot_mask")
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]

    def run(one_branch):
        m = copy.deepcopy(model).cuda()
        m.set_class_embeddings(e_seen.cuda())
        m.train()
        assert m.heads_as_one_branch
        m.heads_as_one_branch = one_branch
        torch.manual_seed(11)
        loss_dict = m(images, tg)
        trainer.total_loss(loss_dict).backward()
        return ({k: float(v.detach()) for k, v in loss_dict.items()},
                {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})

    (la, ga), (lb, gb) = run(False), run(True)
    assert set(la) == set(lb) == {"loss_classifier", "loss_box_reg", "loss_mask", "loss_objectness", "loss_rpn_box_reg"}
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-6 * max(abs(la[k]), 1.0), (k, la[k], lb[k])
    assert set(ga) == set(gb)
    for n in ga:
        assert torch.allclose(ga[n], gb[n], rtol=1e-4, atol=1e-6 * float(ga[n].abs().max()) + 1e-12), n


def test_rpn_shared_selection_matches_two_selections():
    """RPNModule.proposals_train_and_test (one decode + NMS pass per image) returns exactly the proposals of the
    train-mode and the test-mode selection run separately."""
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import to_image_list

    model, e_vocab, e_seen, images, targets = _build("student_teacher_mask_rcnn_uncertainty")
    model = model.cuda()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]
    with torch.no_grad():
        il = to_image_list(images)
        feats = model.backbone(il.tensors)
        head_out = model.rpn.head(feats[0])
        model.rpn.eval()
        test_ref, _ = model.rpn(il, feats, None, head_out=head_out)
        model.rpn.train()
        train_ref, _ = model.rpn(il, feats, tg, compute_loss=False, head_out=head_out)
        train_got, test_got = model.rpn.proposals_train_and_test(il, feats, tg, head_out)
    for a, b in zip(train_got + test_got, train_ref + test_ref):
        assert len(a) == len(b) and len(a) > 0
        assert torch.equal(a.bbox, b.bbox)
        assert torch.equal(a.get_field("objectness"), b.get_field("objectness"))


def test_batched_student_branches_match_sequential_branches():
    """forward_student with ONE pooler + res5 pass over the RoIs of both branches (CombinedROIHeads.forward_branches)
    gives the losses and gradients of the two sequential head calls (same sampler stream, injected noise)."""
    import copy

    model, e_vocab, e_seen, images, targets = _build("student_teacher_mask_rcnn_uncertainty")
    model = model.cuda()
    model.set_class_embeddings(e_seen.cuda())
    model.set_caption_vocab(e_vocab.cuda())
    model.train()
    tg = [t.to("cuda") for t in targets]
    with torch.no_grad():
        frozen = model.forward_frozen(images.cuda(), tg)
    res = {}
    for mode in ("0", "1"):
        m = copy.deepcopy(model)
        m.roi_heads_student.batch_branches = mode == "1"
        m.iter = model.iter
        torch.manual_seed(99)
        eps = torch.randn(1, 4096, 2, 14, 14, generator=torch.Generator().manual_seed(7))
        losses = m.forward_student(frozen, tg, eps=eps)
        sum(losses.values()).backward()
        res[mode] = ({k: float(v.detach()) for k, v in losses.items()},
                     {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    (l0, g0), (l1, g1) = res["0"], res["1"]
    assert set(l0) == set(l1)
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-5 * max(abs(l0[k]), 1e-3), (k, l0[k], l1[k])
    assert set(g0) == set(g1) and len(g0) >= 10
    for n in g0:
        assert (g0[n] - g1[n]).norm().item() <= 2e-4 * g0[n].norm().item() + 1e-9, n


def test_do_train_checkpoint_cadence_and_resume(tmp_path):
    """do_train writes model_<iter>.pth every CHECKPOINT_PERIOD and model_final.pth (trainer.py:172-173, 252-253); a
    second run in the same directory resumes weights, momentum, LR schedule and the iteration counter from the tag file
    EXACTLY (state equality) and continues with the remaining iterations."""
    import copy

    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import solver, trainer
    from cvpr22_cross_modal_pseudo_labeling_amd.utils.checkpoint import DetectronCheckpointer

    model, e_vocab, e_seen, images, targets = _build("zeroshot_mask")
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/zeroshot_mask.yaml"))
    cfg.merge_from_list(["SOLVER.BASE_LR", 1e-4, "SOLVER.LOG_PERIOD", 1])
    cfg.freeze()
    images = images.cuda()
    tg = [t.to("cuda") for t in targets]

    def setup(save_dir, start_model):
        m = copy.deepcopy(start_model).cuda()
        m.set_class_embeddings(e_seen.cuda())
        m.train()
        opt = solver.make_optimizer(cfg, m)
        sched = solver.make_lr_scheduler(cfg, opt)
        return m, opt, sched, DetectronCheckpointer(cfg, m, opt, sched, save_dir=save_dir)

    def train(m, opt, sched, ck, start, max_iter):
        return trainer.do_train(cfg, m, iter([(images, tg)] * (max_iter - start)), opt, sched, max_iter, start_iter=start,
                                checkpointer=ck, checkpoint_period=2)

    run_dir = str(tmp_path / "run")
    m1, opt1, sched1, ck1 = setup(run_dir, model)
    h1 = train(m1, opt1, sched1, ck1, 0, 2)
    assert [i for i, _ in h1] == [1, 2]
    assert sorted(os.listdir(run_dir)) == ["last_checkpoint", "model_0000002.pth", "model_final.pth"]
    assert open(os.path.join(run_dir, "last_checkpoint")).read().endswith("model_final.pth")
    # a fresh process state (weights of the ORIGINAL model) resumes from the tag file
    m2, opt2, sched2, ck2 = setup(run_dir, model)
    assert not torch.equal(m2.state_dict()["rpn.head.conv.weight"], m1.state_dict()["rpn.head.conv.weight"])
    extra = ck2.load("")
    assert extra == {"iteration": 2}
    for (n, a), (_, b) in zip(m2.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), n
    assert sched2.last_epoch == sched1.last_epoch == 2
    assert [g["lr"] for g in opt2.param_groups] == [g["lr"] for g in opt1.param_groups]
    bufs1 = [opt1.state[p]["momentum_buffer"] for g in opt1.param_groups for p in g["params"] if p in opt1.state]
    bufs2 = [opt2.state[p]["momentum_buffer"] for g in opt2.param_groups for p in g["params"] if p in opt2.state]
    assert len(bufs1) == len(bufs2) > 10 and all(torch.equal(a, b) for a, b in zip(bufs1, bufs2))
    h2 = train(m2, opt2, sched2, ck2, int(extra["iteration"]), 4)
    assert [i for i, _ in h2] == [3, 4] and all(v == v for _, d in h2 for v in d.values())
    assert sorted(os.listdir(run_dir)) == ["last_checkpoint", "model_0000002.pth", "model_0000004.pth", "model_final.pth"]
    assert torch.load(os.path.join(run_dir, "model_final.pth"), weights_only=False)["iteration"] == 4


def test_eval_forward_and_inference_collects_detections(tmp_path):
    """Evaluation mode (box PostProcessor with the grouped per-class NMS, mask post-processing) through
    engine.inference: every image comes back in id order as a BoxList with scores / labels / an MxM mask probability,
    at most DETECTIONS_PER_IMG detections, boxes inside the image; the grouped-NMS post-processor gives the detections
    of the per-class loop."""
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import inference
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.roi_heads import PostProcessor

    model, e_vocab, e_seen, images, targets = _build("zeroshot_mask")
    model = model.cuda()
    batches = [(images[:1], None, [1]), (images[1:], None, [0])]
    preds = inference.inference(model, batches, device="cuda", output_folder=str(tmp_path), class_embeddings=e_seen)
    assert len(preds) == 2 and os.path.exists(tmp_path / "predictions.pth")
    total = 0
    for det in preds:
        assert det.bbox.device.type == "cpu" and set(det.fields()) >= {"scores", "labels", "mask"}
        m = det.get_field("mask")
        assert len(det) <= 100 and m.shape[:2] == (len(det), 1) and m.shape[2] == m.shape[3] >= 14
        assert float(m.min()) >= 0.0 and float(m.max()) <= 1.0
        w, h = det.size
        if len(det):
            assert float(det.bbox[:, 0::2].max()) <= w - 1 and float(det.bbox[:, 1::2].max()) <= h - 1
            assert bool((det.get_field("labels") >= 1).all()) and bool((det.get_field("scores") > 0.05).all())
        total += len(det)
    assert total > 0
    # same detections with the reference's per-class loop in the post-processor
    orig = PostProcessor.filter_results
    PostProcessor.filter_results = PostProcessor.filter_results_per_class
    try:
        loop = inference.inference(model, batches, device="cuda", class_embeddings=e_seen)
    finally:
        PostProcessor.filter_results = orig
    for a, b in zip(preds, loop):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("labels"), b.get_field("labels"))
        assert torch.equal(a.get_field("scores"), b.get_field("scores"))


def test_device_prefetcher_stages_batches_on_a_copy_stream():
    """DevicePrefetcher on the GPU: device batches equal the host batches, in order, usable on the consumer's stream
    right away (event hand-over + record_stream), also while the consumer's stream is busy."""
    from cvpr22_cross_modal_pseudo_labeling_amd.data.prefetch import DevicePrefetcher
    from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import make_batch

    host = [make_batch(2, seed=i, height=96, width=128, num_gt=3, num_nouns=2) for i in range(6)]
    busy = torch.randn(4096, 4096, device="cuda")
    sums = []
    for images, targets in DevicePrefetcher(iter(host), "cuda", depth=2):
        for _ in range(3):
            busy = busy @ busy * 1e-4  # keep the consumer stream occupied
        assert images.is_cuda and targets[0].bbox.is_cuda and targets[0].get_field("masks").is_cuda
        sums.append((images.double().sum() + targets[1].get_field("masks").sum() + targets[0].bbox.sum()).item())
    want = [(im.double().sum() + t[1].get_field("masks").sum() + t[0].bbox.sum()).item() for im, t in host]
    assert len(sums) == 6 and all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(sums, want))


def test_streams_are_shared_per_process_not_grown_per_object():
    """HIP multiplexes streams onto a few hardware queues: a process that made a NEW copy / side stream for every prefetcher /
    trainer put its fifth stream on the compute stream's queue (the teacher step as second workload of bench.py: 31 ms instead
    of 22.5).  Every prefetcher of a device shares one copy stream, every trainer one side stream per priority -- and a second
    prefetcher opened after the first was closed still delivers its batches."""
    from cvpr22_cross_modal_pseudo_labeling_amd.data.prefetch import DevicePrefetcher
    from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import make_batch
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import trainer

    host = [make_batch(1, seed=i, height=64, width=96, num_gt=2, num_nouns=2) for i in range(3)]
    a = DevicePrefetcher(iter(host), "cuda", depth=2)
    first = [im.sum().item() for im, _ in a]
    a.close()
    b = DevicePrefetcher(iter(host), "cuda", depth=2)
    assert b.stream is a.stream and b.stream != torch.cuda.current_stream()
    assert [im.sum().item() for im, _ in b] == first
    b.close()
    s1, s2 = trainer.side_stream(-1), trainer.side_stream(-1)
    assert s1 is s2 and s1 != torch.cuda.current_stream() and trainer.side_stream(0) is not s1


def test_student_step_with_text_vocabulary_and_polygon_ground_truth(golden_dir):
    """The round-3 input forms inside the real step: (a) the caption vocabulary given as STRINGS and embedded through the
    BERT word-embedding table (tokenizer over the golden WordPiece vocabulary; per-image nouns from the ``nn_caption``
    string, st_generalized_rcnn.py:318,242) gives exactly the losses of the same vocabulary handed over as the matrix
    ``extract_emb`` produces; (b) polygon ground-truth masks (``SegmentationMask`` mode 'poly' in the reference) run
    through the device rasteriser: finite losses, a mask loss that moves when the polygons change."""
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.language_backbone import BERT, normalize_class_names
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.structures import PolygonMasks

    model, _, e_seen, images, targets = _build("student_teacher_mask_rcnn_uncertainty")
    vocab = open(os.path.join(golden_dir, "wordpiece_vocab.txt")).read().split()
    words = sorted({w for w in vocab if w.isalpha() and len(w) > 2})
    n = len(words)
    names = [f"{words[i % n]}_{words[(i + 1 + i // n) % n]}" for i in range(60)]  # 60 distinct two-word names
    assert len(set(names)) == 60
    model = model.cuda()
    model.bert = BERT(None, vocab_file=os.path.join(golden_dir, "wordpiece_vocab.txt"), vocab_size=len(vocab)).cuda()
    model.set_class_embeddings(e_seen.cuda())
    eps = torch.randn(1, 4096, 2, 14, 14, generator=torch.Generator().manual_seed(3))
    tg = [t.to("cuda") for t in targets]

    def step(tgs):
        torch.manual_seed(11)
        model.iter = 1
        for p in model.parameters():
            p.grad = None
        losses = model(images.cuda(), tgs, eps=eps)
        sum(losses.values()).backward()
        return {k: float(v.detach()) for k, v in losses.items()}

    # (a) strings vs matrix
    model.set_caption_vocab_names(names)
    with_text = []
    for t in tg:
        t2 = t.copy_with_fields(t.fields())
        t2.add_field("nn_caption", "/".join(normalize_class_names(names)[i] for i in t.get_field("ids_cap").tolist()))
        with_text.append(t2)
    l_text = step(with_text)
    matrix = model.bert.extract_emb(normalize_class_names(names))
    assert matrix.shape == (60, 768) and bool(torch.isfinite(matrix).all())
    model.cap_vocab = None
    model.set_caption_vocab(matrix)
    l_matrix = step(tg)
    assert l_text == l_matrix, (l_text, l_matrix)
    assert all(v == v and abs(v) < 1e6 for v in l_text.values())

    # (b) polygon ground truth: the inset rectangles of the synthetic masks as polygons, then a different shape
    def with_polygons(shrink):
        out = []
        for t in tg:
            inst = []
            for b in t.bbox.tolist():
                dx, dy = shrink * (b[2] - b[0]), shrink * (b[3] - b[1])
                inst.append([[b[0] + dx, b[1] + dy, b[2] - dx, b[1] + dy, b[2] - dx, b[3] - dy, b[0] + dx, b[3] - dy]])
            t2 = t.copy_with_fields([f for f in t.fields() if f != "masks"])
            t2.add_field("masks", PolygonMasks(inst, t.size).to("cuda"))
            out.append(t2)
        return out

    l_poly = step(with_polygons(0.1))
    l_poly2 = step(with_polygons(0.3))
    assert all(v == v and abs(v) < 1e6 for v in l_poly.values())
    assert l_poly["loss_mask"] != l_poly2["loss_mask"]                      # the targets come from the polygons
    assert l_poly["loss_classifier"] == l_poly2["loss_classifier"]          # ... and nothing else does
    assert abs(l_poly["loss_mask"] - l_matrix["loss_mask"]) <= 0.25 * abs(l_matrix["loss_mask"])  # same shapes as the binary masks
