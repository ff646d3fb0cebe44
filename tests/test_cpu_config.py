"""CPU (no GPU): the HOST-tensor side of the product -- the reference's CPU-only configuration (MODEL.DEVICE cpu,
BASELINE.json configs[0]).  ``_C`` dispatches on the tensor's device like the reference's module (csrc/ROIAlign.h:11-25,
csrc/nms.h:10-28): host tensors -> ``libovis_cpu.so`` / the reference's torch-op formulas (``_cpu.py``), device tensors -> HIP.
Pinned to the golden vectors of the reference's own CPU kernels and to the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.oracle_backend import oracle_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_header_symbols_exported_and_bound():
    from cvpr22_cross_modal_pseudo_labeling_amd import _cpu

    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ovis_cpu.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(ovis_cpu_\w+)\s*\(", text)))
    lib = ctypes.CDLL(_cpu.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_cpu.SIGNATURES) == names
    assert _cpu.load().ovis_cpu_version().startswith(b"ovis_cpu")


def test_host_roi_align_forward_bit_exact_vs_reference_golden(golden_dir):
    from cvpr22_cross_modal_pseudo_labeling_amd import _C, layers

    z = np.load(os.path.join(golden_dir, "roi_align_forward.npz"))
    x, rois, scale = torch.from_numpy(z["input"]), torch.from_numpy(z["rois"]), float(z["scale"])
    for key, (ph, pw, sr) in {"out_sr0": (14, 14, 0), "out_sr2": (14, 14, 2), "out_7x7_sr0": (7, 7, 0)}.items():
        assert torch.equal(_C.roi_align_forward(x, rois, scale, ph, pw, sr), torch.from_numpy(z[key])), key
    assert torch.equal(layers.ROIAlign((14, 14), scale, 0)(x, rois), torch.from_numpy(z["out_sr0"]))


def test_host_roi_align_matches_oracle_and_is_thread_count_independent(oracle_mod):
    from cvpr22_cross_modal_pseudo_labeling_amd import _C, _cpu

    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 16, 25, 42, generator=g)
    b = torch.randint(0, 3, (70, 1), generator=g).float()
    xy = torch.rand(70, 2, generator=g) * torch.tensor([600.0, 350.0]) - 20.0  # some RoIs start outside the image
    wh = torch.rand(70, 2, generator=g) * 300 + 1
    rois = torch.cat([b, xy, xy + wh], 1)
    rois[5, 3:] = rois[5, 1:3]  # a degenerate RoI (zero extent -> clamped to one cell)
    for sr in (0, 2):
        out = _C.roi_align_forward(x, rois, 1 / 16, 14, 14, sr)
        assert torch.equal(out, oracle_mod.roi_align_forward(x, rois, 1 / 16, 14, 14, sr))
        go = torch.randn(out.shape, generator=g)
        gin = _C.roi_align_backward(go, rois, 1 / 16, 14, 14, 3, 16, 25, 42, sr)
        assert torch.equal(gin, oracle_mod.roi_align_backward(go, rois, 1 / 16, 14, 14, 3, 16, 25, 42, sr))
        # autograd through the layer reaches the same transpose
        xr = x.clone().requires_grad_(True)
        from cvpr22_cross_modal_pseudo_labeling_amd import layers
        layers.ROIAlign((14, 14), 1 / 16, sr)(xr, rois).backward(go)
        assert torch.equal(xr.grad, gin)
    lib = _cpu.load()
    one = torch.empty(70, 16, 14, 14)
    lib.ovis_cpu_roi_align_forward_f32(x.data_ptr(), rois.data_ptr(), one.data_ptr(), 70, 3, 16, 25, 42, 14, 14, 1 / 16, 0, 1)
    assert torch.equal(one, _C.roi_align_forward(x, rois, 1 / 16, 14, 14, 0))
    # empty inputs: correctly shaped, nothing touched
    assert _C.roi_align_forward(x, torch.zeros(0, 5), 1 / 16, 14, 14, 0).shape == (0, 16, 14, 14)
    assert torch.count_nonzero(_C.roi_align_backward(torch.zeros(0, 16, 14, 14), torch.zeros(0, 5), 1 / 16, 14, 14, 3, 16, 25, 42, 0)) == 0
    bad = rois.clone()
    bad[0, 0] = 7.0
    with pytest.raises(RuntimeError):
        _C.roi_align_forward(x, bad, 1 / 16, 14, 14, 0)


@pytest.mark.parametrize("name", ["rpn_like", "dense", "tiny", "one"])
def test_host_nms_exact_vs_reference_golden(golden_dir, name):
    from cvpr22_cross_modal_pseudo_labeling_amd import layers

    z = np.load(os.path.join(golden_dir, "nms.npz"))
    boxes, scores = torch.from_numpy(z[f"{name}_boxes"]), torch.from_numpy(z[f"{name}_scores"])
    assert torch.equal(layers.nms(boxes, scores, float(z[f"{name}_thr"])), torch.from_numpy(z[f"{name}_keep"]))


def test_host_nms_ties_and_empty(oracle_mod):
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    # IoU exactly 0.5: the host kernel suppresses on >= (cpu/nms_cpu.cpp:60)
    boxes = torch.tensor([[0.0, 0.0, 9.0, 9.0], [0.0, 0.0, 9.0, 4.0]])
    assert _C.nms(boxes, torch.tensor([0.9, 0.8]), 0.5).tolist() == [0]
    assert _C.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(400, 2, generator=g) * 300
    boxes = torch.cat([xy, xy + torch.rand(400, 2, generator=g) * 120 + 4], 1)
    scores = torch.randint(0, 20, (400,), generator=g).float() / 20  # many equal scores: ties keep the lower index first
    assert torch.equal(_C.nms(boxes, scores, 0.4), oracle_mod.nms(boxes, scores, 0.4, ge_mode=True))
    keep, num = _C.nms_padded(boxes, scores, 0.4)
    assert torch.equal(keep[: int(num)], _C.nms(boxes, scores, 0.4)) and keep.shape == (400,)


def test_host_focal_layer_vs_reference_golden(golden_dir):
    from cvpr22_cross_modal_pseudo_labeling_amd import layers

    z = np.load(os.path.join(golden_dir, "sigmoid_focal_loss.npz"))
    for sfx, gamma, alpha in (("", 2.0, 0.25), ("2", 1.5, 0.4)):
        logits = torch.from_numpy(z["logits" + sfx]).clone().requires_grad_(True)
        targets = torch.from_numpy(z["targets" + sfx])
        # fp32 log(1 - p) of the plain formula loses digits as |x| grows (the reference's host formula is this one; the golden
        # values are its fp64 evaluation): 2e-5 relative below |x| = 8, 5e-4 at 12
        loss = layers.sigmoid_focal_loss.sigmoid_focal_loss_cpu(logits, targets, gamma, alpha)
        want = torch.from_numpy(z["loss" + sfx])
        ok = logits.detach().abs() < 8
        assert torch.allclose(loss.detach().double()[ok], want[ok], rtol=1e-4, atol=1e-6)
        assert torch.allclose(loss.detach().double(), want, rtol=2e-3, atol=1e-6)
        total = layers.SigmoidFocalLoss(gamma, alpha)(logits, targets)
        assert abs(float(total.detach()) - float(want.sum())) <= 1e-4 * float(want.sum())
        total.backward()
        assert bool(torch.isfinite(logits.grad).all())


def test_host_head_and_loss_entry_points_match_autograd_of_the_reference_formulas():
    import torch.nn.functional as F

    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    g = torch.Generator().manual_seed(2)
    logits = torch.randn(37, 9, generator=g)
    labels = torch.randint(0, 9, (37,), generator=g)
    labels[:10] = 0
    loss, grad = _C.weighted_ce_fwd_bwd(logits, labels, 0.2)
    x = logits.clone().requires_grad_(True)
    w = torch.ones(9)
    w[0] = 0.2
    want = (F.cross_entropy(x, labels, weight=w, reduction="none") / labels.numel()).sum()
    want.backward()
    assert torch.allclose(loss, want.detach(), rtol=1e-6, atol=1e-7) and torch.allclose(grad, x.grad, rtol=1e-5, atol=1e-7)
    mu, sigma = torch.randn(12, 2, 14, 14, generator=g), torch.rand(12, 1, 14, 14, generator=g) + 0.5
    eps = torch.randn(12, 2, 14, 14, generator=g)
    pos = torch.tensor([0, 3, 3, 7, 11])
    chan = torch.tensor([1, 0, 1, 1, 0])
    tg = (torch.rand(5, 14, 14, generator=g) > 0.5).float()
    for s, e in ((sigma, eps), (None, None)):
        for ch in (chan, 1):
            loss, dmu, dsig = _C.mask_bce_stochastic_fwd_bwd(mu, s, e, pos, tg, ch)
            m = mu.clone().requires_grad_(True)
            sg = None if s is None else s.clone().requires_grad_(True)
            zz = m if sg is None else m + e * sg
            want = F.binary_cross_entropy_with_logits(zz[pos, ch], tg)
            want.backward()
            assert torch.allclose(loss, want.detach(), rtol=1e-6)
            assert torch.allclose(dmu, m.grad, rtol=1e-5, atol=1e-8)
            assert (dsig is None) == (s is None) and (s is None or torch.allclose(dsig, sg.grad, rtol=1e-5, atol=1e-8))
    a, b, bias = torch.randn(20, 64, generator=g), torch.randn(7, 64, generator=g), torch.randn(7, generator=g)
    assert torch.allclose(_C.gemm_nt(a, b, bias), F.linear(a, b, bias), rtol=1e-6, atol=1e-6)
    raw, prob, idx = _C.region_noun_align(a, b)
    assert torch.equal(idx, (a @ b.t()).argmax(0)) and torch.allclose(prob, torch.sigmoid(raw))


def test_device_only_ops_refuse_host_tensors_and_nothing_crosses_devices():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C

    with pytest.raises(RuntimeError):  # no host form in the reference either (csrc/SigmoidFocalLoss.h:23)
        _C.sigmoid_focalloss_forward(torch.zeros(2, 4), torch.zeros(2, dtype=torch.int32), 4, 2.0, 0.25)
    with pytest.raises(RuntimeError):
        _C.split_pair(torch.zeros(32, 32))
    with pytest.raises(RuntimeError):
        _C.roi_align_forward_strided_nhwc(torch.zeros(1, 4, 4, 32), torch.zeros(1, 5), 1.0, 14, 14, 0, 2)
    src = open(os.path.join(ROOT, "cvpr22_cross_modal_pseudo_labeling_amd", "_cpu.py")).read()
    assert ".cuda(" not in src and "libovis_hip" not in src and "oracle" not in src


class _DeviceTensorStandIn:
    is_cuda = True
    dtype = torch.float32


def test_host_entry_points_reject_device_tensors():
    from cvpr22_cross_modal_pseudo_labeling_amd import _cpu

    with pytest.raises(RuntimeError):
        _cpu._host(_DeviceTensorStandIn(), "input")


def _cpu_model(name, extra=()):
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

    torch.manual_seed(0)
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, f"configs/coco_cap_det/{name}.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 200,
                         "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 60, "MODEL.RPN.POST_NMS_TOP_N_TEST", 40,
                         "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "SOLVER.BASE_LR", 1e-5] + list(extra))
    cfg.freeze()
    model = build_detection_model(cfg).to(cfg.MODEL.DEVICE)
    e_vocab, e_seen = make_embeddings(n_vocab=50)
    model.set_class_embeddings(e_seen)
    if hasattr(model, "set_caption_vocab"):
        model.set_caption_vocab(e_vocab)
    images, targets = make_batch(2, height=128, width=160, num_gt=3, num_nouns=2, n_vocab=50)
    calibrate_stem_bn(model, images)
    return cfg, model, images, targets


def test_config0_teacher_on_cpu_trains_and_matches_the_oracle_backed_run():
    """BASELINE configs[0]: zeroshot_mask.yaml (R-50 teacher), 2 images, MODEL.DEVICE cpu -- forward, backward and an SGD step
    on the product's host path; the same step with every native entry point routed to the oracle gives the same losses and
    the same updated weights (RoIAlign / NMS are bit-identical, the loss formulas agree to rounding)."""
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer

    results = []
    for use_oracle in (False, True):
        cfg, model, images, targets = _cpu_model("zeroshot_mask")
        model.train()
        opt = solver.make_optimizer(cfg, model)
        red = comm.BucketedGradReducer(model)
        torch.manual_seed(11)
        if use_oracle:
            with oracle_ops():
                losses = trainer.train_step(model, opt, red, images, targets)
        else:
            losses = trainer.train_step(model, opt, red, images, targets)
        red.remove()
        results.append(({k: float(v) for k, v in losses.items()}, {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}))
        # the trainers' hook behind optimizer.step() (GEMM operands of the trainable bottlenecks in one launch) has nothing to
        # prepare on a host model, remembers that, and leaves no plan on any block
        from cvpr22_cross_modal_pseudo_labeling_amd.modeling.backbone import prepare_weights_ahead
        assert model.__dict__["_ovis_prep_plan"] == torch.device("cpu") and prepare_weights_ahead(model) == 0
        assert not any("_prep_plan" in m.__dict__ for m in model.modules())
        assert not model.rpn.runs_ahead([torch.zeros(1)])  # ... nor is there a second stream for the RPN branch to run ahead on
    (la, pa), (lb, pb) = results
    assert set(la) == {"loss_classifier", "loss_box_reg", "loss_mask", "loss_objectness", "loss_rpn_box_reg"}
    for k in la:
        assert np.isfinite(la[k]) and abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    for n in pa:
        assert torch.allclose(pa[n], pb[n], rtol=1e-4, atol=1e-7), n


def test_teacher_frozen_half_on_the_host_is_the_plain_forward():
    """``GeneralizedRCNN.forward_frozen`` / ``forward_student`` (what ``PipelinedTrainer`` drives on the GPU) on host tensors: the
    trunk has no prefix to run ahead there (``forward_prefix`` -> None: the NHWC pair chain is device-only), and the split step is
    the plain forward -- same losses, bit for bit; the trainer itself stays disabled without a GPU."""
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer

    cfg, model, images, targets = _cpu_model("zeroshot_mask")
    model.train()
    torch.manual_seed(11)
    want = {k: float(v) for k, v in model(images, targets).items()}
    torch.manual_seed(11)
    frozen = model.forward_frozen(images, targets)
    assert frozen["prefix"] is None
    got = {k: float(v) for k, v in model.forward_student(frozen, targets).items()}
    assert got == want
    opt = solver.make_optimizer(cfg, model)
    red = comm.BucketedGradReducer(model)
    pipe = trainer.PipelinedTrainer(model, opt, red)
    assert pipe.enabled == torch.cuda.is_available()
    red.remove()


def test_config0_teacher_on_cpu_inference_returns_detections():
    cfg, model, images, targets = _cpu_model("zeroshot_mask")
    model.eval()
    with torch.no_grad():
        out = model(images)
    assert len(out) == 2
    for det in out:
        assert det.bbox.shape[1] == 4 and not det.bbox.is_cuda
        assert det.has_field("scores") and det.has_field("labels")
        assert bool(torch.isfinite(det.bbox).all())


def test_config0_variants_linear_classifier_per_class_boxes_and_masks_on_cpu():
    """The configuration variants of round 4 on the HOST path (MODEL.DEVICE cpu): ``EMBEDDING_BASED False`` (a learned Linear over
    NUM_CLASSES, roi_box_predictors.py:33-40), per-class box deltas (box_head/loss.py:156-160) and class-specific mask logits
    (mask_head/loss.py:131-141: one logit channel per positive through the same fused loss entry point) -- the product's host code
    against the same step with every native entry point routed to the oracle; eval-mode inference returns masks."""
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
    from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import make_batch

    extra = ["MODEL.ROI_BOX_HEAD.EMBEDDING_BASED", False, "MODEL.ROI_BOX_HEAD.NUM_CLASSES", 9, "MODEL.CLS_AGNOSTIC_BBOX_REG", False,
             "MODEL.CLS_AGNOSTIC_MASK", False]
    results = []
    for use_oracle in (False, True):
        torch.manual_seed(0)
        from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
        from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn
        from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

        cfg = get_defaults()
        cfg.merge_from_file(os.path.join(ROOT, "configs/coco_cap_det/zeroshot_mask.yaml"))
        cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 60,
                             "MODEL.RPN.PRE_NMS_TOP_N_TEST", 200, "MODEL.RPN.POST_NMS_TOP_N_TEST", 40,
                             "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "SOLVER.BASE_LR", 1e-5] + extra)
        cfg.freeze()
        model = build_detection_model(cfg)
        pred = model.roi_heads["box"].predictor
        assert pred.cls_score.weight.shape == (9, 2048) and pred.bbox_pred.weight.shape == (36, 2048)
        with pytest.raises(RuntimeError):
            model.set_class_embeddings(torch.zeros(9, 768))  # no embedding head to set
        images, targets = make_batch(2, height=128, width=160, num_gt=3, num_nouns=2, n_vocab=50, n_seen=9)
        calibrate_stem_bn(model, images)
        with torch.no_grad():
            pred.cls_score.weight.mul_(30.0)
            pred.bbox_pred.weight.mul_(30.0)
        model.train()
        opt = solver.make_optimizer(cfg, model)
        red = comm.BucketedGradReducer(model)
        torch.manual_seed(11)
        if use_oracle:
            with oracle_ops():
                losses = trainer.train_step(model, opt, red, images, targets)
        else:
            losses = trainer.train_step(model, opt, red, images, targets)
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        red.remove()
        results.append(({k: float(v) for k, v in losses.items()}, grads, model, images))
    (la, ga, model, images), (lb, gb, _, _) = results
    for k in la:
        assert np.isfinite(la[k]) and abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    for n in ga:
        assert torch.allclose(ga[n], gb[n], rtol=1e-3, atol=1e-7), n
    gm = ga["roi_heads.mask.predictor.mask_fcn_logits.weight"].flatten(1).abs().sum(1)
    assert float(gm[0]) == 0.0 and int((gm > 0).sum()) >= 2  # only the logit channels of the positives' classes train
    model.eval()
    with torch.no_grad():
        det = model(images)
    assert len(det) == 2 and all(d.has_field("mask") and d.has_field("labels") for d in det)
