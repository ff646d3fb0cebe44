"""Generates tests/golden/step_student.npz and step_teacher.npz by running the REFERENCE's own detector classes --
``STGeneralizedRCNN`` (modeling/detector/st_generalized_rcnn.py:27-418: ``prepare_model``, ``generate_pseudo_label``,
``forward`` with its branch split, class-matrix swaps, ``adaptive_lamb`` and dummy losses) and ``GeneralizedRCNN``
(detector/generalized_rcnn.py:16-73) -- built from the reference's real ``config/defaults.py`` + its shipped yaml
files + the overrides of ``step_case.COMMON_OPTS``, on the CPU of the build container.  Run there only:

    python oracle/build_ref.py && python tests/golden/make_step_golden.py

How the reference is made to run here (tests/golden/ref_import.py): stand-in modules for packages this image lacks,
``maskrcnn_benchmark._C`` = the reference's own csrc compiled for the CPU, ``torch.Tensor.cuda`` = identity,
the BERT constructor replaced (it downloads bert-base-uncased) by one that installs a HuggingFace ``BertTokenizer`` over
``step_case.WORDPIECES`` and a seeded embedding table -- ``BERT.forward`` and ``extract_emb`` run unchanged.

What is captured besides the returned losses: the fg / bg samplers' masks (``BalancedPositiveNegativeSampler.__call__``
wrapped), the mask head's noise (``torch.randn`` wrapped), the region x noun score matrix (``torch.einsum`` wrapped),
proposals of both RPN modes, the pseudo labels, and after ``sum(losses).backward()`` a digest of every gradient
(step_case.grad_digest).  ``STGeneralizedRCNN`` is run one image per call (it is only correct there);
``GeneralizedRCNN`` on the two-image batch.  Only inputs / outputs are stored -- no reference source text.
"""
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import step_case as case  # noqa: E402


class Capture:
    """Wraps the three random / intermediate call sites for the duration of one forward."""

    def __init__(self, sampler_cls):
        self.sampler_cls = sampler_cls
        self.samples, self.randn, self.einsum = [], [], []

    def __enter__(self):
        cap = self
        self._call, self._randn, self._einsum = self.sampler_cls.__call__, torch.randn, torch.einsum

        def call(sampler, matched_idxs):
            pos, neg = cap._call(sampler, matched_idxs)
            cap.samples.append((sampler.batch_size_per_image, [p.clone() for p in pos], [n.clone() for n in neg]))
            return pos, neg

        def randn(*a, **k):
            out = cap._randn(*a, **k)
            cap.randn.append(out.clone())
            return out

        def einsum(eq, *ops):
            out = cap._einsum(eq, *ops)
            cap.einsum.append((eq, out.detach().clone()))
            return out

        self.sampler_cls.__call__, torch.randn, torch.einsum = call, randn, einsum
        return self

    def __exit__(self, *exc):
        self.sampler_cls.__call__, torch.randn, torch.einsum = self._call, self._randn, self._einsum


def make_bert_class(RefBERT):
    from transformers import BertTokenizer

    vocab_path = os.path.join(HERE, "step_wordpiece_vocab.txt")
    with open(vocab_path, "w") as f:
        f.write("\n".join(case.WORDPIECES) + "\n")
    hf = BertTokenizer(vocab_path, do_lower_case=True)

    class Tok302:  # the call surface of transformers==3.0.2 that language_backbone/transformers.py:28-32 uses
        def batch_encode_plus(self, text_list, add_special_tokens=True, pad_to_max_length=False, return_special_tokens_mask=False):
            enc = hf(list(text_list), add_special_tokens=add_special_tokens, padding=bool(pad_to_max_length),
                     return_special_tokens_mask=return_special_tokens_mask)
            return {k: v for k, v in enc.items()}

    class LocalBERT(RefBERT):
        _seeded_stand_in = True

        def __init__(self, cfg):
            torch.nn.Module.__init__(self)
            self.tokenizer, self.mlm = Tok302(), False
            self.embeddings = torch.nn.Parameter(torch.zeros(len(case.WORDPIECES), case.EMB_DIM), requires_grad=False)

    return LocalBERT


def load_seeded(model):
    """Every parameter / FrozenBN buffer of the reference model <- step_case.seeded_tensor(its first state-dict name)."""
    first = {}
    with torch.no_grad():
        for name, t in list(model.named_parameters()) + list(model.named_buffers()):
            if id(t) in first or not case.is_seeded(name):
                continue
            first[id(t)] = name
            t.copy_(case.seeded_tensor(name, t.shape))
    # (state-dict name, shape, the name its value was seeded under: a module reachable under two names -- the mask head
    # shares the box head's feature extractor, roi_heads.py:21-22 -- has one tensor and two entries)
    return [(n, tuple(t.shape), first[id(t)]) for n, t in model.state_dict(keep_vars=True).items() if case.is_seeded(n)]


def make_target(BoxList, SegmentationMask, c, with_caption):
    t = BoxList(c["boxes"].clone(), (case.IMAGE_W, case.IMAGE_H), mode="xyxy")
    t.add_field("labels", c["labels"].clone())
    t.add_field("masks", SegmentationMask(c["masks"].clone(), (case.IMAGE_W, case.IMAGE_H), mode="mask"))
    if with_caption:
        t.add_field("nn_caption", c["nn_caption"])
        t.add_field("ids_cap", c["ids_cap"].clone())
        t.add_field("is_det", "Yes")
    return t


def put_boxes(out, key, boxlists, fields=()):
    for i, b in enumerate(boxlists):
        out[f"{key}{i}_bbox"] = b.bbox.detach().numpy()
        for f in fields:
            out[f"{key}{i}_{f}"] = b.get_field(f).detach().numpy()


def put_grads(out, model, prefix="grad"):
    names = []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        d = case.grad_digest(name, p.grad)
        out[f"{prefix}:{name}:values"] = d["values"]
        out[f"{prefix}:{name}:norm_sum"] = np.array([d["norm"], d["sum"]])
        names.append(name)
    out[f"{prefix}_names"] = np.array(names)


def put_samples(out, key, cap, batch_size):
    """Masks of the RoI (or RPN) sampler's calls in call order, one (pos, neg) pair per image."""
    k = 0
    for bs, pos, neg in cap.samples:
        if bs != batch_size:
            continue
        for p, n in zip(pos, neg):
            out[f"{key}{k}_pos"], out[f"{key}{k}_neg"] = p.numpy().astype(np.bool_), n.numpy().astype(np.bool_)
            k += 1
    out[f"{key}_count"] = np.int64(k)


def gen_student():
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.language_backbone import transformers as ref_lb
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    ref_import._namespace_pkg("maskrcnn_benchmark.modeling.detector",
                              os.path.join(ref_import.REF, "maskrcnn_benchmark/modeling/detector"))
    ref_lb.BERT = make_bert_class(ref_lb.BERT)
    from maskrcnn_benchmark.modeling.detector import st_generalized_rcnn as st_mod

    cfg = ref_import.reference_cfg("student_teacher_mask_rcnn_uncertainty.yaml", case.COMMON_OPTS)
    out = {}
    e_seen = case.text_embeddings()
    for img_index in range(2):
        # a fresh model per image: every run is "iteration 0" (prepare_model copies the teacher heads into the student)
        model = st_mod.STGeneralizedRCNN(cfg)
        names = load_seeded(model)
        model.class_names = list(case.SEEN_NAMES)
        model.roi_heads["box"].predictor.set_class_embeddings(e_seen.clone())
        model.train()
        c = case.image_case(img_index, model.cap_vocab)
        target = make_target(BoxList, SegmentationMask, c, True)
        key = f"img{img_index}_"
        # ---- evaluation branch (st_generalized_rcnn.py:409-418) on a FRESH model: the student heads still hold their own
        # weights (the training branch copies the teacher's into them at iteration 0), so a product that evaluated with the
        # teacher heads, or with the caption vocabulary left in the predictor, would not reproduce these detections
        model.eval()
        with torch.no_grad():
            det = model(c["image"][None])[0]
        out[key + "eval_bbox"] = det.bbox.numpy()
        for f in ("scores", "labels", "mask"):
            out[key + "eval_" + f] = det.get_field(f).numpy()
        model.train()
        with Capture(BalancedPositiveNegativeSampler) as cap:
            losses = model(c["image"][None], [target])
            # the frozen half again, piece by piece, for the intermediate values (same modules, same state)
            with torch.no_grad():
                from maskrcnn_benchmark.structures.image_list import to_image_list
                images = to_image_list(c["image"][None])
                feats = model.backbone(images.tensors)
                model.rpn.eval()
                props_test, _ = model.rpn(images, feats, None)
                caps = [target.get_field("nn_caption").split("/")]
                n_einsum = len(cap.einsum)
                pseudo = model.generate_pseudo_label([feats[0]], props_test, caps, [target])
                scores_pw = [o for eq, o in cap.einsum[n_einsum:] if eq == "pd,wd->pw"][0]
        n_roi_calls = sum(1 for bs, _, _ in cap.samples if bs == cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
        assert n_roi_calls == 2, n_roi_calls  # pseudo-label branch, then ground-truth branch
        total = sum(losses.values())
        total.backward()
        assert set(losses) == {"loss_box_reg", "loss_classifier", "loss_mask", "loss_box_reg_pseudo",
                               "loss_classifier_pseudo", "loss_mask_pseudo"}
        out[key + "features"] = feats[0].numpy()
        put_boxes(out, key + "proposals_test", props_test, ("objectness",))
        out[key + "region_noun_scores"] = scores_pw.numpy()
        out[key + "aligned_idx"] = scores_pw.argmax(0).numpy()
        top2 = scores_pw.topk(2, dim=0).values
        out[key + "aligned_margin"] = (top2[0] - top2[1]).numpy()
        pl = pseudo[0]
        out[key + "pseudo_bbox"] = pl.bbox.numpy()
        for f in ("labels", "scores", "consistencies", "embs"):
            out[key + "pseudo_" + f] = pl.get_field(f).numpy()
        pm = pl.get_field("masks").get_mask_tensor()
        pm = pm[None] if pm.dim() == 2 else pm
        out[key + "pseudo_masks_packed"] = np.packbits(pm.numpy().astype(np.bool_), axis=-1)
        # train-mode proposals of the ground-truth branch (ground-truth boxes appended, rpn/inference.py:51-74)
        with torch.no_grad():
            model.rpn.train()
            with Capture(BalancedPositiveNegativeSampler):
                props_train, _ = model.rpn(images, feats, [target])
        put_boxes(out, key + "proposals_train", props_train)
        put_samples(out, key + "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
        eps = [r for r in cap.randn if r.dim() == 5]
        assert len(eps) == 1
        out[key + "mask_eps"] = eps[0].numpy()
        out[key + "avg_uncertain"] = np.float64(model.roi_heads_student["mask"].avg_uncertain.item())
        out[key + "adaptive_lamb"] = np.float64(float(model.adaptive_lamb))
        for k, v in losses.items():
            out[key + k] = np.float64(v.item())
        put_grads(out, model, key + "grad")
        print(key, {k: round(v.item(), 6) for k, v in losses.items()}, "lamb", float(model.adaptive_lamb),
              "margin", out[key + "aligned_margin"], "eval detections", len(det), "classes", len(set(det.get_field("labels").tolist())))
    out["state_names"] = np.array([n for n, _, _ in names])
    out["state_shapes"] = np.array([",".join(map(str, s)) for _, s, _ in names])
    out["state_seeded_as"] = np.array([c for _, _, c in names])
    out["cap_vocab"] = np.array(model.cap_vocab)
    out["opts"] = np.array([str(o) for o in case.COMMON_OPTS])
    np.savez_compressed(os.path.join(HERE, "step_student.npz"), **out)


FULL_OPTS = ["MODEL.DEVICE", "cpu"]   # the shipped configuration itself: full R-50-C4, 6000 -> 1000 / 12000 -> 2000 proposals, 512 RoIs


def gen_student_full(index=0, out_name="step_student_full.npz", digests=True):
    """ONE image at BASELINE size (3 x 800 x 1333) through the reference's STGeneralizedRCNN with the SHIPPED configuration
    (full R-50-C4 -- 55 M seeded weights --, 1000 test-mode / 2000 train-mode proposals, 512 sampled RoIs per branch):
    proposals, aligned regions + margins, pseudo labels, sampler draws, mask noise, losses, gradient digests ->
    tests/golden/step_student_full.npz.  About a minute on 8 cores."""
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.language_backbone import transformers as ref_lb
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.image_list import to_image_list
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    # (what gen_student sets up when the whole script runs; needed when this generator runs alone)
    ref_import._namespace_pkg("maskrcnn_benchmark.modeling.detector",
                              os.path.join(ref_import.REF, "maskrcnn_benchmark/modeling/detector"))
    if not getattr(ref_lb.BERT, "_seeded_stand_in", False):
        ref_lb.BERT = make_bert_class(ref_lb.BERT)
    from maskrcnn_benchmark.modeling.detector import st_generalized_rcnn as st_mod

    torch.set_num_threads(os.cpu_count())
    cfg = ref_import.reference_cfg("student_teacher_mask_rcnn_uncertainty.yaml", FULL_OPTS)
    model = st_mod.STGeneralizedRCNN(cfg)
    names = load_seeded(model)
    model.class_names = list(case.SEEN_NAMES)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.train()
    c = case.image_case(index, model.cap_vocab, size=(case.FULL_H, case.FULL_W), n_gt=7, n_nouns=5)
    t = BoxList(c["boxes"].clone(), (case.FULL_W, case.FULL_H), mode="xyxy")
    t.add_field("labels", c["labels"].clone())
    t.add_field("masks", SegmentationMask(c["masks"].clone(), (case.FULL_W, case.FULL_H), mode="mask"))
    t.add_field("nn_caption", c["nn_caption"])
    t.add_field("ids_cap", c["ids_cap"].clone())
    t.add_field("is_det", "Yes")
    out, key = {}, f"img{index}_"
    with Capture(BalancedPositiveNegativeSampler) as cap:
        losses = model(c["image"][None], [t])
        with torch.no_grad():
            images = to_image_list(c["image"][None])
            feats = model.backbone(images.tensors)
            model.rpn.eval()
            props_test, _ = model.rpn(images, feats, None)
            n_einsum = len(cap.einsum)
            pseudo = model.generate_pseudo_label([feats[0]], props_test, [t.get_field("nn_caption").split("/")], [t])
            scores_pw = [o for eq, o in cap.einsum[n_einsum:] if eq == "pd,wd->pw"][0]
            model.rpn.train()
            props_train, _ = model.rpn(images, feats, [t])
    sum(losses.values()).backward()
    f = feats[0]
    out[key + "feature_stats"] = np.array([float(f.mean()), float(f.std()), float(f.abs().max())])
    idx = torch.from_numpy(_rng_index("full_features", f.numel(), 4096))
    out[key + "feature_samples"] = f.reshape(-1)[idx].numpy()
    put_boxes(out, key + "proposals_test", props_test, ("objectness",))
    put_boxes(out, key + "proposals_train", props_train)
    out[key + "aligned_idx"] = scores_pw.argmax(0).numpy()
    top2 = scores_pw.topk(2, dim=0).values
    out[key + "aligned_margin"] = (top2[0] - top2[1]).numpy()
    out[key + "aligned_scores"] = top2[0].numpy()
    pl = pseudo[0]
    out[key + "pseudo_bbox"] = pl.bbox.numpy()
    for fld in ("labels", "scores", "consistencies", "embs"):
        out[key + "pseudo_" + fld] = pl.get_field(fld).numpy()
    pm = pl.get_field("masks").get_mask_tensor()
    out[key + "pseudo_masks_packed"] = np.packbits((pm[None] if pm.dim() == 2 else pm).numpy().astype(np.bool_), axis=-1)
    put_samples(out, key + "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
    assert out[key + "roi_sample_count"] == 2
    eps = [r for r in cap.randn if r.dim() == 5]
    assert len(eps) == 1
    out[key + "mask_eps"] = eps[0].numpy()
    out[key + "avg_uncertain"] = np.float64(model.roi_heads_student["mask"].avg_uncertain.item())
    out[key + "adaptive_lamb"] = np.float64(float(model.adaptive_lamb))
    for k, v in losses.items():
        out[key + k] = np.float64(v.item())
    if digests:   # (the second image's file serves the two-image combination test: outputs of the frozen half, draws, losses)
        put_grads(out, model, key + "grad")
        out["state_names"] = np.array([n for n, _, _ in names])
        out["state_shapes"] = np.array([",".join(map(str, s)) for _, s, _ in names])
        out["state_seeded_as"] = np.array([c_ for _, _, c_ in names])
    print("full size", {k: round(v.item(), 6) for k, v in losses.items()}, "proposals", len(props_test[0]), len(props_train[0]),
          "positives", [int(out[key + f"roi_sample{b}_pos"].sum()) for b in (0, 1)], "margins", out[key + "aligned_margin"],
          "features mean/std/max", out[key + "feature_stats"])
    if not digests:
        for k in [k for k in out if k.startswith(key + "feature_samples")]:
            del out[k]
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    torch.set_num_threads(1)


def gen_student_full_second():
    """The SECOND image of the bench's two-image batch at BASELINE size through the reference's one-image-per-call class ->
    tests/golden/step_student_full_img1.npz; with step_student_full.npz (image 0) the product's 2-image batch at 3 x 800 x 1333
    is held to the combination of the two reference runs (tests/test_step_golden.py)."""
    gen_student_full(index=1, out_name="step_student_full_img1.npz", digests=False)


def gen_teacher_full():
    """The two-image batch at BASELINE size through the reference's GeneralizedRCNN with the SHIPPED zeroshot_mask.yaml (trunk
    trainable from layer2, RPN trained: 12000 -> 2000 train-mode proposals, 256 RPN / 512 RoI samples per image): sampler
    draws, proposals, the five losses, gradient digests -> tests/golden/step_teacher_full.npz.  RoIAlign backward = the oracle's
    (see gen_teacher).  A few minutes on 8 cores."""
    import oracle
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.detector.generalized_rcnn import GeneralizedRCNN
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.image_list import to_image_list
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    import maskrcnn_benchmark.layers  # noqa: F401

    ref_roi_align = sys.modules["maskrcnn_benchmark.layers.roi_align"]

    class _CWithBackward:
        def __getattr__(self, name):
            return getattr(sys.modules["maskrcnn_benchmark._C"], name)

        @staticmethod
        def roi_align_backward(grad, rois, scale, ph, pw, n, c, h, w, sampling_ratio):
            return oracle.roi_align_backward(grad, rois, scale, ph, pw, n, c, h, w, sampling_ratio)

    ref_roi_align._C = _CWithBackward()
    torch.set_num_threads(os.cpu_count())
    cfg = ref_import.reference_cfg("zeroshot_mask.yaml", FULL_OPTS)
    model = GeneralizedRCNN(cfg)
    names = load_seeded(model)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.train()
    cases = [case.image_case(i, ["-"] * 1203, size=(case.FULL_H, case.FULL_W), n_gt=7, n_nouns=5) for i in range(2)]
    targets = []
    for c in cases:
        t = BoxList(c["boxes"].clone(), (case.FULL_W, case.FULL_H), mode="xyxy")
        t.add_field("labels", c["labels"].clone())
        t.add_field("masks", SegmentationMask(c["masks"].clone(), (case.FULL_W, case.FULL_H), mode="mask"))
        targets.append(t)
    images = torch.stack([c["image"] for c in cases])
    out = {}
    with Capture(BalancedPositiveNegativeSampler) as cap:
        losses = model(images, targets)
    sum(losses.values()).backward()
    put_samples(out, "rpn_sample", cap, cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE)
    put_samples(out, "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
    assert out["rpn_sample_count"] == 2 and out["roi_sample_count"] == 2
    for i in range(2):   # 63 000 anchors per image: store the drawn indices, not the masks
        for kind in ("pos", "neg"):
            m = out.pop(f"rpn_sample{i}_{kind}")
            out[f"rpn_sample{i}_{kind}_index"] = np.nonzero(m)[0].astype(np.int32)
        out[f"rpn_sample{i}_anchors"] = np.int64(m.shape[0])
    with torch.no_grad():
        il = to_image_list(images)
        feats = model.backbone(il.tensors)
        with Capture(BalancedPositiveNegativeSampler):
            props, _ = model.rpn(il, feats, targets)
    f = feats[0]
    out["feature_stats"] = np.array([float(f.mean()), float(f.std()), float(f.abs().max())])
    out["feature_samples"] = f.reshape(-1)[torch.from_numpy(_rng_index("full_features", f.numel(), 4096))].numpy()
    put_boxes(out, "proposals_train", props)
    for k, v in losses.items():
        out[k] = np.float64(v.item())
    put_grads(out, model)
    out["state_names"] = np.array([n for n, _, _ in names])
    out["state_shapes"] = np.array([",".join(map(str, s)) for _, s, _ in names])
    out["state_seeded_as"] = np.array([c_ for _, _, c_ in names])
    print("teacher, full size", {k: round(v.item(), 6) for k, v in losses.items()}, "proposals", [len(p) for p in props],
          "positives", [int(out[f"roi_sample{b}_pos"].sum()) for b in (0, 1)], [len(out[f"rpn_sample{b}_pos_index"]) for b in (0, 1)])
    np.savez_compressed(os.path.join(HERE, "step_teacher_full.npz"), **out)
    torch.set_num_threads(1)


def _rng_index(name, numel, n):
    return np.sort(np.random.default_rng([zlib.crc32(name.encode()), 0]).choice(numel, n, replace=False)).astype(np.int64)


VARIANTS = {  # configuration switches of STGeneralizedRCNN.forward's loss composition (st_generalized_rcnn.py:332-361) ...
    "no_reweight": ["MODEL.REWEIGHT", False],                        # pseudo losses x LAMBDA_PSEUDO_LABEL (0.1), mask too
    "no_pseudo_mask": ["MODEL.NO_PSEUDO_MASK", True],                # loss_mask_pseudo x 0
    "no_uncertainty": ["MODEL.UNCERTAINTY", False],                  # no sigma branch, no noise; x LAMBDA_PSEUDO_LABEL
    "lambda_half": ["MODEL.REWEIGHT", False, "MODEL.LAMBDA_PSEUDO_LABEL", 0.5],
    # ... and of the optimisation step around it (engine/trainer.py:117,135-141; solver/build.py:8-37)
    "sigma_lr": ["SOLVER.UNCERTAINTY_LR_FACTOR", 0.25],              # uncertain_pred's own learning-rate factor
    "clip_grad": ["SOLVER.CLIP_GRAD_NORM_AT", 5.0],                  # total gradient norm here is ~40: the clip is active
    "accumulate2": ["SOLVER.GRADIENT_ACCUMULATION_STEPS", 2],        # two micro-steps, losses / 2, one optimizer step
}


def reference_iteration(cfg, model, optimizer, scheduler, run_forward, iteration, zero_grad=None):
    """One pass of the reference's loop body, engine/trainer.py:110-141 (amp at O0 is the identity)."""
    loss_dict = run_forward()
    losses = sum(loss for loss in loss_dict.values())
    losses = losses / float(cfg.SOLVER.GRADIENT_ACCUMULATION_STEPS)
    losses.backward()
    if iteration % cfg.SOLVER.GRADIENT_ACCUMULATION_STEPS == 0:
        if cfg.SOLVER.CLIP_GRAD_NORM_AT > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.CLIP_GRAD_NORM_AT)
        optimizer.step()
        scheduler.step()
        (zero_grad or optimizer.zero_grad)()
    return loss_dict


def gen_student_variants():
    """The same image-0 step under the switches above: the six losses (and what the product needs to replay the run) ->
    tests/golden/step_student_variants.npz."""
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.detector import st_generalized_rcnn as st_mod
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    out = {}
    for name, opts in VARIANTS.items():
        cfg = ref_import.reference_cfg("student_teacher_mask_rcnn_uncertainty.yaml", list(case.COMMON_OPTS) + opts)
        model = st_mod.STGeneralizedRCNN(cfg)
        load_seeded(model)
        model.class_names = list(case.SEEN_NAMES)
        model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
        model.train()
        from maskrcnn_benchmark.solver import make_lr_scheduler, make_optimizer
        c = case.image_case(0, model.cap_vocab)
        target = make_target(BoxList, SegmentationMask, c, True)
        optimizer = make_optimizer(cfg, model)            # tools/train_net.py:58-59
        scheduler = make_lr_scheduler(cfg, optimizer)
        by_id = {id(p): n for n, p in model.named_parameters()}
        key = name + "_"
        out[key + "group_names"] = np.array([by_id[id(g["params"][0])] for g in optimizer.param_groups])
        out[key + "group_lr_wd"] = np.array([[g["initial_lr"], g["weight_decay"]] for g in optimizer.param_groups])
        k_acc = cfg.SOLVER.GRADIENT_ACCUMULATION_STEPS
        before, grads = None, {}
        with Capture(BalancedPositiveNegativeSampler) as cap:
            for it in range(1, k_acc + 1):
                if it == k_acc:  # keep the gradients the optimizer is about to consume (zero_grad follows the step)
                    step, zero = optimizer.step, optimizer.zero_grad

                    def recording_step():
                        for n, p in model.named_parameters():
                            if p.grad is not None:
                                grads[n] = p.grad.detach().clone()
                        return step()

                    optimizer.step = recording_step
                losses = reference_iteration(cfg, model, optimizer, scheduler, lambda: model(c["image"][None], [target]), it)
                if before is None:  # the student heads as the first forward left them (teacher copy at iteration 0)
                    before = {n: p.detach().clone() for n, p in model.named_parameters()}
                    if k_acc == 1:   # (the step already happened: undo is not possible -- recompute "before" from the teacher)
                        before = None
        if before is None:
            teacher = dict(model.roi_heads.named_parameters())
            before = {n: (teacher[n[len("roi_heads_student."):]] if n.startswith("roi_heads_student.") else p).detach().clone()
                      for n, p in model.named_parameters()}
        put_samples(out, key + "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
        assert out[key + "roi_sample_count"] == 2 * k_acc
        eps = [r for r in cap.randn if r.dim() == 5]
        assert len(eps) == (k_acc if cfg.MODEL.UNCERTAINTY else 0)
        for i, e in enumerate(eps):
            out[key + f"mask_eps{i}"] = e.numpy()
        for k, v in losses.items():   # of the last micro-step
            out[key + k] = np.float64(v.item())
        names = []
        for n, g in grads.items():    # what optimizer.step() consumed (accumulated and clipped)
            dg = case.grad_digest(n, g, n=case.VARIANT_DIGEST)
            out[f"{key}grad:{n}:values"], out[f"{key}grad:{n}:norm_sum"] = dg["values"], np.array([dg["norm"], dg["sum"]])
            names.append(n)
        out[key + "grad_names"] = np.array(names)
        for n, p in model.named_parameters():   # the update itself: parameter after the step minus before
            if n in grads:
                dd = case.grad_digest(n, p.detach() - before[n], n=case.VARIANT_DIGEST)
                out[f"{key}delta:{n}:values"], out[f"{key}delta:{n}:norm_sum"] = dd["values"], np.array([dd["norm"], dd["sum"]])
        out[key + "delta_names"] = np.array(names)
        out[key + "lr_after"] = np.array([g["lr"] for g in optimizer.param_groups])
        out[key + "opts"] = np.array([str(o) for o in opts])
        print(name, {k: round(v.item(), 6) for k, v in losses.items()}, "lr", optimizer.param_groups[0]["lr"])
    # the schedule itself (solver/lr_scheduler.py:10-52) at a few iterations of the shipped student configuration
    probe = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=cfg.SOLVER.BASE_LR)
    sched = make_lr_scheduler(cfg, probe)
    its = [0, 1, 2, 100, 499, 500, 501, 19999, 20000, 20001, 49999, 50000, 69999]
    lrs = []
    for it in its:
        sched.last_epoch = it
        lrs.append(sched.get_lr()[0])
    out["schedule_iterations"], out["schedule_lr"] = np.array(its), np.array(lrs)
    np.savez_compressed(os.path.join(HERE, "step_student_variants.npz"), **out)


def gen_teacher():
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.detector.generalized_rcnn import GeneralizedRCNN
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    # The one op of the teacher step the reference cannot run on a CPU (csrc/ROIAlign.h:44 "Not implemented on the CPU"):
    # the RoIAlign backward is served by the oracle's restatement of ROIAlign_cuda.cu:178-254, which tests/test_oracle.py
    # pins as the exact adjoint of the reference's own compiled forward kernel.
    import oracle
    import maskrcnn_benchmark.layers  # noqa: F401
    ref_roi_align = sys.modules["maskrcnn_benchmark.layers.roi_align"]  # (the package re-exports a function of this name)

    class _CWithBackward:
        def __getattr__(self, name):
            return getattr(sys.modules["maskrcnn_benchmark._C"], name)

        @staticmethod
        def roi_align_backward(grad, rois, scale, ph, pw, n, c, h, w, sampling_ratio):
            return oracle.roi_align_backward(grad, rois, scale, ph, pw, n, c, h, w, sampling_ratio)

    ref_roi_align._C = _CWithBackward()
    cfg = ref_import.reference_cfg("zeroshot_mask.yaml", case.COMMON_OPTS)
    model = GeneralizedRCNN(cfg)
    names = load_seeded(model)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.train()
    from maskrcnn_benchmark.data.datasets.helper.lvis_v1_categories import LVIS_CATEGORIES  # noqa: F401  (names only)
    cases = [case.image_case(i, ["-"] * 1203) for i in range(2)]
    targets = [make_target(BoxList, SegmentationMask, c, False) for c in cases]
    images = torch.stack([c["image"] for c in cases])
    out = {}
    model.eval()   # evaluation branch (generalized_rcnn.py:56-73): detections of the two-image batch
    with torch.no_grad():
        dets = model(images)
    for i, det in enumerate(dets):
        out[f"eval{i}_bbox"] = det.bbox.numpy()
        for f in ("scores", "labels", "mask"):
            out[f"eval{i}_{f}"] = det.get_field(f).numpy()
    model.train()
    with Capture(BalancedPositiveNegativeSampler) as cap:
        losses = model(images, targets)
    sum(losses.values()).backward()
    assert set(losses) == {"loss_box_reg", "loss_classifier", "loss_mask", "loss_objectness", "loss_rpn_box_reg"}
    put_samples(out, "rpn_sample", cap, cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE)
    put_samples(out, "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
    assert out["rpn_sample_count"] == 2 and out["roi_sample_count"] == 2
    with torch.no_grad():
        from maskrcnn_benchmark.structures.image_list import to_image_list
        il = to_image_list(images)
        feats = model.backbone(il.tensors)
        with Capture(BalancedPositiveNegativeSampler):
            props, _ = model.rpn(il, feats, targets)
    out["features"] = feats[0].numpy()
    put_boxes(out, "proposals_train", props)
    for k, v in losses.items():
        out[k] = np.float64(v.item())
    put_grads(out, model)
    out["state_names"] = np.array([n for n, _, _ in names])
    out["state_shapes"] = np.array([",".join(map(str, s)) for _, s, _ in names])
    out["state_seeded_as"] = np.array([c for _, _, c in names])
    out["opts"] = np.array([str(o) for o in case.COMMON_OPTS])
    print("teacher", {k: round(v.item(), 6) for k, v in losses.items()})
    np.savez_compressed(os.path.join(HERE, "step_teacher.npz"), **out)


TEACHER_VARIANTS = {  # head configurations of the reference's GeneralizedRCNN other than the shipped one
    # a plain Mask R-CNN: linear classifier, per-class box regression, one mask per class
    # (roi_box_predictors.py:33-40,70-72; box_head/loss.py:156-160; mask_head/loss.py:131-141)
    "plain_mask_rcnn": ["MODEL.ROI_BOX_HEAD.EMBEDDING_BASED", False, "MODEL.CLS_AGNOSTIC_BBOX_REG", False,
                        "MODEL.CLS_AGNOSTIC_MASK", False],
    # the mask head pooling for itself instead of re-using the box head's res5 features (roi_heads.py:21-22,57-64)
    "own_mask_extractor": ["MODEL.ROI_MASK_HEAD.SHARE_BOX_FEATURE_EXTRACTOR", False],
    # trainable emb_pred (the shipped teacher freezes it, roi_box_predictors.py:52-56)
    "train_emb_pred": ["MODEL.ROI_BOX_HEAD.FREEZE_EMB_PRED", False],
}


def gen_teacher_variants():
    """The small two-image teacher step under the head configurations above -> tests/golden/step_teacher_variants.npz."""
    import oracle
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.detector.generalized_rcnn import GeneralizedRCNN
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    out = {}
    for name, opts in TEACHER_VARIANTS.items():
        cfg = ref_import.reference_cfg("zeroshot_mask.yaml", list(case.COMMON_OPTS) + opts)
        model = GeneralizedRCNN(cfg)
        names = load_seeded(model)
        if cfg.MODEL.ROI_BOX_HEAD.EMBEDDING_BASED:
            model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
        model.train()
        cases = [case.image_case(i, ["-"] * 1203) for i in range(2)]
        targets = [make_target(BoxList, SegmentationMask, c, False) for c in cases]
        images = torch.stack([c["image"] for c in cases])
        key = name + "_"
        with Capture(BalancedPositiveNegativeSampler) as cap:
            losses = model(images, targets)
        sum(losses.values()).backward()
        put_samples(out, key + "rpn_sample", cap, cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE)
        put_samples(out, key + "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
        with torch.no_grad():
            from maskrcnn_benchmark.structures.image_list import to_image_list
            il = to_image_list(images)
            with Capture(BalancedPositiveNegativeSampler):
                props, _ = model.rpn(il, model.backbone(il.tensors), targets)
        put_boxes(out, key + "proposals_train", props)
        for k, v in losses.items():
            out[key + k] = np.float64(v.item())
        gnames = []
        for n, p in model.named_parameters():
            if p.grad is not None:
                dg = case.grad_digest(n, p.grad, n=case.VARIANT_DIGEST)
                out[f"{key}grad:{n}:values"], out[f"{key}grad:{n}:norm_sum"] = dg["values"], np.array([dg["norm"], dg["sum"]])
                gnames.append(n)
        out[key + "grad_names"] = np.array(gnames)
        out[key + "state_names"] = np.array([n for n, _, _ in names])
        out[key + "state_shapes"] = np.array([",".join(map(str, s)) for _, s, _ in names])
        out[key + "state_seeded_as"] = np.array([c_ for _, _, c_ in names])
        out[key + "opts"] = np.array([str(o) for o in opts])
        print("teacher variant", name, {k: round(v.item(), 6) for k, v in losses.items()}, len(gnames), "gradients")
    np.savez_compressed(os.path.join(HERE, "step_teacher_variants.npz"), **out)


def gen_teacher_fixed_rpn():
    """GeneralizedRCNN with MODEL.RPN.DONT_TRAIN True (generalized_rcnn.py:32-35,53-54): the RPN frozen and in eval mode
    inside the training step -- test-mode proposals, three losses -> tests/golden/step_teacher_fixed_rpn.npz."""
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.detector.generalized_rcnn import GeneralizedRCNN
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    cfg = ref_import.reference_cfg("zeroshot_mask.yaml", list(case.COMMON_OPTS) + ["MODEL.RPN.DONT_TRAIN", True])
    model = GeneralizedRCNN(cfg)
    names = load_seeded(model)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.train()
    # The eval-mode RPN selects its boxes OUTSIDE no_grad (rpn/rpn.py:175-192) on features that require grad here; its in-place
    # clip (structures/bounding_box.py:216) on a view of the decoded boxes was a deprecation warning in the reference's
    # torch 1.7 and is an error today.  Box coordinates carry no gradient into the heads (RoIAlign does not differentiate its
    # rois), so running the frozen RPN under no_grad changes nothing but makes the step runnable.
    model.rpn.forward = torch.no_grad()(model.rpn.forward)
    cases = [case.image_case(i, ["-"] * 1203) for i in range(2)]
    targets = [make_target(BoxList, SegmentationMask, c, False) for c in cases]
    images = torch.stack([c["image"] for c in cases])
    out = {}
    with Capture(BalancedPositiveNegativeSampler) as cap:
        losses = model(images, targets)
    sum(losses.values()).backward()
    assert set(losses) == {"loss_box_reg", "loss_classifier", "loss_mask"}
    assert not any(p.requires_grad for p in model.rpn.parameters())
    put_samples(out, "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
    assert out["roi_sample_count"] == 2 and len(cap.samples) == 1  # the RPN's own sampler never ran
    for i in range(2):
        out[f"proposal_count{i}"] = np.int64(out[f"roi_sample{i}_pos"].shape[0])
    for k, v in losses.items():
        out[k] = np.float64(v.item())
    put_grads(out, model)
    out["state_names"] = np.array([n for n, _, _ in names])
    out["state_shapes"] = np.array([",".join(map(str, s)) for _, s, _ in names])
    out["state_seeded_as"] = np.array([c for _, _, c in names])
    print("teacher, fixed RPN", {k: round(v.item(), 6) for k, v in losses.items()}, [int(out[f"proposal_count{i}"]) for i in range(2)])
    np.savez_compressed(os.path.join(HERE, "step_teacher_fixed_rpn.npz"), **out)


def pretraining_checkpoint_keys(model_state):
    """A synthetic caption-pretraining checkpoint for ``model_state`` (name -> shape): what the README's pretrained OVR-CNN
    weights look like from the loader's side -- a DistributedDataParallel-wrapped model (``module.``) with a ResNet-50 body
    INCLUDING layer4 (the res5 heads take their weights from it by suffix match), the grounding head's vision-to-language
    projection (-> ``emb_pred``), and pieces no detector key matches.  {checkpoint key: shape}."""
    ck = {}
    for k, shape in model_state.items():
        if k.startswith("backbone.body."):
            ck["module." + k] = shape
        elif k.startswith("roi_heads.box.feature_extractor.head.layer4."):
            ck["module.backbone.body." + k[len("roi_heads.box.feature_extractor.head."):]] = shape
        elif k.startswith("roi_heads.box.predictor.emb_pred."):
            ck["module.mmss_heads.GroundingHead.v2l_projection." + k.rsplit(".", 1)[1]] = shape
    ck["module.mmss_heads.TransformerHead.heads.cls.weight"] = (7, 5)
    ck["module.mmss_heads.GroundingHead.t2v_unused.bias"] = (3,)
    return ck


def gen_checkpoint_map():
    """Which checkpoint tensor the reference's ``DetectronCheckpointer`` (utils/checkpoint.py:103-131: prefix strip, key
    rewrites; utils/model_serialization.py:10-89: longest-suffix match) puts into which model tensor, built exactly as
    tools/train_net.py:78-86 builds it from the shipped yaml files -> tests/golden/step_checkpoint_map.json."""
    import json

    import torch.hub
    if not hasattr(torch.hub, "_download_url_to_file"):   # utils/model_zoo.py:5-12 imports the private name of torch < 1.12;
        torch.hub._download_url_to_file = torch.hub.download_url_to_file  # the download path is never taken here
    from maskrcnn_benchmark.modeling.detector import st_generalized_rcnn as st_mod
    from maskrcnn_benchmark.modeling.detector.generalized_rcnn import GeneralizedRCNN
    from maskrcnn_benchmark.utils.checkpoint import DetectronCheckpointer

    out = {}
    for name, yaml_name, cls in (("student", "student_teacher_mask_rcnn_uncertainty.yaml", st_mod.STGeneralizedRCNN),
                                 ("teacher", "zeroshot_mask.yaml", GeneralizedRCNN)):
        cfg = ref_import.reference_cfg(yaml_name, case.COMMON_OPTS)
        model = cls(cfg)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        ck = pretraining_checkpoint_keys(shapes)
        with torch.no_grad():
            for v in model.state_dict().values():
                v.fill_(-1)
        loaded = {k: torch.full(shape, float(i)) for i, (k, shape) in enumerate(ck.items())}
        chk = DetectronCheckpointer(cfg, model, None, None, "", False, backbone_prefix=cfg.MODEL.BACKBONE_PREFIX,
                                    load_emb_pred_from=(cfg.MODEL.MMSS_HEAD.DEFAULT_HEAD if cfg.MODEL.LOAD_EMB_PRED_FROM_MMSS_HEAD
                                                        else None), load_classifier=cfg.MODEL.LOAD_CLASSIFIER,
                                    replace_substr_dict={})
        chk._load_model({"model": loaded})
        keys = list(ck)
        mapping = {}
        for k, v in model.state_dict().items():
            val = float(v.reshape(-1)[0])
            assert bool((v == val).all())
            mapping[k] = keys[int(val)] if val >= 0 else None
        out[name] = {"checkpoint": {k: list(s) for k, s in ck.items()}, "loaded_from": mapping,
                     "backbone_prefix": cfg.MODEL.BACKBONE_PREFIX, "default_head": cfg.MODEL.MMSS_HEAD.DEFAULT_HEAD}
        n_hit = sum(1 for v in mapping.values() if v is not None)
        print("checkpoint map", name, len(ck), "checkpoint keys ->", n_hit, "of", len(mapping), "model tensors")
    with open(os.path.join(HERE, "step_checkpoint_map.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def gen_uncertainty_freeze():
    """MODEL.UNCERTAINTY_TRAIN_ITER (st_generalized_rcnn.py:77,197-199,402-406, defaults.py:45): the sigma branch stops training
    when the model's own counter reaches it -- the counter is bumped twice by the first forward, so 3 means "inside the second
    forward", whose backward still reaches the branch (its graph was built before the switch).  Four iterations of the reference
    loop.  The reference's pinned torch 1.7.1 (and apex) zero gradients in place: ``optimizer.zero_grad()`` leaves the frozen
    branch a ZERO gradient, so weight decay and momentum keep moving it; this container's torch would set the gradient to None
    and skip it -- the loop below asks for the pinned behaviour explicitly (``set_to_none=False``)."""
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    from maskrcnn_benchmark.modeling.language_backbone import transformers as ref_lb
    from maskrcnn_benchmark.solver import make_lr_scheduler, make_optimizer
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    ref_import._namespace_pkg("maskrcnn_benchmark.modeling.detector",
                              os.path.join(ref_import.REF, "maskrcnn_benchmark/modeling/detector"))
    if not getattr(ref_lb.BERT, "_seeded_stand_in", False):
        ref_lb.BERT = make_bert_class(ref_lb.BERT)
    from maskrcnn_benchmark.modeling.detector import st_generalized_rcnn as st_mod

    opts = ["MODEL.UNCERTAINTY_TRAIN_ITER", 3]
    cfg = ref_import.reference_cfg("student_teacher_mask_rcnn_uncertainty.yaml", list(case.COMMON_OPTS) + opts)
    model = st_mod.STGeneralizedRCNN(cfg)
    load_seeded(model)
    model.class_names = list(case.SEEN_NAMES)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.train()
    c = case.image_case(0, model.cap_vocab)
    target = make_target(BoxList, SegmentationMask, c, True)
    optimizer = make_optimizer(cfg, model)
    scheduler = make_lr_scheduler(cfg, optimizer)
    watched = ["roi_heads_student.mask.predictor.uncertain_pred.weight", "roi_heads_student.mask.predictor.uncertain_pred.bias",
               "roi_heads_student.mask.predictor.mask_fcn_logits.weight", "roi_heads_student.box.predictor.emb_pred.weight"]
    params = dict(model.named_parameters())
    out = {"opts": np.array([str(o) for o in opts]), "watched": np.array(watched), "iterations": np.int64(4)}
    with Capture(BalancedPositiveNegativeSampler) as cap:
        for it in range(1, 5):
            state = {}

            def run_forward():
                losses = model(c["image"][None], [target])
                state["flag"] = params[watched[0]].requires_grad      # after the forward, before the backward
                state["iter"] = int(model.iter)
                state["before"] = {n: params[n].detach().clone() for n in watched}   # (the first forward copies the teacher heads)
                return losses

            losses = reference_iteration(cfg, model, optimizer, scheduler, run_forward, it,
                                         zero_grad=lambda: optimizer.zero_grad(set_to_none=False))
            key = f"it{it}_"
            out[key + "sigma_trainable_after_forward"] = np.bool_(state["flag"])
            out[key + "model_iter_after_forward"] = np.int64(state["iter"])
            for k, v in losses.items():
                out[key + k] = np.float64(v.item())
            out[key + "delta_names"] = np.array(watched)
            for n in watched:
                dd = case.grad_digest(n, params[n].detach() - state["before"][n], n=case.VARIANT_DIGEST)
                out[f"{key}delta:{n}:values"], out[f"{key}delta:{n}:norm_sum"] = dd["values"], np.array([dd["norm"], dd["sum"]])
            print("freeze it", it, "sigma trainable", state["flag"], "iter", state["iter"],
                  "|d sigma.w|", float(out[f"{key}delta:{watched[0]}:norm_sum"][0]), {k: round(v.item(), 5) for k, v in losses.items()})
    put_samples(out, "roi_sample", cap, cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
    assert out["roi_sample_count"] == 8
    eps = [r for r in cap.randn if r.dim() == 5]
    assert len(eps) == 4
    for i, e in enumerate(eps):
        out[f"mask_eps{i}"] = e.numpy()
    np.savez_compressed(os.path.join(HERE, "step_student_freeze.npz"), **out)


def gen_gt_box_eval():
    """MODEL.GT_BOX_EVAL True (roi_heads.py:25-49, box_head/inference.py:82-89,177-181): the evaluation branch of both
    detectors with the ground-truth boxes as the proposals of the heads -- one detection per ground-truth box, scored
    ``prob[own class] + 1.1``, class-major order, masks of the own class."""
    from maskrcnn_benchmark.modeling.language_backbone import transformers as ref_lb
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    ref_import._namespace_pkg("maskrcnn_benchmark.modeling.detector",
                              os.path.join(ref_import.REF, "maskrcnn_benchmark/modeling/detector"))
    if not getattr(ref_lb.BERT, "_seeded_stand_in", False):
        ref_lb.BERT = make_bert_class(ref_lb.BERT)
    from maskrcnn_benchmark.modeling.detector.generalized_rcnn import GeneralizedRCNN
    from maskrcnn_benchmark.modeling.detector import st_generalized_rcnn as st_mod

    opts = list(case.COMMON_OPTS) + ["MODEL.GT_BOX_EVAL", True]
    out = {}
    cfg = ref_import.reference_cfg("student_teacher_mask_rcnn_uncertainty.yaml", opts)
    model = st_mod.STGeneralizedRCNN(cfg)
    load_seeded(model)
    model.class_names = list(case.SEEN_NAMES)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.eval()
    cases = [case.image_case(i, model.cap_vocab, n_gt=4) for i in range(2)]
    targets = [make_target(BoxList, SegmentationMask, c, True) for c in cases]
    with torch.no_grad():
        dets = model(torch.stack([c["image"] for c in cases]), targets)
    for i, det in enumerate(dets):
        assert len(det) == len(targets[i])
        out[f"student{i}_bbox"] = det.bbox.numpy()
        for f in ("scores", "labels", "mask"):
            out[f"student{i}_{f}"] = det.get_field(f).numpy()
    cfg = ref_import.reference_cfg("zeroshot_mask.yaml", opts)
    model = GeneralizedRCNN(cfg)
    load_seeded(model)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.eval()
    cases = [case.image_case(i, ["-"] * 1203, n_gt=4) for i in range(2)]
    targets = [make_target(BoxList, SegmentationMask, c, False) for c in cases]
    with torch.no_grad():
        dets = model(torch.stack([c["image"] for c in cases]), targets)
    for i, det in enumerate(dets):
        assert len(det) == len(targets[i])
        out[f"teacher{i}_bbox"] = det.bbox.numpy()
        for f in ("scores", "labels", "mask"):
            out[f"teacher{i}_{f}"] = det.get_field(f).numpy()
        print("gt_box_eval teacher", i, det.get_field("labels").tolist(), det.get_field("scores").tolist())
    # the same pass with MODEL.ROI_MASK_HEAD.POSTPROCESS_MASKS (mask_head/inference.py:56-57,207-213): masks pasted into the
    # image at the threshold instead of the 14 x 14 probabilities
    cfg = ref_import.reference_cfg("zeroshot_mask.yaml", opts + ["MODEL.ROI_MASK_HEAD.POSTPROCESS_MASKS", True,
                                                                 "MODEL.ROI_MASK_HEAD.POSTPROCESS_MASKS_THRESHOLD", 0.45])
    model = GeneralizedRCNN(cfg)
    load_seeded(model)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.eval()
    with torch.no_grad():
        dets = model(torch.stack([c["image"] for c in cases]), targets)
    for i, det in enumerate(dets):
        m = det.get_field("mask")
        assert m.dtype == torch.bool and m.shape == (len(targets[i]), 1, case.IMAGE_H, case.IMAGE_W)
        out[f"pasted{i}_mask_packed"] = np.packbits(m.numpy(), axis=-1)
        out[f"pasted{i}_bbox"] = det.bbox.numpy()
        print("pasted masks", i, m.flatten(1).sum(1).tolist())
    np.savez_compressed(os.path.join(HERE, "step_gt_box_eval.npz"), **out)


def main():
    torch.set_num_threads(1)
    torch.manual_seed(20260101)
    ref_import.install()
    torch.Tensor.cuda = lambda self, *a, **k: self  # box_head/loss.py:42,173, language_backbone/transformers.py:60
    gens = [gen_student, gen_student_variants, gen_teacher, gen_teacher_variants, gen_teacher_fixed_rpn, gen_checkpoint_map,
            gen_student_full, gen_teacher_full, gen_gt_box_eval, gen_uncertainty_freeze, gen_student_full_second]
    only = sys.argv[1:]  # e.g. ``make_step_golden.py gen_gt_box_eval``: that file alone (each generator builds its own models)
    for g in gens:
        if not only or g.__name__ in only:
            if g in (gen_uncertainty_freeze, gen_student_full_second):
                # their committed files were made by runs of these generators alone (fresh seed): the sampler draws and the mask
                # noise it records follow the global generator, so a full run re-seeds here to reproduce it byte for byte
                torch.manual_seed(20260101)
            g()
    for f in ("step_student.npz", "step_student_variants.npz", "step_teacher.npz", "step_teacher_fixed_rpn.npz", "step_student_full.npz",
              "step_student_full_img1.npz", "step_teacher_full.npz", "step_teacher_variants.npz", "step_gt_box_eval.npz", "step_student_freeze.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
