"""Meta-architectures: the teacher (``GeneralizedRCNN``) and the student-teacher step
(``STGeneralizedRCNN``) with cross-modal pseudo-labelling.

Counterparts: maskrcnn_benchmark/modeling/detector/generalized_rcnn.py:16-73,
detector/st_generalized_rcnn.py:27-88 (construction / freezing), :164-177 (``combine_embs``),
:190-216 (``prepare_model`` / ``extract_emb``), :218-275 (``generate_pseudo_label``),
:284-418 (``forward``), detector/detectors.py:7-16.

What differs from the reference, on purpose (DESIGN.md "Deviations"):
  * text embeddings come from a fixed matrix (``set_caption_vocab``) instead of a per-iteration BERT
    tokenizer + embedding lookup: the vocabulary is constant, so they are computed once
    (SURVEY 8f-3); noun embeddings of an image are rows of that matrix (``ids_cap``);
  * per-image slicing is by image index into the feature batch (the reference indexes the list of
    feature LEVELS with the image index and only works at 1 image / process -- SURVEY D4);
  * the student heads run on any device the tensors live on (no ``.cuda()``), errors are not swallowed
    by a bare ``except`` (a failure in the student pass raises instead of silently training on the
    dummy loss);
  * the exemplar bank is kept as an (empty) dict: ``update_exemplars`` is commented out upstream
    (st_generalized_rcnn.py:325-326), so ``combine_embs`` reduces to row normalisation.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn

from .. import _C
from .backbone import Backbone
from .roi_heads import CombinedROIHeads, Masker
from .rpn import RPNModule
from .language_backbone import BERT, normalize_class_names
from .structures import BoxList, PastedMasks, to_image_list


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = Backbone(cfg)
        self.rpn = RPNModule(cfg, self.backbone.out_channels)
        self.roi_heads = CombinedROIHeads(cfg, self.backbone.out_channels)
        # generalized_rcnn.py:32-35,53-54: MODEL.RPN.DONT_TRAIN freezes the RPN and keeps it in eval mode inside a training
        # step -- test-mode proposal selection (no ground-truth boxes appended), no RPN losses
        self.fix_rpn = bool(cfg.MODEL.RPN.DONT_TRAIN)
        if self.fix_rpn:
            for p in self.rpn.parameters():
                p.requires_grad = False
        self.heads_as_one_branch = True  # False = ``CombinedROIHeads.forward`` in training too (cross-check / A-B switch)

    def set_class_embeddings(self, embs):
        self.roi_heads["box"].predictor.set_class_embeddings(embs)

    def forward_frozen(self, images, targets=None):
        """The part of a training step no trainable parameter feeds: stem + the frozen leading stages of the trunk
        (FREEZE_CONV_BODY_AT).  ``PipelinedTrainer`` runs it for the next batch on a side stream beside this batch's
        backward; ``forward_student`` continues from it with the values the un-split ``forward`` computes."""
        images = to_image_list(images)
        return {"images": images, "prefix": self.backbone.body.forward_prefix(images.tensors)}

    def forward_student(self, frozen, targets):
        return self.forward(frozen["images"], targets, prefix=frozen["prefix"])

    def forward(self, images, targets=None, prefix=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        if self.fix_rpn:
            self.rpn.eval()
        images = to_image_list(images)
        features = self.backbone(images.tensors) if prefix is None else self.backbone.body(images.tensors, prefix=prefix)
        if self.rpn.runs_ahead(features):  # trainable RPN on a device: its backward runs ahead on a second stream
            proposals, proposal_losses, features = self.rpn.forward_ahead(images, features, targets)
        else:
            proposals, proposal_losses = self.rpn(images, features, targets)
        heads = self.roi_heads
        predictor = heads["box"].predictor
        if (self.training and self.heads_as_one_branch and heads.branches_batchable(features[0])
                and torch.is_tensor(getattr(predictor, "cls_score", None))):
            # the student's batched-branch form with ONE branch: the last res5 block hands the positives' maps to the mask
            # head itself, and their gradient enters its backward as dense maps -- no 411 MB zero tensor + scatter + dense
            # read in front of the res5 backward
            result = None
            (detector_losses,) = heads.forward_branches(features[0], [dict(
                image_ids=list(range(len(proposals))), proposals=proposals, targets=targets, cls_embs=predictor.cls_score)])
        else:
            _, result, detector_losses = heads(features, proposals, targets, is_eval_func=True)
        if self.training:
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            return losses
        return result


class STGeneralizedRCNN(nn.Module):
    LOSS_NAMES = ("loss_box_reg", "loss_classifier", "loss_mask")

    def __init__(self, cfg):
        super().__init__()
        self.backbone = Backbone(cfg)
        self.rpn = RPNModule(cfg, self.backbone.out_channels)
        self.roi_heads = CombinedROIHeads(cfg, self.backbone.out_channels, is_teacher=True)
        self.roi_heads_student = CombinedROIHeads(cfg, self.backbone.out_channels)
        assert cfg.MODEL.RPN.DONT_TRAIN, "the student-teacher step keeps the RPN frozen"
        self.lambda_exemplar = nn.Parameter(torch.zeros(1), requires_grad=True)
        self.mask_on = cfg.MODEL.MASK_ON
        for m in (self.rpn, self.backbone, self.roi_heads):
            for p in m.parameters():
                p.requires_grad = False
        self.exemplars = {}
        self.masker = Masker(threshold=0.5, padding=1)
        self.lambda_pseudo_label = cfg.MODEL.LAMBDA_PSEUDO_LABEL
        self.adaptive_lamb = "N/A"
        self.uncertainty = cfg.MODEL.UNCERTAINTY
        self.resume = cfg.MODEL.RESUME
        self.uncertainty_train_iter = cfg.MODEL.UNCERTAINTY_TRAIN_ITER
        self.no_pseudo_mask = cfg.MODEL.NO_PSEUDO_MASK
        self.reweight = cfg.MODEL.REWEIGHT
        self.iter = 0
        self.cap_embs = None  # [V, emb_dim] unit-norm caption-vocabulary (LVIS) embeddings
        self.cap_vocab = None  # ... or the vocabulary's names, embedded through ``self.bert`` (set_caption_vocab_names)
        # st_generalized_rcnn.py:45: the frozen BERT word-embedding table + tokenizer (state-dict key ``bert.embeddings``)
        self.bert = BERT(cfg)

    def never_used_parameters(self):
        """Trainable parameters no loss depends on (``update_exemplars`` is commented out upstream): the gradient
        reducer does not wait for their hooks (engine/comm.py)."""
        return [self.lambda_exemplar]

    # -- text side ----------------------------------------------------------------------------------
    def set_caption_vocab(self, embs):
        """Embeddings of the caption vocabulary (the reference re-extracts them from BERT every
        iteration, st_generalized_rcnn.py:190-191,202-209)."""
        self.cap_embs = F.normalize(embs.float(), dim=-1)

    def set_caption_vocab_names(self, names):
        """The caption vocabulary as strings (the reference hard-wires the 1203 LVIS category names,
        st_generalized_rcnn.py:71-76): embedded through the BERT word-embedding table by ``prepare_model``
        (st_generalized_rcnn.py:190-191) -- once per table version, not once per iteration."""
        self.cap_vocab = normalize_class_names(names)
        self.cap_embs = None

    def extract_emb(self, words):
        """st_generalized_rcnn.py:202-209."""
        return self.bert.extract_emb(words)

    def set_class_embeddings(self, embs):
        """Seen-class matrix with the all-zero background row 0 (engine/trainer.py:85-90)."""
        self.roi_heads["box"].predictor.set_class_embeddings(embs)
        # own reference: generate_pseudo_label swaps the teacher predictor's matrix for a dummy while it runs, and the
        # frozen half may be running on another thread (engine/trainer.py::PipelinedTrainer)
        self._seen_cls = self.roi_heads["box"].predictor.cls_score

    def _noun_embs(self, target):
        """[W, D] embeddings of an image's caption nouns: given (``cap_embs`` field), extracted from the strings
        (``nn_caption`` = 'noun/noun/...', st_generalized_rcnn.py:318,242) or rows of the vocabulary matrix (``ids_cap``)."""
        if target.has_field("cap_embs"):
            return target.get_field("cap_embs")
        by_text = self.cap_vocab is not None or self.cap_embs is None  # a vocabulary given as a matrix has no strings
        if by_text and target.has_field("nn_caption") and isinstance(target.get_field("nn_caption"), str):
            return self.bert.extract_emb(target.get_field("nn_caption").split("/"))
        return self.cap_embs[target.get_field("ids_cap")]

    def combine_embs(self, embs):
        # exemplar bank is empty (see module docstring) -> st_generalized_rcnn.py:165-166.  The matrices are constants of
        # the run: the normalised form is computed once per (tensor, version), so the predictor's operand caches hit
        key = (id(embs), embs._version, embs.device)
        cache = self.__dict__.setdefault("_combined", {})
        hit = cache.get(id(embs))
        if hit is None or hit[0] != key or hit[1] is not embs:
            if len(cache) > 8:
                cache.clear()
            hit = (key, embs, F.normalize(embs, dim=-1))
            cache[id(embs)] = hit
        return hit[2]

    def prepare_text(self):
        if self.cap_vocab is not None:
            self.cap_embs = self.bert.extract_emb(self.cap_vocab)  # cached: a no-op while the table is unchanged

    def prepare_model(self):
        self.prepare_text()
        student = self.roi_heads_student["box"].predictor
        seen = getattr(self, "_seen_cls", None)
        if seen is None:
            seen = self.roi_heads["box"].predictor.cls_score
        if student.cls_score is None:  # every pass sets the matrix it needs; this only covers first use
            student.cls_score = seen
        if self.iter == 0 and not self.resume:
            self.roi_heads_student.load_state_dict(copy.deepcopy(self.roi_heads.state_dict()), strict=False)
            self.iter += 1

    def compute_dummy_loss(self):
        loss = 0.0
        for p in self.roi_heads_student.parameters():
            loss = loss + torch.sum(p) * 0.0
        return loss

    # -- teacher: region <-> noun alignment -----------------------------------------------------------
    @torch.no_grad()
    def generate_pseudo_label(self, features, proposals, noun_embs, targets):
        teacher = self.roi_heads
        class_embs = teacher["box"].predictor.cls_score
        teacher["box"].predictor.set_class_embeddings(features[0].new_zeros((1, teacher["box"].predictor.emb_dim)))
        teacher.eval()
        package_x, results, _ = teacher(features, proposals, None, bbox_only=True)
        cls_embs = teacher["box"].predictor.embed(package_x["bbox"]).split([len(p) for p in proposals])
        pseudo_labels = []
        for emb_img, w_cap, result_img, target_img in zip(cls_embs, noun_embs, results, targets):
            if w_cap.shape[0] == 0:
                pseudo_labels.append(BoxList(emb_img.new_zeros((0, 4)), result_img.size))
                continue
            # einsum('pd,wd->pw') -> max over regions -> sigmoid, without materialising the [P,W] matrix
            aligned, prob, idx = _C.region_noun_align(emb_img, w_cap)
            pl = result_img[idx]
            pl.add_field("labels", target_img.get_field("ids_cap"))
            pl.add_field("scores", prob)
            pl.add_field("consistencies", aligned * 0.0 + 1.0)
            pl.add_field("embs", emb_img[idx])
            pseudo_labels.append(pl)
        if self.mask_on:
            _, results, _ = teacher(features, pseudo_labels, None, bbox_only=False)
            for res, pl in zip(results, pseudo_labels):
                if pl.bbox.is_cuda:
                    # the pasted image-size masks (Masker, mask_head/inference.py:124-205) are only ever cropped and
                    # resized back to 14x14 by the student's mask loss: keep (probability map, box) and let
                    # _C.project_pasted_masks evaluate the pixels it needs -- same targets, no H x W canvases
                    w, h = pl.size
                    pl.add_field("masks", PastedMasks(res.get_field("mask")[:, 0], pl.bbox, (h, w),
                                                      self.masker.threshold, self.masker.padding))
                else:
                    pl.add_field("masks", self.masker(res.get_field("mask"), pl)[:, 0])  # [W,H,W] bool
        teacher["box"].predictor.set_class_embeddings(class_embs)
        return pseudo_labels

    # -- step ----------------------------------------------------------------------------------------------
    def forward(self, images, targets=None, eps=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        features = self.backbone(images.tensors)
        student = self.roi_heads_student
        if not self.training:
            self.rpn.eval()
            proposals, _ = self.rpn(images, features, None)
            student["box"].predictor.set_class_embeddings(self.combine_embs(self.roi_heads["box"].predictor.cls_score))
            _, result, _ = student(features, proposals, targets, is_eval_func=True)
            return result

        frozen = self.forward_frozen(images, targets, features=features)
        return self.forward_student(frozen, targets, eps=eps)

    @torch.no_grad()
    def forward_frozen(self, images, targets, features=None):
        """Everything of the training step that only uses FROZEN modules (trunk, RPN, teacher heads; __init__ turns
        their gradients off): features, proposals of both branches and the teacher's pseudo labels.  It does not
        depend on the student's weights, so ``engine.trainer.PipelinedTrainer`` runs it for batch i+1 on a side
        stream while the student forward / backward of batch i occupies the main stream."""
        images = to_image_list(images)
        if features is None:
            features = self.backbone(images.tensors)
        feat = features[0]
        out = {"feat": feat}
        self.prepare_text()
        idxs_cap = [i for i, t in enumerate(targets) if t.has_field("ids_cap") and len(t.get_field("ids_cap")) > 0]
        out["idxs_cap"] = idxs_cap
        idxs_gt = [i for i, t in enumerate(targets) if t.has_field("is_det") and t.get_field("is_det") == "Yes"]
        head_out = self.rpn.head(feat) if (idxs_cap or idxs_gt) else None  # one frozen head pass for both selections
        proposals = proposals_target = None
        if idxs_cap and idxs_gt:
            # test-mode proposals (pseudo branch) and train-mode proposals (ground-truth branch) from ONE decode + NMS
            # pass per image (rpn.py::RPNPostProcessor.forward_with); the reference runs the whole RPN twice
            proposals_target, proposals = self.rpn.proposals_train_and_test(images, features, targets, head_out)
        if idxs_cap:
            if proposals is None:
                self.rpn.eval()
                proposals, _ = self.rpn(images, features, None, head_out=head_out)
            cap_features = [feat if idxs_cap == list(range(feat.shape[0])) else feat[idxs_cap]]  # no copy for "all images"
            cap_proposals = [proposals[i] for i in idxs_cap]
            cap_targets = [targets[i] for i in idxs_cap]
            noun_embs = [self._noun_embs(t) for t in cap_targets]
            out["cap_features"] = cap_features
            out["cap_proposals"] = cap_proposals
            out["pseudo_targets"] = self.generate_pseudo_label(cap_features, cap_proposals, noun_embs, cap_targets)
        out["idxs_gt"] = idxs_gt
        if idxs_gt:
            if proposals_target is None:
                self.rpn.train()
                proposals_target, _ = self.rpn(images, features, targets, compute_loss=False, head_out=head_out)
            out["gt_features"] = [feat if idxs_gt == list(range(feat.shape[0])) else feat[idxs_gt]]
            out["gt_proposals"] = [proposals_target[i] for i in idxs_gt]
        return out

    def forward_student(self, frozen, targets, eps=None):
        """The trainable half of the step: both student passes and their losses on the outputs of ``forward_frozen``."""
        student = self.roi_heads_student
        self.prepare_model()
        # the all-parameter zero loss (st_generalized_rcnn.py:277-282) only enters the graph when a branch has no
        # images; it is ~3 launches per parameter, so it is built on first use
        dummy = []

        def dummy_loss():
            if not dummy:
                dummy.append(self.compute_dummy_loss())
            return dummy[0]

        loss_pseudo, loss_gt = {}, {}
        batched = bool(frozen["idxs_cap"] and frozen["idxs_gt"]) and student.branches_batchable(frozen["feat"])
        if batched:
            # both branches share the student heads: ONE pooler + res5 pass (and one backward) over the RoIs of both
            gt_targets = [targets[i] for i in frozen["idxs_gt"]]
            loss_pseudo, loss_gt = student.forward_branches(frozen["feat"], [
                dict(image_ids=frozen["idxs_cap"], proposals=frozen["cap_proposals"], targets=frozen["pseudo_targets"],
                     cls_embs=self.combine_embs(self.cap_embs), compute_uncertain=self.uncertainty, eps=eps),
                dict(image_ids=frozen["idxs_gt"], proposals=frozen["gt_proposals"], targets=gt_targets,
                     cls_embs=self.combine_embs(self._seen_cls), compute_uncertain=False, eps=None)])
        # ---- pseudo branch: images that come with caption nouns ------------------------------------------
        if frozen["idxs_cap"]:
            if not batched:
                student["box"].predictor.set_class_embeddings(self.combine_embs(self.cap_embs))
                _, _, loss_pseudo = student(frozen["cap_features"], frozen["cap_proposals"], frozen["pseudo_targets"],
                                            compute_uncertain=self.uncertainty, eps=eps)
            for k in loss_pseudo:
                if self.uncertainty and self.reweight:
                    if "mask" not in k:
                        self.adaptive_lamb = 0.01 / student["mask"].avg_uncertain.detach()
                        loss_pseudo[k] = loss_pseudo[k] * self.adaptive_lamb
                else:
                    loss_pseudo[k] = loss_pseudo[k] * self.lambda_pseudo_label
        losses = {}
        for k in self.LOSS_NAMES:
            v = loss_pseudo[k] if k in loss_pseudo else dummy_loss()
            if "mask" in k and self.no_pseudo_mask:
                v = v * 0.0
            losses[f"{k}_pseudo"] = v

        # ---- seen-class branch: images with box / mask ground truth -----------------------------------------
        if frozen["idxs_gt"] and not batched:
            gt_targets = [targets[i] for i in frozen["idxs_gt"]]
            student["box"].predictor.set_class_embeddings(self.combine_embs(self._seen_cls))
            _, _, loss_gt = student(frozen["gt_features"], frozen["gt_proposals"], gt_targets, compute_uncertain=False)
        for k in self.LOSS_NAMES:
            losses[k] = loss_gt[k] if k in loss_gt else dummy_loss()

        self.iter += 1
        if self.uncertainty and self.iter == self.uncertainty_train_iter and self.mask_on:
            student["mask"].predictor.uncertain_pred.requires_grad_(False)
        return losses


_META_ARCHITECTURES = {"GeneralizedRCNN": GeneralizedRCNN, "STGeneralizedRCNN": STGeneralizedRCNN}


# Keys config.py carries so that the reference's yaml files merge, whose non-default values select reference subsystems
# outside the training step this package replaces (SURVEY §2 "out of scope"): refuse them instead of building the default
# silently.  (R-50-C4 / FPN switches are refused where the body and the RPN are built.)
_ONLY_VALUE = {
    "MODEL.RPN_ONLY": False, "MODEL.RETINANET_ON": False, "MODEL.KEYPOINT_ON": False,
    "MODEL.RPN.RPN_HEAD": "SingleConvRPNHead",
    "MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor",
    "MODEL.ROI_MASK_HEAD.FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor",
    "MODEL.ROI_BOX_HEAD.PREDICTOR": "FastRCNNPredictor", "MODEL.ROI_MASK_HEAD.PREDICTOR": "MaskRCNNC4Predictor",
    "MODEL.RESNETS.TRANS_FUNC": "BottleneckWithFixedBatchNorm", "MODEL.RESNETS.STEM_FUNC": "StemWithFixedBatchNorm",
    "DTYPE": "float32",
}


def check_supported(cfg):
    for key, only in _ONLY_VALUE.items():
        node = cfg
        for part in key.split("."):
            node = getattr(node, part)
        if node != only:
            raise NotImplementedError(f"{key} = {node!r}: only {only!r} is built (the hot path of the shipped configurations)")


def build_detection_model(cfg):
    check_supported(cfg)
    return _META_ARCHITECTURES[cfg.MODEL.META_ARCHITECTURE](cfg)
