"""Generates tests/golden/*.npz from the REFERENCE itself (run in the build container only).

Sources of truth:
  * oracle/_ref/ref_C.so   -- the reference's own CPU kernels (roi_align_forward, nms), built by
                              oracle/build_ref.py from /root/reference.
  * /root/reference Python -- imported with two sys.modules stubs (apex.amp.float_function =
                              identity, maskrcnn_benchmark._C = ref_C) exactly as SURVEY.md 8(c)
                              describes; used for sigmoid_focal_loss_cpu, smooth_l1_loss,
                              FrozenBatchNorm2d, the box / mask predictors and losses.
Only inputs and expected outputs are stored; no reference source text is copied.

    python tests/golden/make_golden.py          # rewrites the fixtures in place
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference"


def import_reference():
    ref_c = oracle.ref_module()
    assert ref_c is not None, "run `python oracle/build_ref.py` first"
    apex = types.ModuleType("apex")
    amp = types.ModuleType("apex.amp")
    amp.float_function = lambda f: f
    apex.amp = amp
    sys.modules["apex"] = apex
    sys.modules["apex.amp"] = amp
    sys.path.insert(0, REF)
    import maskrcnn_benchmark  # noqa: F401

    sys.modules["maskrcnn_benchmark._C"] = ref_c
    maskrcnn_benchmark._C = ref_c
    return ref_c


def rand_rois(g, r, n_img, img_w, img_h, wmin, wmax):
    b = torch.randint(0, n_img, (r, 1), generator=g).float()
    x1 = torch.rand(r, 1, generator=g) * (img_w * 0.8)
    y1 = torch.rand(r, 1, generator=g) * (img_h * 0.8)
    w = torch.rand(r, 1, generator=g) * (wmax - wmin) + wmin
    h = torch.rand(r, 1, generator=g) * (wmax - wmin) + wmin
    return torch.cat([b, x1, y1, (x1 + w).clamp(max=img_w - 1), (y1 + h).clamp(max=img_h - 1)], 1)


def gen_roi_align(ref_c):
    g = torch.Generator().manual_seed(1234)
    n, c, h, w = 2, 8, 25, 42
    x = torch.randn(n, c, h, w, generator=g)
    rois = rand_rois(g, 56, n, 672, 400, 8, 330)
    edge = torch.tensor([
        [0, -40.0, -30.0, 20.0, 25.0],      # sticks out top-left
        [1, 600.0, 350.0, 800.0, 500.0],    # sticks out bottom-right
        [0, 100.0, 100.0, 100.5, 100.2],    # degenerate -> forced 1x1
        [1, 0.0, 0.0, 671.0, 399.0],        # whole image (gh=2, gw=3)
        [0, 900.0, 900.0, 950.0, 950.0],    # fully outside: every sample invalid -> zeros
        [1, -500.0, -500.0, 1500.0, 1200.0],  # far larger than the map (gh=8, gw=9)
        [0, 333.3, 111.1, 340.9, 390.7],    # thin and tall
        [1, 15.99, 16.01, 239.99, 240.01],  # cell-boundary coordinates
    ])
    rois = torch.cat([rois, edge], 0)
    out = {"input": x.numpy(), "rois": rois.numpy(), "scale": np.float32(1 / 16)}
    for sr in (0, 2):
        out[f"out_sr{sr}"] = ref_c.roi_align_forward(x, rois, 1 / 16, 14, 14, sr).numpy()
    out["out_7x7_sr0"] = ref_c.roi_align_forward(x, rois, 1 / 16, 7, 7, 0).numpy()
    np.savez_compressed(os.path.join(HERE, "roi_align_forward.npz"), **out)


def gen_nms(ref_c):
    g = torch.Generator().manual_seed(4321)
    out = {}
    for name, k, thr in (("rpn_like", 1500, 0.7), ("dense", 700, 0.5), ("tiny", 5, 0.5), ("one", 1, 0.3)):
        xy = torch.rand(k, 2, generator=g) * torch.tensor([600.0, 360.0])
        wh = torch.rand(k, 2, generator=g) * 200 + 8
        if name == "dense":  # heavy overlap: many suppressions per survivor
            xy = xy * 0.15 + 100
        boxes = torch.cat([xy, xy + wh], 1)
        scores = torch.rand(k, generator=g)
        assert scores.unique().numel() == k  # no ties: sort order is unambiguous
        keep = ref_c.nms(boxes, scores, thr)
        out[f"{name}_boxes"], out[f"{name}_scores"] = boxes.numpy(), scores.numpy()
        out[f"{name}_thr"], out[f"{name}_keep"] = np.float32(thr), keep.numpy()
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **out)


def gen_focal():
    from maskrcnn_benchmark.layers.sigmoid_focal_loss import sigmoid_focal_loss_cpu

    g = torch.Generator().manual_seed(77)
    num, c = 257, 80
    logits = (torch.randn(num, c, generator=g) * 3).requires_grad_(True)
    targets = torch.randint(-1, c + 1, (num,), generator=g, dtype=torch.int32)
    d_losses = torch.rand(num, c, generator=g)
    # The reference's Python formula takes log(1 - sigmoid(x)) literally, which loses precision in
    # fp32 for x >~ 8; its CUDA kernel uses the stable form.  The fixture therefore stores the
    # reference formula evaluated in fp64 (fp32 inputs promoted), which both agree with.
    logits = logits.detach().double().requires_grad_(True)
    d_losses = d_losses.double()
    loss = sigmoid_focal_loss_cpu(logits, targets, 2.0, 0.25)
    (grad,) = torch.autograd.grad(loss, logits, d_losses)
    # second case: non-default gamma/alpha and a class count that is not a multiple of 4
    logits2 = (torch.randn(64, 7, generator=g) * 2).requires_grad_(True)
    targets2 = torch.randint(-1, 8, (64,), generator=g, dtype=torch.int32)
    d2 = torch.rand(64, 7, generator=g)
    logits2 = logits2.detach().double().requires_grad_(True)
    d2 = d2.double()
    loss2 = sigmoid_focal_loss_cpu(logits2, targets2, 1.5, 0.4)
    (grad2,) = torch.autograd.grad(loss2, logits2, d2)
    np.savez_compressed(os.path.join(HERE, "sigmoid_focal_loss.npz"),
                        logits=logits.detach().float().numpy(), targets=targets.numpy(),
                        d_losses=d_losses.float().numpy(), loss=loss.detach().numpy(), grad=grad.numpy(),
                        logits2=logits2.detach().float().numpy(), targets2=targets2.numpy(),
                        d_losses2=d2.float().numpy(), loss2=loss2.detach().numpy(), grad2=grad2.numpy())


# --------------------------------------------------------------------------------------------------
# Python-side components of the hot path (SURVEY.md 8a rows a6, a8, a9, a10 + RPN / anchors / paste)
# --------------------------------------------------------------------------------------------------
def _ns(**kw):
    return types.SimpleNamespace(**kw)


def _ref_cfg(uncertainty=True):
    return _ns(MODEL=_ns(
        CLS_AGNOSTIC_BBOX_REG=True, CLS_AGNOSTIC_MASK=True, UNCERTAINTY=uncertainty,
        ROI_BOX_HEAD=_ns(EMBEDDING_BASED=True, EMB_DIM=48, FREEZE_EMB_PRED=False, NUM_CLASSES=49,
                         LOSS_WEIGHT_BACKGROUND=0.2),
        ROI_MASK_HEAD=_ns(CONV_LAYERS=(24, 24, 24, 24), RESOLUTION=14),
        ROI_HEADS=_ns(FG_IOU_THRESHOLD=0.5, BG_IOU_THRESHOLD=0.5)))


def gen_heads():
    """FastRCNNPredictor / MaskRCNNC4Predictor / losses, run through the reference's own modules."""
    for m in ("cv2", "pycocotools", "pycocotools.mask"):
        sys.modules.setdefault(m, types.ModuleType(m))
    if not hasattr(np, "float"):
        np.float = float  # numpy>=1.24 dropped the alias rpn/anchor_generator.py:227-228 still uses
    from maskrcnn_benchmark.modeling.roi_heads.box_head.roi_box_predictors import FastRCNNPredictor
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.roi_mask_predictors import MaskRCNNC4Predictor
    from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import FastRCNNLossComputation
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.matcher import Matcher
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou
    from maskrcnn_benchmark.layers import smooth_l1_loss, FrozenBatchNorm2d
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator
    from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.inference import paste_mask_in_image
    import torch.nn.functional as F

    out = {}
    g = torch.Generator().manual_seed(99)
    cfg = _ref_cfg()

    # ---- box predictor (roi_box_predictors.py:62-81) for the three class-matrix sizes of the step
    torch.manual_seed(5)
    pred = FastRCNNPredictor(cfg, 96, False)  # small dims keep the fixture small; the math is size-agnostic
    with torch.no_grad():
        pred.emb_pred.bias.normal_(0, 0.01, generator=g)
        pred.bbox_pred.bias.normal_(0, 0.01, generator=g)
    x = torch.randn(24, 96, 7, 7, generator=g)
    out["pred_x"] = x.numpy()
    for k, v in pred.state_dict().items():
        out["pred_" + k] = v.numpy()
    for c in (1, 49, 1203):
        e = F.normalize(torch.randn(c, 48, generator=g), dim=-1)
        if c > 1:
            e[0] = 0
        pred.set_class_embeddings(e)
        logits, box = pred(x)
        out[f"pred_cls{c}"], out[f"pred_logits{c}"] = e.numpy(), logits.detach().numpy()
    out["pred_box"] = box.detach().numpy()

    # ---- box loss arithmetic (box_head/loss.py:125-185); `.cuda()` is hard-coded there
    torch.Tensor.cuda = lambda self, *a, **k: self
    loss_eval = FastRCNNLossComputation.__new__(FastRCNNLossComputation)
    loss_eval.cls_agnostic_bbox_reg, loss_eval.bg_weight = True, 0.2
    P = 40
    labels = torch.randint(0, 49, (P,), generator=g)
    labels[::3] = 0
    reg_t = torch.randn(P, 4, generator=g)
    prop = BoxList(torch.zeros(P, 4), (100, 100))
    prop.add_field("labels", labels)
    prop.add_field("regression_targets", reg_t)
    loss_eval._proposals = [prop]
    cl, br = torch.randn(P, 49, generator=g), torch.randn(P, 8, generator=g) * 0.3
    lc, lb = loss_eval([cl], [br], None)
    out.update(boxloss_logits=cl.numpy(), boxloss_reg=br.numpy(), boxloss_labels=labels.numpy(),
               boxloss_targets=reg_t.numpy(), boxloss_cls=lc.numpy(), boxloss_box=lb.numpy())

    # ---- mask predictor with the uncertainty branch (roi_mask_predictors.py:41-65), eps captured
    torch.manual_seed(6)
    mp = MaskRCNNC4Predictor(cfg, 64)
    with torch.no_grad():
        mp.uncertain_pred.weight.normal_(0, 0.05, generator=g)
    xm = torch.randn(6, 64, 7, 7, generator=g) * 0.2
    out["mask_x"] = xm.numpy()
    for k, v in mp.state_dict().items():
        out["maskpred_" + k] = v.numpy()
    mp.train()
    torch.manual_seed(1234)
    eps = torch.randn(1, 6, 2, 14, 14)  # exactly what reparameterize() draws next: std = logits*0+scale is [P,2,M,M]
    torch.manual_seed(1234)
    logits5, scale = mp(xm, True)
    out.update(mask_eps=eps.numpy(), mask_logits5=logits5.detach().numpy(), mask_scale=scale.detach().numpy())
    mp.eval()
    out["mask_logits_eval"] = mp(xm).detach().numpy()
    # mask loss arithmetic (mask_head/loss.py:117-148) on channel 1 of positives
    tgt = (torch.rand(6, 14, 14, generator=g) > 0.5).float()
    flat = torch.flatten(logits5, 0, 1)
    out["mask_targets"] = tgt.numpy()
    out["mask_loss"] = F.binary_cross_entropy_with_logits(flat[torch.arange(6), torch.ones(6, dtype=torch.long)], tgt,
                                                          reduction="none").mean().detach().numpy()

    # ---- small numeric helpers
    coder = BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))
    xy = torch.rand(30, 2, generator=g) * 300
    ref_b = torch.cat([xy, xy + torch.rand(30, 2, generator=g) * 200 + 4], 1)
    xy2 = xy + torch.randn(30, 2, generator=g) * 10
    prop_b = torch.cat([xy2, xy2 + torch.rand(30, 2, generator=g) * 200 + 4], 1)
    enc = coder.encode(ref_b, prop_b)
    codes = torch.randn(30, 8, generator=g)
    codes[0, 2] = 50.0  # exercises the bbox_xform_clip clamp
    out.update(coder_ref=ref_b.numpy(), coder_prop=prop_b.numpy(), coder_enc=enc.numpy(), coder_codes=codes.numpy(),
               coder_dec=coder.decode(codes, prop_b).numpy())
    iou = boxlist_iou(BoxList(ref_b[:7], (600, 600)), BoxList(prop_b, (600, 600)))
    out["iou"] = iou.numpy()
    out["match_plain"] = Matcher(0.5, 0.5, False)(iou).numpy()
    out["match_rpn"] = Matcher(0.7, 0.3, True)(iou).numpy()
    a, b = torch.randn(50, 4, generator=g), torch.randn(50, 4, generator=g)
    out.update(sl1_a=a.numpy(), sl1_b=b.numpy(), sl1_beta1_sum=smooth_l1_loss(a, b, beta=1, size_average=False).numpy(),
               sl1_beta9_mean=smooth_l1_loss(a, b).numpy())
    bn = FrozenBatchNorm2d(5)
    for n_ in ("weight", "bias", "running_mean"):
        getattr(bn, n_).copy_(torch.randn(5, generator=g))
    bn.running_var.copy_(torch.rand(5, generator=g) + 0.5)
    xb = torch.randn(2, 5, 3, 4, generator=g)
    out.update(bn_x=xb.numpy(), bn_y=bn(xb).numpy(), **{"bn_" + k: v.numpy() for k, v in bn.state_dict().items()})

    # ---- anchors + RPN proposal selection on one small feature map (rpn/inference.py:76-123)
    ag = AnchorGenerator((32, 64, 128, 256, 512), (0.5, 1.0, 2.0), (16,), 0)
    out["cell_anchors"] = list(ag.cell_anchors)[0].numpy()
    n, A, H, W = 2, 15, 9, 12
    feat = torch.zeros(n, 1, H, W)
    il = _ns(image_sizes=[(H * 16, W * 16), (H * 16 - 10, W * 16 - 7)])
    anchors = ag(il, [feat])
    out["anchors_img1"] = anchors[1][0].bbox.numpy()
    out["anchors_vis1"] = anchors[1][0].get_field("visibility").numpy()
    obj = torch.randn(n, A, H, W, generator=g)
    reg = torch.randn(n, A * 4, H, W, generator=g) * 0.2
    pp = RPNPostProcessor(pre_nms_top_n=600, post_nms_top_n=50, nms_thresh=0.7, min_size=0)
    res = pp.forward_for_single_feature_map([a_[0] for a_ in anchors], obj, reg)
    out.update(rpn_obj=obj.numpy(), rpn_reg=reg.numpy())
    for i, r in enumerate(res):
        out[f"rpn_boxes{i}"], out[f"rpn_scores{i}"] = r.bbox.numpy(), r.get_field("objectness").numpy()

    # ---- RPN loss (rpn/loss.py:21-131): targets of every anchor and the two losses; the sampler's quotas cover every
    # candidate (batch 10^6: randperm()[:k] with k = all), so the values do not depend on the random stream
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler as RefSampler
    from maskrcnn_benchmark.modeling.rpn.loss import RPNLossComputation as RefRPNLoss, generate_rpn_labels
    rl = RefRPNLoss(Matcher(0.7, 0.3, allow_low_quality_matches=True), RefSampler(10 ** 6, 0.5), BoxCoder(weights=(1.0, 1.0, 1.0, 1.0)),
                    generate_rpn_labels)
    rpn_gt = [BoxList(torch.tensor([[10.0, 12, 70, 90], [60, 30, 150, 120], [100, 5, 130, 40]]), (W * 16, H * 16), mode="xyxy"),
              BoxList(torch.tensor([[5.0, 5, 175, 130], [20, 60, 60, 100]]), (W * 16 - 7, H * 16 - 10), mode="xyxy")]
    lab, tgt = rl.prepare_targets([a_[0] for a_ in anchors], rpn_gt)
    lo, lb = rl([[a_[0]] for a_ in anchors], [obj], [reg], rpn_gt)
    for i in range(2):
        out[f"rpnloss_gt{i}"], out[f"rpnloss_labels{i}"], out[f"rpnloss_targets{i}"] = rpn_gt[i].bbox.numpy(), lab[i].numpy(), tgt[i].numpy()
    out["rpnloss_objectness"], out["rpnloss_box"] = np.float32(lo.item()), np.float32(lb.item())

    # ---- Masker paste (mask_head/inference.py:124-160)
    m = torch.rand(14, 14, generator=g)
    for i, box in enumerate(([10.3, 20.7, 90.2, 70.9], [-5.0, -8.0, 30.0, 25.0], [100.0, 60.0, 159.0, 119.0])):
        out[f"paste_box{i}"] = np.array(box, dtype=np.float32)
        out[f"paste_out{i}"] = paste_mask_in_image(m, torch.tensor(box), 120, 160).numpy()
    out["paste_mask"] = m.numpy()

    # ---- fg / bg sampler (balanced_positive_negative_sampler.py:19-68): the masks when the quotas cover every candidate
    # (deterministic) and the COUNTS when they do not (the subsets themselves are random)
    from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
    lab = torch.zeros(900, dtype=torch.int64)
    perm = torch.randperm(900, generator=g)
    lab[perm[:130]] = torch.randint(1, 49, (130,), generator=g)
    lab[perm[130:200]] = -1
    pos_all, neg_all = BalancedPositiveNegativeSampler(4096, 0.25)([lab])
    pos_q, neg_q = BalancedPositiveNegativeSampler(512, 0.25)([lab])
    pos_f, neg_f = BalancedPositiveNegativeSampler(256, 1.0)([lab])
    out.update(sampler_labels=lab.numpy(), sampler_pos_all=pos_all[0].numpy(), sampler_neg_all=neg_all[0].numpy(),
               sampler_counts_512_025=np.array([int(pos_q[0].sum()), int(neg_q[0].sum())]),
               sampler_counts_256_100=np.array([int(pos_f[0].sum()), int(neg_f[0].sum())]))
    np.savez_compressed(os.path.join(HERE, "heads.npz"), **out)


def c2_blob_names():
    """Blob names of a Detectron R-50-C4 Mask R-CNN checkpoint / an ImageNet-pretrained R-50 (the checkpoints the
    reference's configs name), incl. solver momentum blobs."""
    names = ["conv1_w", "res_conv1_bn_s", "res_conv1_bn_b", "fc1000_w", "fc1000_b", "pred_w", "pred_b",
             "conv_rpn_w", "conv_rpn_b", "rpn_cls_logits_w", "rpn_cls_logits_b", "rpn_bbox_pred_w", "rpn_bbox_pred_b",
             "cls_score_w", "cls_score_b", "bbox_pred_w", "bbox_pred_b", "conv5_mask_w", "conv5_mask_b",
             "mask_fcn_logits_w", "mask_fcn_logits_b", "conv1_w_momentum", "res2_0_branch2a_w_momentum"]
    for stage, blocks in ((2, 3), (3, 4), (4, 6), (5, 3)):
        for b in range(blocks):
            for br in ("2a", "2b", "2c") + (("1",) if b == 0 else ()):
                names += [f"res{stage}_{b}_branch{br}_w", f"res{stage}_{b}_branch{br}_bn_s", f"res{stage}_{b}_branch{br}_bn_b"]
    return names


def gen_c2_names():
    """Name pairs (C2 blob -> torch parameter) produced by the reference's own translation, catalog URLs and cache file
    names: tests/golden/c2_names.json (data only)."""
    import json

    six = types.ModuleType("torch._six")
    six.PY3 = True
    sys.modules["torch._six"] = six
    torch._six = six
    import importlib.util

    # config/__init__ needs yacs (absent here); paths_catalog.py itself only imports os / copy
    spec = importlib.util.spec_from_file_location("ref_paths_catalog",
                                                  os.path.join(REF, "maskrcnn_benchmark/config/paths_catalog.py"))
    catalog_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(catalog_mod)
    ModelCatalog = catalog_mod.ModelCatalog
    from maskrcnn_benchmark.utils import c2_model_loading as c2

    names = [n for n in c2_blob_names() if n not in ("pred_w", "pred_b")]  # pred_* and fc1000_* collide in one file
    mapped = c2._rename_weights_for_resnet({n: np.zeros(1, np.float32) for n in names}, c2._C2_STAGE_NAMES["R-50"])
    # the function returns an OrderedDict in sorted-original-key order without the momentum blobs
    kept = [n for n in sorted(names) if "_momentum" not in n]
    pairs = dict(zip(kept, mapped.keys()))
    alt = c2._rename_weights_for_resnet({"pred_w": np.zeros(1, np.float32), "pred_b": np.zeros(1, np.float32)},
                                        c2._C2_STAGE_NAMES["R-50"])
    pairs.update(dict(zip(["pred_b", "pred_w"], alt.keys())))
    catalog = {n: ModelCatalog.get(n) for n in ("ImageNetPretrained/MSRA/R-50", "ImageNetPretrained/MSRA/R-101",
                                                "ImageNetPretrained/FAIR/20171220/X-101-32x8d",
                                                "Caffe2Detectron/COCO/35858791/e2e_mask_rcnn_R-50-C4_1x",
                                                "Caffe2Detectron/COCO/37697547/e2e_keypoint_rcnn_R-50-FPN_1x")}
    with open(os.path.join(HERE, "c2_names.json"), "w") as f:
        json.dump({"names": pairs, "momentum": [n for n in names if "_momentum" in n], "catalog": catalog}, f, indent=1,
                  sort_keys=True)


WORDPIECES = ("[PAD] [UNK] [CLS] [SEP] [MASK] a b c d e f g h i j k l m n o p q r s t u v w x y z ##a ##b ##c ##d ##e ##f ##g "
              "##h ##i ##j ##k ##l ##m ##n ##o ##p ##q ##r ##s ##t ##u ##v ##w ##x ##y ##z cat dog traffic light sign stop "
              "fire hydrant hot teddy bear hair drier ##s ##ing ##ed ##er ##board skate surf snow tennis racket wine glass "
              "cell phone potted plant dining table sports ball baseball bat glove parking meter bench bird horse sheep cow "
              "elephant zebra giraffe back ##pack umbrella hand ##bag tie suit ##case fr ##is ##bee ski ##is kite person bicycle "
              "car motor ##cycle air ##plane bus train truck boat").split()


def gen_text():
    """tests/golden/text_embed.npz + tests/golden/wordpiece_vocab.txt: the reference's own ``BERT.forward``
    (language_backbone/transformers.py:27-68) and ``STGeneralizedRCNN.extract_emb`` (st_generalized_rcnn.py:202-209) on a
    small WordPiece vocabulary and a random embedding table.  The reference constructor downloads bert-base-uncased, so
    the object is built without it (``__new__``) and given the pieces its ``forward`` reads: ``tokenizer`` (HuggingFace
    BertTokenizer over the synthetic vocabulary, behind an adapter that accepts transformers 3.0.2's
    ``batch_encode_plus(..., pad_to_max_length=True)`` call), ``embeddings`` and ``mlm``; ``extract_emb`` is taken from the
    reference file's syntax tree (its module imports yacs-dependent code and cannot be imported here) and run unchanged."""
    import ast

    from transformers import BertTokenizer

    vocab_path = os.path.join(HERE, "wordpiece_vocab.txt")
    with open(vocab_path, "w") as f:
        f.write("\n".join(WORDPIECES) + "\n")
    hf = BertTokenizer(vocab_path, do_lower_case=True)

    class Tok302:  # the call surface of transformers==3.0.2 that transformers.py:28-32 uses
        def batch_encode_plus(self, text_list, add_special_tokens=True, pad_to_max_length=False, return_special_tokens_mask=False):
            enc = hf(list(text_list), add_special_tokens=add_special_tokens, padding=bool(pad_to_max_length),
                     return_special_tokens_mask=return_special_tokens_mask)
            return {k: v for k, v in enc.items()}

    from maskrcnn_benchmark.modeling.language_backbone.transformers import BERT as RefBERT

    g = torch.Generator().manual_seed(77)
    table = torch.randn(len(WORDPIECES), 768, generator=g) * 0.05
    ref = RefBERT.__new__(RefBERT)
    torch.nn.Module.__init__(ref)
    ref.tokenizer, ref.mlm = Tok302(), False
    ref.embeddings = torch.nn.Parameter(table, requires_grad=False)
    src = open(os.path.join(REF, "maskrcnn_benchmark/modeling/detector/st_generalized_rcnn.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "extract_emb")
    ns = {"torch": torch, "F": torch.nn.functional}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "st_generalized_rcnn.py", "exec"), ns)
    holder = types.SimpleNamespace(bert=ref)
    words = ["cat", "traffic light", "Fire Hydrant", "teddy bears", "hair drier", "skateboard", "xylophone zq", "a",
             "potted plant", "sports ball", "backpack", "frisbee", "wine glasses", "motorcycle", "snowboarding"]
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # transformers.py:60 hard-codes .cuda()
    try:
        enc = ref(words)
        emb = ns["extract_emb"](holder, words)
    finally:
        torch.Tensor.cuda = cuda
    # data/datasets/helper/parser.py:10-20 (its module imports spacy / nltk: the function is taken from the syntax tree too)
    psrc = open(os.path.join(REF, "maskrcnn_benchmark/data/datasets/helper/parser.py")).read()
    pfn = next(n for n in ast.walk(ast.parse(psrc)) if isinstance(n, ast.FunctionDef) and n.name == "normalize_class_names")
    pns = {}
    exec(compile(ast.Module(body=[pfn], type_ignores=[]), "parser.py", "exec"), pns)
    raw_names = ["aerosol_can", "Bow_(decorative_ribbons)", "T-shirt", "hot/dog", "CD_player", "pop_(soda)", "plain"]
    np.savez_compressed(os.path.join(HERE, "text_embed.npz"), table=table.numpy(), words=np.array(words),
                        names_raw=np.array(raw_names), names_normalized=np.array(pns["normalize_class_names"](raw_names)),
                        input_ids=enc["input_ids"].numpy(), special_tokens_mask=enc["special_tokens_mask"].numpy(),
                        attention_mask=enc["attention_mask"].numpy(), embeddings=emb.detach().numpy())


def main():
    torch.set_num_threads(1)
    ref_c = import_reference()
    gen_roi_align(ref_c)
    gen_nms(ref_c)
    gen_focal()
    gen_heads()
    gen_c2_names()
    gen_text()
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
