"""The whole-step parity case shared by ``make_step_golden.py`` (which runs the REFERENCE's own detector classes on it in
the build container) and by the tests (which run the product on it): configuration overrides, inputs and weights.

Nothing here touches the reference.  Inputs and weights are functions of their NAME and a seed through numpy's
``default_rng`` (stream-stable by numpy's compatibility policy), so the fixtures hold outputs only: a reduced
R-50-C4 with every channel count a multiple of 32 (the pair-layout GEMM route's granularity, so the GPU test runs the
same kernels as the full-size model) is still ~5 M weights -- too large to commit, trivial to regenerate.
"""
import zlib

import numpy as np
import torch

IMAGE_H, IMAGE_W = 128, 160
EMB_DIM = 768  # st_generalized_rcnn.py:220 hard-codes torch.zeros((1, 768))
N_SEEN = 49

# KEY VALUE overrides applied on top of the shipped yaml by BOTH sides
COMMON_OPTS = [
    "MODEL.DEVICE", "cpu",
    "MODEL.RESNETS.STEM_OUT_CHANNELS", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 32,
    "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128,
    "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (32, 32, 32, 32),
    "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 400, "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 80,
    "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300, "MODEL.RPN.POST_NMS_TOP_N_TEST", 60,
    # below the candidate counts: the fg / bg samplers really draw (the draws are part of the fixture)
    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 40, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 64,
    "MODEL.RPN.MIN_SIZE", 8,
]

# a small WordPiece vocabulary (every lower-case letter is a piece, so no word collapses to [UNK] as a whole)
WORDPIECES = ("[PAD] [UNK] [CLS] [SEP] [MASK] a b c d e f g h i j k l m n o p q r s t u v w x y z ##a ##b ##c ##d ##e ##f ##g "
              "##h ##i ##j ##k ##l ##m ##n ##o ##p ##q ##r ##s ##t ##u ##v ##w ##x ##y ##z 0 1 2 3 4 5 6 7 8 9 ##0 ##1 ##2 ##3 "
              "##4 ##5 ##6 ##7 ##8 ##9 - ' . , & cat dog traffic light sign stop fire hydrant hot teddy bear hair drier ##s "
              "##ing ##ed ##er ##board skate surf snow tennis racket wine glass cell phone potted plant dining table sports "
              "ball baseball bat glove parking meter bench bird horse sheep cow elephant zebra giraffe back ##pack umbrella "
              "hand ##bag tie suit ##case fr ##is ##bee ski kite person bicycle car motor ##cycle air ##plane bus train truck "
              "boat bottle cup fork knife spoon bowl banana apple sandwich orange broccoli carrot pizza donut cake chair "
              "couch bed toilet tv laptop mouse remote keyboard microwave oven toaster sink refrigerator book clock vase "
              "scissors tooth ##brush").split()

SEEN_NAMES = ["__background__"] + [f"seen class {i}" for i in range(1, N_SEEN)]  # only their COUNT is read (st_generalized_rcnn.py:371)


def _rng(name, seed=0):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def seeded_tensor(name, shape, seed=0):
    """Value of the parameter / buffer called ``name`` (reference state-dict names).  Scales keep activations O(1) through
    the 16 residual blocks and give RPN scores / region embeddings real spread, so that top-k, NMS and the per-noun
    argmax are decided by margins far above fp32 round-off."""
    g = _rng(name, seed)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("lambda_exemplar"):
        v = np.zeros(shape)
    elif name.endswith("bert.embeddings"):
        v = g.standard_normal(shape) * 0.05
    elif leaf == "running_var":
        v = g.uniform(0.8, 1.2, shape)
    elif leaf == "running_mean":
        v = g.standard_normal(shape) * 0.05
    elif len(shape) == 1 and leaf == "weight":      # FrozenBatchNorm2d scale; the last one of a block damps the residual sum
        v = g.uniform(0.8, 1.2, shape) * (0.35 if ".bn3." in name else 0.01 if "stem.bn1" in name else 1.0)  # pixels are O(100)
    elif leaf == "bias":
        v = g.standard_normal(shape) * (0.05 if ".bn" in name or "downsample.1" in name else 0.01)
        if "uncertain_pred" in name:
            v = v + 1.0                               # roi_mask_predictors.py:39
    else:                                             # convolution / linear / transposed-convolution weights
        fan_in = shape[0] if "conv5_mask" in name else int(np.prod(shape[1:]))
        v = g.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        if "rpn.head.bbox_pred" in name:
            v *= 0.1
        if "uncertain_pred" in name:
            v *= 0.2
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def seeded_state(named_shapes, seed=0):
    """{name: tensor} for an iterable of (name, shape)."""
    return {n: seeded_tensor(n, s, seed) for n, s in named_shapes}


def is_seeded(name):
    """Entries of a reference state dict that carry weights (the anchor tables are geometry, not weights)."""
    return "anchor_generator" not in name


def text_embeddings(seed=0):
    """Seen-class matrix [49, 768] with the all-zero background row (data/datasets/coco.py:85-91), unit-norm rows."""
    e = _rng("seen_class_embeddings", seed).standard_normal((N_SEEN, EMB_DIM))
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    e[0] = 0
    return torch.from_numpy(e.astype(np.float32))


FULL_H, FULL_W = 800, 1333   # the BASELINE image size (full-size fixture: step_student_full.npz)


def image_case(index, vocab_names, seed=0, size=None, n_gt=3, n_nouns=3):
    """One image of the case: pixels, ground-truth boxes (xyxy) / labels / rectangular binary masks, caption nouns (names of
    the caption vocabulary and their 0-based ids).  ``vocab_names`` = the normalised caption vocabulary.  ``size`` = (H, W),
    default the small case's 128 x 160 (its random stream is unchanged by the other arguments' defaults)."""
    IMAGE_H, IMAGE_W = size or (globals()["IMAGE_H"], globals()["IMAGE_W"])
    big = size is not None
    g = _rng(f"image{index}" + (f"@{IMAGE_H}x{IMAGE_W}" if big else ""), seed)
    img = (g.uniform(0, 255, (3, IMAGE_H, IMAGE_W)) - np.array([102.9801, 115.9465, 122.7717])[:, None, None])
    # smooth structure so that features differ across the map (pure noise averages out in the strided stem)
    yy, xx = np.mgrid[0:IMAGE_H, 0:IMAGE_W]
    for _ in range(24 if big else 6):
        cx, cy, r = g.uniform(0, IMAGE_W), g.uniform(0, IMAGE_H), g.uniform(10, 50) * (5 if big else 1)
        img += (g.uniform(-120, 120, (3, 1, 1)) * (((xx - cx) ** 2 + (yy - cy) ** 2) < r * r))
    k = 5 if big else 1
    x1 = g.uniform(0, IMAGE_W - 60 * k, n_gt)
    y1 = g.uniform(0, IMAGE_H - 60 * k, n_gt)
    w = g.uniform(24, 90, n_gt) * k
    h = g.uniform(24, 80, n_gt) * k
    boxes = np.stack([x1, y1, np.minimum(x1 + w, IMAGE_W - 1), np.minimum(y1 + h, IMAGE_H - 1)], 1).round()
    labels = g.integers(1, N_SEEN, n_gt)
    masks = np.zeros((n_gt, IMAGE_H, IMAGE_W), np.uint8)
    for i, (bx1, by1, bx2, by2) in enumerate(boxes.astype(int)):
        dx, dy = (bx2 - bx1) // 6, (by2 - by1) // 6
        masks[i, by1 + dy:by2 - dy + 1, bx1 + dx:bx2 - dx + 1] = 1
    ids_cap = np.sort(g.choice(len(vocab_names), n_nouns, replace=False))
    return {
        "image": torch.from_numpy(img.astype(np.float32)),
        "boxes": torch.from_numpy(boxes.astype(np.float32)),
        "labels": torch.from_numpy(labels.astype(np.int64)),
        # bool, like the pseudo masks the Masker pastes: BinaryMaskList.resize casts the interpolated crop back with
        # ``type_as`` (segmentation_mask.py:150-155) -- "any weight on a set pixel" for bool.  For uint8 the cast truncates,
        # and whether an interior cell (true value 1) comes out as 1.0 or 0.99999994 depends on the interpolation kernel's
        # rounding (the reference's own CPU and CUDA kernels differ there); tests/test_components.py covers that dtype
        # away from the edge
        "masks": torch.from_numpy(masks).bool(),
        "ids_cap": torch.from_numpy(ids_cap.astype(np.int64)),
        "nn_caption": "/".join(vocab_names[i] for i in ids_cap),
    }


VARIANT_DIGEST = 256  # entries per tensor kept by the configuration-variant fixtures (seven runs: keep them small)


def grad_digest(name, grad, n=2048, seed=0):
    """What the fixtures keep of one gradient tensor: its L2 norm, its sum and the entries at ``n`` seeded positions (all
    of them when the tensor is smaller)."""
    flat = grad.detach().reshape(-1).double()
    idx = digest_index(name, flat.numel(), n, seed)
    return {"norm": float(flat.norm()), "sum": float(flat.sum()), "values": flat[idx].float().numpy()}


def digest_index(name, numel, n=2048, seed=0):
    if numel <= n:
        return torch.arange(numel)
    return torch.from_numpy(np.sort(_rng("digest:" + name, seed).choice(numel, n, replace=False)).astype(np.int64))
