"""TEST-ONLY: the tiny student-teacher / teacher configuration shared by the whole-step parity tests (GPU HIP ops vs the
oracle-backed CPU run) and by the world-size-2 gloo test of the real model: 160x192 images, a few hundred proposals,
batch sizes large enough that the fg/bg samplers take every candidate (deterministic sampling)."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_tiny(name, batch=2):
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

    torch.manual_seed(0)
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, f"configs/coco_cap_det/{name}.yaml"))
    cfg.merge_from_list(["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 400, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300,
                         "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 80, "MODEL.RPN.POST_NMS_TOP_N_TEST", 60,
                         "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 4096, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 100000,
                         "MODEL.RPN.POSITIVE_FRACTION", 1.0,
                         # random-init RPN boxes pile up as 1-px slivers on the image border; encode() divides by
                         # their width, which turns 1e-4 px of fp32 noise into O(1) target differences
                         "MODEL.RPN.MIN_SIZE", 16])
    cfg.freeze()
    model = build_detection_model(cfg)
    e_vocab, e_seen = make_embeddings(n_vocab=60)
    images, targets = make_batch(batch, height=160, width=192, num_gt=3, num_nouns=3, n_vocab=60)
    calibrate_stem_bn(model, images)
    # random-init region embeddings are nearly identical across regions; widen them so that the teacher's
    # per-noun argmax over regions is decided by a margin far above fp32 round-off (index outputs must be
    # compared exactly, and a near-tie would make the two sides pick different pseudo boxes)
    with torch.no_grad():
        for m in ([model.roi_heads] + ([model.roi_heads_student] if hasattr(model, "roi_heads_student") else [])):
            m["box"].predictor.emb_pred.weight.mul_(100.0)
    model.train()
    return model, e_vocab, e_seen, images, targets
