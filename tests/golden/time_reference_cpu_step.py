"""Times the REFERENCE's own student-teacher training step on CPU cores at BASELINE size (build container only: the
reference does not travel to the GPU box, so this is not part of bench.py's cpu_baseline -- it is the number that grounds its
``full_step_estimate``).

The reference's ``STGeneralizedRCNN`` (real defaults.py + student_teacher_mask_rcnn_uncertainty.yaml, full R-50-C4, 1203 LVIS
names through the stand-in BERT table) runs forward + backward + its own ``make_optimizer`` SGD step on ONE 3 x 800 x 1333 image
(it is only correct at one image per process) with the stand-ins of tests/golden/ref_import.py; native ops = the
reference's compiled CPU kernels (RoIAlign forward, NMS).  The student step needs no RoIAlign backward (frozen trunk).

    python tests/golden/time_reference_cpu_step.py [threads] > profiles/r5_reference_cpu_step_build_container.txt
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_step_golden as M  # noqa: E402
import ref_import  # noqa: E402
import step_case as case  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    ref_import.install()
    torch.Tensor.cuda = lambda self, *a, **k: self
    from maskrcnn_benchmark.modeling.language_backbone import transformers as ref_lb
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask

    ref_import._namespace_pkg("maskrcnn_benchmark.modeling.detector",
                              os.path.join(ref_import.REF, "maskrcnn_benchmark/modeling/detector"))
    ref_lb.BERT = M.make_bert_class(ref_lb.BERT)
    from maskrcnn_benchmark.modeling.detector import st_generalized_rcnn as st_mod
    from maskrcnn_benchmark.solver import make_optimizer

    cfg = ref_import.reference_cfg("student_teacher_mask_rcnn_uncertainty.yaml", ["MODEL.DEVICE", "cpu"])
    model = st_mod.STGeneralizedRCNN(cfg)
    with torch.no_grad():
        model.bert.embeddings.normal_(0, 0.05)
        # FrozenBN identity statistics + pixels of O(100): calibrate the stem like data/synthetic.py does
        g = np.random.default_rng(0)
    H, W = 800, 1333
    img = torch.from_numpy((g.uniform(0, 255, (3, H, W)) - np.array([102.9801, 115.9465, 122.7717])[:, None, None]).astype(np.float32))
    with torch.no_grad():
        stem = model.backbone.body.stem
        y = torch.nn.functional.conv2d(img[None], stem.conv1.weight, None, stem.conv1.stride, stem.conv1.padding)
        stem.bn1.running_mean.copy_(y.mean((0, 2, 3)))
        stem.bn1.running_var.copy_(y.var((0, 2, 3)))
    model.class_names = list(case.SEEN_NAMES)
    model.roi_heads["box"].predictor.set_class_embeddings(case.text_embeddings())
    model.train()
    n_gt = 7
    xy = g.uniform(0, 1, (n_gt, 2)) * np.array([W - 400, H - 400])
    wh = g.uniform(32, 368, (n_gt, 2))
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], 1).astype(np.float32)).floor()
    masks = torch.zeros(n_gt, H, W, dtype=torch.bool)
    for i, b in enumerate(boxes.long().tolist()):
        masks[i, b[1] + 3:b[3] - 3, b[0] + 3:b[2] - 3] = True
    t = BoxList(boxes, (W, H), mode="xyxy")
    t.add_field("labels", torch.from_numpy(g.integers(1, 49, n_gt)))
    t.add_field("masks", SegmentationMask(masks, (W, H), mode="mask"))
    ids = np.sort(g.choice(1203, 5, replace=False))
    t.add_field("nn_caption", "/".join(model.cap_vocab[i] for i in ids))
    t.add_field("ids_cap", torch.from_numpy(ids))
    t.add_field("is_det", "Yes")
    optimizer = make_optimizer(cfg, model)
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        losses = model(img[None], [t])
        t1 = time.perf_counter()
        sum(losses.values()).backward()
        t2 = time.perf_counter()
        optimizer.step()
        optimizer.zero_grad()
        t3 = time.perf_counter()
        times.append((t1 - t0, t2 - t1, t3 - t2))
        print(f"# iteration {it}: forward {t1 - t0:.2f} s, backward {t2 - t1:.2f} s, SGD {t3 - t2:.2f} s; losses "
              + " ".join(f"{k}={float(v):.4f}" for k, v in losses.items()), flush=True)
    fwd, bwd, sgd = (float(np.median([x[i] for x in times[1:]])) for i in range(3))
    total = fwd + bwd + sgd
    print(f"reference STGeneralizedRCNN training step, 1 image 3x800x1333, full R-50-C4, {threads} CPU threads "
          f"(build container, torch {torch.__version__} CPU convolutions, reference CPU RoIAlign / NMS kernels):")
    print(f"  forward {fwd:.2f} s + backward {bwd:.2f} s + SGD {sgd:.2f} s = {total:.2f} s per image = {1.0 / total:.4f} images/s")


if __name__ == "__main__":
    main()
