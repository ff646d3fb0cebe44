"""Loss trajectory of a training step over N optimisation steps at the config's real learning rate.

    python tools/experiments/loss_trajectory.py STEPS PATH WORKLOAD > out.json
      PATH     split     the product path: NHWC + bf16 hi/lo split GEMMs (three-term products, fp32 accumulation)
               fp32      per-layer fp32 convolutions (MIOpen / rocBLAS; ``nhwc = False`` on every module that has the switch)
               fp32_eps  the fp32 path with every trainable parameter perturbed by one fp32 rounding (x * (1 +- 2^-24)) at
                         step 0: the divergence floor of fp32 training itself on this (chaotic) problem
      WORKLOAD student | teacher

Same seeds, same synthetic batch every step.  ``compare`` mode turns the three JSON files into the table committed under
profiles/:   python tools/experiments/loss_trajectory.py compare split.json fp32.json fp32_eps.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(steps, path, workload):
    import torch
    from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults
    from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings
    from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer
    from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model

    name = "student_teacher_mask_rcnn_uncertainty" if workload == "student" else "zeroshot_mask"
    cfg = get_defaults()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "coco_cap_det", name + ".yaml"))
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 2])
    cfg.freeze()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = build_detection_model(cfg).to(dev)
    e_vocab, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, seed=1234, device=dev)
    model.set_class_embeddings(e_seen)
    if hasattr(model, "set_caption_vocab"):
        model.set_caption_vocab(e_vocab)
    images, targets = make_batch(2, device=dev, seed=1234)
    calibrate_stem_bn(model, images)
    if path != "split":
        for m in model.modules():
            for attr in ("nhwc", "train_nhwc"):
                if hasattr(m, attr):
                    setattr(m, attr, False)
    if path == "fp32_eps":
        g = torch.Generator(device=dev).manual_seed(99)
        with torch.no_grad():
            for p in model.parameters():
                if p.requires_grad:
                    sign = torch.randint(0, 2, p.shape, device=dev, generator=g).float() * 2 - 1
                    p.mul_(1 + sign * 2.0 ** -24)
    model.train()
    opt = solver.make_optimizer(cfg, model)
    sch = solver.make_lr_scheduler(cfg, opt)
    red = comm.BucketedGradReducer(model)
    out = []
    for i in range(steps):
        torch.manual_seed(1000 + i)  # same sampling / noise stream in every run
        ld = trainer.train_step(model, opt, red, images, targets, sch)
        out.append({k: float(v) for k, v in ld.items()})
    print(json.dumps({"path": path, "workload": workload, "lr": cfg.SOLVER.BASE_LR, "losses": out}))


def compare(split_json, fp32_json, eps_json):
    a, b, c = (json.load(open(p)) for p in (split_json, fp32_json, eps_json))
    n = min(len(a["losses"]), len(b["losses"]), len(c["losses"]))
    names = list(a["losses"][0].keys())
    print(f"# {a['workload']} step, config learning rate ({a['lr']}), {n} SGD steps on one synthetic batch, same seeds")
    print("# split = NHWC + bf16 hi/lo split GEMMs (product path); fp32 = per-layer fp32 convolutions; fp32' = the fp32 path with")
    print("# every trainable parameter perturbed by ONE fp32 rounding at step 0 (the divergence floor of fp32 training itself)")
    print("# per window of steps: max over the window of |x - fp32| / max(|fp32|, 1e-3) for x = split and x = fp32'")
    print("# %9s  %-28s %12s %12s %14s %14s" % ("steps", "loss", "fp32 (last)", "split (last)", "split vs fp32", "fp32' vs fp32"))
    w = max(1, n // 10)
    for lo in range(0, n, w):
        hi = min(lo + w, n)
        for k in names:
            ds = max(abs(a["losses"][i][k] - b["losses"][i][k]) / max(abs(b["losses"][i][k]), 1e-3) for i in range(lo, hi))
            de = max(abs(c["losses"][i][k] - b["losses"][i][k]) / max(abs(b["losses"][i][k]), 1e-3) for i in range(lo, hi))
            print(f"  {lo:4d}-{hi - 1:4d}  {k:28s} {b['losses'][hi - 1][k]:12.6f} {a['losses'][hi - 1][k]:12.6f} {ds:14.2e} {de:14.2e}")
    tot = lambda d, i: sum(d["losses"][i].values())  # noqa: E731
    print("# total loss, mean over the last tenth of the run: fp32 %.5f  split %.5f  fp32' %.5f" % tuple(
        sum(tot(d, i) for i in range(n - w, n)) / w for d in (b, a, c)))
    finite = all(all(v == v and abs(v) != float("inf") for v in row.values()) for d in (a, b, c) for row in d["losses"])
    print("# all losses finite:", finite)


if __name__ == "__main__":
    if sys.argv[1] == "compare":
        compare(*sys.argv[2:5])
    else:
        run(int(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else "split", sys.argv[3] if len(sys.argv) > 3 else "student")
