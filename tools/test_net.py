"""Inference entry point with the reference's command line (tools/test_net.py:27-116):

    python -m torch.distributed.run --nproc-per-node N tools/test_net.py --config-file CFG [--ckpt FILE] KEY VALUE ...

One process per GPU; the weights come from ``--ckpt`` / ``MODEL.WEIGHT`` / the last checkpoint of OUTPUT_DIR
(utils/checkpoint.py).  There is no dataset access in this build: the images are the synthetic COCO-shaped stream
(image ids 0..N-1 sharded round-robin over the ranks like ``InferenceSampler``), the detections of all ranks are
gathered on rank 0 and saved as ``<OUTPUT_DIR>/inference/synthetic/predictions.pth``; the COCO / LVIS scoring of
``evaluate`` needs annotation files and pycocotools and is outside this build.
"""
import argparse
import logging
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import calibrate_stem_bn, make_batch, make_embeddings  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, inference  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.utils.checkpoint import DetectronCheckpointer  # noqa: E402


def main():
    parser = argparse.ArgumentParser(description="MI355X-native detection inference (synthetic data)")
    parser.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    parser.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    parser.add_argument("--ckpt", default=None, help="checkpoint to test instead of MODEL.WEIGHT / the last one of OUTPUT_DIR")
    parser.add_argument("--num-images", type=int, default=16, help="size of the synthetic test set")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE overrides")
    args = parser.parse_args()

    world = int(os.environ.get("WORLD_SIZE", 1))
    cfg = get_defaults()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts or [])
    cfg.freeze()
    device = torch.device(cfg.MODEL.DEVICE, args.local_rank) if cfg.MODEL.DEVICE == "cuda" else torch.device(cfg.MODEL.DEVICE)
    if cfg.MODEL.DEVICE == "cuda":
        torch.cuda.set_device(args.local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl" if cfg.MODEL.DEVICE == "cuda" else "gloo", init_method="env://")
        comm.synchronize()
    logging.basicConfig(level=logging.INFO if comm.get_rank() == 0 else logging.WARNING,
                        format="%(asctime)s %(name)s %(levelname)s: %(message)s")
    logger = logging.getLogger("ovis.inference")
    logger.info("Using %d GPUs\n%s", world, args)

    model = build_detection_model(cfg).to(device)
    checkpointer = DetectronCheckpointer(cfg, model, save_dir=cfg.OUTPUT_DIR)
    weight = args.ckpt or cfg.MODEL.WEIGHT
    extra = checkpointer.load(weight, use_latest=args.ckpt is None)
    if not weight and not extra and not checkpointer.has_checkpoint():
        images, _ = make_batch(1, device=device, seed=7)
        calibrate_stem_bn(model, images)  # random init only: give the frozen BN usable statistics
    _, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, device=device)

    ims = max(cfg.TEST.IMS_PER_BATCH // world, 1)
    ids = list(range(comm.get_rank(), args.num_images, world))  # samplers/distributed.py-style round-robin shard

    def batches():
        for k in range(0, len(ids), ims):
            chunk = ids[k:k + ims]
            images = torch.cat([make_batch(1, device=device, seed=5000 + i)[0] for i in chunk], 0)
            yield images, None, chunk

    out = os.path.join(cfg.OUTPUT_DIR, "inference", "synthetic") if cfg.OUTPUT_DIR else None
    preds = inference.inference(model, batches(), "synthetic", device, out, class_embeddings=e_seen, logger=logger)
    if preds is not None:
        n = sum(len(p) for p in preds)
        logger.info("%d images, %d detections (fields: %s)", len(preds), n, sorted(preds[0].fields()) if preds else [])
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
