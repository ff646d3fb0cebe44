"""Training entry point with the reference's command line (tools/train_net.py:162-234):

    python -m torch.distributed.run --nproc-per-node N tools/train_net.py --config-file CFG [--skip-test] KEY VALUE ...

One process per GPU (RCCL via torch.distributed, env:// rendezvous), config = defaults <- yaml <- trailing overrides,
frozen before use.  There is no dataset access in this build, so the data stream is the synthetic COCO-shaped generator
(cvpr22_cross_modal_pseudo_labeling_amd/data/synthetic.py).  MODEL.WEIGHT (a ``.pth`` in the reference's wire format),
OUTPUT_DIR resume and SOLVER.CHECKPOINT_PERIOD behave as in the reference (utils/checkpoint.py: checkpoints are written and
resumed from whenever OUTPUT_DIR is set; ``--no-checkpoints`` opts out); evaluation and
TensorBoard are outside the hot-path scope.
"""
import argparse
import logging
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL between ranks needs it

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cvpr22_cross_modal_pseudo_labeling_amd.engine import launch  # noqa: E402  (no torch, no GPU)

# pin this rank to its share of the cores (of its GPU's NUMA node when sysfs tells) before torch / HIP start their threads
AFFINITY = launch.apply_rank_affinity() if __name__ == "__main__" else {"cpus": None, "source": "not applied"}

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cvpr22_cross_modal_pseudo_labeling_amd.config import get_defaults  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.data.prefetch import DevicePrefetcher  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.data.synthetic import (SyntheticBatches, calibrate_stem_bn, make_batch,  # noqa: E402
                                                                       make_embeddings)
from cvpr22_cross_modal_pseudo_labeling_amd.engine import comm, solver, trainer  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import build_detection_model  # noqa: E402
from cvpr22_cross_modal_pseudo_labeling_amd.utils.checkpoint import DetectronCheckpointer  # noqa: E402


def train(cfg, local_rank, distributed, max_iter, ims_per_gpu, save_checkpoints=False):
    device = torch.device(cfg.MODEL.DEVICE, local_rank) if cfg.MODEL.DEVICE == "cuda" else torch.device(cfg.MODEL.DEVICE)
    model = build_detection_model(cfg).to(device)
    e_vocab, e_seen = make_embeddings(cfg.MODEL.ROI_BOX_HEAD.EMB_DIM, device=device)
    model.set_class_embeddings(e_seen)  # engine/trainer.py:85-90
    if hasattr(model, "set_caption_vocab"):
        model.set_caption_vocab(e_vocab)
    optimizer = solver.make_optimizer(cfg, model)
    scheduler = solver.make_lr_scheduler(cfg, optimizer)
    # tools/train_net.py:76-87: resume from OUTPUT_DIR's last checkpoint, else initialise from MODEL.WEIGHT
    checkpointer = DetectronCheckpointer(
        cfg, model, optimizer, scheduler, cfg.OUTPUT_DIR if save_checkpoints else "", comm.get_rank() == 0,
        backbone_prefix=cfg.MODEL.BACKBONE_PREFIX,
        load_emb_pred_from=(cfg.MODEL.MMSS_HEAD.DEFAULT_HEAD if cfg.MODEL.LOAD_EMB_PRED_FROM_MMSS_HEAD else None),
        load_classifier=cfg.MODEL.LOAD_CLASSIFIER)
    resumed = checkpointer.has_checkpoint()
    extra = checkpointer.load(cfg.MODEL.WEIGHT, load_trainer_state=cfg.MODEL.LOAD_TRAINER_STATE)
    start_iter = int(extra.get("iteration", 0)) if cfg.MODEL.LOAD_TRAINER_STATE else 0
    if resumed and hasattr(model, "roi_heads_student") and not cfg.MODEL.RESUME:
        # st_generalized_rcnn.py:197-200 copies the teacher heads over the student at iteration 0 unless MODEL.RESUME
        logging.getLogger("ovis.trainer").warning(
            "resuming from %s with MODEL.RESUME False: the student heads in the checkpoint will be overwritten by the "
            "teacher's at the first iteration (set MODEL.RESUME True to keep them)", cfg.OUTPUT_DIR)
    if distributed:
        comm.broadcast_parameters(model)

    # host batches come from DATALOADER.NUM_WORKERS worker processes (the training process' interpreter lock is busy feeding
    # two HIP streams) and are staged through pinned memory on a copy stream, two batches ahead
    loader = torch.utils.data.DataLoader(SyntheticBatches(ims_per_gpu, seed0=1234, rank=comm.get_rank()), batch_size=None,
                                         num_workers=cfg.DATALOADER.NUM_WORKERS,
                                         prefetch_factor=2 if cfg.DATALOADER.NUM_WORKERS > 0 else None,
                                         persistent_workers=cfg.DATALOADER.NUM_WORKERS > 0)
    data = DevicePrefetcher(loader, device, depth=2)
    if not cfg.MODEL.WEIGHT and not checkpointer.has_checkpoint():  # random init only: give the frozen BN usable statistics
        images, _ = make_batch(1, device=device, seed=7)
        calibrate_stem_bn(model, images)
    try:
        return trainer.do_train(cfg, model, data, optimizer, scheduler, max_iter, start_iter=start_iter,
                                checkpointer=checkpointer if save_checkpoints else None,
                                checkpoint_period=cfg.SOLVER.CHECKPOINT_PERIOD)
    finally:
        data.close()


def main():
    parser = argparse.ArgumentParser(description="MI355X-native detection training (synthetic data)")
    parser.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    parser.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    parser.add_argument("--skip-test", dest="skip_test", action="store_true", help="accepted for compatibility")
    parser.add_argument("--max-iter", type=int, default=None, help="override SOLVER.MAX_ITER for a short run")
    parser.add_argument("--no-checkpoints", action="store_true",
                        help="do not write model_<iter>.pth / model_final.pth / last_checkpoint under OUTPUT_DIR and do not "
                             "resume from them (the reference always does both, tools/train_net.py:76-88)")
    parser.add_argument("--save-checkpoints", action="store_true", help="accepted for compatibility: saving is the default")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE overrides")
    args = parser.parse_args()

    num_gpus = int(os.environ.get("WORLD_SIZE", 1))
    distributed = num_gpus > 1
    cfg = get_defaults()
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts or [])
    cfg.freeze()
    if cfg.MODEL.DEVICE == "cuda":
        torch.cuda.set_device(args.local_rank)
        # The training process does almost no CPU math, but every multi-threaded host op (a staging memcpy, a reduction over a
        # host tensor) wakes an OpenMP team as wide as the machine whose threads then SPIN for their blocktime -- on a
        # 256-thread host that starved the data-loader workers: next(loader) 31 instead of 6 ms per 2-image batch, 58 instead
        # of 33 ms per iteration (tools/experiments/loader_consume_probe.py).  A small team for the rank's core share.
        if "OMP_NUM_THREADS" not in os.environ:
            torch.set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)) // 4)))
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl" if cfg.MODEL.DEVICE == "cuda" else "gloo", init_method="env://")
        comm.synchronize()
    logging.basicConfig(level=logging.INFO if comm.get_rank() == 0 else logging.WARNING,
                        format="%(asctime)s %(name)s %(levelname)s: %(message)s")
    logging.getLogger("ovis.trainer").info("Using %d GPUs\n%s", num_gpus, args)
    logging.getLogger("ovis.trainer").info("rank %d host cores: %s (%s)", comm.get_rank(), AFFINITY["cpus"], AFFINITY["source"])
    ims_per_gpu = max(cfg.SOLVER.IMS_PER_BATCH // num_gpus, 1)
    logging.getLogger("ovis.trainer").info("%d images per GPU and iteration (SOLVER.IMS_PER_BATCH %d / %d GPUs)", ims_per_gpu,
                                           cfg.SOLVER.IMS_PER_BATCH, num_gpus)
    train(cfg, args.local_rank, distributed, args.max_iter or cfg.SOLVER.MAX_ITER, ims_per_gpu,
          save_checkpoints=bool(cfg.OUTPUT_DIR) and not args.no_checkpoints)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
