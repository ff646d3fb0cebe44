"""Checkpoint wire format of the reference, so that its published / intermediate ``.pth`` files load here.

Counterparts: maskrcnn_benchmark/utils/model_serialization.py:10-89 (suffix alignment of state-dict keys, ``module.``
prefix of DistributedDataParallel) and utils/checkpoint.py:14-154 (``Checkpointer`` file layout: ``model_<iter>.pth``
holding ``{"model", "optimizer", "scheduler", **extras}`` + a ``last_checkpoint`` tag file; ``DetectronCheckpointer``
key-rewrite rules).  ``catalog://`` names, URLs and Caffe2 ``.pkl`` files go through ``utils/weight_sources.py``.
"""
import logging
import os

import torch

from . import weight_sources


def strip_prefix_if_present(state_dict, prefix):
    """Drop ``prefix`` from every key when ALL keys carry it (a DataParallel / DDP checkpoint), else leave the dict."""
    if not state_dict or not all(k.startswith(prefix) for k in state_dict):
        return state_dict
    return type(state_dict)((k[len(prefix):], v) for k, v in state_dict.items())


def align_and_update_state_dicts(model_state_dict, loaded_state_dict, replace_substr_dict=None, logger=None):
    """For every model key pick the loaded key that is its LONGEST suffix (after the substring rewrites), the
    reference's rule for weights saved under fewer / other module prefixes; unmatched model keys keep their values."""
    renamed = {}
    for key in sorted(loaded_state_dict):
        new = key
        for old, sub in (replace_substr_dict or {}).items():
            if old in key:
                new = new.replace(old, sub)
        renamed.setdefault(new, key)  # first (sorted) original wins on a rename collision, like max() over the match row
    logger = logger or logging.getLogger(__name__)
    matched = {}
    for key in sorted(model_state_dict):
        # candidate suffixes: the key itself and everything after each '.', longest first
        parts = key.split(".")
        for i in range(len(parts)):
            cand = ".".join(parts[i:])
            if cand in renamed:
                src = renamed[cand]
                model_state_dict[key] = loaded_state_dict[src]
                matched[key] = src
                logger.debug("%s loaded from %s of shape %s", key, src, tuple(loaded_state_dict[src].shape))
                break
    return matched


def load_state_dict(model, loaded_state_dict, replace_substr_dict=None):
    """Strict load of the model's own state dict after the suffix alignment (model_serialization.py:77-89)."""
    model_state_dict = model.state_dict()
    loaded_state_dict = strip_prefix_if_present(loaded_state_dict, "module.")
    matched = align_and_update_state_dicts(model_state_dict, loaded_state_dict, replace_substr_dict)
    model.load_state_dict(model_state_dict)
    return matched


class Checkpointer:
    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=True, logger=None,
                 replace_substr_dict=None):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.save_dir, self.save_to_disk = save_dir, save_to_disk
        self.logger = logger or logging.getLogger(__name__)
        self.replace_substr_dict = dict(replace_substr_dict or {})

    # -- writing ------------------------------------------------------------------------------------------
    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return None
        data = {"model": self.model.state_dict()}
        if self.optimizer is not None:
            data["optimizer"] = self.optimizer.state_dict()
        if self.scheduler is not None:
            data["scheduler"] = self.scheduler.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, f"{name}.pth")
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(path)
        return path

    # -- reading ------------------------------------------------------------------------------------------
    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                return f.read().strip()
        except OSError:
            return ""

    def _load_file(self, f):
        loaded = torch.load(f, map_location=torch.device("cpu"))
        return loaded if "model" in loaded else dict(model=loaded)

    def load(self, f=None, use_latest=True, load_trainer_state=True):
        if use_latest and self.has_checkpoint():
            f = self.get_checkpoint_file()  # a run directory resumes from its own last checkpoint
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        self.logger.info("Loading checkpoint from %s", f)
        checkpoint = self._load_file(f)
        load_state_dict(self.model, checkpoint.pop("model"), self.replace_substr_dict)
        if load_trainer_state and "optimizer" in checkpoint and self.optimizer is not None:
            self.optimizer.load_state_dict(checkpoint.pop("optimizer"))
        if load_trainer_state and "scheduler" in checkpoint and self.scheduler is not None:
            self.scheduler.load_state_dict(checkpoint.pop("scheduler"))
        return checkpoint  # whatever else was stored (e.g. "iteration")


class DetectronCheckpointer(Checkpointer):
    """Key-rewrite rules of utils/checkpoint.py:103-131: ``backbone_prefix`` is removed from loaded keys, the
    vision-to-language projection of an MMSS pre-training head becomes the box predictor's ``emb_pred``, another
    predictor class name replaces ``FastRCNNPredictor`` and ``load_classifier=False`` keeps ``cls_score`` out."""

    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=True, logger=None,
                 replace_substr_dict=None, backbone_prefix="", load_emb_pred_from=None, load_classifier=True):
        rules = dict(replace_substr_dict or {})
        if backbone_prefix:
            rules[backbone_prefix] = ""
        if load_emb_pred_from is not None:
            rules[f"mmss_heads.{load_emb_pred_from}.v2l_projection"] = "roi_heads.box.predictor.emb_pred"
        if cfg.MODEL.ROI_BOX_HEAD.PREDICTOR != "FastRCNNPredictor":
            rules["FastRCNNPredictor"] = cfg.MODEL.ROI_BOX_HEAD.PREDICTOR
        if not load_classifier:
            rules["predictor.cls_score"] = "predictor.DONT_LOAD.cls_score"
        super().__init__(model, optimizer, scheduler, save_dir, save_to_disk, logger, rules)
        self.cfg = cfg

    def _load_file(self, f):
        # catalog:// name -> URL -> model cache; Caffe2 / Detectron .pkl -> translated blob names (utils/checkpoint.py:132-154)
        if weight_sources.is_foreign(f):
            return weight_sources.resolve(f, self.cfg, is_main_process=self.save_to_disk)
        return super()._load_file(f)
