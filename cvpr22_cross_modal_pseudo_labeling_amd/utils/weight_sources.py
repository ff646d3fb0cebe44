"""Weight sources other than a local ``.pth``: ``catalog://`` names, URLs and Caffe2 / Detectron ``.pkl`` files.

Counterparts in the reference: ``DetectronCheckpointer._load_file`` (utils/checkpoint.py:132-154), the model half of
``config/paths_catalog.py:340-398`` (``ModelCatalog``), ``utils/model_zoo.py:18-61`` (``cache_url``) and
``utils/c2_model_loading.py:10-206`` (blob-name translation of the C2 ResNet checkpoints).  This box has no network:
a URL resolves to its place in the model cache and loads from there when the file is present; otherwise the download
is attempted and its failure says where to put the file.

The C2 translation here is a small grammar instead of the reference's ordered list of string replacements: a blob name
is parsed (``res<stage>_<block>_branch<which>[_bn]_<w|b|s>``, stem, RPN / box / mask head blobs) and the torch name is
built from the parts; ``tests/golden/c2_names.json`` holds name pairs produced by the reference's own function.
"""
import logging
import os
import pickle
import re
import shutil
import sys
import urllib.parse
import urllib.request

import torch

DETECTRON_URL = "https://dl.fbaipublicfiles.com/detectron"

# ImageNet-pretrained backbones (paths_catalog.py:342-348)
IMAGENET_MODELS = {
    "MSRA/R-50": "ImageNetPretrained/MSRA/R-50.pkl",
    "MSRA/R-50-GN": "ImageNetPretrained/47261647/R-50-GN.pkl",
    "MSRA/R-101": "ImageNetPretrained/MSRA/R-101.pkl",
    "MSRA/R-101-GN": "ImageNetPretrained/47592356/R-101-GN.pkl",
    "FAIR/20171220/X-101-32x8d": "ImageNetPretrained/20171220/X-101-32x8d.pkl",
}
# Detectron 12_2017 baselines: <model id>/<config name> -> signature (paths_catalog.py:351-363)
DETECTRON_BASELINES = {
    "35857197/e2e_faster_rcnn_R-50-C4_1x": "01_33_49.iAX0mXvW",
    "35857345/e2e_faster_rcnn_R-50-FPN_1x": "01_36_30.cUF7QR7I",
    "35857890/e2e_faster_rcnn_R-101-FPN_1x": "01_38_50.sNxI7sX7",
    "36761737/e2e_faster_rcnn_X-101-32x8d-FPN_1x": "06_31_39.5MIHi1fZ",
    "35858791/e2e_mask_rcnn_R-50-C4_1x": "01_45_57.ZgkA7hPB",
    "35858933/e2e_mask_rcnn_R-50-FPN_1x": "01_48_14.DzEQe4wC",
    "35861795/e2e_mask_rcnn_R-101-FPN_1x": "02_31_37.KqyEK4tT",
    "36761843/e2e_mask_rcnn_X-101-32x8d-FPN_1x": "06_35_59.RZotkLKI",
    "37129812/e2e_mask_rcnn_X-152-32x8d-FPN-IN5k_1.44x": "09_35_36.8pzTQKYK",
    "37697547/e2e_keypoint_rcnn_R-50-FPN_1x": "08_42_54.kdzV35ao",
}


def catalog_url(name):
    """``catalog://<name>`` -> URL (``ModelCatalog.get``)."""
    if name.startswith("ImageNetPretrained/"):
        key = name[len("ImageNetPretrained/"):]
        if key not in IMAGENET_MODELS:
            raise RuntimeError(f"model not present in the catalog {name}")
        return f"{DETECTRON_URL}/{IMAGENET_MODELS[key]}"
    if name.startswith("Caffe2Detectron/COCO/"):
        key = name[len("Caffe2Detectron/COCO/"):]
        if key not in DETECTRON_BASELINES:
            raise RuntimeError(f"model not present in the catalog {name}")
        model_id, config = key.split("/")
        tag = "keypoints_" if "keypoint" in name else ""
        tail = f"output/train/{tag}coco_2014_train%3A{tag}coco_2014_valminusminival/generalized_rcnn/model_final.pkl"
        return f"{DETECTRON_URL}/{model_id}/12_2017_baselines/{config}.yaml.{DETECTRON_BASELINES[key]}/{tail}"
    raise RuntimeError(f"model not present in the catalog {name}")


def cache_path(url, model_dir=None):
    """Where ``cache_url`` keeps the file of ``url`` (model_zoo.py:40-48): ``$TORCH_MODEL_ZOO`` or ``$TORCH_HOME/models``."""
    if model_dir is None:
        home = os.path.expanduser(os.getenv("TORCH_HOME", "~/.torch"))
        model_dir = os.getenv("TORCH_MODEL_ZOO", os.path.join(home, "models"))
    path = urllib.parse.urlparse(url).path
    filename = os.path.basename(path)
    if filename == "model_final.pkl":  # every Detectron baseline ends in this name: keep the whole path
        filename = path.replace("/", "_")
    return os.path.join(model_dir, filename)


def cache_url(url, model_dir=None, is_main_process=True):
    """Local file of ``url``: the cached copy when it exists, else downloaded by the main process."""
    target = cache_path(url, model_dir)
    if not os.path.exists(target) and is_main_process:
        sys.stderr.write(f'Downloading: "{url}" to {target}\n')
        try:
            with urllib.request.urlopen(url, timeout=60) as response:  # nothing is created before the server answers
                os.makedirs(os.path.dirname(target), exist_ok=True)
                partial = target + ".partial"
                with open(partial, "wb") as out:
                    shutil.copyfileobj(response, out, 1 << 20)
                os.replace(partial, target)
        except Exception as e:  # no network on the training boxes: say where the file belongs
            raise RuntimeError(f"cannot fetch {url} ({e}); place the file at {target}") from e
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()  # the other ranks read the file the main process fetched (model_zoo.py:59 synchronize)
    if not os.path.exists(target):
        raise RuntimeError(f"{url} is not in the model cache; place the file at {target}")
    return target


_SUFFIX = {"w": "weight", "b": "bias", "s": "weight"}  # an affine / FrozenBN scale is the module's weight
_BRANCH = {"2a": "1", "2b": "2", "2c": "3"}
_HEAD_BLOBS = {  # blobs outside the residual stages (c2_model_loading.py:19-27,88-96): C2 stem -> torch module path
    "conv1": "conv1", "res_conv1_bn": "bn1", "conv1_bn": "bn1", "fc1000": "fc1000", "pred": "fc1000",
    "conv_rpn": "rpn.head.conv", "rpn_cls_logits": "rpn.head.cls_logits", "rpn_bbox_pred": "rpn.head.bbox_pred",
    "bbox_pred": "bbox_pred", "cls_score": "cls_score", "conv5_mask": "conv5_mask", "mask_fcn_logits": "mask_fcn_logits",
    "fc6": "fc6", "fc7": "fc7",
}
_RESIDUAL = re.compile(r"res(\d)_(\d+)_branch(1|2a|2b|2c)(_bn)?_([wbs])")
_PLAIN = re.compile(r"(.+)_([wbs])")


def c2_name_to_torch(blob):
    """Torch parameter name of a C2 ResNet blob (the result of c2_model_loading.py:10-121 for the single-level C4 / C5
    models this package builds), or None for blobs that carry no weight (``*_momentum``)."""
    if "_momentum" in blob:
        return None
    m = _RESIDUAL.fullmatch(blob)
    if m:
        stage, block, which, bn, kind = m.groups()
        base = f"layer{int(stage) - 1}.{block}"
        if which == "1":
            return f"{base}.downsample.{'1' if bn else '0'}.{_SUFFIX[kind]}"
        return f"{base}.{'bn' if bn else 'conv'}{_BRANCH[which]}.{_SUFFIX[kind]}"
    m = _PLAIN.fullmatch(blob)
    if m and m.group(1) in _HEAD_BLOBS:
        return f"{_HEAD_BLOBS[m.group(1)]}.{_SUFFIX[m.group(2)]}"
    raise RuntimeError(f"C2 blob {blob!r}: not a ResNet-C4 / C5 weight this package knows (FPN / GN / keypoint checkpoints "
                       "belong to models it does not build)")


def load_c2_pickle(path, stage_with_dcn=()):
    """``{"model": state_dict}`` of a Caffe2 / Detectron ``.pkl`` (c2_model_loading.py:132-206): ``blobs`` (or the dict
    itself) of numpy arrays, names translated; in stages built with deformable convolutions the 3x3's weights belong to
    the wrapped ``conv2.conv`` (c2_model_loading.py:145-170)."""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    blobs = data["blobs"] if "blobs" in data else data
    log = logging.getLogger(__name__)
    state = {}
    for blob in sorted(blobs):
        name = c2_name_to_torch(blob)
        if name is None:
            continue
        for stage, with_dcn in enumerate(stage_with_dcn, 1):
            if with_dcn and name.startswith(f"layer{stage}.") and ".conv2." in name:
                name = name.replace(".conv2.", ".conv2.conv.")
        log.debug("C2 name: %s mapped name: %s", blob, name)
        state[name] = torch.from_numpy(blobs[blob])
    return dict(model=state)


def is_foreign(f):
    return f.startswith("catalog://") or f.startswith("http") or f.endswith(".pkl")


def resolve(f, cfg, is_main_process=True):
    """``DetectronCheckpointer._load_file`` for non-``.pth`` sources (utils/checkpoint.py:132-154): catalog name -> URL ->
    cached file; a ``.pkl`` goes through the C2 translation, anything else is a torch checkpoint."""
    if f.startswith("catalog://"):
        f = catalog_url(f[len("catalog://"):])
    if f.startswith("http"):
        f = cache_url(f, is_main_process=is_main_process)
    if f.endswith(".pkl"):
        return load_c2_pickle(f, tuple(cfg.MODEL.RESNETS.STAGE_WITH_DCN))
    loaded = torch.load(f, map_location=torch.device("cpu"))
    return loaded if "model" in loaded else dict(model=loaded)
