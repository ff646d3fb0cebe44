"""Config surface of ``tools/train_net.py``: the yacs-style tree of
maskrcnn_benchmark/config/defaults.py:21-581, restricted to the keys the hot path reads (same
names, same defaults), with ``merge_from_file`` (yaml), ``merge_from_list`` (KEY VALUE pairs from the
command line), ``freeze`` and ``clone``.  yacs itself is not a dependency.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"config is frozen; cannot set {name}")
        self[name] = value

    def freeze(self, flag=True):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze(flag)

    def defrost(self):
        self.freeze(False)

    def clone(self):
        c = copy.deepcopy(self)
        return c

    def __deepcopy__(self, memo):
        c = CfgNode()
        for k, v in self.items():
            dict.__setitem__(c, k, copy.deepcopy(v, memo))
        object.__setattr__(c, "_frozen", object.__getattribute__(self, "_frozen"))
        return c

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {path}{k}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{path}{k} must be a mapping")
                self[k]._merge(v, path + k + ".")
            else:
                dict.__setitem__(self, k, _coerce(v, self[k], path + k))

    def merge_from_file(self, filename):
        with open(filename) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("override list must be KEY VALUE pairs")
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {key}")
            if isinstance(value, str):
                try:
                    value = ast.literal_eval(value)
                except (ValueError, SyntaxError):
                    pass
            dict.__setitem__(node, parts[-1], _coerce(value, node[parts[-1]], key))


def _coerce(value, old, key):
    if isinstance(value, str) and isinstance(old, (tuple, list)):
        value = ast.literal_eval(value)
    if isinstance(old, tuple) and isinstance(value, list):
        value = tuple(value)
    if isinstance(old, list) and isinstance(value, tuple):
        value = list(value)
    if isinstance(old, float) and isinstance(value, int) and not isinstance(value, bool):
        value = float(value)
    if old is not None and value is not None and type(old) is not type(value):
        raise ValueError(f"type mismatch for {key}: {type(old).__name__} vs {type(value).__name__}")
    return value


def get_defaults():
    """Defaults of the keys on the hot path (defaults.py line numbers in comments)."""
    return CfgNode({
        "MODEL": {
            "RPN_ONLY": False, "MASK_ON": False, "RETINANET_ON": False, "KEYPOINT_ON": False,  # :24-27
            "DEVICE": "cuda", "META_ARCHITECTURE": "GeneralizedRCNN",  # :28-29
            "CLS_AGNOSTIC_BBOX_REG": False, "CLS_AGNOSTIC_MASK": False, "GT_BOX_EVAL": False,  # :30-32
            "WEIGHT": "", "BACKBONE_PREFIX": "", "LOAD_TRAINER_STATE": True,  # :37-39
            "LOAD_EMB_PRED_FROM_MMSS_HEAD": False, "LOAD_CLASSIFIER": True,  # :40-41
            "MMSS_HEAD": {"TYPES": ("GroundingHead",), "DEFAULT_HEAD": "GroundingHead"},  # :138-140 (checkpoint key rewrite only)
            "LANGUAGE_BACKBONE": {"TYPE": "BERT-Base", "FREEZE": True, "EMBEDDING_PATH": "",  # :130-135
                                  "ADD_POSITION_EMBEDDING": False, "FT_EMB": False},
            "LAMBDA_PSEUDO_LABEL": 0.0, "UNCERTAINTY": False, "RESUME": False,  # :42-44
            "UNCERTAINTY_TRAIN_ITER": 10000, "NO_PSEUDO_MASK": False, "REWEIGHT": True,  # :45-47
            "BACKBONE": {"CONV_BODY": "R-50-C4", "FREEZE_CONV_BODY_AT": 2},  # :125-128
            "RPN": {  # :216-262
                "USE_FPN": False, "ANCHOR_SIZES": (32, 64, 128, 256, 512), "ANCHOR_STRIDE": (16,),
                "ASPECT_RATIOS": (0.5, 1.0, 2.0), "STRADDLE_THRESH": 0, "FG_IOU_THRESHOLD": 0.7,
                "BG_IOU_THRESHOLD": 0.3, "BATCH_SIZE_PER_IMAGE": 256, "POSITIVE_FRACTION": 0.5,
                "PRE_NMS_TOP_N_TRAIN": 12000, "PRE_NMS_TOP_N_TEST": 6000, "POST_NMS_TOP_N_TRAIN": 2000,
                "POST_NMS_TOP_N_TEST": 1000, "NMS_THRESH": 0.7, "MIN_SIZE": 0,
                "RPN_HEAD": "SingleConvRPNHead", "DONT_TRAIN": False,
            },
            "ROI_HEADS": {  # :268-297
                "USE_FPN": False, "FG_IOU_THRESHOLD": 0.5, "BG_IOU_THRESHOLD": 0.5,
                "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "BATCH_SIZE_PER_IMAGE": 512,
                "POSITIVE_FRACTION": 0.25, "SCORE_THRESH": 0.05, "NMS": 0.5, "DETECTIONS_PER_IMG": 100,
            },
            "ROI_BOX_HEAD": {  # :300-322
                "FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor", "PREDICTOR": "FastRCNNPredictor",
                "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_SCALES": (1.0 / 16,),
                "NUM_CLASSES": 81, "EMB_DIM": 300, "EMBEDDING_BASED": False, "LOSS_WEIGHT_BACKGROUND": 1.0,
                "FREEZE_EMB_PRED": False, "FREEZE_FEATURE_EXTRACTOR": False,
            },
            "ROI_MASK_HEAD": {  # :324-340
                "FEATURE_EXTRACTOR": "ResNet50Conv5ROIFeatureExtractor", "PREDICTOR": "MaskRCNNC4Predictor",
                "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "POOLER_SCALES": (1.0 / 16,),
                "CONV_LAYERS": (256, 256, 256, 256), "RESOLUTION": 14, "SHARE_BOX_FEATURE_EXTRACTOR": True,
                "POSTPROCESS_MASKS": False, "POSTPROCESS_MASKS_THRESHOLD": 0.5,
            },
            "RESNETS": {  # :359-386
                "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True,
                "TRANS_FUNC": "BottleneckWithFixedBatchNorm", "STEM_FUNC": "StemWithFixedBatchNorm",
                "RES5_DILATION": 1, "BACKBONE_OUT_CHANNELS": 1024, "RES2_OUT_CHANNELS": 256,
                "STEM_OUT_CHANNELS": 64, "STAGE_WITH_DCN": (False, False, False, False),
                "WITH_MODULATED_DCN": False, "DEFORMABLE_GROUPS": 1,
            },
        },
        "INPUT": {  # :54-76
            "MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333,
            "PIXEL_MEAN": [102.9801, 115.9465, 122.7717], "PIXEL_STD": [1.0, 1.0, 1.0], "TO_BGR255": True,
        },
        "DATASETS": {"TRAIN": (), "TEST": (), "DATASET_CLASS": "COCODataset",
                     "DATASET_ARGS": {"LOAD_EMBEDDINGS": False, "EMB_KEY": "GloVE", "EMB_DIM": 300}},
        "DATALOADER": {"NUM_WORKERS": 4, "SIZE_DIVISIBILITY": 0, "ASPECT_RATIO_GROUPING": True},
        "SOLVER": {  # :491-528
            "MAX_ITER": 40000, "BASE_LR": 0.001, "BIAS_LR_FACTOR": 2, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005,
            "WEIGHT_DECAY_BIAS": 0, "GAMMA": 0.1, "STEPS": (30000,), "WARMUP_FACTOR": 1.0 / 3,
            "WARMUP_ITERS": 500, "WARMUP_METHOD": "linear", "CHECKPOINT_PERIOD": 10000, "TEST_PERIOD": 10000,
            "LOG_PERIOD": 20, "IMS_PER_BATCH": 16, "CLIP_GRAD_NORM_AT": -1.0, "GRADIENT_ACCUMULATION_STEPS": 1,
            "SKIP_VAL_LOSS": False, "UNCERTAINTY_LR_FACTOR": 1.0,
        },
        "TEST": {"IMS_PER_BATCH": 8, "DETECTIONS_PER_IMG": 100},
        "OUTPUT_DIR": ".",
        "DTYPE": "float32",
    })


cfg = get_defaults()
