"""Import machinery for running the REFERENCE's own detector classes in the build container (never on the GPU box,
never from the product): ``/root/reference`` is put on ``sys.path`` behind a handful of ``sys.modules`` stand-ins for
packages this image lacks.  Used by ``make_step_golden.py`` only; the fixtures it writes are data.

Stand-ins (each only as wide as the reference's import statements / call sites need):
  * ``apex.amp.float_function``            identity decorator (layers/roi_align.py:8, nms.py:5)
  * ``maskrcnn_benchmark._C``              oracle/_ref/ref_C.so = the reference's csrc compiled for the CPU
  * ``yacs.config.CfgNode``                attribute dict with clone / merge_from_file / merge_from_list / freeze
  * ``cv2``, ``pycocotools(.mask)``        empty modules (structures/segmentation_mask.py:1,7: polygon paths unused here)
  * ``spacy``, ``nltk(.corpus)``, ``tqdm`` empty modules (data/datasets/helper/parser.py:3-6: LVISParser unused here)
  * ``maskrcnn_benchmark.data(.datasets(.helper))``  bare namespace packages so that ``helper/lvis_v1_categories.py``
    and ``helper/parser.py`` load without ``data/__init__`` (torchvision)
"""
import ast
import copy
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


class CfgNode(dict):
    """The slice of yacs.config.CfgNode the reference touches (config/defaults.py:4, tools/train_net.py:197-199)."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    # yacs semantics (yacs/config.py: _decode_cfg_value, _check_and_coerce_cfg_value_type, merge_from_list): string values are
    # decoded with ast.literal_eval when they parse, a replacement must have the type of the value it replaces (tuple <->
    # list is cast to the original's type), a key that does not exist is an error.  A value that would silently change a
    # key's type -- and with it the configuration the fixture is generated for -- raises here as it does under yacs.
    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return CfgNode(v)
        if not isinstance(v, str):
            return v
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    @staticmethod
    def _check_and_coerce(replacement, original, key):
        if type(replacement) is type(original):
            return replacement
        for from_type, to_type in ((list, tuple), (tuple, list)):
            if type(replacement) is from_type and type(original) is to_type:
                return to_type(replacement)
        raise ValueError(f"Type mismatch ({type(original)} vs. {type(replacement)}) with values ({original!r} vs. "
                         f"{replacement!r}) for config key: {key}")

    def _merge(self, other, path=""):
        for k, v in other.items():
            full = path + k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            v = self._decode(copy.deepcopy(v))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{full} must be a mapping")
                self[k]._merge(v, full + ".")
            else:
                self[k] = self._check_and_coerce(v, self[k], full)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "override list must be KEY VALUE pairs"
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {k}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {k}")
            node[parts[-1]] = self._check_and_coerce(self._decode(v), node[parts[-1]], k)


def flatten_cfg(node, prefix=""):
    """{DOTTED.KEY: value} of a config tree; tuples as lists (JSON has one sequence type)."""
    out = {}
    for k, v in node.items():
        if isinstance(v, dict):
            out.update(flatten_cfg(v, prefix + k + "."))
        else:
            out[prefix + k] = list(v) if isinstance(v, tuple) else v
    return out


def _empty(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _namespace_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def install():
    """Returns the reference's compiled CPU module after making ``import maskrcnn_benchmark...`` work."""
    sys.path.insert(0, ROOT)
    import oracle

    ref_c = oracle.ref_module()
    assert ref_c is not None, "run `python oracle/build_ref.py` first"
    amp = _empty("apex.amp", float_function=lambda f: f)
    _empty("apex", amp=amp)
    yacs = _empty("yacs")
    yacs.config = _empty("yacs.config", CfgNode=CfgNode)
    _empty("cv2")
    pc = _empty("pycocotools")
    pc.mask = _empty("pycocotools.mask")
    _empty("spacy")
    nltk = _empty("nltk")
    nltk.corpus = _empty("nltk.corpus", wordnet=None)
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except ImportError:
            _empty("tqdm", tqdm=lambda x, *a, **k: x)
    if not hasattr(np, "float"):
        np.float = float  # rpn/anchor_generator.py:227-228 uses the alias numpy >= 1.24 dropped
    six = _empty("torch._six", PY3=True, string_classes=(str,))
    torch._six = six
    sys.path.insert(0, REF)
    import maskrcnn_benchmark

    sys.modules["maskrcnn_benchmark._C"] = ref_c
    maskrcnn_benchmark._C = ref_c
    mb = os.path.join(REF, "maskrcnn_benchmark")
    _namespace_pkg("maskrcnn_benchmark.data", os.path.join(mb, "data"))
    _namespace_pkg("maskrcnn_benchmark.data.datasets", os.path.join(mb, "data", "datasets"))
    _namespace_pkg("maskrcnn_benchmark.data.datasets.helper", os.path.join(mb, "data", "datasets", "helper"))
    return ref_c


def load_file_as(name, relpath):
    """Loads one reference file as module ``name`` without running its package's ``__init__``
    (detector/__init__ -> detectors.py:3 -> mmss_gcnn is not importable)."""
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_cfg(yaml_name, opts=()):
    """The reference's real defaults.py + one of its shipped yaml files + KEY VALUE overrides."""
    from maskrcnn_benchmark.config import cfg

    c = cfg.clone()
    c.merge_from_file(os.path.join(REF, "configs", "coco_cap_det", yaml_name))
    c.merge_from_list(list(opts))
    c.freeze()
    return c
