"""The reference's MERGED configuration trees as data: ``maskrcnn_benchmark/config/defaults.py`` + each shipped yaml file
(+ the override lists the step fixtures are generated under), flattened to {DOTTED.KEY: value} ->
``tests/golden/ref_cfg.json``.  ``tests/test_cfg_vs_reference.py`` holds the product's ``config.get_defaults()`` + the same
yaml + the same overrides to it key by key -- a fixture generated under a silently different configuration (a stand-in
``CfgNode`` that coerces differently from yacs) would show up there.  Build container only (imports /root/reference).

    python tests/golden/make_cfg_golden.py
"""
import json
import os
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import step_case as case  # noqa: E402

YAMLS = ("student_teacher_mask_rcnn_uncertainty.yaml", "zeroshot_mask.yaml")
# the override lists of make_step_golden.py (reduced model / shipped model on the CPU)
OPTS = {"shipped": [], "full_cpu": ["MODEL.DEVICE", "cpu"], "reduced_cpu": list(case.COMMON_OPTS)}


def dump():
    ref_import.install()
    out = {}
    for y in YAMLS:
        for tag, opts in OPTS.items():
            out[f"{y}|{tag}"] = {"opts": [list(o) if isinstance(o, tuple) else o for o in opts],
                                 "cfg": ref_import.flatten_cfg(ref_import.reference_cfg(y, opts))}
        # what the reference's yaml file SETS, as parsed values (data, not the file's text): the product's own copy drops the
        # comments and MODEL.WEIGHT (the author's local checkpoint path); the test merges these values into the product's
        # defaults as well, so the reference's file itself is known to be accepted
        with open(os.path.join(ref_import.REF, "configs", "coco_cap_det", y)) as f:
            out[f"{y}|yaml_values"] = yaml.safe_load(f)
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "ref_cfg.json")
    with open(path, "w") as f:
        json.dump(dump(), f, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")
