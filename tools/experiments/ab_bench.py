"""A/B of one product switch inside bench.py on the SAME box (boxes of the pool differ by ~2.5 %):
    python tools/experiments/ab_bench.py <patch> [bench args...]
runs bench.py alternately unpatched (A) and with the named monkeypatch applied (B), three times each, and prints ms per step.
Patches: nopool (the res5 head's average pooling as a separate pass), rpnloss_ops (the RPN loss as its tensor-op sequence;
use with --workload teacher), lib:<variant> (a variant library of build_variants.sh), side_prio0 (the frozen half's stream at normal priority), no_prefix (teacher step without the frozen trunk prefix run ahead on the side stream; --workload teacher), gated_nosplit (gated data gradients never K-sliced, the form before round 4's _ws entry point; --workload teacher), rpnloss_one_stream (the teacher's RPN loss on the main stream, after the selection; --workload teacher), rpn_no_ahead (the teacher's RPN backward inside the step's backward instead of run ahead on the second stream; --workload teacher), teacher_heads_plain (the teacher's heads through CombinedROIHeads.forward instead of the one-branch batched form; --workload teacher), drop_trunk_dw (WRONG RESULTS, a bound: the bottlenecks' weight gradients are computed but not handed to autograd -- what their AccumulateGrad adds cost), memo_weight_prep (WRONG RESULTS, a bound: pair forms of the trainable weights computed once and never again -- what the per-step weight preparation costs the step), inline_weight_prep (every trainable block prepares its own GEMM operands in its forward instead of finding them prepared behind the optimizer step), nms_full_scan (the RPN's NMS reduce over all 12000 candidates instead of stopping at post_nms_top_n survivors), block_backward_by_calls (the trunk's identity bottlenecks issue their backward launch by launch on one stream instead of through the one-call, two-stream entry point; --workload teacher), own_branch_stream (the RPN branch and the trunk's weight gradients on a side stream of their own instead of the look-ahead half's; --workload teacher), begin_first (the gradient buffers zeroed and the reducer re-armed in FRONT of the forward, the order until round 6, instead of behind its launches), torch_topk (the RPN's sorted top-k through the tensor library instead of the one-radix-sort op), foreach_sgd (the optimizer step as six multi-tensor passes instead of the fused launch)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PATCHES = {
    "none": "",
    "nopool": "from cvpr22_cross_modal_pseudo_labeling_amd import _C; _C.split_gemm_pair_pool_supported = lambda *a: False",
    "rpnloss_ops": ("from cvpr22_cross_modal_pseudo_labeling_amd.modeling import rpn as _r; "
                    "_r.RPNLossComputation._call_device = _r.RPNLossComputation._call_tensor_ops"),
    "side_prio0": "import torch; _S = torch.cuda.Stream; torch.cuda.Stream = lambda *a, **k: _S(*a, **dict(k, priority=0))",
    "no_prefix": "from cvpr22_cross_modal_pseudo_labeling_amd.modeling.detector import GeneralizedRCNN as _G; del _G.forward_frozen",
    "gated_nosplit": ("from cvpr22_cross_modal_pseudo_labeling_amd import _C; _g = _C.split_gemm_pair_gated; "
                      "_C.split_gemm_pair_gated = lambda *a, **k: _g(*a, **dict(k, config=8))"),
    "rpnloss_one_stream": ("from cvpr22_cross_modal_pseudo_labeling_amd.modeling import rpn as _r; _i = _r.RPNModule.__init__\n"
                           "def _init(self, *a, **k):\n    _i(self, *a, **k); self.loss_beside_selection = self.backward_ahead = False\n"
                           "_r.RPNModule.__init__ = _init"),
    "rpn_no_ahead": ("from cvpr22_cross_modal_pseudo_labeling_amd.modeling import rpn as _r; _i = _r.RPNModule.__init__\n"
                     "def _init(self, *a, **k):\n    _i(self, *a, **k); self.backward_ahead = False\n"
                     "_r.RPNModule.__init__ = _init"),
    "teacher_heads_plain": ("from cvpr22_cross_modal_pseudo_labeling_amd.modeling import detector as _d; _i = _d.GeneralizedRCNN.__init__\n"
                            "def _init(self, *a, **k):\n    _i(self, *a, **k); self.heads_as_one_branch = False\n"
                            "_d.GeneralizedRCNN.__init__ = _init"),
    "drop_trunk_dw": ("from cvpr22_cross_modal_pseudo_labeling_amd.layers import pair_bottleneck as _p; _b = _p._BottleneckPair.backward\n"
                      "def _bw(ctx, *g):\n    out = list(_b(ctx, *g))\n    for i in (3, 6, 9, 12):\n        out[i] = None\n    return tuple(out)\n"
                      "_p._BottleneckPair.backward = staticmethod(_bw)"),
    # (needs tools/experiments/python_patches/trunk_dw_beside.patch applied)
    "dw_beside": "from cvpr22_cross_modal_pseudo_labeling_amd.layers import pair_bottleneck as _p; _p.DW_BESIDE_ROWS = 40000",
    "memo_weight_prep": ("from cvpr22_cross_modal_pseudo_labeling_amd import _C; _w = _C.weight_prep_pair; _m = {}\n"
                         "def _memo(w, scale=None, want_transposed=False):\n"
                         "    k = (w.data_ptr(), 0 if scale is None else scale.data_ptr(), bool(want_transposed))\n"
                         "    if k not in _m:\n        _m[k] = _w(w, scale, want_transposed)\n    return _m[k]\n"
                         "_C.weight_prep_pair = _memo"),
    "inline_weight_prep": ("from cvpr22_cross_modal_pseudo_labeling_amd.engine import trainer as _t; _i = _t.StepPolicy.__init__\n"
                           "def _init(self, *a, **k):\n    _i(self, *a, **k); self.prepare_weights_ahead = False\n"
                           "_t.StepPolicy.__init__ = _init"),
    # (needs tools/experiments/python_patches/nms_stop_at_max_keep.patch applied)
    "nms_full_scan": ("from cvpr22_cross_modal_pseudo_labeling_amd import _C; _n = _C.nms_presorted_batched\n"
                      "_C.nms_presorted_batched = lambda *a, **k: _n(*a, **dict(k, max_keep=0))"),
    "block_backward_by_calls": "from cvpr22_cross_modal_pseudo_labeling_amd.layers import pair_bottleneck as _p; _p.ONE_CALL_BACKWARD_ROWS = 0",
    "own_branch_stream": ("from cvpr22_cross_modal_pseudo_labeling_amd.engine import trainer as _t\n"
                          "_t.branch_stream = lambda: _t.side_stream(0)"),
    "begin_first": ("from cvpr22_cross_modal_pseudo_labeling_amd.engine import trainer as _t; _P = _t.PipelinedTrainer; _i = _P.__init__\n"
                    "def _init(self, model, *a, **k):\n"
                    "    _i(self, model, *a, **k)\n"
                    "    fs = model.forward_student\n"
                    "    def first(frozen, targets, **kw):\n"
                    "        self.policy.begin(self.reducer); self.policy.begin = lambda r: None\n"
                    "        try:\n            return fs(frozen, targets, **kw)\n        finally:\n            del self.policy.begin\n"
                    "    model.forward_student = first\n"
                    "_P.__init__ = _init"),
    "torch_topk": "from cvpr22_cross_modal_pseudo_labeling_amd import _C; _C.topk_sorted = lambda s, k: s.topk(k, dim=1, sorted=True)",
    "foreach_sgd": "from cvpr22_cross_modal_pseudo_labeling_amd.engine import solver as _s; _s.GroupFusedSGD.native = False",
}


def _patch_code(patch):
    if patch.startswith("lib:"):  # a variant library of tools/experiments/build_variants.sh
        path = os.path.join(ROOT, "tools", "experiments", "variants", f"libovis_hip_{patch[4:]}.so")
        return f"from cvpr22_cross_modal_pseudo_labeling_amd import _lib; _lib.LIB_PATH = {path!r}"
    return PATCHES[patch]


def run(patch, args):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.argv = ['bench.py'] + {args!r}\n{_patch_code(patch)}\n"
            "import bench\nbench.main()\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print(out.stderr[-2000:])
        raise SystemExit("bench failed")
    d = json.loads(line[-1])
    return d["ms_per_step"], d["ms_per_step_spread"]


if __name__ == "__main__":
    patch = sys.argv[1]
    args = sys.argv[2:] or ["--steps", "40", "--warmup", "8", "--no-cpu-baseline"]
    for i in range(3):
        for name in ("none", patch):
            ms, spread = run(name, args)
            print(f"round {i} {'A (shipped)' if name == 'none' else 'B (' + patch + ')':14s} {ms:7.3f} ms/step  {spread}", flush=True)
