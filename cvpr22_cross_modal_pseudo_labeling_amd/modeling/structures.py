"""Box containers and box arithmetic with the reference's +1-pixel convention.

Behavioural counterpart of maskrcnn_benchmark/structures/bounding_box.py:9-255 (``BoxList``),
structures/boxlist_ops.py:9-129 (``boxlist_nms / remove_small_boxes / boxlist_iou / cat_boxlist``)
and structures/image_list.py:7-72, reduced to what the training hot path touches: boxes are always
``xyxy`` float32 tensors on the device, fields are plain tensors (or python objects), and binary
instance masks are a ``[G, H, W]`` tensor field instead of a ``SegmentationMask`` wrapper.
"""
import torch

from ..layers import nms as _nms

TO_REMOVE = 1.0


class BoxList:
    """xyxy boxes of one image + named per-box fields.  ``size`` is (width, height)."""

    def __init__(self, bbox, image_size, mode="xyxy"):
        bbox = torch.as_tensor(bbox, dtype=torch.float32)
        if bbox.dim() != 2 or bbox.size(-1) != 4:
            raise ValueError(f"bbox should be [N,4], got {tuple(bbox.shape)}")
        if mode == "xywh":  # bounding_box.py:83-93
            x, y, w, h = bbox.unbind(1)
            bbox = torch.stack((x, y, x + (w - TO_REMOVE).clamp(min=0), y + (h - TO_REMOVE).clamp(min=0)), 1)
        elif mode != "xyxy":
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox = bbox
        self.size = tuple(image_size)
        self.mode = "xyxy"
        self.extra_fields = {}

    # -- fields ---------------------------------------------------------------
    def add_field(self, name, value):
        self.extra_fields[name] = value

    def get_field(self, name):
        return self.extra_fields[name]

    def has_field(self, name):
        return name in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size)
        for f in [fields] if isinstance(fields, str) else fields:
            if self.has_field(f):
                out.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError(f"field '{f}' not found")
        return out

    # -- tensor-like ------------------------------------------------------------
    def __len__(self):
        return self.bbox.shape[0]

    def __getitem__(self, item):
        if torch.is_tensor(item) and item.dtype == torch.bool:
            item = torch.nonzero(item).squeeze(1)  # ONE nonzero instead of one per indexed field
        out = BoxList(self.bbox[item], self.size)
        for k, v in self.extra_fields.items():
            if torch.is_tensor(v):
                out.add_field(k, v[item] if v.dim() > 0 and v.shape[0] == len(self) else v)
            elif isinstance(v, PolygonMasks):  # per-box polygon lists follow the boxes (bounding_box.py:233-242)
                out.add_field(k, v[item])
            else:
                out.add_field(k, v)
        return out

    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size)
        for k, v in self.extra_fields.items():
            out.add_field(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def area(self):
        b = self.bbox
        return (b[:, 2] - b[:, 0] + TO_REMOVE) * (b[:, 3] - b[:, 1] + TO_REMOVE)

    def clip_to_image(self, remove_empty=True):  # bounding_box.py:214-225
        w, h = self.size
        b = self.bbox
        b[:, 0].clamp_(min=0, max=w - TO_REMOVE)
        b[:, 1].clamp_(min=0, max=h - TO_REMOVE)
        b[:, 2].clamp_(min=0, max=w - TO_REMOVE)
        b[:, 3].clamp_(min=0, max=h - TO_REMOVE)
        if remove_empty:
            keep = (b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])
            return self[keep]
        return self

    def __repr__(self):
        return f"BoxList(num_boxes={len(self)}, image_width={self.size[0]}, image_height={self.size[1]})"


class PastedMasks:
    """Binary instance masks of one image that are DEFINED by per-instance probability maps and boxes through the
    Masker paste (mask_head/inference.py:100-160) -- the pseudo labels' masks (st_generalized_rcnn.py:266-271) -- kept in
    that form: the student's mask targets are computed from the maps directly (``_C.project_pasted_masks``), so the
    H x W canvases are only built when something asks for them (``materialize``).  probs [G, M, M], boxes [G, 4],
    image_size = (height, width)."""

    def __init__(self, probs, boxes, image_size, threshold=0.5, padding=1):
        assert padding == 1, "the device kernel implements the reference's Masker(padding=1)"
        self.probs, self.boxes, self.image_size, self.threshold, self.padding = probs, boxes, tuple(image_size), threshold, padding

    def __len__(self):
        return self.probs.shape[0]

    def to(self, device):
        return PastedMasks(self.probs.to(device), self.boxes.to(device), self.image_size, self.threshold, self.padding)

    def materialize(self):
        """bool [G, H, W]: what Masker(threshold, padding) pastes."""
        from .roi_heads import Masker
        h, w = self.image_size
        return Masker(self.threshold, self.padding)(self.probs[:, None], BoxList(self.boxes, (w, h)))[:, 0]


class PolygonMasks:
    """Polygon ground-truth masks of one image, the reference's ``SegmentationMask(polygons, size, mode='poly')``
    (structures/segmentation_mask.py:348-478 over PolygonInstance :208-347) in the flat form the device kernel reads:
    ``coords`` float32 [T] -- the (x, y) pairs of every polygon back to back --, ``polygon_start`` int32 [NP + 1] (offsets
    in floats) and ``instance_start`` int32 [G + 1] (polygon ranges of the G instances).  ``size`` = (width, height).
    Polygons with fewer than 3 vertices are dropped at construction, as ``PolygonInstance.__init__`` does.  The mask
    head's targets come from ``_C.project_polygon_masks`` (crop -> resize -> rasterise per positive, one launch);
    ``convert_to_binarymask`` rasterises whole-image masks with the same kernel."""

    def __init__(self, instances, size, _flat=None):
        self.size = tuple(size)
        if _flat is not None:
            self.coords, self.polygon_start, self.instance_start = _flat
            return
        coords, pstart, istart = [], [0], [0]
        for polys in instances:
            for poly in polys:
                flat = [float(v) for v in (poly.tolist() if hasattr(poly, "tolist") else poly)]
                if len(flat) >= 6:  # 3 * 2 coordinates (segmentation_mask.py:227)
                    coords.extend(flat[: len(flat) // 2 * 2])
                    pstart.append(len(coords))
            istart.append(len(pstart) - 1)
        self.coords = torch.tensor(coords, dtype=torch.float32)
        self.polygon_start = torch.tensor(pstart, dtype=torch.int32)
        self.instance_start = torch.tensor(istart, dtype=torch.int32)

    def __len__(self):
        return self.instance_start.numel() - 1

    def to(self, device):
        return PolygonMasks(None, self.size, (self.coords.to(device), self.polygon_start.to(device),
                                              self.instance_start.to(device)))

    def instances(self):
        """Back to the nested-list form: per instance a list of flat float32 polygons (host tensors)."""
        c, ps, ist = self.coords.cpu(), self.polygon_start.tolist(), self.instance_start.tolist()
        return [[c[ps[q]:ps[q + 1]] for q in range(ist[g], ist[g + 1])] for g in range(len(self))]

    def __getitem__(self, item):
        """Instances selected by an int, a slice, an index tensor or a boolean mask (SegmentationMask.__getitem__,
        segmentation_mask.py:437-458) -- re-packed on the host: ground-truth lists are a handful of instances."""
        inst = self.instances()
        if isinstance(item, int):
            picked = [inst[item]]
        elif isinstance(item, slice):
            picked = inst[item]
        else:
            item = torch.as_tensor(item).cpu()
            idx = torch.nonzero(item).squeeze(1).tolist() if item.dtype == torch.bool else item.reshape(-1).tolist()
            picked = [inst[i] for i in idx]
        return PolygonMasks(picked, self.size).to(self.coords.device)

    def transpose(self, method):
        """FLIP_LEFT_RIGHT (0) / FLIP_TOP_BOTTOM (1): segmentation_mask.py:250-268 (``dim - p - 1`` on one coordinate)."""
        if method not in (0, 1):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        c = self.coords.clone()
        c[method::2] = self.size[method] - self.coords[method::2] - TO_REMOVE
        return PolygonMasks(None, self.size, (c, self.polygon_start, self.instance_start))

    def convert_to_binarymask(self):
        """uint8 [G, height, width]: every instance rasterised over the whole image (segmentation_mask.py:326-334)."""
        from .. import _C
        w, h = self.size
        g = len(self)
        if not self.coords.is_cuda:  # host polygons (dataset side, MODEL.DEVICE cpu): any image size, libovis_cpu.so
            from .. import _cpu
            return _cpu.polygons_to_masks(self.coords, self.polygon_start, self.instance_start, self.size)
        if max(w, h) > 64:
            raise NotImplementedError("whole-image rasterisation of DEVICE polygons is wired for maps up to 64 x 64 (the training "
                                      "path rasterises per positive at the mask resolution); convert on the host: "
                                      "``masks.to('cpu').convert_to_binarymask()``")
        if w != h:
            raise NotImplementedError("square maps only (crop + resize to M x M is the training path)")
        boxes = torch.tensor([[0.0, 0.0, float(w), float(h)]], device=self.coords.device).expand(g, 4).contiguous()
        idx = torch.arange(g, device=self.coords.device)
        return _C.project_polygon_masks(self.coords, self.polygon_start, self.instance_start, idx, boxes, self.size, w).to(torch.uint8)


def cat_boxlist(boxlists):  # boxlist_ops.py:107-129
    size = boxlists[0].size
    fields = set(boxlists[0].fields())
    assert all(b.size == size and set(b.fields()) == fields for b in boxlists)
    out = BoxList(torch.cat([b.bbox for b in boxlists], 0), size)
    for f in fields:
        out.add_field(f, torch.cat([b.get_field(f) for b in boxlists], 0))
    return out


def box_iou(a, b):
    """IoU matrix [len(a), len(b)] of two xyxy tensors (boxlist_ops.py:53-89)."""
    area_a = (a[:, 2] - a[:, 0] + TO_REMOVE) * (a[:, 3] - a[:, 1] + TO_REMOVE)
    area_b = (b[:, 2] - b[:, 0] + TO_REMOVE) * (b[:, 3] - b[:, 1] + TO_REMOVE)
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt + TO_REMOVE).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area_a[:, None] + area_b - inter)


def boxlist_iou(boxlist1, boxlist2):
    if boxlist1.size != boxlist2.size:
        raise RuntimeError(f"boxlists should have same image size, got {boxlist1}, {boxlist2}")
    return box_iou(boxlist1.bbox, boxlist2.bbox)


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):  # boxlist_ops.py:9-31
    if nms_thresh <= 0:
        return boxlist
    keep = _nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep.to(boxlist.bbox.device)]


def remove_small_boxes(boxlist, min_size):  # boxlist_ops.py:34-49
    b = boxlist.bbox
    ws = b[:, 2] - b[:, 0] + TO_REMOVE
    hs = b[:, 3] - b[:, 1] + TO_REMOVE
    return boxlist[(ws >= min_size) & (hs >= min_size)]


class ImageList:
    """Batched images padded to a common size (image_list.py:7-72)."""

    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = list(image_sizes)  # (height, width) per image

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    if isinstance(tensors, ImageList):
        return tensors
    if torch.is_tensor(tensors):
        if tensors.dim() == 3:
            tensors = tensors[None]
        return ImageList(tensors, [tuple(t.shape[-2:]) for t in tensors])
    max_size = [max(s) for s in zip(*[img.shape for img in tensors])]
    if size_divisible > 0:
        import math

        max_size[1] = int(math.ceil(max_size[1] / size_divisible) * size_divisible)
        max_size[2] = int(math.ceil(max_size[2] / size_divisible) * size_divisible)
    batched = tensors[0].new_zeros((len(tensors), *max_size))
    for img, pad in zip(tensors, batched):
        pad[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
    return ImageList(batched, [tuple(im.shape[-2:]) for im in tensors])
