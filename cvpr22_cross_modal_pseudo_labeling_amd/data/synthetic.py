"""Synthetic COCO-shaped inputs for the hot path (SURVEY.md 8(d) "Synthetic inputs").

There is no network / dataset here, so batches are generated: images ``[B,3,800,1333]`` (U[0,255] minus
INPUT.PIXEL_MEAN), G ground-truth boxes per image with rectangular binary masks and seen-class labels,
W caption nouns per image as caption-vocabulary ids (``ids_cap``, 0-based LVIS ids as in
maskrcnn_benchmark/data/datasets/coco_cap_det.py:166-171), and unit-norm text embeddings that stand in
for the BERT word-embedding rows (language_backbone/transformers.py:24,67).
"""
import torch
import torch.utils.data

from ..modeling.structures import BoxList

LVIS_VOCAB = 1203
COCO_SEEN_PLUS_BG = 49


def make_embeddings(emb_dim=768, seed=1234, device="cpu", n_vocab=LVIS_VOCAB, n_seen=COCO_SEEN_PLUS_BG):
    g = torch.Generator().manual_seed(seed)
    e_vocab = torch.nn.functional.normalize(torch.randn(n_vocab, emb_dim, generator=g), dim=-1)
    e_seen = torch.nn.functional.normalize(torch.randn(n_seen, emb_dim, generator=g), dim=-1)
    e_seen[0] = 0  # background row (datasets/coco.py:85-91)
    return e_vocab.to(device), e_seen.to(device)


def make_batch(batch, device="cpu", seed=1234, height=800, width=1333, num_gt=7, num_nouns=5,
               n_vocab=LVIS_VOCAB, n_seen=COCO_SEEN_PLUS_BG,
               pixel_mean=(102.9801, 115.9465, 122.7717)):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(batch, 3, height, width, generator=g) * 255.0
    images -= torch.tensor(pixel_mean).view(1, 3, 1, 1)
    targets = []
    for _ in range(batch):
        wh = torch.rand(num_gt, 2, generator=g) * torch.tensor([min(368.0, width * 0.5), min(368.0, height * 0.5)]) + 32
        xy = torch.rand(num_gt, 2, generator=g) * (torch.tensor([float(width), float(height)]) - wh - 1)
        boxes = torch.cat([xy, xy + wh], 1).floor()
        t = BoxList(boxes, (width, height))
        t.add_field("labels", torch.randint(1, n_seen, (num_gt,), generator=g))
        masks = torch.zeros(num_gt, height, width, dtype=torch.bool)
        for i, b in enumerate(boxes.tolist()):
            dx, dy = 0.1 * (b[2] - b[0]), 0.1 * (b[3] - b[1])
            masks[i, int(b[1] + dy):int(b[3] - dy) + 1, int(b[0] + dx):int(b[2] - dx) + 1] = True
        t.add_field("masks", masks)
        t.add_field("ids_cap", torch.randperm(n_vocab, generator=g)[:num_nouns])
        t.add_field("is_det", "Yes")
        targets.append(t.to(device))
    return images.to(device), targets


class SyntheticBatches(torch.utils.data.IterableDataset):
    """Endless stream of host batches for ``torch.utils.data.DataLoader(batch_size=None, num_workers=N)``: batch ``it`` is
    ``make_batch(seed = seed0 + 1000 * it + rank)`` whichever worker PROCESS produces it (worker w of N takes
    it = w, w + N, ...; the loader collects the workers round-robin, so the order is it = 0, 1, 2, ...).  Worker
    processes keep the generation (dozens of small host ops per batch) off the training process' interpreter lock --
    the role of DATALOADER.NUM_WORKERS in the reference (data/build.py:154-172)."""

    def __init__(self, batch, seed0=1234, rank=0, **make_batch_kwargs):
        self.batch, self.seed0, self.rank, self.kw = batch, seed0, rank, make_batch_kwargs

    def __iter__(self):
        info = torch.utils.data.get_worker_info()
        it, step = (0, 1) if info is None else (info.id, info.num_workers)
        if info is not None:
            torch.set_num_threads(1)
        while True:
            yield make_batch(self.batch, device="cpu", seed=self.seed0 + 1000 * it + self.rank, **self.kw)
            it += step


@torch.no_grad()
def calibrate_stem_bn(model, images):
    """Random-init stand-in for pretrained FrozenBN statistics: sets the stem's ``running_mean/var`` from the
    conv1 response to ``images`` so that activations are O(1) through the trunk (with identity statistics
    and raw 0..255 pixels they blow up and every RPN box degenerates to the image border)."""
    stem = model.backbone.body.stem
    y = torch.nn.functional.conv2d(images[:1], stem.conv1.weight, None, stem.conv1.stride, stem.conv1.padding)
    stem.bn1.running_mean.copy_(y.mean((0, 2, 3)))
    stem.bn1.running_var.copy_(y.var((0, 2, 3)))  # in-place: bumps the buffer versions every folded-weight cache keys on
