"""Module forms of deformable RoI pooling (maskrcnn_benchmark/layers/dcn/deform_pool_module.py:6-150): the plain op, the
variant that predicts its own offsets from a first undeformed pooling, and the modulated one that also predicts a
per-bin mask.  Parameter names (``offset_fc``, ``mask_fc``) are the reference's, so its checkpoints load."""
from torch import nn

from .deform_pool_func import deform_roi_pooling


class DeformRoIPooling(nn.Module):
    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None, sample_per_part=4,
                 trans_std=.0):
        super().__init__()
        self.spatial_scale, self.out_size, self.out_channels, self.no_trans = spatial_scale, out_size, out_channels, no_trans
        self.group_size = group_size
        self.part_size = out_size if part_size is None else part_size
        self.sample_per_part, self.trans_std = sample_per_part, trans_std

    def pool(self, data, rois, offset, no_trans):
        return deform_roi_pooling(data, rois, offset, self.spatial_scale, self.out_size, self.out_channels, no_trans,
                                  self.group_size, self.part_size, self.sample_per_part, self.trans_std)

    def forward(self, data, rois, offset):
        if self.no_trans:
            offset = data.new_empty(0)
        return self.pool(data, rois, offset, self.no_trans)


def _fc_stack(n_in, hidden, n_out, depth):
    layers, c = [], n_in
    for _ in range(depth):
        layers += [nn.Linear(c, hidden), nn.ReLU(inplace=True)]
        c = hidden
    last = nn.Linear(c, n_out)
    nn.init.zeros_(last.weight)  # the predicted offsets / mask logits start at zero
    nn.init.zeros_(last.bias)
    return layers + [last]


class DeformRoIPoolingPack(DeformRoIPooling):
    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None, sample_per_part=4,
                 trans_std=.0, deform_fc_channels=1024):
        super().__init__(spatial_scale, out_size, out_channels, no_trans, group_size, part_size, sample_per_part, trans_std)
        self.deform_fc_channels = deform_fc_channels
        if not no_trans:
            self.offset_fc = nn.Sequential(*_fc_stack(out_size * out_size * out_channels, deform_fc_channels,
                                                      out_size * out_size * 2, 2))

    def predicted_offset(self, data, rois):
        n = rois.shape[0]
        x = self.pool(data, rois, data.new_empty(0), True)
        return x, self.offset_fc(x.view(n, -1)).view(n, 2, self.out_size, self.out_size)

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self.pool(data, rois, data.new_empty(0), True)
        _, offset = self.predicted_offset(data, rois)
        return self.pool(data, rois, offset, False)


class ModulatedDeformRoIPoolingPack(DeformRoIPoolingPack):
    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None, sample_per_part=4,
                 trans_std=.0, deform_fc_channels=1024):
        super().__init__(spatial_scale, out_size, out_channels, no_trans, group_size, part_size, sample_per_part, trans_std,
                         deform_fc_channels)
        if not no_trans:
            self.mask_fc = nn.Sequential(*_fc_stack(out_size * out_size * out_channels, deform_fc_channels,
                                                    out_size * out_size, 1), nn.Sigmoid())

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self.pool(data, rois, data.new_empty(0), True)
        n = rois.shape[0]
        x, offset = self.predicted_offset(data, rois)
        mask = self.mask_fc(x.view(n, -1)).view(n, 1, self.out_size, self.out_size)
        return self.pool(data, rois, offset, False) * mask
