# torchvision is not installed here; GIoU restated from the published torchvision 0.20.1 formula
# (ops/giou_loss.py + ops/_utils.py::_loss_inter_union). PARITY UNPINNED for this function.
import torch


def generalized_box_iou_loss(boxes1, boxes2, reduction="none", eps=1e-7):
    if not boxes1.is_floating_point():
        boxes1 = boxes1.float()
    if not boxes2.is_floating_point():
        boxes2 = boxes2.float()
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xkis1, ykis1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xkis2, ykis2 = torch.min(x2, x2g), torch.min(y2, y2g)
    intsctk = torch.zeros_like(x1)
    mask = (ykis2 > ykis1) & (xkis2 > xkis1)
    intsctk[mask] = (xkis2[mask] - xkis1[mask]) * (ykis2[mask] - ykis1[mask])
    unionk = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsctk
    iouk = intsctk / (unionk + eps)
    xc1, yc1, xc2, yc2 = torch.min(x1, x1g), torch.min(y1, y1g), torch.max(x2, x2g), torch.max(y2, y2g)
    area_c = (xc2 - xc1) * (yc2 - yc1)
    miouk = iouk - ((area_c - unionk) / (area_c + eps))
    loss = 1 - miouk
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss
