"""Oracle (TEST INFRASTRUCTURE ONLY): PyTorch-CPU restatement of the reference's losses.

Restates loss_fns.py:6-204, utils.py:65-85 (polar_transform) and
utils.py:353-371 (cwh_to_xy_torch).  ``dark_loss`` is written in the dense,
mask-multiplied form the HIP kernel uses (no boolean-mask gathers, so no
dynamic shapes); it is pinned against the reference's gather formulation by
the golden fixtures.
"""
import math

import torch
import torch.nn.functional as F


def cnn_loss(scores, y, params=None):
    """loss_fns.py:6-8."""
    logp = F.log_softmax(scores, dim=1)
    return -logp[torch.arange(y.shape[0]), y].sum() / y.shape[0]


def margin_terms(r, t):
    """Shared margin form: t*relu(.9-r)^2 + .5*(1-t)*relu(r-.1)^2 (loss_fns.py:12-16,192-195)."""
    return t * F.relu(0.9 - r) ** 2 + 0.5 * (1.0 - t) * F.relu(r - 0.1) ** 2


def capsule_loss(scores, y, params, x=None, recon=None):
    """loss_fns.py:11-23."""
    onehot = F.one_hot(y, params.n_classes).to(scores.dtype)
    total = margin_terms(scores, onehot).sum()
    if params.recon:
        total = total + params.recon_coef * ((x - recon) ** 2).sum()
    return total / y.shape[0]


def polar_transform(t):
    """utils.py:69-85.  t [...,5] = (r,x,y,w,h) -> (r [...], phi [...,5]).

    Angles: (x*pi, y*pi, h*pi, w*2pi) -- note h before w (utils.py:74).
    phi = (s1, s1 c2, s1 s2 c3, s1 s2 s3 c4, s1 s2 s3 s4); not unit norm.
    """
    r, x, y, w, h = t.unbind(-1)
    a1, a2, a3, a4 = x * math.pi, y * math.pi, h * math.pi, w * math.pi * 2
    s1, s2, s3, s4 = a1.sin(), a2.sin(), a3.sin(), a4.sin()
    c2, c3, c4 = a2.cos(), a3.cos(), a4.cos()
    phi = torch.stack([s1, s1 * c2, s1 * s2 * c3, s1 * s2 * s3 * c4, s1 * s2 * s3 * s4], dim=-1)
    return r, phi


def darkcapsule_loss(caps, y, params, x=None, recon=None):
    """loss_fns.py:187-204.  caps [B,g,g,5], y [B,g,g,5+C] (float64 labels cast to float)."""
    y = y.to(caps.dtype)
    y_r, y_phi = polar_transform(y[..., :5])
    cap_r = (caps ** 2).sum(dim=-1) ** 0.5
    loss = (margin_terms(cap_r, y_r).sum() - (caps * y_phi).sum()) / y.shape[0]
    if params.recon:
        loss = loss + ((x - recon) ** 2).sum()
    return loss


def darkcapsule2_loss(caps, y, params):
    """loss_fns.py:145-160."""
    y = y.to(caps.dtype)
    caps = caps * math.sqrt(2)
    y_r, y_phi = polar_transform(y[..., :5])
    cap_r = (caps ** 2).sum(dim=-1) ** 0.5
    obj = margin_terms(cap_r, y_r).sum()
    coord = -(caps[..., :5] * y_phi).sum()
    cls = ((caps[..., 5:] - y[..., 5:]) ** 2).sum()
    return (obj + coord + cls) / y.shape[0]


def darkcapsule3_loss(caps, y, params, x=None, recon=None):
    """loss_fns.py:163-184.  caps [B,g,g,C,21]."""
    y = y.to(caps.dtype)
    caps = caps * math.sqrt(2)
    y_r, y_phi = polar_transform(y[..., :5])
    t = y[..., 5:] * y_r.unsqueeze(-1)
    cap_r = (caps[..., 5:] ** 2).sum(dim=-1) ** 0.5
    loss = (margin_terms(cap_r, t).sum() - (caps[..., :5] * y_phi.unsqueeze(3)).sum()) / y.shape[0]
    if params.recon:
        loss = loss + ((x - recon) ** 2).sum()
    return loss


def cwh_to_xyxy(cwh, img_size, n_grid):
    """utils.py:353-371 (3-argument version): normalised (xc,yc,w,h) -> corner box, detached."""
    cell = 1.0 * img_size / n_grid
    half_w, half_h = cwh[..., 2] * img_size / 2, cwh[..., 3] * img_size / 2
    cx, cy = cwh[..., 0] * cell, cwh[..., 1] * cell
    return torch.stack([cx - half_w, cy - half_h, cx + half_w, cy + half_h], dim=-1).detach()


def iou_xyxy(a, b):
    """loss_fns.py:26-58 with broadcasting."""
    lt = torch.max(a[..., :2], b[..., :2])
    rb = torch.min(a[..., 2:], b[..., 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    return inter / (area_a + area_b - inter)


def dark_loss(y_pred, y_true, params):
    """loss_fns.py:60-142, dense form.  Returns (loss, avg_iou); the reference stores avg_iou on params.

    y_pred [Bt,g,g,5*nb+C], y_true [Bt,g,g,5+C].  Object cells: y_true[...,0]==1;
    no-object cells: ==0 (a cell that is neither contributes nothing, loss_fns.py:79-80).
    Responsible box = first argmax of IoU over the nb boxes (torch.max, loss_fns.py:104).
    """
    y_true = y_true.to(y_pred.dtype)
    nb, C = params.n_boxes, params.n_classes
    Bt, g = y_true.shape[0], y_true.shape[1]
    boxes = y_pred[..., :5 * nb].reshape(Bt, g, g, nb, 5)
    tbox = y_true[..., :5]
    obj = (tbox[..., 0] == 1).to(y_pred.dtype)                      # [Bt,g,g]
    noobj = (tbox[..., 0] == 0).to(y_pred.dtype)

    pc = boxes[..., 0]                                              # [Bt,g,g,nb]
    iou = iou_xyxy(cwh_to_xyxy(boxes[..., 1:5], params.darknet_input, g),
                   cwh_to_xyxy(tbox[..., None, 1:5], params.darknet_input, g))   # [Bt,g,g,nb]
    iou = torch.where(obj[..., None] > 0, iou, torch.zeros_like(iou))   # only object cells are used
    best_iou, best = iou.max(dim=-1)
    resp = F.one_hot(best, nb).to(y_pred.dtype)                     # [Bt,g,g,nb]

    noobj_pc = (noobj[..., None] * pc ** 2).sum() + (obj[..., None] * (1 - resp) * pc ** 2).sum()
    obj_pc = (obj[..., None] * resp * (pc - best_iou[..., None].detach()) ** 2).sum()
    dxy = boxes[..., 1:3] - tbox[..., None, 1:3]
    obj_xy = (obj[..., None, None] * resp[..., None] * dxy ** 2).sum()
    # sqrt only where it is used: masked-out boxes may be anything (incl. negative)
    sel = (obj[..., None] * resp)[..., None]
    safe_wh = torch.where(sel > 0, boxes[..., 3:5], torch.ones_like(boxes[..., 3:5]))
    safe_t = torch.where(obj[..., None] > 0, tbox[..., 3:5], torch.ones_like(tbox[..., 3:5]))
    dwh = safe_wh.sqrt() - safe_t[..., None, :].sqrt()
    obj_wh = (sel * dwh ** 2).sum()
    obj_cls = 0.0
    if C != 0:
        obj_cls = (obj[..., None] * (y_true[..., 5:] - y_pred[..., 5 * nb:]) ** 2).sum()

    loss = (params.l_coord * obj_xy + params.l_coord * obj_wh + obj_pc
            + params.l_noobj * noobj_pc + obj_cls) / Bt
    n_obj = obj.sum()
    avg_iou = (obj * best_iou).sum() / n_obj
    return loss, avg_iou.detach()
