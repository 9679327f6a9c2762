"""Detection metrics of the reference on the device: metrics.detect_acc / detect_and_recog_acc / detect_AP
(metrics.py:193-282).

The reference decodes both arrays to boxes with numpy and matches them with two nested Python loops per image
(metrics.py:136-147) every `eval_every` epochs; here the decoding (`cy_yolo_decode_boxes`) and the IoU matching
(`cy_detect_confusion`, one block per image) stay on the GPU and only TP / FP / FN come back.
"""
import numpy as np
import torch

from . import utils
from ._lib import call


def _confusion_of_boxes(gt, pr, y, y_hat, params, iou_th):
    (n1, gi, gxy, _), (n2, pi, pxy, _) = gt, pr
    batch = int(y.shape[0])
    g = int(y.shape[1])
    nb_max = max((int(y.shape[3]) - int(params.n_classes)) // 5, (int(y_hat.shape[3]) - int(params.n_classes)) // 5)
    out = torch.zeros(4, dtype=torch.int32, device='cuda')
    call('cy_detect_confusion', gi.data_ptr() if n1 else None, gxy.data_ptr() if n1 else None, n1,
         pi.data_ptr() if n2 else None, pxy.data_ptr() if n2 else None, n2, batch, float(iou_th), g * g * nb_max,
         out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    tp, fp, fn, bad = [int(v) for v in out.cpu().numpy()]
    if bad:
        raise AssertionError('malformed box (x1 > x2 or y1 > y2) in %d case(s)' % bad)
    return tp, fp, fn


def detect_confusion(y, y_hat, params, conf_th=0.5, iou_th=0.5):
    """(TP, FP, FN) summed over the batch; raises AssertionError on a malformed box like metrics.py:114-119."""
    return _confusion_of_boxes(utils.decode_boxes_device(y, params, None, conf_th),
                               utils.decode_boxes_device(y_hat, params, None, conf_th), y, y_hat, params, iou_th)


def precision_and_recall(tp, fp, fn):
    """metrics.py:150-160."""
    return (tp / (tp + fp) if tp + fp else 0.0), (tp / (tp + fn) if tp + fn else 0.0)


def detect_acc(y, y_hat, params):
    """metrics.py:245-262: F1 of the detector at confidence 0.5 / IoU 0.5."""
    p, r = precision_and_recall(*detect_confusion(y, y_hat, params))
    return 2 * p * r / (p + r + 1e-8)


def detect_and_recog_confusion(y, y_hat, params, conf_th=0.5, iou_th=0.5):
    """(TP, FP, FN) of metrics.py:264-280: the reference loops over classes and images and matches the boxes of one
    (image, class) pair at a time; here every pair is one block of the same kernel -- the boxes are sorted by the key
    image * n_classes + class on the device, which makes each pair a contiguous range."""
    C = int(params.n_classes)
    if C <= 0:
        raise ValueError('detect_and_recog_acc needs a classifying head (n_classes > 0)')
    sets = []
    for arr in (y, y_hat):
        n, idx, xy, cls = utils.decode_boxes_device(arr, params, None, conf_th)
        if n:
            key, order = torch.sort(idx.long() * C + cls.long(), stable=True)
            idx, xy = key.to(torch.int32).contiguous(), xy[order].contiguous()
        sets.append((n, idx, xy, cls))
    batch, g = int(y.shape[0]), int(y.shape[1])
    nb_max = max((int(y.shape[3]) - C) // 5, (int(y_hat.shape[3]) - C) // 5)
    (n1, gi, gxy, _), (n2, pi, pxy, _) = sets
    out = torch.zeros(4, dtype=torch.int32, device='cuda')
    call('cy_detect_confusion', gi.data_ptr() if n1 else None, gxy.data_ptr() if n1 else None, n1,
         pi.data_ptr() if n2 else None, pxy.data_ptr() if n2 else None, n2, batch * C, float(iou_th), g * g * nb_max,
         out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    tp, fp, fn, bad = [int(v) for v in out.cpu().numpy()]
    if bad:
        raise AssertionError('malformed box (x1 > x2 or y1 > y2) in %d case(s)' % bad)
    return tp, fp, fn


def detect_and_recog_acc(y, y_hat, params, show=False, save=False):
    """metrics.py:264-282: F1 of detection + recognition; the registry's metric of darknet_r and darkcapsule
    (main.py:262-264)."""
    p, r = precision_and_recall(*detect_and_recog_confusion(y, y_hat, params))
    return 2 * p * r / (p + r + 1e-8)


def recog_acc(y, y_hat, params):
    """metrics.py:9-11."""
    y, y_hat = [t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t) for t in (y, y_hat)]
    return np.sum(y == np.argmax(y_hat, axis=1)) / len(y)


def average_precision(p, r):
    """metrics.py:180-190: 11-point interpolated average precision."""
    out = []
    for level in np.linspace(0.0, 1.0, 11):
        args = np.argwhere(r >= level).flatten()
        out.append(max(p[args]) if len(args) else 0.0)
    return np.mean(out)


def detect_AP(y, y_hat, params, show=False, save=False, save_dir=None):
    """metrics.py:193-243 (plots are out of scope): boxes are decoded once per confidence threshold and matched on the
    device for each of the 10 IoU thresholds."""
    iou_ths, conf_ths = np.linspace(0.5, 0.95, 10), np.linspace(0, 1, 100)
    yt = torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y).to(device='cuda', dtype=torch.float32)
    ht = torch.as_tensor(np.asarray(y_hat) if not torch.is_tensor(y_hat) else y_hat).to(device='cuda', dtype=torch.float32)
    prec, rec = np.zeros((10, 100)), np.zeros((10, 100))
    for k, conf_th in enumerate(conf_ths):
        gt = utils.decode_boxes_device(yt, params, None, conf_th)
        pr = utils.decode_boxes_device(ht, params, None, conf_th)
        for i, iou_th in enumerate(iou_ths):
            prec[i, k], rec[i, k] = precision_and_recall(*_confusion_of_boxes(gt, pr, yt, ht, params, iou_th))
    return np.mean(np.array([average_precision(prec[i], rec[i]) for i in range(10)]))
