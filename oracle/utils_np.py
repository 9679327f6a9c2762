"""CPU restatement (numpy) of the box decoding of the reference's utils.py -- TEST INFRASTRUCTURE ONLY.

Imported by tests/ (and nothing in the product).  Follows utils.py:233-252 (denorm_boxes_cwh_vec), 254-269
(cwh_to_xy_vec) and 288-334 (y_to_boxes_vec); pinned by tests/golden/boxes.npz, which holds outputs of the reference
itself (tests/golden/make_golden.py boxes).
"""
import numpy as np


def denorm_boxes_cwh_vec(image_hw, n_grid, norm_cwh, grid_indices):
    """utils.py:233-252: cell-relative (xc, yc) and image-relative (w, h) -> pixels."""
    image_hw = np.array(image_hw).reshape(-1, 2)
    image_wh = image_hw[:, [1, 0]]
    grids_wh = 1. * image_wh / n_grid
    cwh = norm_cwh * np.concatenate((grids_wh, image_wh), axis=1)
    cwh[:, 0:2] += grid_indices[:, [1, 0]] * grids_wh
    return cwh


def cwh_to_xy_vec(cwh):
    """utils.py:254-269."""
    xy = np.zeros_like(cwh)
    xy[:, 0] = cwh[:, 0] - cwh[:, 2] / 2
    xy[:, 1] = cwh[:, 1] - cwh[:, 3] / 2
    xy[:, 2] = cwh[:, 0] + cwh[:, 2] / 2
    xy[:, 3] = cwh[:, 1] + cwh[:, 3] / 2
    return xy


def y_to_boxes_vec(y, n_classes, darknet_input, image_hw=None, conf_th=0.5):
    """utils.py:288-334: boxes above the confidence threshold in np.argwhere order -> (image index, xyxy, class)."""
    batch, g, _, D = y.shape
    nb = int((D - n_classes) / 5)
    boxes = y[:, :, :, 0:5 * nb].reshape(batch, g, g, nb, 5)
    mask = boxes[:, :, :, :, 0] > conf_th
    indices = np.argwhere(mask)
    cwh = boxes[mask, 1:5]
    image_indices = indices[:, 0]
    hw = (darknet_input, darknet_input) if image_hw is None else image_hw[image_indices]
    xy = cwh_to_xy_vec(denorm_boxes_cwh_vec(hw, g, cwh, indices[:, 1:3]))
    classes = None
    if n_classes != 0:
        classes = np.argmax(y[:, :, :, 5 * nb:][indices[:, 0], indices[:, 1], indices[:, 2]], axis=1)
    return image_indices, xy, classes


def calc_iou_individual(gt_box, pred_box):
    """metrics.py:99-133 (a malformed box raises, like the reference)."""
    x1_t, y1_t, x2_t, y2_t = gt_box
    x1_p, y1_p, x2_p, y2_p = pred_box
    if (x1_p > x2_p) or (y1_p > y2_p) or (x1_t > x2_t) or (y1_t > y2_t):
        raise AssertionError('malformed box')
    if x2_t < x1_p or x2_p < x1_t or y2_t < y1_p or y2_p < y1_t:
        return 0.0
    inter = (min(x2_t, x2_p) - max(x1_t, x1_p)) * (min(y2_t, y2_p) - max(y1_t, y1_p))
    return inter / ((x2_t - x1_t) * (y2_t - y1_t) + (x2_p - x1_p) * (y2_p - y1_p) - inter)


def single_img_confusion(y_, y_hat_, iou_th):
    """metrics.py:136-147: a ground-truth / predicted box counts as hit when ANY partner overlaps it by more than iou_th."""
    n1, n2 = y_.shape[0], y_hat_.shape[0]
    gt_hit, pred_hit = set(), set()
    for i in range(n1):
        for j in range(n2):
            if calc_iou_individual(y_[i], y_hat_[j]) > iou_th:
                gt_hit.add(i)
                pred_hit.add(j)
    return len(gt_hit), n2 - len(pred_hit), n1 - len(gt_hit)


def detect_confusion(y, y_hat, darknet_input, conf_th=0.5, iou_th=0.5):
    """TP, FP, FN of metrics.py:245-258 (n_classes = 0: detector head)."""
    yi, yb, _ = y_to_boxes_vec(y, 0, darknet_input, conf_th=conf_th)
    hi, hb, _ = y_to_boxes_vec(y_hat, 0, darknet_input, conf_th=conf_th)
    tot = np.zeros(3, dtype=np.int64)
    for j in range(y.shape[0]):
        tot += np.array(single_img_confusion(yb[yi == j], hb[hi == j], iou_th))
    return tot


def detect_acc(y, y_hat, darknet_input):
    """metrics.py:245-262: F1 of the detector."""
    tp, fp, fn = [int(v) for v in detect_confusion(y, y_hat, darknet_input)]
    p = tp / (tp + fp) if tp + fp else 0.0
    r = tp / (tp + fn) if tp + fn else 0.0
    return 2 * p * r / (p + r + 1e-8)


def average_precision(p, r):
    """metrics.py:180-190: 11-point interpolated average precision."""
    out = []
    for level in np.linspace(0.0, 1.0, 11):
        args = np.argwhere(r >= level).flatten()
        out.append(max(p[args]) if len(args) else 0.0)
    return np.mean(out)


def detect_AP(y, y_hat, darknet_input):
    """metrics.py:193-243 without the plotting: mean over 10 IoU thresholds of the 11-point AP over 100 confidence thresholds."""
    aps = []
    for iou_th in np.linspace(0.5, 0.95, 10):
        ps, rs = [], []
        for conf_th in np.linspace(0, 1, 100):
            tp, fp, fn = [int(v) for v in detect_confusion(y, y_hat, darknet_input, conf_th, iou_th)]
            ps.append(tp / (tp + fp) if tp + fp else 0.0)
            rs.append(tp / (tp + fn) if tp + fn else 0.0)
        aps.append(average_precision(np.array(ps), np.array(rs)))
    return np.mean(np.array(aps))


def detect_and_recog_confusion(y, y_hat, n_classes, darknet_input, conf_th=0.5, iou_th=0.5):
    """TP, FP, FN of metrics.py:264-280: boxes are matched per image AND per class."""
    yi, yb, yc = y_to_boxes_vec(y, n_classes, darknet_input, conf_th=conf_th)
    hi, hb, hc = y_to_boxes_vec(y_hat, n_classes, darknet_input, conf_th=conf_th)
    tot = np.zeros(3, dtype=np.int64)
    for c in range(n_classes):
        for j in range(y.shape[0]):
            tot += np.array(single_img_confusion(yb[(yi == j) * (yc == c)], hb[(hi == j) * (hc == c)], iou_th))
    return tot


def detect_and_recog_acc(y, y_hat, n_classes, darknet_input):
    """metrics.py:264-282: F1 of detection + recognition (the registry's metric of darknet_r and darkcapsule, main.py:262-264)."""
    tp, fp, fn = [int(v) for v in detect_and_recog_confusion(y, y_hat, n_classes, darknet_input)]
    p = tp / (tp + fp) if tp + fp else 0.0
    r = tp / (tp + fn) if tp + fn else 0.0
    return 2 * p * r / (p + r + 1e-8)

