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
