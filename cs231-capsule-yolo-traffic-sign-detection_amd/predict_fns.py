"""Eval-mode forward wrappers with the reference's names (predict_fns.py:60-73 and the forward part of
10-58).  Box decoding / drawing / cropping (cv2 post-processing, predict_fns.py:44-58) is out of scope."""
import os

import numpy as np
import torch

from . import utils


def _restore(model, model_dir, params, restore_file):
    path = os.path.join(model_dir, restore_file + '.pth.tar')
    print("Restoring parameters from {}".format(path))
    utils.load_checkpoint(path, model, params)


def class_pred(x, model, model_dir, params, restore_file):
    """predict_fns.py:60-73: x NHWC numpy -> (scores, argmax classes)."""
    _restore(model, model_dir, params, restore_file)
    model.eval()
    with torch.no_grad():
        xt = torch.from_numpy(x).float().permute(0, 3, 1, 2).contiguous().to(device=params.device)
        y_hat = model(xt).data.cpu().numpy()
    return y_hat, np.argmax(y_hat, axis=1)


def dark_forward(x, model, model_dir, params, restore_file):
    """predict_fns.py:38-43: eval forward of the detector on already-resized NHWC images -> y_hat numpy."""
    _restore(model, model_dir, params, restore_file)
    model.eval()
    with torch.no_grad():
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous().to(device=params.device, dtype=torch.float32)
        return model(xt).data.cpu().numpy()
