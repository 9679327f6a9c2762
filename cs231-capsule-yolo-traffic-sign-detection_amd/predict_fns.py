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


def _eval_forward_batched(x, model, params, batch_size):
    """The reference pushes the WHOLE set through the model in one call (predict_fns.py:40-43, 65-69: x [N,3,448,448]).
    In eval mode every sample is independent (BatchNorm uses the running statistics, folded into the conv weights here:
    ops.fold_eval_bn), so the set goes through in chunks of `batch_size` with identical results and bounded memory."""
    n = int(x.shape[0])
    bs = n if not batch_size else int(batch_size)
    outs = []
    model.eval()
    with torch.no_grad():
        for lo in range(0, n, max(bs, 1)):
            xt = torch.from_numpy(np.ascontiguousarray(x[lo:lo + bs])).to(device=params.device, dtype=torch.float32)
            outs.append(model(xt.permute(0, 3, 1, 2).contiguous()).data.cpu().numpy())
    return np.concatenate(outs, axis=0) if len(outs) != 1 else outs[0]


def class_pred(x, model, model_dir, params, restore_file, batch_size=1024):
    """predict_fns.py:60-73: x NHWC numpy -> (scores, argmax classes).  batch_size (new, optional): chunk of the set per forward
    call; None = the whole set at once like the reference."""
    _restore(model, model_dir, params, restore_file)
    y_hat = _eval_forward_batched(x, model, params, batch_size)
    return y_hat, np.argmax(y_hat, axis=1)


def dark_forward(x, model, model_dir, params, restore_file, batch_size=32):
    """predict_fns.py:38-43: eval forward of the detector on already-resized NHWC images -> y_hat numpy (batched like class_pred)."""
    _restore(model, model_dir, params, restore_file)
    return _eval_forward_batched(x, model, params, batch_size)
