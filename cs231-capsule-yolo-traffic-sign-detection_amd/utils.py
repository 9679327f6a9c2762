"""Training-side utilities with the reference's names (utils.py:14-60, 122-148)."""
import json
import os
import shutil

import numpy as np
import torch


class Params():
    """utils.py:14-31."""

    def __init__(self, json_path):
        self.update(json_path)

    def save(self, json_path):
        with open(json_path, 'w') as f:
            json.dump({k: v for k, v in self.__dict__.items() if isinstance(v, (int, float, str, bool, list))}, f,
                      indent=4)

    def update(self, json_path):
        with open(json_path) as f:
            self.__dict__.update(json.load(f))

    @property
    def dict(self):
        return self.__dict__


def save_checkpoint(state, is_best, checkpoint):
    """utils.py:40-49: <dir>/last.pth.tar (+ best.pth.tar copy)."""
    filepath = os.path.join(checkpoint, 'last.pth.tar')
    if not os.path.exists(checkpoint):
        os.makedirs(checkpoint)
    torch.save(state, filepath)
    if is_best:
        shutil.copyfile(filepath, os.path.join(checkpoint, 'best.pth.tar'))


def load_checkpoint(checkpoint, model, params=None, optimizer=None):
    """utils.py:52-60 (the reference raises a str on a missing file; a real exception here)."""
    if not os.path.exists(checkpoint):
        raise FileNotFoundError("File doesn't exist {}".format(checkpoint))
    ckpt = torch.load(checkpoint, map_location='cpu', weights_only=False)
    model.load_state_dict(ckpt['state_dict'])
    if optimizer and 'optim_dict' in ckpt:
        optimizer.load_state_dict(ckpt['optim_dict'])
    return ckpt


def load_data(data_dir, is_small=False, npy=False):
    """utils.py:91-113: pickle (X, Y) tuples or *_X.npy / *_Y.npy pairs."""
    import pickle
    from . import config
    tr = data_dir + (config.tr_sm_d if is_small else config.tr_d)
    ev = data_dir + (config.ev_sm_d if is_small else config.ev_d)
    if not npy:
        with open(tr, 'rb') as f:
            x_tr, y_tr = pickle.load(f)
        with open(ev, 'rb') as f:
            x_ev, y_ev = pickle.load(f)
        return x_tr, y_tr, x_ev, y_ev
    tr, ev = tr.split('.')[0], ev.split('.')[0]
    return np.load(tr + '_X.npy'), np.load(tr + '_Y.npy'), np.load(ev + '_X.npy'), np.load(ev + '_Y.npy')


def center_rgb(x):
    """utils.py:122-123."""
    return (x - 128) / 128


def shuffle(x, y):
    """utils.py:146-148."""
    i = np.random.permutation(len(y))
    return x[i], y[i]


def decode_boxes_device(y, params, image_hw=None, conf_th=0.5):
    """`cy_yolo_decode_boxes` with everything left on the device: (n, image_idx int32[n], xy float64[n,4], cls int32[n] | None)."""
    from ._lib import call
    yt = torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y).to(device='cuda', dtype=torch.float32).contiguous()
    batch, g, _, D = yt.shape
    C = int(params.n_classes)
    nb = int((D - C) / 5)
    if nb < 1:          # e.g. DarkCapsuleNet's [B,g,g,5] output with n_classes = 43: the reference's reshape fails here too
        raise ValueError('y has %d values per cell: no box left after %d class scores (utils.py:291-293)' % (D, C))
    cap = batch * g * g * nb
    dev = yt.device
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    idx = torch.empty(cap, dtype=torch.int32, device=dev)
    xy = torch.empty((cap, 4), dtype=torch.float64, device=dev)
    cls = torch.empty(cap, dtype=torch.int32, device=dev) if C else None
    hw = None
    if image_hw is not None:
        hw = torch.as_tensor(np.ascontiguousarray(np.asarray(image_hw).reshape(batch, 2)), dtype=torch.int64).to(dev)
    side = float(params.darknet_input)
    call('cy_yolo_decode_boxes', yt.data_ptr(), hw.data_ptr() if hw is not None else None, side, side, batch, g, nb, C,
         float(conf_th), count.data_ptr(), idx.data_ptr(), xy.data_ptr(), cls.data_ptr() if C else None, cap,
         torch.cuda.current_stream().cuda_stream)
    n = int(count.item())
    return n, idx[:n], xy[:n], (cls[:n] if C else None)


def y_to_boxes_vec(y, params, image_hw=None, conf_th=0.5):
    """utils.py:288-334 on the device (`cy_yolo_decode_boxes`): y is the network output / ground truth as a numpy
    array or a tensor, image_hw an optional (batch, 2) array of (height, width).  Returns (image_indices, xy, classes)
    as numpy arrays like the reference (classes is None when params.n_classes == 0)."""
    n, idx, xy, cls = decode_boxes_device(y, params, image_hw, conf_th)
    return (idx.cpu().numpy().astype(np.int64), xy.cpu().numpy(),
            cls.cpu().numpy().astype(np.int64) if cls is not None else None)
