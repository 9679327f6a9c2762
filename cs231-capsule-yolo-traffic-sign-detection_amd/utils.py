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
