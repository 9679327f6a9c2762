#!/usr/bin/env python
"""Drop-in for the reference's ``python main.py --model <name> --mode train|overfit|predict`` (main.py:22-373)
on the MI355X-native hot path.

Same flags (including the inverted ``--recon`` store_false, main.py:32), same ``experiments/<model>/params.json``
keys, same registry shape ``name -> (ModelCls, loss_fn, predict_fn, metric)`` (main.py:258-265), same step loop
(main.py:55-80, metrics every ``--eval_every`` epochs) and the same artefacts: checkpoints in
``model_dir + str(train_frac)`` like main.py:188, ``losses_tr.npy`` / ``losses_ev.npy`` / ``metrics_tr.npy`` /
``metrics_ev.npy`` in model_dir.  New, optional:
  --synthetic N     train on N synthetic GTSRB/GTSDB-shaped samples (the reference ships no data)
  --n_epochs E, --batch_size B   override params.json
  --fix_ckpt_dir    save checkpoints into model_dir, where --restore looks for them
  params.json keys ``n_iter`` (routing iterations, default 3), ``sync_bn`` (data parallel only)
  --model darkcapsule2 | darkcapsule3   the reference's unwired variants with their losses (models.py:271-337, 403-463)
Data-parallel: launch with ``python -m torch.distributed.run --nproc-per-node N main.py ...``; every rank takes
its equal shard of each global batch, gradients are averaged with one RCCL all-reduce per step, epoch losses are
averaged over the ranks before the LR scheduler sees them, metrics run on the gathered predictions, rank 0 writes.
tensorboardX and torchsummary are outside the hot path; the registry's metrics (main.py:259-264) run through
capsyolo_amd.metrics (detection metrics on the device), ``--no_metric`` skips them like the reference.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import capsyolo_amd  # noqa: E402,F401
from capsyolo_amd import config, dp, metrics, synth, utils  # noqa: E402
from capsyolo_amd.input_pipeline import DeviceFeeder, quantize_if_exact  # noqa: E402
from capsyolo_amd.loss_fns import (capsule_loss, cnn_loss, dark_loss, darkcapsule2_loss, darkcapsule3_loss,  # noqa: E402
                                   darkcapsule_loss)
from capsyolo_amd.models import CapsuleNet, ConvNet, DarkCapsuleNet, DarkCapsuleNet2, DarkCapsuleNet3, DarkNet  # noqa: E402
from capsyolo_amd.optim import Adam  # noqa: E402
from capsyolo_amd.predict_fns import class_pred, dark_forward  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('--model', default='cnn', help=' | '.join(config.model_names))
parser.add_argument('--mode', default='train', help='train | predict | overfit')
parser.add_argument('--summary', default=True, help='if summarize model', action='store_true')
parser.add_argument('--seed', type=int, default=0, help='random seed')
parser.add_argument('--lr', type=float, default=1e-3, help='learning rate')
parser.add_argument('--dropout', type=float, default=-1, help='dropout rate')
parser.add_argument('--train_frac', type=float, default=1, help='fraction of train data')
parser.add_argument('--restore', default=None, help="last | best")
parser.add_argument('--combine', default=None, help="darknet_r | darknet_d")
parser.add_argument('--recon', help='if use reconstruction loss', action='store_false')
parser.add_argument('--recon_coef', default=5e-4, help='reconstruction coefficient')
parser.add_argument('--eval_every', default=1, type=int, help='evaluate metric every # epochs')
parser.add_argument('--fine_tune', default=-1, type=int, help='number of fixed layer in fine tuning')
parser.add_argument('--no_metric', help='do not compute metric', action='store_true')
parser.add_argument('--model_dir', default=None, help='model dir')
parser.add_argument('--show', default=False, help='save result', action='store_true')
parser.add_argument('--npy', default=False, help='data is npy file', action='store_true')
parser.add_argument('--synthetic', type=int, default=0, help='use N synthetic samples instead of data/')
parser.add_argument('--n_epochs', type=int, default=0, help='override params.json n_epochs')
parser.add_argument('--batch_size', type=int, default=0, help='override params.json batch_size')
parser.add_argument('--graph', action='store_true',
                    help='capture the training step (forward + loss + backward + Adam) once in a HIP graph and replay it per batch: '
                         'for the launch-bound small models (capsule); single process only')
parser.add_argument('--fix_ckpt_dir', action='store_true',
                    help='save checkpoints into model_dir (where --restore reads) instead of model_dir + str(train_frac)')

model_loss_predict = {
    'cnn': (ConvNet, cnn_loss, class_pred, metrics.recog_acc),                    # main.py:259-260
    'capsule': (CapsuleNet, capsule_loss, class_pred, metrics.recog_acc),
    'darknet_d': (DarkNet, dark_loss, dark_forward, metrics.detect_acc),          # main.py:261; on the device here
    'darknet_r': (DarkNet, dark_loss, dark_forward, metrics.detect_and_recog_acc),        # main.py:262
    'darkcapsule': (DarkCapsuleNet, darkcapsule_loss, None, metrics.detect_and_recog_acc),   # main.py:264 (the later key wins)
    # the reference's unwired variants with their own losses (models.py:271-337, 403-463; loss_fns.py:145-184)
    'darkcapsule2': (DarkCapsuleNet2, darkcapsule2_loss, None, None),
    'darkcapsule3': (DarkCapsuleNet3, darkcapsule3_loss, None, None),
}


def _batches(x, y, batch_size):
    n_batch = (len(y) + batch_size - 1) // batch_size
    return n_batch, zip(np.array_split(x, n_batch), np.array_split(y, n_batch))        # main.py:45-47


_warned = set()


def _shard(x_bch, y_bch, params):
    """This rank's contiguous, EQUAL shard of a global batch.  np.array_split makes near-equal batches (600 / 32 ->
    11 x 32 + 8 x 31, main.py:47); equal shards are what makes the mean of the ranks' gradients the global-batch
    gradient (each loss divides by its local batch), so up to world-1 samples of a batch that does not divide are left
    out, with one warning."""
    rank, world = params.rank, params.world
    if world > 1:
        per = len(y_bch) // world
        if len(y_bch) % world and 'rem' not in _warned and rank == 0:
            _warned.add('rem')
            print('data parallel: batches of %d samples on %d ranks: %d sample(s) per such batch are not used'
                  % (len(y_bch), world, len(y_bch) % world), flush=True)
        x_bch, y_bch = x_bch[rank * per:(rank + 1) * per], y_bch[rank * per:(rank + 1) * per]
    return x_bch, y_bch


def _feed(it, params):
    """main.py:57-59 (H2D + float + NHWC->NCHW) through the buffered device-side pipeline; batches smaller than
    the number of ranks are dropped on every rank alike."""
    shards = [_shard(x_np, y_np, params) for x_np, y_np in it]
    shards = [(x_np, y_np) for x_np, y_np in shards if len(y_np) > 0]
    if params.device == 'cpu':          # only the plain-torch `cnn` baseline gets here (main() refuses the others)
        return _host_batches(shards)
    return DeviceFeeder(shards, params.device)


def _host_batches(shards):
    """main.py:57-59 as written, for the plain-torch baseline model on a machine without a GPU."""
    for x_np, y_np in shards:
        if x_np.dtype == np.uint8:      # quantize_if_exact() stored the centred set as bytes
            x_np = (x_np.astype(np.float32) - 128.0) * np.float32(0.0078125)
        yield torch.from_numpy(x_np).float().permute(0, 3, 1, 2).contiguous(), torch.from_numpy(y_np)


def _forward(model, loss_fn, x_bch, y_bch, params):
    if params.model == 'capsule' and params.recon:                                       # main.py:61-66
        y_hat, recon = model(x_bch, y_bch, True)
        return y_hat, loss_fn(y_hat, y_bch, params, x_bch, recon)
    y_hat = model(x_bch)
    return y_hat, loss_fn(y_hat, y_bch, params)


def _gather(t, params):
    """Concatenate a per-rank tensor over the ranks (equal shards) -- metrics are computed on the global predictions."""
    if params.world == 1:
        return t
    import torch.distributed as dist
    parts = [torch.empty_like(t) for _ in range(params.world)]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts)


def _metric(metric, y_true, y_hat, params):
    """main.py:82-91 / 129-137: the metric on at most config.max_metric_samples samples drawn like the reference does
    (np.random.choice with replacement, only when there are more samples than that)."""
    if metric is None or not y_hat:
        return -1
    y_true, y_hat = _gather(torch.cat(y_true), params), _gather(torch.cat(y_hat), params)
    n, cap = y_true.shape[0], getattr(config, 'max_metric_samples', 1000)
    if n > cap:
        idx = torch.from_numpy(np.random.choice(n, cap).astype(np.int64)).to(y_true.device)
        y_true, y_hat = y_true[idx], y_hat[idx]
    try:
        return float(metric(y_true, y_hat, params))
    except ValueError as e:     # e.g. darkcapsule's [B,g,g,5] output has no class scores for its registry metric
        if params.rank == 0:    # (the reference stops here, metrics.py:267-268; pass --no_metric to skip the attempt)
            print('metric not computed: %s' % e)
        return -1


def _mean_over_ranks(value, params):
    """Epoch losses are per-rank means over equal shards: their mean over the ranks is the global epoch loss.  Every rank
    must see the SAME number (ReduceLROnPlateau steps on it: a rank-local loss would let the replicas' learning rates
    drift apart)."""
    if params.world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=params.device if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t)
    return float(t.item()) / params.world


def train(x, y, model, optimizer, loss_fn, metric, params, bucket, if_eval=True):
    """main.py:42-95.  The step loop keeps the predictions on the device (no per-step D2H copy: the reference's
    y_hat.append(... .cpu().numpy()), main.py:68, exists only to feed the metric below) and reads the loss value once
    per step like the reference's progress bar does (main.py:74-77)."""
    model.train()
    x, y = utils.shuffle(x, y)
    n_batch, it = _batches(x, y, params.batch_size)
    avg_loss, avg_iou, y_hat, y_true = 0.0, 0.0, [], []
    want_metric = if_eval and metric is not None
    graphed = getattr(params, 'graph', False) and params.world == 1 and params.device != 'cpu'
    for x_bch, y_bch in _feed(it, params):
        if graphed:
            # the whole step as one graph replay (capsyolo_amd/graph_step.py); captured on the first batch of this shape
            key = (id(model), id(optimizer), tuple(x_bch.shape), tuple(y_bch.shape))
            steps = params.__dict__.setdefault('_graph_steps', {})
            if key not in steps:                          # (one capture per batch shape: the ragged last batch gets its own)
                from capsyolo_amd.graph_step import GraphedStep
                steps[key] = GraphedStep(model, lambda m, xb, yb: _forward(m, loss_fn, xb, yb, params), optimizer, (x_bch, y_bch))
            y_hat_bch, loss = steps[key](x_bch, y_bch)
            if want_metric:
                y_hat.append(y_hat_bch.detach().clone())  # the graph's output tensor is overwritten by the next replay
                y_true.append(y_bch.detach().clone())
            avg_loss += loss.item() / n_batch
            if params.model == 'darknet_d':
                avg_iou += params.avg_iou.item() / n_batch
            continue
        y_hat_bch, loss = _forward(model, loss_fn, x_bch, y_bch, params)
        if want_metric:
            y_hat.append(y_hat_bch.detach())
            y_true.append(y_bch.detach().clone())        # the feeder's slot buffers are overwritten by later batches
        optimizer.zero_grad()
        loss.backward()
        bucket.allreduce_mean()
        optimizer.step()
        avg_loss += loss.item() / n_batch
        if params.model == 'darknet_d':
            avg_iou += params.avg_iou.item() / n_batch
    score = _metric(metric, y_true, y_hat, params) if want_metric else -1
    if params.model == 'darknet_d' and params.rank == 0:
        print('train avg iou: {:05.3f}'.format(avg_iou), flush=True)
    return avg_loss, score


def evaluate(x, y, model, loss_fn, metric, params, if_eval=True):
    """main.py:98-143."""
    model.eval()
    n_batch, it = _batches(x, y, params.batch_size)
    avg_loss, y_hat, y_true = 0.0, [], []
    want_metric = if_eval and metric is not None
    with torch.no_grad():
        for x_bch, y_bch in _feed(it, params):
            y_hat_bch, loss = _forward(model, loss_fn, x_bch, y_bch, params)
            avg_loss += loss.item() / n_batch
            if want_metric:
                y_hat.append(y_hat_bch.detach().clone())
                y_true.append(y_bch.detach().clone())
    return avg_loss, (_metric(metric, y_true, y_hat, params) if want_metric else -1)


def checkpoint_dir(model_dir, params):
    """The reference saves checkpoints to model_dir + str(train_frac) (e.g. experiments/darkcapsule1, main.py:188) but
    restores from model_dir (main.py:149).  Same here by default; --fix_ckpt_dir saves where --restore reads."""
    return model_dir if getattr(params, 'fix_ckpt_dir', False) else model_dir + str(params.train_frac)


def train_and_evaluate(model, optimizer, loss_fn, metric, params, data, model_dir, restore_file=None):
    """main.py:146-217."""
    if restore_file is not None:
        utils.load_checkpoint(os.path.join(model_dir, restore_file + '.pth.tar'), model, params, optimizer)
    x_tr, y_tr, x_ev, y_ev = data
    to_frac = int(y_tr.shape[0] * params.train_frac)
    x_tr, y_tr = x_tr[:to_frac], y_tr[:to_frac]
    # centred images that are exactly (uint8 - 128) / 128 are kept as bytes (once per data set): 4-8x less to shuffle
    # and to move over PCIe, converted back on the device (input_pipeline.py)
    x_tr, x_ev = [q if q is not None else x for q, x in ((quantize_if_exact(x), x) for x in (x_tr, x_ev))]
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'min', factor=params.lr_decay)
    bucket = dp.GradBucket(model)
    losses_tr, losses_ev, metrics_tr, metrics_ev = [], [], [], []
    best_metric_ev, best_loss_ev = float('-inf'), float('inf')
    for epoch in range(params.n_epochs):
        if_eval = (epoch + 1) % params.eval_every == 0                                   # main.py:168
        loss_tr, metric_tr = train(x_tr, y_tr, model, optimizer, loss_fn, metric, params, bucket, if_eval)
        loss_ev, metric_ev = evaluate(x_ev, y_ev, model, loss_fn, metric, params, if_eval)
        loss_tr, loss_ev = _mean_over_ranks(loss_tr, params), _mean_over_ranks(loss_ev, params)
        scheduler.step(loss_tr)                                                          # plateau on the TRAIN loss (main.py:174)
        is_best = metric_ev > best_metric_ev
        best_metric_ev = max(best_metric_ev, metric_ev)
        best_loss_ev = min(best_loss_ev, loss_ev)
        if params.rank == 0:
            utils.save_checkpoint({'epoch': epoch + 1, 'state_dict': model.state_dict(),
                                   'optim_dict': optimizer.state_dict()}, is_best=is_best,
                                  checkpoint=checkpoint_dir(model_dir, params))
            if if_eval:
                print('epoch {} | train loss: {:05.3f} | eval loss: {:05.3f} | best eval loss: {:05.3f} | '
                      'train metric: {:05.3f} | eval metric: {:05.3f} | best eval metric {:05.3f}'.format(
                          epoch + 1, loss_tr, loss_ev, best_loss_ev, metric_tr, metric_ev, best_metric_ev), flush=True)
                metrics_tr.append(metric_tr)
                metrics_ev.append(metric_ev)
                np.save(os.path.join(model_dir, 'metrics_tr'), metrics_tr)
                np.save(os.path.join(model_dir, 'metrics_ev'), metrics_ev)
        losses_tr.append(loss_tr)
        losses_ev.append(loss_ev)
        if params.rank == 0:
            np.save(os.path.join(model_dir, 'losses_tr'), losses_tr)
            np.save(os.path.join(model_dir, 'losses_ev'), losses_ev)
    return losses_tr, losses_ev


def load_params(model_dir, args):
    """main.py:227-241 (params.device is decided here, the JSON value is ignored)."""
    params = utils.Params(os.path.join(model_dir, 'params.json'))
    params.device = 'cuda' if torch.cuda.is_available() else 'cpu'
    params.seed = args.seed
    if args.dropout >= 0:
        params.dropout = args.dropout
    if not hasattr(params, 'dropout'):
        params.dropout = 0.0
    params.model = args.model
    params.recon = args.recon
    params.recon_coef = float(args.recon_coef)
    params.eval_every = args.eval_every
    params.train_frac = args.train_frac
    if args.n_epochs:
        params.n_epochs = args.n_epochs
    if args.batch_size:
        params.batch_size = args.batch_size
    params.fix_ckpt_dir = args.fix_ckpt_dir
    params.graph = args.graph
    return params


def synthetic_data(args, params):
    n = args.synthetic
    n_ev = max(params.batch_size, n // 4)
    if args.model in ('cnn', 'capsule'):
        mk = lambda k, first: (synth.images(k, 32, first=first), synth.gtsrb_labels(k, params.n_classes, first=first))
    else:
        if params.darknet_input != 32 * params.n_grid and args.model in ('darkcapsule', 'darkcapsule3'):
            raise SystemExit('darkcapsule needs darknet_input = 32 * n_grid (models.py:393); got %d and %d'
                             % (params.darknet_input, params.n_grid))
        mk = lambda k, first: (synth.images(k, params.darknet_input, first=first),
                               synth.gtsdb_labels(k, params.n_grid, params.n_classes, first=first))
    x_tr, y_tr = mk(n, 0)
    x_ev, y_ev = mk(n_ev, n)
    return x_tr, y_tr, x_ev, y_ev


def main(argv=None):
    args = parser.parse_args(argv)
    if args.model not in config.model_names:
        print("Did not recognize model, choose from: ", *config.model_names)
        sys.exit()
    data_dir, model_dir = config.data_dir[args.model], config.model_dir[args.model]
    if args.model_dir is not None:
        model_dir = args.model_dir
    params = load_params(model_dir, args)
    params.rank, params.world, local_rank = dp.init_from_env()
    if params.world > 1 and params.batch_size % params.world:
        raise SystemExit('data parallel: batch_size %d is not divisible by the %d ranks' % (params.batch_size, params.world))
    if params.world > 1 and getattr(params, 'sync_bn', False):     # optional params.json key (new; data parallel only)
        from capsyolo_amd import ops as _ops
        _ops.SYNC_BN = True
    if params.device == 'cuda':
        import torch.distributed as dist
        torch.cuda.set_device(dp.local_device_index(local_rank, dist.get_backend() if dist.is_initialized() else 'nccl'))
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if params.device == 'cuda':
        torch.cuda.manual_seed(args.seed)
    elif args.model != 'cnn':
        raise SystemExit('model %s runs on hand-written gfx950 kernels only; no GPU is visible' % args.model)

    model_cls, loss_fn, predict_fn, metric = model_loss_predict[args.model]
    if args.no_metric:                                                                 # main.py:87,133
        metric = None
    model = model_cls(params).to(device=params.device)
    dp.broadcast_parameters(model)
    if args.fine_tune > 0:
        model.load_weights('./darknet19_weights.npz', 18)                              # main.py:273-278
        for name, param in model.named_parameters():
            if int(name.split('.')[1].split('_')[1]) <= params.fine_tune:
                param.requires_grad = False
    trainable = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.Adam(trainable, lr=args.lr) if args.model == 'cnn' else Adam(trainable, lr=args.lr)

    if args.mode in ('train', 'overfit'):
        if args.synthetic:
            data = synthetic_data(args, params)
        else:
            if args.mode == 'overfit':
                raise SystemExit('--mode overfit needs the real dataset under %s (or use --synthetic 3)' % data_dir)
            data = utils.load_data(data_dir, False, npy=args.npy)
        return train_and_evaluate(model, optimizer, loss_fn, metric, params, data, model_dir, restore_file=args.restore)
    if args.mode == 'predict':
        raise SystemExit('predict mode needs the raw GTSDB images and cv2 post-processing (out of scope, SURVEY section 2); '
                         'the eval-mode forward is capsyolo_amd.predict_fns.class_pred / dark_forward')


if __name__ == '__main__':
    main()
