"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the product path has no
fallback and never touches the oracle, host-side geometry / batching / checkpoint / parameter logic."""
import os
import re
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import REPO, make_params

import capsyolo_amd  # noqa: F401
from capsyolo_amd import _lib, config, models, ops, synth, utils
from oracle import models as OM

PKG = os.path.join(REPO, 'cs231-capsule-yolo-traffic-sign-detection_amd')


def test_cabi_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, 'include', 'capsyolo_hip.h')).read()
    declared = set(re.findall(r'\b((?:cy|capsyolo)_[a-z0-9_]+)\s*\(', header))
    declared = {d for d in declared if not d.endswith('_t')}
    assert len(declared) >= 35
    lib = _lib.load()                                  # loads without a GPU; no compute call is made here
    for name in sorted(declared):
        assert hasattr(lib, name), 'libcapsyolo_hip.so does not export %s' % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.capsyolo_abi_version() == _lib.ABI_VERSION == 5


def test_ctypes_structs_match_header_field_order():
    header = open(os.path.join(REPO, 'include', 'capsyolo_hip.h')).read()
    for cname, struct in (('cy_conv_gemm_t', _lib.ConvGemm), ('cy_conv_wgrad_t', _lib.ConvWgrad),
                          ('cy_routing_fwd_t', _lib.RoutingFwd), ('cy_routing_bwd_t', _lib.RoutingBwd)):
        body = re.search(r'typedef struct \{([^{}]*)\}\s*%s;' % cname, header, re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(',')
            first = names[0].split()[-1].lstrip('*')
            fields.append(first)
            fields.extend(n.strip().lstrip('*') for n in names[1:])
        assert fields == [f[0] for f in struct._fields_], cname


def test_split_reduction_plans_without_a_gpu():
    """The workspace queries of the split reductions are host arithmetic (no launch): cy_conv_gemm_ws_floats plans with 256 CUs when no
    device answers -- CapsuleNet's primary-capsule convolution at batch 32 (models.py:60-62 fused to 256 -> 128, 8x8 / stride 2 on
    24 x 24: M = 2592 pixels, 42 tiles of 128 x 64, K = 20736 = 648 K tiles) gets 12 shares; a launch with the statistics epilogue
    and a launch that fills the chip get none; cy_wino_split_ws_floats never splits a launch with an epilogue."""
    import ctypes as C
    a = _lib.ConvGemm(X=1, Wp=1, Y=1, bias=None, stats=None, xs_b=24 * 24 * 256, xs_y=24 * 256, xs_x=256, xs_c=1, B=32, Hi=24, Wi=24, Cin=256,
                      Ho=9, Wo=9, N=128, TH=8, TW=8, in_stride=2, dy0=0, dx0=0, dstep=1, Hy=9, Wy=9, out_stride=1, out_oy=0, out_ox=0, act=0)
    assert _lib.query('cy_conv_gemm_ws_floats', C.byref(a)) == 12 * 2592 * 128
    a.stats = 1                                           # BatchNorm statistics in the epilogue: the finished sums are needed there
    assert _lib.query('cy_conv_gemm_ws_floats', C.byref(a)) == 0
    a.stats, a.B = None, 3200                             # 4200 tiles: nothing to split
    assert _lib.query('cy_conv_gemm_ws_floats', C.byref(a)) == 0
    assert _lib.query('cy_wino_split_ws_floats', 16, 13, 13, 1024, 512, 0) == 0
    assert _lib.query('cy_wino_split_ws_floats', 16, 13, 13, 1024, 512, 1) in (0, 2 * 16 * 13 * 13 * 512)   # (0 without a device to count CUs on)


def test_product_never_imports_the_oracle_and_has_no_fallback():
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f
    # a CPU tensor is rejected loudly, not silently computed some other way
    with pytest.raises(_lib.HipExtensionError):
        ops.conv_forward(torch.zeros(1, 4, 4, 32), torch.zeros(32, 32, 3, 3), None, 3, 1, 1)
    with pytest.raises(_lib.HipExtensionError):
        ops.routing(torch.zeros(2, 8, 8), torch.zeros(1, 8, 3, 8, 16))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', os.path.join(PKG, 'does_not_exist.so'))
    with pytest.raises(_lib.HipExtensionError, match='no CPU fallback'):
        _lib.load()


@pytest.mark.parametrize('k,stride,pad,Hi', [(3, 1, 1, 7), (4, 2, 1, 8), (8, 2, 0, 12), (1, 1, 0, 5), (3, 2, 1, 9),
                                             (9, 1, 0, 11), (5, 3, 2, 10)])
def test_dgrad_class_decomposition(k, stride, pad, Hi):
    """numpy emulation of the cy_conv_gemm contract fed with ops.dgrad_classes must equal conv_transpose."""
    rng = np.random.default_rng(k * 10 + stride)
    Cin, Cout = 2, 3
    Ho = (Hi + 2 * pad - k) // stride + 1
    W = rng.standard_normal((Cout, Cin, k, k))
    dz = rng.standard_normal((1, Cout, Ho, Ho))
    ref = F.conv_transpose2d(torch.from_numpy(dz), torch.from_numpy(W), stride=stride, padding=pad,
                             output_padding=Hi - ((Ho - 1) * stride - 2 * pad + k)).numpy()
    dx = np.full((1, Cin, Hi, Hi), np.nan)
    for c in ops.dgrad_classes(Hi, Hi, k, stride, pad):
        for oy in range(c['Ho']):
            for ox in range(c['Wo']):
                acc = np.zeros(Cin)
                for a in range(c['TH']):
                    for b in range(c['TW']):
                        iy, ix = oy + c['dy0'] + a * c['dstep'], ox + c['dx0'] + b * c['dstep']
                        if 0 <= iy < Ho and 0 <= ix < Ho:
                            acc += dz[0, :, iy, ix] @ W[:, :, c['kh0'] + a * c['kstep'], c['kw0'] + b * c['kstep']]
                dx[0, :, oy * c['out_stride'] + c['out_oy'], ox * c['out_stride'] + c['out_ox']] = acc
    assert not np.isnan(dx).any()                      # every input pixel belongs to exactly one class
    np.testing.assert_allclose(dx, ref, rtol=1e-12, atol=1e-12)


def test_state_dict_keys_and_shapes_match_the_oracle():
    p = make_params(n_grid=2)
    pairs = [(models.DarkCapsuleNet(p), OM.DarkCapsuleNet(p)), (models.CapsuleNet(p), OM.CapsuleNet(p)),
             (models.DarkNet(make_params(n_boxes=2, n_classes=0)), OM.DarkNet(make_params(n_boxes=2, n_classes=0))),
             (models.DarkCapsuleNet3(make_params(n_classes=3)), OM.DarkCapsuleNet3(make_params(n_classes=3))),
             (models.ConvNet(p), OM.ConvNet(p))]
    for ours, ref in pairs:
        a, b = ours.state_dict(), ref.state_dict()
        assert list(a.keys()) == list(b.keys()), type(ours).__name__
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        ours.load_state_dict(b)                       # reference-format checkpoints load unchanged


def test_params_checkpoint_and_batching(tmp_path):
    pj = tmp_path / 'params.json'
    pj.write_text('{"batch_size": 32, "n_classes": 43, "lr_decay": 0.1}')
    params = utils.Params(str(pj))
    assert params.batch_size == 32 and params.dict['n_classes'] == 43
    net = models.DarkCapsuleNet(make_params())
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    utils.save_checkpoint({'epoch': 1, 'state_dict': net.state_dict(), 'optim_dict': opt.state_dict()}, True,
                          str(tmp_path / 'ck'))
    assert (tmp_path / 'ck' / 'last.pth.tar').exists() and (tmp_path / 'ck' / 'best.pth.tar').exists()
    net2 = models.DarkCapsuleNet(make_params())
    ck = utils.load_checkpoint(str(tmp_path / 'ck' / 'last.pth.tar'), net2, params)
    assert ck['epoch'] == 1
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)
    with pytest.raises(FileNotFoundError):
        utils.load_checkpoint(str(tmp_path / 'nope.pth.tar'), net2, params)
    # main.py:45-47 batching: near-equal splits
    x, y = np.zeros((600, 2)), np.arange(600)
    n_batch = (600 + 31) // 32
    sizes = [len(b) for b in np.array_split(y, n_batch)]
    assert n_batch == 19 and sizes == [32] * 11 + [31] * 8
    np.random.seed(0)
    xs, ys = utils.shuffle(x, y)
    assert sorted(ys.tolist()) == list(range(600))


def test_synthetic_data_shapes_and_world_size_independence():
    x = synth.images(4, 64)
    assert x.shape == (4, 64, 64, 3) and x.dtype == np.float32 and -1.0 <= x.min() and x.max() < 1.0
    y = synth.gtsdb_labels(6, 13, 43)
    assert y.shape == (6, 13, 13, 48) and y.dtype == np.float64
    n_obj = (y[..., 0] == 1).sum(axis=(1, 2))
    assert ((n_obj >= 1) & (n_obj <= 3)).all() and (y[y[..., 0] == 1][:, 5:].sum(axis=1) == 1).all()
    # shards of a global batch equal slices of the single-process batch
    np.testing.assert_array_equal(synth.images(4, 32, first=4), synth.images(8, 32)[4:])
    np.testing.assert_array_equal(synth.gtsdb_labels(3, 7, 43, first=2), synth.gtsdb_labels(5, 7, 43)[2:])
    np.testing.assert_array_equal(synth.gtsrb_labels(3, first=5), synth.gtsrb_labels(8)[5:])


def test_registry_and_cli_surface():
    import importlib.util
    spec = importlib.util.spec_from_file_location('cy_main', os.path.join(REPO, 'main.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert set(m.model_loss_predict) == set(config.model_names)
    args = m.parser.parse_args(['--model', 'darkcapsule', '--recon', '--no_metric'])
    assert args.recon is False and args.no_metric is True and args.lr == 1e-3         # store_false semantics kept
    assert m.parser.parse_args([]).recon is True
    for name in config.model_names:
        assert os.path.exists(os.path.join(REPO, config.model_dir[name], 'params.json'))
    p = m.load_params(os.path.join(REPO, 'experiments', 'darkcapsule_416'), args)
    assert p.n_grid == 13 and p.darknet_input == 416 and p.n_iter == 3 and p.recon is False


def test_input_pipeline_quantize_if_exact():
    """Centred image arrays of the reference's builders ((u8 - 128) / 128, float32 or float64) go back to uint8
    losslessly; anything else (augmented data) is refused."""
    from capsyolo_amd.input_pipeline import quantize_if_exact
    rng = np.random.default_rng(3)
    u = rng.integers(0, 256, (2, 5, 7, 3), dtype=np.uint8)
    for dt in (np.float32, np.float64):
        x = ((u.astype(np.float64) - 128.0) / 128.0).astype(dt)
        q = quantize_if_exact(x)
        assert q is not None and q.dtype == np.uint8 and np.array_equal(q, u)
        back = ((q.astype(np.float32) - 128.0) * np.float32(0.0078125))
        assert np.array_equal(back, x.astype(np.float32))
    assert quantize_if_exact(u) is u
    x = ((u.astype(np.float64) - 128.0) / 128.0)
    x[0, 0, 0, 0] += 1e-3
    assert quantize_if_exact(x) is None
    assert quantize_if_exact(x * 3.0) is None
    assert quantize_if_exact(np.zeros((0, 4, 4, 3), np.float32)) is None


def test_winograd_wait_counts_cover_the_emitted_code():
    """tools/check_lds_waits.py: every counted `s_waitcnt lgkmcnt(N)` of the kernels that read LDS with untracked inline asm (Winograd forward, bf16 conv, routing rows) is a lower bound
    of the LDS instructions hipcc really emitted between a fragment read and its first use (static check on the ISA)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        import pytest
        pytest.skip('hipcc not available')
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_lds_waits.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "check_lds_waits: ok" in r.stdout and r.stdout.count("0 not covered") >= 5


def test_asm_mfma_destinations_are_not_touched_before_their_last_pass():
    """tools/check_mfma_hazards.py: in the ISA hipcc emits for the kernels whose MFMAs are inline asm (Winograd 3x3 / 4x4-stride-2,
    bf16 GEMM, direct GEMM), no instruction reads or writes an MFMA's destination registers fewer than passes + 4 wait states behind
    it unless it is the next MFMA of the accumulate chain -- the hazard class hipcc does not pad for asm statements (gfx950 has no
    interlock there; symptom: run-to-run differences in lanes 48-63).  Plus the checker's own self-test on a synthetic listing."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('cy_chk_mfma', os.path.join(root, 'tools', 'check_mfma_hazards.py'))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    early = ['\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]', '\tv_add_f32_e32 v9, v8, v8', '\tv_accvgpr_read_b32 v10, a15']
    n, bad = chk.check_body(early, 'synthetic')
    assert n == 1 and len(bad) == 1 and bad[0][3] == 1 and bad[0][4] == 12          # one state behind an 8-pass MFMA: needs 12
    padded = [early[0], '\ts_nop 7', '\ts_nop 3', early[2]]
    assert chk.check_body(padded, 'synthetic')[1] == []
    chain = [early[0], '\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]'] + ['\ts_nop 7', '\ts_nop 3', early[2]]
    assert chk.check_body(chain, 'synthetic')[1] == []                            # the accumulate chain itself needs no states
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_mfma_hazards.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert 'check_mfma_hazards: ok' in r.stdout and r.stdout.count(' 0 closer') >= 4


def test_hand_counted_vmcnt_waits_cover_every_use_of_an_asm_loaded_register():
    """tools/check_vmcnt.py: winograd4.hip loads its B-operand ring and its input patch with inline-asm global loads (hipcc's own
    bookkeeping waited vmcnt(0) at the loop header) and waits with counted `s_waitcnt vmcnt(N)` statements.  The checker replays
    the emitted instruction stream (prologue, first and steady-state chunk, across a tile's drain) with the in-order queue of
    outstanding vector-memory operations: no instruction may touch the destination of a load that is still in the queue.  Plus a
    self-test on a synthetic listing (a wait that is one too weak must be found)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('cy_chk_vmcnt', os.path.join(root, 'tools', 'check_vmcnt.py'))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    loop = ['.LBB0_1:'] + ['\tv_mfma_f32_16x16x4_f32 a[0:3], v20, v21, a[0:3]'] * 144 + ['\ts_cbranch_scc1 .LBB0_1', '\ts_endpgm']
    pro = ['\tglobal_load_dwordx4 v[2:5], v0, s[0:1]', '\tglobal_load_dwordx4 v[6:9], v0, s[0:1] offset:1024']
    ok = pro + ['\ts_waitcnt vmcnt(1)', '\tv_add_f32_e32 v10, v2, v2'] + loop
    weak = pro + ['\ts_waitcnt vmcnt(2)', '\tv_add_f32_e32 v10, v2, v2'] + loop
    assert chk.check_kernel('synthetic', ok)[2] == []
    assert len(chk.check_kernel('synthetic', weak)[2]) == 1
    # conditional stores between a load and its wait: the path WITHOUT the store is the binding one (fewer younger operations)
    cond = ['\tglobal_load_dwordx4 v[2:5], v0, s[0:1]', '\ts_cbranch_execz .LBB0_0', '\tglobal_store_dword v0, v1, s[0:1]', '.LBB0_0:']
    assert chk.check_kernel('synthetic', cond + ['\ts_waitcnt vmcnt(0)', '\tv_add_f32_e32 v10, v2, v2'] + loop)[2] == []
    assert len(chk.check_kernel('synthetic', cond + ['\ts_waitcnt vmcnt(1)', '\tv_add_f32_e32 v10, v2, v2'] + loop)[2]) == 1
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_vmcnt.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    # every fused Winograd kernel with asm loads: 3 of winograd4.hip, 3 of winograd4_wgrad.hip, 7 of winograd4_s2.hip
    assert 'check_vmcnt: ok' in r.stdout and r.stdout.count(', 0 uses') == r.stdout.count(' loads and ') == 13
