"""The oracle against the committed golden vectors (guards the oracle against drift) and host-side model logic."""
import os

import numpy as np

from oracle import msdn as O
from oracle import tf13_ops as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_oracle_reproduces_op_kats():
    g = np.load(os.path.join(GOLD, 'op_kats.npz'))
    for name in ('same5', 's2valid', 'cout63', 'cout1', 'cin3s4'):
        st, same = (int(v) for v in g[f'conv_{name}_geom'])
        pad = 'SAME' if same else 'VALID'
        x, w, b, dz = (g[f'conv_{name}_{k}'] for k in ('x', 'w', 'b', 'dz'))
        np.testing.assert_allclose(T.conv2d_fwd(x, w, b, st, pad, relu=True), g[f'conv_{name}_y'], rtol=1e-5, atol=1e-6)
        dw, db = T.conv2d_bwd_filter(x, dz, w.shape, st, pad)
        np.testing.assert_allclose(dw, g[f'conv_{name}_dw'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(db, g[f'conv_{name}_db'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(T.conv2d_bwd_data(dz, w, x.shape, st, pad), g[f'conv_{name}_dx'], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(T.maxpool2x2_fwd(g['pool_x']), g['pool_y'])
    np.testing.assert_array_equal(T.maxpool2x2_bwd(g['pool_x'], g['pool_dy']), g['pool_dx'])
    np.testing.assert_array_equal(T.resize_bilinear_tf1(g['resize_up_x'], 55, 74), g['resize_up_y'])
    np.testing.assert_array_equal(T.resize_bilinear_tf1(g['resize_dn_x'], 23, 30), g['resize_dn_y'])
    assert abs(T.silog_loss_fwd(g['loss_o'], g['loss_t']) - g['loss_value']) < 1e-5 * abs(g['loss_value'])
    np.testing.assert_allclose(T.silog_loss_bwd(g['loss_o'], g['loss_t']), g['loss_grad'], rtol=1e-5, atol=1e-3)


def test_oracle_reproduces_msdn_golden():
    g = np.load(os.path.join(GOLD, 'msdn_b2.npz'))
    a = O.forward(O.init_params(int(g['seed_params'])), g['images'], g['depths'], g['keep'])
    rel = lambda x, y: np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y)
    assert rel(a['coarse'], g['coarse']) < 1e-5 and rel(a['fine'], g['fine']) < 1e-5
    assert a['coarse'].shape == (2, 55, 74, 1) and a['fine'].shape == (2, 55, 74, 1)


def test_phase_schedule_and_variable_inventory():
    # src/models.py:301-305,348-365 at B=32: boundaries 62,500 and 109,375
    assert O.phase_of(0, 32) == 1 and O.phase_of(62499, 32) == 1
    assert O.phase_of(62500, 32) == 2 and O.phase_of(109374, 32) == 2
    assert O.phase_of(109375, 32) == 3
    shapes = O.param_shapes()
    assert sum(int(np.prod(s)) for s in shapes.values()) == 70877171            # 270.4 MiB
    assert sum(int(np.prod(s)) for n, s in shapes.items() if n.startswith('coarse')) == 70757734
    from ann3depth_amd import models
    assert models.phase_of(62500, 32) == 2 and models.phase_of(109375, 32) == 3 and models.phase_of(3, 4) == 1
    # the product's variable table is the oracle's (same TF names and shapes)
    prod = {c.name + '/kernel': (c.k, c.k, c.cin, c.cout) for c in models.MSDN_CONVS}
    prod.update({n + '/kernel': (i, o) for n, i, o in models.MSDN_DENSES})
    for n, s in prod.items():
        assert tuple(shapes[n]) == tuple(s), n
    assert len(prod) * 2 == len(shapes)
