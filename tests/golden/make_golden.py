"""Generates tests/golden/*.npz from the numpy oracle (oracle/), the only source of truth available: the reference
ships no vectors and TensorFlow 1.3 cannot run here (parity unpinned — see oracle/tf13_ops.py).

    python tests/golden/make_golden.py

Fixtures are DATA: seeded inputs and the oracle's outputs.  `msdn_b2.npz` stores seeds plus the 55x74 maps only
(the 283 MB of weights are regenerated from seed 3000 by oracle.msdn.init_params)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import msdn as O          # noqa: E402
from oracle import tf13_ops as T      # noqa: E402


def op_kats():
    rng = np.random.default_rng(424242)
    out = {}
    # conv edge cases of SURVEY 8c: SAME 5x5 borders, stride-2 VALID 13x18 -> 6x8, Cout = 63 / 1, Cin = 3 stride 4
    cases = {'same5': (2, 9, 11, 8, 12, 5, 1, 'SAME'), 's2valid': (2, 13, 18, 8, 12, 3, 2, 'VALID'),
             'cout63': (1, 20, 24, 3, 63, 9, 2, 'VALID'), 'cout1': (2, 7, 9, 16, 1, 5, 1, 'SAME'),
             'cin3s4': (2, 35, 47, 3, 16, 11, 4, 'VALID')}
    for name, (n, h, w, c, k, ks, st, pad) in cases.items():
        x = rng.standard_normal((n, h, w, c)).astype(np.float32)
        wt = (rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)
        b = rng.standard_normal(k).astype(np.float32)
        y = T.conv2d_fwd(x, wt, b, st, pad, relu=True)
        dz = rng.standard_normal(y.shape).astype(np.float32)
        dw, db = T.conv2d_bwd_filter(x, dz, wt.shape, st, pad)
        dx = T.conv2d_bwd_data(dz, wt, x.shape, st, pad)
        out.update({f'conv_{name}_{k_}': v for k_, v in dict(x=x, w=wt, b=b, y=y, dz=dz, dw=dw, db=db, dx=dx,
                                                              geom=np.array([st, pad == 'SAME'])).items()})
    # odd-width pooling 37 -> 18 with ties at zero
    x = np.maximum(rng.standard_normal((2, 27, 37, 8)), 0).astype(np.float32)
    dy = rng.standard_normal((2, 13, 18, 8)).astype(np.float32)
    out.update(pool_x=x, pool_y=T.maxpool2x2_fwd(x), pool_dy=dy, pool_dx=T.maxpool2x2_bwd(x, dy))
    # resize at the clamp edge (upscale 6x8 -> 55x74 as in BASELINE config 1) and the 480->228 downscale pattern
    x = rng.random((1, 6, 8, 1)).astype(np.float32)
    out.update(resize_up_x=x, resize_up_y=T.resize_bilinear_tf1(x, 55, 74))
    x = rng.random((1, 48, 64, 3)).astype(np.float32)
    out.update(resize_dn_x=x, resize_dn_y=T.resize_bilinear_tf1(x, 23, 30))
    # loss with o < 0 (NaN -> 0), t == 0, positive o
    o = (rng.standard_normal((4, 4070)) * 0.05).astype(np.float32)
    t = (rng.integers(0, 256, (4, 4070)) / 255).astype(np.float32)
    out.update(loss_o=o, loss_t=t, loss_value=np.float32(T.silog_loss_fwd(o, t)), loss_grad=T.silog_loss_bwd(o, t))
    # Adam, reference setting beta2 = 1 and a learning setting
    var = rng.standard_normal(1003).astype(np.float32)
    g = rng.standard_normal((3, 1003)).astype(np.float32)
    for tag, b2 in (('ref', 1.0), ('learn', 0.999)):
        opt = T.AdamTF1(0.1, 0.9, b2)
        v = {'w': var.copy()}
        for i in range(3):
            opt.apply(v, {'w': g[i]})
        out.update({f'adam_{tag}_var': v['w'], f'adam_{tag}_m': opt.m['w'], f'adam_{tag}_v': opt.v['w']})
    out.update(adam_var0=var, adam_g=g)
    return out


def msdn_b2():
    B = 2
    rng = np.random.default_rng(1000)
    img = (rng.integers(0, 256, (B, 48, 64, 3)) / 255).astype(np.float32)       # BASELINE config 1 stored sizes
    dep = (rng.integers(0, 256, (B, 6, 8, 1)) / 255).astype(np.float32)
    keep = rng.random((B, 4096)) >= 0.5
    a = O.forward(O.init_params(3000), img, dep, keep)
    return dict(seed_params=3000, images=img, depths=dep, keep=keep, coarse=a['coarse'], fine=a['fine'],
                loss_coarse=np.float32(a['loss_coarse']), loss_fine=np.float32(a['loss_fine']))


if __name__ == '__main__':
    np.savez_compressed(os.path.join(HERE, 'op_kats.npz'), **op_kats())
    np.savez_compressed(os.path.join(HERE, 'msdn_b2.npz'), **msdn_b2())
    for f in ('op_kats.npz', 'msdn_b2.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
