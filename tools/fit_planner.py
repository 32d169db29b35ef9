"""Offline check of the plan_gemm cost model against gpurun_out/sweep_full.json: prints, per problem, the measured
time of the model's pick relative to the best measured configuration."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {  # name: (bm, bn, eff)
    '128x128': (128, 128, 1.00), '128x96': (128, 96, 1.00), '128x64': (128, 64, 0.92), '128x32': (128, 32, 0.60),
    '64x64': (64, 64, 0.98), '32x128': (32, 128, 0.90), '64x128': (64, 128, 1.00), '128x128w8': (128, 128, 1.15),
    '128x64w8': (128, 64, 1.05), 'G128x128w8': (128, 128, 1.19), 'G128x64w8': (128, 64, 1.09),
}
SLOTS = 512


def model(M, N, K, bm, bn, eff, want):
    nk = (K + 31) // 32
    if want > 1 and want > nk // 2:
        return None
    kps = -(-nk // want)
    splitk = -(-nk // kps)
    tiles = -(-M // bm) * -(-N // bn)
    blocks = tiles * splitk
    t_b = bm * bn * kps * 32.0 / (96.5e3 * eff)
    if blocks <= 256:
        f = 0.62
    elif blocks <= SLOTS:
        f = 1.0
    else:
        f = max(blocks / SLOTS + 0.08, 1.45)
    t = t_b * f
    if splitk > 1:
        t += 2.5 + M * N * 4.0 * (splitk + 1) / 3.0e6
    return t, splitk


def main():
    data = json.load(open(os.path.join(ROOT, 'gpurun_out', 'sweep_full.json')))
    tot_best = tot_pick = 0
    for p in data:
        meas = {(cn, sk): t for cn, sk, t in p['results']}
        best = min(meas.items(), key=lambda kv: kv[1])
        cands = []
        for cn, (bm, bn, eff) in CFG.items():
            for want in (1, 2, 4, 8, 16, 32, 64, 128, 256):
                r = model(p['M'], p['N'], p['K'], bm, bn, eff, want)
                if r is None:
                    continue
                cands.append((r[0], cn, want))
        cands.sort()
        pick = None
        for t, cn, want in cands:
            if (cn, want) in meas:
                pick = (cn, want, t)
                break
        tp = meas[(pick[0], pick[1])]
        tot_best += best[1]
        tot_pick += tp
        print(f"{p['layer']:9s} {p['mode']:6s} M={p['M']:7d} N={p['N']:5d} K={p['K']:7d} best {best[0][0]:>10s} sk{best[0][1]:<3d} "
              f"{best[1]:7.0f}us | pick {pick[0]:>10s} sk{pick[1]:<3d} model {pick[2]:7.0f} meas {tp:7.0f}us  x{tp / best[1]:.2f}")
    print(f'sum best {tot_best:.0f}us  sum pick {tot_pick:.0f}us')


if __name__ == '__main__':
    main()
