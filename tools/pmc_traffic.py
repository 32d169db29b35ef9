"""Builds profiles/<round>_pmc_traffic.json from two rocprofv3 counter passes of the bench command:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py <args>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py <args>
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv \
           profiles/r01_pmc_traffic.json

Per kernel: average FETCH_SIZE / WRITE_SIZE (KiB) per launch and hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 — on
gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section).  Check the correction on a
kernel of known traffic: adam_frozen_kernel reads g + m and writes m."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            tot[r['Kernel_Name']] += float(r['Counter_Value'])
            n[r['Kernel_Name']] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main(fetch_csv, write_csv, out, how=''):
    f, w = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(f) & set(w)):
        name = k.replace('void ', '').split('(')[0]
        kernels[name] = {'fetch_size_kb': round(f[k][0], 1), 'write_size_kb': round(w[k][0], 1),
                         'launches_sampled': min(f[k][1], w[k][1]),
                         'hbm_bytes_per_launch': int((2 * f[k][0] + w[k][0]) * 1024)}
    json.dump({'_how': how or __doc__, 'kernels': kernels}, open(out, 'w'), indent=1)
    return kernels


if __name__ == '__main__':
    ks = main(*sys.argv[1:4])
    for name in ('a3d::adam_frozen_kernel', 'a3d::igemm_kernel<2, 128, 128, 4, 8, 32, 4, 4>'):
        print(name, ks.get(name))
