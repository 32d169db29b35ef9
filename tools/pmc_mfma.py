"""Builds profiles/<round>_pmc_mfma_busy.json from one rocprofv3 counter pass of the bench command:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
              --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fine --no-dp-rank --also ""
    python tools/pmc_mfma.py gpurun_out/pmc_mfma/m_counter_collection.csv gpurun_out/pmc_mfma/m_kernel_trace.csv profiles/rNN_pmc_mfma_busy.json

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the kernel during which a
SIMD's matrix pipe is executing (GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import collections
import csv
import json
import sys


def main(counters, trace, out):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(counters)):
        k = r['Kernel_Name']
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            n[k] += 1
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    kernels = {}
    for k, c in tot.items():
        if not any(t in k for t in ('igemm', 'dense', 'conv3', 'fewch', 'stencil1_bwd_kernel')):
            continue
        name = k.replace('void ', '').split('(')[0]
        m = n[k]
        gui = c['GRBM_GUI_ACTIVE'] / m
        busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / m
        kernels[name] = {'launches_sampled': m, 'avg_us': round(sum(dur[k]) / max(len(dur[k]), 1) / 1e3, 1),
                         'mfma_insts_per_launch': int(c['SQ_INSTS_MFMA'] / m), 'mfma_busy_cycles_per_launch': int(busy),
                         'gui_active_cycles_per_xcd': int(gui / 8), 'mfma_busy_frac': round(busy / (1024 * gui / 8), 3) if gui else None,
                         'lds_bank_conflict_cycles': int(c['SQ_LDS_BANK_CONFLICT'] / m)}
    json.dump({'_how': __doc__, 'kernels': kernels}, open(out, 'w'), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['launches_sampled'])[:8]:
        print(k, v['launches_sampled'], v['avg_us'], v['mfma_busy_frac'], v['lds_bank_conflict_cycles'])


if __name__ == '__main__':
    main(*sys.argv[1:4])
