"""Which kernels wait for every store?  Reads the ISA of one or more .s files (hipcc -S --cuda-device-only) and lists the
kernels in which consecutive global / buffer stores are separated by `s_waitcnt vmcnt(0)`: each such store waits for the
round trip of the one before it (vmcnt counts loads AND stores, in order).  The usual cause is a per-value epilogue whose
uniform branches (activation, dropout, output type) or mask loads sit inside the store loop: the compiler closes each
value's control flow with a full wait.  DESIGN.md 3.1j.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude --cuda-device-only -S ann3depth_amd/csrc/igemm_fwd.hip -o /tmp/f.s
    python tools/isa/store_waits.py /tmp/f.s [substring of the kernel name]"""
import re
import sys


def kernels(path):
    name, out = None, {}
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if name is not None:
            out[name].append(line)
            if line.startswith('.Lfunc_end'):
                name = None
    return out


def main():
    files = [a for a in sys.argv[1:] if a.endswith('.s')]
    pat = next((a for a in sys.argv[1:] if not a.endswith('.s')), '')
    for path in files:
        for k, lines in kernels(path).items():
            if pat not in k:
                continue
            stores = [i for i, l in enumerate(lines) if re.search(r'\b(global|buffer)_store', l)]
            waits = [i for i, l in enumerate(lines) if 's_waitcnt vmcnt(0)' in l]
            ser = sum(1 for a, b in zip(stores, stores[1:]) if any(a < w < b for w in waits))
            if len(stores) >= 3 and (pat or ser >= 2):
                print(f'{path.split("/")[-1]:18s} {len(lines):6d} lines  stores {len(stores):4d}  followed by a full wait {ser:4d}  {k[:110]}')


if __name__ == '__main__':
    main()
