"""The C-ABI library loads without a GPU and exports exactly what include/a3d.h declares (no compute calls here)."""
import os
import re
import subprocess

from ann3depth_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'a3d.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(a3d_[a-z0-9_]+)\s*\(', text))


def test_header_symbols_are_exported_and_bound():
    lib = _lib.load()
    declared = declared_symbols()
    assert len(declared) >= 30
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (a3d_[a-z0-9_]+)', out))
    assert declared <= exported, declared - exported
    assert exported <= declared, f'exported but undeclared: {exported - declared}'
    assert set(_lib.SIGNATURES) == declared
    for name in declared:
        assert getattr(lib, name).argtypes is not None
    assert b'gfx950' in lib.a3d_version()


def test_the_documented_binding_stub_matches_the_struct_the_library_reads():
    """VERDICT r3 (weak 12): INTEGRATION.md's ctypes example of a3d_conv_desc fell two fields behind include/a3d.h.  The
    field list is taken from the document itself, rebuilt as a ctypes struct, and its size and field names must be the
    library's (a3d_sizeof_conv_desc) and the header's."""
    import ctypes
    lib = _lib.load()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r"class ConvDesc\(ctypes\.Structure\):.*?_fields_ = \[\(n, ctypes\.c_int32\) for n in\s*\((.*?)\)\]", doc, flags=re.S)
    assert m, 'INTEGRATION.md no longer shows the ConvDesc stub'
    doc_fields = re.findall(r"'(\w+)'", m.group(1))

    class DocDesc(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int32) for n in doc_fields]
    assert ctypes.sizeof(DocDesc) == lib.a3d_sizeof_conv_desc() == ctypes.sizeof(_lib.ConvDesc)
    header = open(os.path.join(ROOT, 'include', 'a3d.h')).read()
    body = re.search(r'typedef struct a3d_conv_desc \{(.*?)\} a3d_conv_desc;', header, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    header_fields = [f.strip() for decl in re.findall(r'int32_t ([^;]+);', body) for f in decl.split(',')]
    assert doc_fields == header_fields == [n for n, _ in _lib.ConvDesc._fields_]
    assert 'a3d_sizeof_conv_desc' in doc


def test_staging_pool_slots_are_range_checked():
    """a3d_records_decode / a3d_h2d_gather index pinned pools with caller-supplied slot numbers; a slot outside
    [0, nslots) or a non-positive dimension is A3D_EINVAL before a byte is written or a copy enqueued (no GPU needed: the
    checks come first)."""
    import ctypes

    import numpy as np
    lib = _lib.load()
    img = np.full((2, 2, 3), 0.25, np.float32)
    dep = np.full((1, 1, 1), 0.25, np.float32)
    buf = ctypes.create_string_buffer(4096)
    n = lib.a3d_example_write(img.ctypes.data, 2, 2, 3, dep.ctypes.data, 1, 1, 1, buf, 4096)
    assert 0 < n < 4096
    frame = np.frombuffer(buf.raw[:n], np.uint8).copy()
    nslots = 3
    ipool = np.full((nslots + 2, 2, 2, 3), 7.0, np.float32)            # two guard slots after the pool proper
    dpool = np.full((nslots + 2, 1, 1, 1), 7.0, np.float32)
    frames = (ctypes.c_void_p * 1)(frame.ctypes.data)
    lens = (ctypes.c_size_t * 1)(frame.size)
    kinds = (ctypes.c_int32 * 1)()
    dims = (ctypes.c_int64 * 6)(2, 2, 3, 1, 1, 1)

    def decode(slot, dims=dims, n=nslots):
        return lib.a3d_records_decode(frames, lens, 1, 1, dims, None, ipool.ctypes.data, None, dpool.ctypes.data,
                                      (ctypes.c_int32 * 1)(slot), n, kinds)
    assert decode(1) == 0
    np.testing.assert_array_equal(ipool[1], img + np.float32(0.5))
    for bad in (-1, nslots, nslots + 1, 2 ** 31 - 1):
        assert decode(bad) == -1 and 'slot' in _lib.last_error()
    assert (ipool[[0, 2, 3, 4]] == 7.0).all() and (dpool[[0, 2, 3, 4]] == 7.0).all()     # nothing outside slot 1 was touched
    assert decode(0, n=0) == -1
    for j in range(6):
        d = list(dims)
        d[j] = 0
        assert decode(0, dims=(ctypes.c_int64 * 6)(*d)) == -1
        d[j] = -4
        assert decode(0, dims=(ctypes.c_int64 * 6)(*d)) == -1
    dst = np.zeros((1, 12), np.float32)
    for bad in (-1, nslots, 10 ** 6):
        rc = lib.a3d_h2d_gather(dst.ctypes.data, ipool.ctypes.data, (ctypes.c_int32 * 1)(bad), 1, nslots, 48, None)
        assert rc == -1 and 'slot' in _lib.last_error()


def test_errors_are_reported_not_thrown():
    lib = _lib.load()
    d = _lib.ConvDesc(n=1, h=8, w=8, c=4, k=4, r=3, s=3, stride=3, pad_t=1, pad_l=1, ho=8, wo=8, ldx=4, ldy=4)
    assert lib.a3d_conv2d_fwd_ws_bytes(d) == 0                       # invalid descriptor: no workspace answer
    rc = lib.a3d_conv2d_fwd(d, None, None, None, None, 0, None, 0, None)
    assert rc == -1 and 'stride' in _lib.last_error()                # A3D_EINVAL before anything touches a device
    rc = lib.a3d_adam_apply_tf1(0, None, None, None, None, 0, 0, 0, 0, 0, 0, 1, None)
    assert rc == -1


def test_gemm_code_objects_are_gfx950_mfma(tmp_path):
    """The shipped library holds gfx950 code objects whose disassembly carries the matrix-core instructions the
    design names (fp32 32x32x2 and bf16 32x32x16 MFMA, the LDS-DMA load and the transposing LDS read): it is not a
    host fallback.  The bundle is extracted from a COPY under tmp_path (llvm-objdump writes next to its input)."""
    import glob
    import shutil
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        import pytest
        pytest.skip('llvm-objdump not in this image')
    copy = shutil.copy(_lib.LIB_PATH, tmp_path / 'liba3d.so')
    out = subprocess.run([objdump, '--offloading', str(copy)], capture_output=True, text=True, cwd=tmp_path).stdout
    assert 'gfx950' in out
    objs = glob.glob(str(tmp_path / 'liba3d.so.*gfx950*'))
    assert objs, 'no gfx950 code object in the offload bundle'
    seen = set()
    wanted = ('v_mfma_f32_32x32x2_f32', 'v_mfma_f32_32x32x16_bf16', 'ds_read_b64_tr_b16', 'global_load_lds_dwordx4')
    for o in objs:
        if os.path.getsize(o) == 0:
            continue
        asm = subprocess.run([objdump, '-d', '--mcpu=gfx950', o], capture_output=True, text=True).stdout
        seen |= {w for w in wanted if w in asm}
    assert seen == set(wanted), set(wanted) - seen


def test_planner_handles_every_layer_shape_on_the_host():
    """The tile / split-K planner runs on the host inside the *_ws_bytes queries: every MSDN (B = 32, 64) and DCNF
    (B = 16 -> 768 patches) layer, every direction and precision must get a plan with a bounded workspace."""
    from ann3depth_amd import models, ops
    lib = _lib.load()
    import ctypes
    descs = []
    for B in (1, 32, 64):
        for prec in ('fp32', 'bf16x3', 'bf16'):
            descs += [ops.conv_desc(B, 228, 304, 3, 96, 11, 11, 4, 'VALID', precision=prec),
                      ops.conv_desc(B, 27, 37, 96, 256, 5, 5, 1, 'SAME', precision=prec),
                      ops.conv_desc(B, 13, 18, 256, 384, 3, 3, 1, 'SAME', precision=prec),
                      ops.conv_desc(B, 13, 18, 384, 384, 3, 3, 1, 'SAME', precision=prec),
                      ops.conv_desc(B, 13, 18, 384, 256, 3, 3, 2, 'VALID', precision=prec),
                      ops.conv_desc(B, 228, 304, 3, 63, 9, 9, 2, 'VALID', precision=prec),
                      ops.conv_desc(B, 55, 74, 64, 64, 5, 5, 1, 'SAME', ldx=64, precision=prec),
                      ops.conv_desc(B, 55, 74, 64, 1, 5, 5, 1, 'SAME', precision=prec)]
    P = 768
    h = 100
    for n, ci, co, k in models.DCNF_CONVS:
        descs.append(ops.conv_desc(P, h, h, ci, co, k, k, 1, 'VALID'))
        h = h - k + 1
        if n in models.DCNF_POOL_AFTER:
            h //= 2
    for d in descs:
        for fn in (lib.a3d_conv2d_fwd_ws_bytes, lib.a3d_conv2d_bwd_data_ws_bytes, lib.a3d_conv2d_bwd_filter_ws_bytes):
            ws = fn(ctypes.byref(d))
            assert 0 <= ws <= 200 << 20, (d.n, d.c, d.k, d.r, fn.__name__, ws)
    # bf16-stored operands (BASELINE config 5): the LDS-DMA kernel's plans — never-split forward / bwd-data, split-K slabs of the
    # filter gradient + the column-sum partials, the dense layers' 64-row tiles with their slabs
    X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
    for B in (2, 32, 64):
        for shape in [(B, 27, 37, 96, 256, 5, 5, 1, 'SAME'), (B, 13, 18, 256, 384, 3, 3, 1, 'SAME'),
                      (B, 13, 18, 384, 384, 3, 3, 1, 'SAME'), (B, 13, 18, 384, 256, 3, 3, 2, 'VALID'),
                      (B, 55, 74, 64, 64, 5, 5, 1, 'SAME'), (B, 1, 1, 12288, 4096, 1, 1, 1, 'VALID')]:
            d = ops.conv_desc(*shape, precision='bf16')
            for fn, bits in ((lib.a3d_conv2d_fwd_ws_bytes, X | W | Y), (lib.a3d_conv2d_bwd_data_ws_bytes, X | W | Y),
                             (lib.a3d_conv2d_bwd_filter_ws_bytes, X | Y)):
                ws = fn(ctypes.byref(ops.with_storage(d, bits)))
                assert 0 <= ws <= 200 << 20, (shape, fn.__name__, ws)
    d = ops.with_storage(ops.conv_desc(64, 27, 37, 96, 256, 5, 5, 1, 'SAME', precision='bf16'), X | Y)
    assert lib.a3d_conv2d_bwd_filter_ws_bytes(ctypes.byref(d)) >= 2 * 2400 * 256 * 4      # split-K slabs of the LDS-DMA bwd-filter
    for m, k, n in [(32, 12288, 4096), (32, 4096, 4070), (64, 12288, 4096), (768, 12544, 128), (768, 128, 16), (768, 16, 1)]:
        for fn in (lib.a3d_dense_fwd_ws_bytes, lib.a3d_dense_bwd_data_ws_bytes, lib.a3d_dense_bwd_filter_ws_bytes):
            assert 0 <= fn(m, k, n) <= 200 << 20


def test_the_library_reads_its_environment_only_behind_the_tuning_gate():
    """VERDICT r4 item 7 / SURVEY 8b ("keeps no global state"): in the shipped liba3d.so, getenv is called from a3d::tuning()
    (A3D_TUNING itself, once) and a3d::tune_int() (which returns its default without touching the environment unless the
    process was started with A3D_TUNING=1) and from nowhere else; and no source file but capi.cc spells getenv."""
    dis = subprocess.run(['objdump', '-d', '-C', '--no-show-raw-insn', '-j', '.text', _lib.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    fn, callers = None, set()
    for line in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
        if m:
            fn = m.group(1)
        elif 'getenv@plt' in line or re.search(r'call.*<getenv', line):
            callers.add(fn)
    assert callers, 'no getenv call found at all: did the disassembly work?'
    assert all(c.startswith('a3d::tuning()') or c.startswith('a3d::tune_int(') for c in callers), callers
    csrc = os.path.join(ROOT, 'ann3depth_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.cc', '.h')) and name != 'capi.cc':
            assert 'getenv' not in open(os.path.join(csrc, name)).read(), name
