"""CPU: the C-ABI library loads, exports every symbol include/ssunet_hip.h declares, and the
ctypes table in ssunet-gan_amd/_lib.py matches the header's parameter counts.  No compute calls."""
import ctypes
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'ssunet_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(?:int|int64_t|const char\*)\s+(ssg_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',')])
        decls[m.group(1)] = n
    return decls


def test_header_has_no_torch_types_and_is_extern_c():
    src = open(HEADER).read()
    assert 'extern "C"' in src
    for bad in ('torch', 'at::', 'Tensor', 'std::', 'hipStream_t'):
        assert bad not in re.sub(r'/\*.*?\*/', '', src, flags=re.S), bad


def test_library_exports_every_declared_symbol(pkg):
    decls = _declared()
    assert len(decls) >= 35
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), 'libssunet_hip.so does not export %s' % name


def test_ctypes_table_matches_header(pkg):
    decls = _declared()
    sig = pkg._lib.SIGNATURES
    missing = set(decls) - set(sig) - {'ssg_last_error'}
    extra = set(sig) - set(decls)
    assert not missing and not extra, (missing, extra)
    for name, n in decls.items():
        if name in sig:
            assert len(sig[name]) == n, '%s: header has %d params, ctypes table %d' % (name, n, len(sig[name]))
    lib = pkg._lib.load()
    assert lib.ssg_abi_version() == 9 == pkg._lib.ABI_VERSION
    assert lib.ssg_last_error() is not None


def test_struct_layouts_match_header(pkg):
    """Field order/count of the two descriptor structs (sizes checked against the C compiler's
    layout rules via ctypes)."""
    src = open(HEADER).read()
    conv = src[src.index('typedef struct {', src.index('kmode 1')):src.index('} ssg_conv_desc;')]
    names = re.findall(r'(\w+)(?:\[SSG_MAX_TAPS\])?\s*[;,]', re.sub(r'/\*.*?\*/', '', conv, flags=re.S))
    got = [f[0] for f in pkg._lib.ConvDesc._fields_]
    assert names == got, (names, got)
    wg = src[src.index('typedef struct {', src.index('} ssg_conv_desc;')):src.index('} ssg_wgrad_desc;')]
    names = re.findall(r'(\w+)(?:\[SSG_MAX_TAPS\])?\s*[;,]', re.sub(r'/\*.*?\*/', '', wg, flags=re.S))
    got = [f[0] for f in pkg._lib.WgradDesc._fields_]
    assert names == got, (names, got)


def test_bad_arguments_return_status_not_crash(pkg):
    """Validation happens on the host before any launch, so this is safe without a GPU."""
    lib = pkg._lib
    d = lib.ConvDesc()
    rc = lib.load().ssg_conv2d_igemm_f32(ctypes.byref(d), None)
    assert rc != 0 and b'conv' in lib.load().ssg_last_error()
    try:
        lib.call('ssg_conv2d_igemm_f32', ctypes.byref(d), None)
        assert False
    except RuntimeError as e:
        assert 'ssg_conv2d_igemm_f32 failed' in str(e)


def test_fused_spade_entry_rejects_unsupported_descriptors(pkg):
    """ssg_spade_conv_modulate_ok is a pure host-side predicate; the launcher refuses what the predicate refuses (no GPU needed)."""
    lib = pkg._lib
    d = lib.ConvDesc()
    assert lib.call('ssg_spade_conv_modulate_ok', ctypes.byref(d)) == 0
    assert lib.call('ssg_conv2d_workspace_bytes', ctypes.byref(d)) == 0
    rc = lib.load().ssg_spade_conv_modulate_f32(ctypes.byref(d), None, 0, None, 0, None)
    assert rc != 0 and b'spade_conv_modulate' in lib.load().ssg_last_error()


def test_struct_sizes_and_offsets_match_the_c_compiler(pkg, tmp_path):
    """The three structs that cross the boundary by pointer (ssg_conv_desc, ssg_wgrad_desc, ssg_bn_fin): size and every field
    offset as gcc lays the header out == what the ctypes mirrors in _lib.py say (a maintainer's cgo / ctypes stub binds the
    header, the package binds the mirrors: both must describe the same bytes)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    lib = pkg._lib
    structs = (('ssg_conv_desc', lib.ConvDesc), ('ssg_wgrad_desc', lib.WgradDesc), ('ssg_bn_fin', lib.BnFin))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HEADER, 'int main(void) {']
    for cname, mirror in structs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in mirror._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', '-o', str(exe), str(src)], check=True, capture_output=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, mirror in structs:
        assert int(out[cname]) == ctypes.sizeof(mirror), (cname, out[cname], ctypes.sizeof(mirror))
        for f in mirror._fields_:
            assert int(out['%s.%s' % (cname, f[0])]) == getattr(mirror, f[0]).offset, (cname, f[0])
    # field names of ssg_bn_fin in header order
    hdr = open(HEADER).read()
    body = hdr[hdr.index('typedef struct ssg_bn_fin {'):hdr.index('} ssg_bn_fin;')]
    names = re.findall(r'(\w+)\s*[;,]', re.sub(r'/\*.*?\*/', '', body, flags=re.S))
    assert names == [f[0] for f in lib.BnFin._fields_], names
