"""Regenerates the binding block of INTEGRATION.md (between the GENERATED markers) from mnexp_amd/_lib.py, so that the
struct a reference maintainer copies is the one the library reads.   python tools/gen_integration_snippet.py [--check]"""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN, END = '# --- BEGIN GENERATED (tools/gen_integration_snippet.py) ---', '# --- END GENERATED ---'
CT = {C.c_int32: 'C.c_int32', C.c_uint32: 'C.c_uint32', C.c_float: 'C.c_float', C.c_size_t: 'C.c_size_t',
      C.c_void_p: 'C.c_void_p'}


def struct_text(name, cls, comment):
    rows = ['class %s(C.Structure):      # %s' % (name, comment), '    _fields_ = [']
    line = '        '
    for fname, ftype in cls._fields_:
        item = "('%s', %s), " % (fname, CT[ftype])
        if len(line) + len(item) > 116:
            rows.append(line.rstrip())
            line = '        '
        line += item
    rows.append(line.rstrip().rstrip(',') + ']')
    return '\n'.join(rows)


def generated():
    from mnexp_amd import _lib
    out = [BEGIN,
           'ABI_VERSION = %d           # include/lstur.h: LSTUR_ABI_VERSION; lib.lstur_abi_version() must return it' % _lib.ABI_VERSION,
           struct_text('Cfg', _lib.Cfg, 'include/lstur.h: lstur_cfg -- every field, in order; struct_size = C.sizeof(Cfg)'),
           struct_text('Sizes', _lib.Sizes, 'include/lstur.h: lstur_sizes'),
           struct_text('Buffers', _lib.Buffers, 'include/lstur.h: lstur_buffers (unused optional tables: None)'),
           'ARCH = %r' % (_lib.ARCH,),
           'SCORE = %r' % (_lib.SCORE,), END]
    return '\n'.join(out)


def main():
    path = os.path.join(ROOT, 'INTEGRATION.md')
    text = open(path).read()
    a, b = text.index(BEGIN), text.index(END) + len(END)
    new = text[:a] + generated() + text[b:]
    if '--check' in sys.argv:
        sys.exit(0 if new == text else 'INTEGRATION.md is stale: run python tools/gen_integration_snippet.py')
    open(path, 'w').write(new)


if __name__ == '__main__':
    main()
