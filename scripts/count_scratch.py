"""Counts the scratch (register spill) instructions in the gfx950 code objects of a shared library: extracts the
.hip_fatbin section, splits the clang offload bundles, disassembles every code object with llvm-objdump.
    python scripts/count_scratch.py modl_amd/libmodl_hip.so modl_amd/libmodl_hip_diag.so
The product library must have none (tests/test_abi.py); the diagnostics build carries the one-wavefront coordinate-descent
kernel's 8 / 16-coefficients-per-lane variants, which spill."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def count(path):
    tmp = tempfile.mkdtemp(prefix='modl_co_')
    fb = os.path.join(tmp, 'fatbin')
    subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fb])
    data = open(fb, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    total = nco = pos = 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from('<Q', data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from('<QQQ', data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if 'gfx950' in triple and s > 0:
                co = os.path.join(tmp, 'co_%d.co' % nco)
                nco += 1
                open(co, 'wb').write(data[i + o:i + o + s])
                d = subprocess.run([OBJDUMP, '-d', co], capture_output=True, text=True).stdout
                total += len(re.findall(r'\bscratch_(?:load|store)', d))
                os.remove(co)
        pos = i + 1
    os.remove(fb)
    os.rmdir(tmp)
    return nco, total


if __name__ == '__main__':
    for p in sys.argv[1:]:
        print(p, 'code objects %d, scratch instructions %d' % count(p))
