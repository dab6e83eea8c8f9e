#!/usr/bin/env python3
"""Lists every kernel of libtvae_hip.so that uses private (scratch) memory, from the code objects embedded in the
library (no GPU needed): python profiles/tools/scratch_audit.py [path/to/libtvae_hip.so]
Scratch is a performance smell (spills / arrays the compiler could not keep in registers) and, on this pool, a
correctness hazard when several processes share one GPU (profiles/README.md, round 3): the library should list NONE.
Also counts packed-fp32 vector instructions (v_pk_fma_f32 ...), which the build disables (csrc/Makefile: NOPK)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'target-vae_amd', 'csrc', 'build', 'libtvae_hip.so')
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
data = open(so, 'rb').read()
offs = [m.start() for m in re.finditer(b'\x7fELF\x02\x01\x01\x40', data)]      # ELF64, little endian, OS/ABI 64 = AMDGPU HSA
found = []
total = 0
npk = 0
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
for o in offs:
    with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
        f.write(data[o:])
        path = f.name
    txt = subprocess.run([READELF, '--notes', path], capture_output=True, text=True).stdout
    dis = subprocess.run([OBJDUMP, '-d', path], capture_output=True, text=True).stdout
    npk += len(re.findall(r'\bv_pk_(?:fma|mul|add|mov)_(?:f32|b32)\b', dis))
    os.unlink(path)
    for blk in re.split(r'\n\s+- ', txt):
        n = re.search(r'\.name:\s+(\S+)', blk)
        p = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk)
        if n and p:
            total += 1
            if int(p.group(1)) > 0:
                found.append((n.group(1), int(p.group(1))))
print(f'{len(offs)} code objects, {total} kernels, {len(found)} with scratch, {npk} packed-fp32 instructions')
for n, p in sorted(found):
    dem = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    print(f'  {p:5d} B/lane  {dem[:150]}')
sys.exit(1 if (found or npk) else 0)
