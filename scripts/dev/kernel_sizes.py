"""Every device kernel of a built library with its code size: the gfx950 code objects are cut out of the fat binary
(clang-offload-bundler magic inside .hip_fatbin) and their FUNC symbols listed (llvm-readelf), names demangled.

  python scripts/dev/kernel_sizes.py geeco_amd/libgeeco_hip.so [--csv]
"""
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin/'


def code_objects(path):
  data = open(path, 'rb').read()
  out, pos = [], 0
  magic = b'__CLANG_OFFLOAD_BUNDLE__'
  while True:
    i = data.find(magic, pos)
    if i < 0:
      break
    n = int.from_bytes(data[i + 24:i + 32], 'little')
    p = i + 32
    for _ in range(n):
      off, size, tl = (int.from_bytes(data[p + 8 * k:p + 8 * k + 8], 'little') for k in range(3))
      triple = data[p + 24:p + 24 + tl].decode()
      p += 24 + tl
      if 'gfx950' in triple and size:
        out.append(data[i + off:i + off + size])
    pos = i + 24
  return out


def kernels(path):
  rows = []
  for co in code_objects(path):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
      f.write(co)
      f.flush()
      txt = subprocess.run([LLVM + 'llvm-readelf', '--symbols', '--wide', f.name], capture_output=True, text=True).stdout
    kd = {m.group(1) for m in re.finditer(r'OBJECT\s+\w+\s+\w+\s+\d+\s+(\S+)\.kd\b', txt)}
    seen = set()                # (.dynsym and .symtab both list a kernel)
    for m in re.finditer(r'^\s*\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\w+\s+\w+\s+\d+\s+(\S+)', txt, re.M):
      if m.group(2) in kd and m.group(2) not in seen:
        seen.add(m.group(2))
        rows.append((int(m.group(1)), m.group(2)))
  names = subprocess.run(['c++filt'], input='\n'.join(n for _, n in rows), capture_output=True, text=True).stdout.split('\n')
  return sorted(((s, re.sub(r'^void ', '', re.sub(r'\(.*\)$', '', d))) for (s, _), d in zip(rows, names)), reverse=True)


if __name__ == '__main__':
  import os
  ks = kernels(sys.argv[1])
  print('# %s: %d bytes on disk, %d device kernels, %d bytes of kernel code' % (sys.argv[1], os.path.getsize(sys.argv[1]), len(ks), sum(s for s, _ in ks)))
  for s, n in ks:
    print('%8d  %s' % (s, n))
