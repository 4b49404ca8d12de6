#!/usr/bin/env python3
"""What the compiler made of the kernels in libe3unet.so: per kernel its register / scratch budget (code-object metadata) and an
instruction histogram (llvm-objdump of the embedded gfx950 code objects).

The performance of several kernels depends on properties of the generated code that no runtime test sees:
`volatile` LDS reads that keep hipcc from pairing into ds_read2_b64 in conv3_wino4_kernel, zero scratch in the 512-register kernels,
the MFMA opcode each family is built on.  tests/test_isa.py asserts them on every build (CPU tier; hipcc cross-compiles without a GPU).

    python tools/isa_check.py                 # table of every kernel
    python tools/isa_check.py wino4           # kernels whose demangled name contains the pattern, with their histograms
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(so_path):
    """The gfx950 ELF images of every translation unit (uncompressed clang offload bundles in .hip_fatbin)."""
    d = open(so_path, 'rb').read()
    out = []
    for m in re.finditer(MAGIC, d):
        off = m.start()
        n = struct.unpack_from('<Q', d, off + 24)[0]
        p = off + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from('<QQQ', d, p)
            p += 24
            triple = d[p:p + ts]
            p += ts
            if b'gfx950' in triple and sz:
                out.append(d[off + o:off + o + sz])
    return out


def _tool(name):
    exe = os.path.join(LLVM, name)
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    return exe


def demangle(names):
    import shutil
    filt = os.path.join(LLVM, 'llvm-cxxfilt')
    if not os.path.exists(filt):
        filt = shutil.which('c++filt')
    if not filt or not names:
        return {n: n for n in names}
    r = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.splitlines()))


_META_KEYS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size',
              'max_flat_workgroup_size')


def kernel_metadata(elf_path):
    """{mangled name: {vgpr_count, agpr_count, ..., private_segment_fixed_size}} from the NT_AMDGPU_METADATA note."""
    txt = subprocess.run([_tool('llvm-readelf'), '--notes', elf_path], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r'\s+(?:- )?\.(\w+):\s+(.*)$', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if line.lstrip().startswith('- .') and re.match(r'  - \.', line):       # first key of a kernel entry (two-space list item)
            cur = {}
            kernels[id(cur)] = cur
        if cur is None:
            continue
        if k == 'name':
            cur['name'] = v
        elif k in _META_KEYS:
            try:
                cur[k] = int(v)
            except ValueError:
                pass
    return {c['name']: c for c in kernels.values() if 'name' in c}


def kernel_histograms(elf_path):
    """{mangled name: {mnemonic(+ ' lds' for LDS-DMA loads): count}} from the disassembly."""
    txt = subprocess.run([_tool('llvm-objdump'), '-d', '--mcpu=gfx950', elf_path], capture_output=True, text=True, check=True).stdout
    hists, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            cur = hists.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        m = re.match(r'^\s+([a-z_0-9]+)\b(.*?)//', line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith('buffer_load') and re.search(r'\blds\b', m.group(2)):
            op += ' lds'
        cur[op] = cur.get(op, 0) + 1
    return hists


def inspect(so_path=None):
    """[{name (demangled), mangled, meta: {...}, hist: {...}}] for every kernel of the library."""
    so_path = so_path or os.path.join(ROOT, 'elektronn3_amd', 'libe3unet.so')
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, img in enumerate(code_objects(so_path)):
            p = os.path.join(tmp, f'co{i}.elf')
            open(p, 'wb').write(img)
            meta = kernel_metadata(p)
            hist = kernel_histograms(p)
            for name, md in meta.items():
                out.append({'mangled': name, 'meta': md, 'hist': hist.get(name, {})})
    dm = demangle([k['mangled'] for k in out])
    for k in out:
        k['name'] = dm.get(k['mangled'], k['mangled'])
    return out


def wide_store_hazards(so_path=None, window=3):
    """[(kernel, store, overwriting instruction)]: buffer_store_dwordx3/x4 (or format_xyz/xyzw) with an SGPR soffset whose data registers a VALU
    instruction writes within the next `window` instructions.  LLVM's hazard recognizer inserts the store-data wait state only when the soffset is NOT
    a register (the gfx9 guides' exemption); gfx950 shows the hazard for SGPR offsets as well (tools/repro_soffset.hip)."""
    so_path = so_path or os.path.join(ROOT, 'elektronn3_amd', 'libe3unet.so')
    found = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, img in enumerate(code_objects(so_path)):
            p = os.path.join(tmp, f'co{i}.elf')
            open(p, 'wb').write(img)
            txt = subprocess.run([_tool('llvm-objdump'), '-d', '--mcpu=gfx950', p], capture_output=True, text=True, check=True).stdout.splitlines()
            name = None
            for n, line in enumerate(txt):
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    name = m.group(1)
                    continue
                m = re.match(r'\s+(buffer_store_(?:dwordx[34]|format_xyzw?))\s+v\[(\d+):(\d+)\], \S+ s\[\d+:\d+\], (s\d+|m0|vcc_lo|vcc_hi)\b', line)
                if not m:
                    continue
                lo, hi = int(m.group(2)), int(m.group(3))
                for nxt in txt[n + 1:n + 1 + window]:
                    ins = re.sub(r'\s*//.*', '', nxt).strip()
                    if re.match(r'(s_cbranch|s_branch|s_endpgm|s_barrier)', ins):
                        break
                    w = re.match(r'(v_(?!cmp|cmpx|readfirstlane|readlane|accvgpr_write|mfma)\S+)\s+v\[?(\d+)(?::(\d+))?\]?', ins)
                    if w:
                        a = int(w.group(2))
                        b = int(w.group(3) or a)
                        if not (b < lo or a > hi):
                            found.append((name, re.sub(r'\s*//.*', '', line).strip(), ins))
                            break
    return found


def mfma_ops(hist):
    return {op: n for op, n in hist.items() if op.startswith('v_mfma') or op.startswith('v_smfma')}


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else None
    ks = inspect()
    print(f'{len(ks)} kernels')
    print(f'{"vgpr":>5} {"agpr":>5} {"sgpr":>5} {"vspill":>6} {"sspill":>6} {"scratch":>8} {"lds":>7}  name')
    for k in sorted(ks, key=lambda k: k['name']):
        if pat and pat not in k['name']:
            continue
        m = k['meta']
        print(f'{m.get("vgpr_count", 0):5d} {m.get("agpr_count", 0):5d} {m.get("sgpr_count", 0):5d} {m.get("vgpr_spill_count", 0):6d} {m.get("sgpr_spill_count", 0):6d} '
              f'{m.get("private_segment_fixed_size", 0):8d} {m.get("group_segment_fixed_size", 0):7d}  {k["name"][:150]}')
        if pat:
            h = k['hist']
            keys = sorted(h, key=lambda o: -h[o])
            print('        ' + ', '.join(f'{o} {h[o]}' for o in keys if o.startswith(('v_mfma', 'ds_', 'buffer_', 'global_', 'scratch_', 'v_accvgpr', 'v_pk_', 's_waitcnt', 's_barrier'))))


if __name__ == '__main__':
    main()
