#!/usr/bin/env python
"""Disassemble the device code of a shared library (every offload bundle of .hip_fatbin, gfx950) and report, per kernel,
the instructions the library must not contain (tests/test_host_cpu.py runs this on the build):

  * packed fp32 arithmetic whose op_sel takes src0's LOW and src1's HIGH register for the low result
    (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[0,1(,x)]): on gfx950 the low result comes back as if src1 were
    zero in lanes 48-63 while waves that issue bf16 MFMAs (v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16) run next to
    it -- tools/ubench/pk_vs_mfma.hip, profiles/r04_pk_vs_mfma_*.txt; every other op_sel / op_sel_hi form tested clean;
  * LDS-DMA (global_load_lds_* / buffer_load_* ... lds): off the product path since round 4 (kernels.h).

    python tools/scan_isa.py [path/to/lib.so]      -> prints a JSON dict {kind: {kernel: count}}"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
PK = re.compile(r'\b(v_pk_(?:add|mul|fma)_f32)\b(.*)')
OPSEL = re.compile(r'op_sel:\[([0-9,]+)\]')
DMA = re.compile(r'global_load_lds_|buffer_load_[a-z0-9_]+ .*\blds\b')


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-objdump'))


def scan(so_path):
    tmp = tempfile.mkdtemp()
    out = {'pk_op_sel_01': collections.Counter(), 'lds_dma': collections.Counter(), 'kernels': 0, 'pk_total': 0}
    try:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so_path], check=True)
        blob = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', blob)] + [len(blob)]
        for i in range(len(starts) - 1):
            part, co = os.path.join(tmp, 'b%d.bin' % i), os.path.join(tmp, 'b%d.co' % i)
            with open(part, 'wb') as f:
                f.write(blob[starts[i]:starts[i + 1]])
            subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o',
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + part, '--output=' + co, '--unbundle'],
                           check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--demangle', co], check=True,
                                 stdout=subprocess.PIPE, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.*)>:', line)
                if m:
                    cur = m.group(1).split('(')[0].replace('void ', '')
                    out['kernels'] += 1
                    continue
                m = PK.search(line)
                if m:
                    out['pk_total'] += 1
                    sel = OPSEL.search(m.group(2))
                    if sel:
                        bits = sel.group(1).split(',')
                        if bits[0] == '0' and bits[1] == '1':
                            out['pk_op_sel_01'][cur] += 1
                if DMA.search(line):
                    out['lds_dma'][cur] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out['pk_op_sel_01'] = dict(out['pk_op_sel_01'])
    out['lds_dma'] = dict(out['lds_dma'])
    return out


if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, 'litepose_amd', 'lib', 'liblitepose_amd.so')
    print(json.dumps(scan(p), indent=1, sort_keys=True))
