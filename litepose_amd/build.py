"""Build liblitepose_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m litepose_amd.build            # rebuild if sources are newer than the .so
    python -m litepose_amd.build --flavour regstage # diagnostics: lib/liblitepose_amd_regstage.so, the same
                                            # sources with -DLP_NO_LDS_DMA -DLP_CLAIM_CU (fused blocks stage weights through
                                            # registers and own their CUs); loaded instead of the library when
                                            # LP_NATIVE_FLAVOUR=regstage is set

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'liblitepose_amd.so')
OBJDIR = os.path.join(os.path.dirname(HERE), 'build', 'obj')

# (source, extra flags).  ae_kernels: every fp op must round like the NumPy/torch CPU
# expression it restates -> no FMA contraction.
# NOPK: no packed fp32 code generation (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  DESIGN 5b: on gfx950 a packed fp32
# instruction whose op_sel routes a source's HIGH register into the LOW result returns src0 + 0 in lanes 48-63 when waves
# that issue v_mfma_f32_32x32x16_bf16 run next to it (tools/ubench/pk_vs_mfma.hip reproduces it in seconds) -- the rare
# wrong batch of the two-network schedule.  hipcc builds such forms from ordinary float2 arithmetic; files without
# hand-placed packed FMAs are compiled without the feature, tests/test_host_cpu.py scans the rest of the library.
NOPK = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
SOURCES = [
    ('engine.cpp', []),
    ('ae_api.cpp', []),
    ('net_kernels.hip', []),
    ('mb16_kernels.hip', []),
    ('mbtile_kernels.hip', []),
    ('mbtile_bf16.hip', []),
    ('stem_kernels.hip', []),
    ('bf16_kernels.hip', []),
    ('ae_kernels.hip', ['-ffp-contract=off'] + NOPK),
    ('ae_mid_kernels.hip', ['-ffp-contract=off'] + NOPK),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value',
          '-Wno-pass-failed']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def flavour_lib(flavour):
    return os.path.join(LIBDIR, 'liblitepose_amd_%s.so' % flavour) if flavour else LIB


def needs_build(flavour=None):
    lib = flavour_lib(flavour)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'litepose_amd.h'))
    deps.append(os.path.abspath(__file__))                 # the flags live here
    return any(os.path.getmtime(d) > t for d in deps)


# regstage: the fused blocks stage weights through registers and claim whole CUs (the A/B that cleared LDS-DMA);
# diag: the library plus the self-checking dwpw_kernel<3, ..., DIAG> of round 4's hunt (option "diag_dwpw", lp_diag_read) --
# the one kernel that keeps v_pk_add_f32 op_sel:[0,1] on purpose, which is why the product library does not link it
# trace (round 6): per-phase shader-clock sums of mbtb_kernel / mbtq_kernel / mbtd_kernel (lp_phase_trace_read)
# gtrace (round 6, built on demand only): group_kernel prints its per-phase clock sums (profiles/r06_group_phase_trace.txt)
FLAVOURS = {'regstage': ['-DLP_NO_LDS_DMA', '-DLP_CLAIM_CU'], 'diag': ['-DLP_DIAG_BUILD'], 'trace': ['-DLP_PHASE_TRACE'], 'gtrace': ['-DLP_GROUP_TRACE']}


def build(force=False, verbose=True, flavour=None):
    if flavour is not None:
        if not force and not needs_build(flavour):
            return flavour_lib(flavour)
        return _build(flavour_lib(flavour), OBJDIR + '_' + flavour, FLAVOURS[flavour],
                      'kernel_resources_%s.json' % flavour, verbose)
    if not force and not needs_build():
        return LIB
    return _build(LIB, OBJDIR, [], 'kernel_resources.json', verbose)


def _build(LIB, OBJDIR, defines, report, verbose):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()

    resources = {}

    def compile_one(item):
        src, extra = item
        obj = os.path.join(OBJDIR, src + '.o')
        # the device compiler reports registers / scratch / LDS of every kernel: collected into
        # lib/kernel_resources.json (a kernel that needs scratch must stay off the path, csrc/kernels.h uses_scratch)
        rep = ['-Rpass-analysis=kernel-resource-usage'] if src.endswith('.hip') else []
        cmd = [hipcc] + COMMON + defines + extra + rep + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stdout))
        cur = None
        for line in r.stdout.splitlines():
            if 'remark:' not in line:
                continue
            t = line.split('remark:', 1)[1].strip()
            if t.startswith('Function Name:'):
                cur = resources.setdefault(t.split(':', 1)[1].split('[')[0].strip(), {'source': src})
            elif cur is not None:
                for key, name in (('VGPRs:', 'vgprs'), ('AGPRs:', 'agprs'), ('ScratchSize [bytes/lane]:', 'scratch'),
                                  ('LDS Size [bytes/block]:', 'static_lds'), ('SGPRs:', 'sgprs')):
                    if t.startswith(key):
                        cur[name] = int(t[len(key):].split('[')[0].strip())
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    try:
        import json
        names = sorted(resources)
        dem = subprocess.run(['c++filt'], input='\n'.join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
        out = {}
        for mangled, pretty in zip(names, dem if len(dem) == len(names) else names):
            out[pretty.split('(')[0].replace('void ', '')] = resources[mangled]
        with open(os.path.join(LIBDIR, report), 'w') as f:
            json.dump(out, f, indent=1, sort_keys=True)
    except Exception as e:                  # the report is a diagnostic, never a build failure
        print('kernel resource report skipped:', e)
    # link under a temporary name, scan THAT, and only then move it into place: whatever goes wrong in the
    # scan -- an unverified packed-fp32 form, or the scanner itself failing (objcopy / offload-bundler / objdump) -- the
    # library an import would load is never an unscanned one
    tmp = LIB + '.unscanned'
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', tmp]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout)
    try:
        _scan(tmp, LIB, allow_diag=('-DLP_DIAG_BUILD' in defines), verbose=verbose)
    except Exception:
        if os.path.exists(tmp):
            os.replace(tmp, LIB + '.rejected')
        raise
    os.replace(tmp, LIB)
    if verbose:
        print('built', LIB)
    return LIB


def _scan(lib, final, allow_diag, verbose):
    """The packed-fp32 rule of DESIGN 5b, enforced where the library is made: disassemble what was just
    linked (tools/scan_isa.py) and refuse a library that contains a packed fp32 instruction whose operand routing the
    reproducer has not cleared next to bf16 MFMAs -- a compiler update that starts emitting such a form fails the build,
    not one batch in ten thousand.  The result travels with the .so (lib/isa_scan*.json).  No ROCm LLVM tools -> loud
    warning, and tests/test_host_cpu.py still scans wherever the tools exist."""
    import importlib.util
    import json
    tool = os.path.join(os.path.dirname(HERE), 'tools', 'scan_isa.py')
    out = os.path.splitext(final)[0].replace('liblitepose_amd', 'isa_scan') + '.json'
    strict = bool(os.environ.get('LP_REQUIRE_ISA_SCAN'))          # release builds: a missing scanner is an error
    if not os.path.exists(tool):
        if strict:
            raise RuntimeError('LP_REQUIRE_ISA_SCAN: tools/scan_isa.py missing')
        print('WARNING: tools/scan_isa.py missing, %s not scanned for unverified packed-fp32 forms' % final)
        return
    spec = importlib.util.spec_from_file_location('scan_isa', tool)
    si = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(si)
    if not si.tools_present():
        if strict:
            raise RuntimeError('LP_REQUIRE_ISA_SCAN: ROCm LLVM tools (llvm-objdump, clang-offload-bundler) not found')
        print('WARNING: ROCm LLVM tools not found, %s not scanned for unverified packed-fp32 forms' % final)
        return
    r = si.scan(lib)
    with open(out, 'w') as f:
        json.dump(r, f, indent=1, sort_keys=True)
    bad = {k: v for k, v in r['pk_unverified'].items()
           if not (allow_diag and k.startswith('lp::dwpw_kernel<') and k.endswith(', true>'))}
    if bad:
        raise RuntimeError('packed fp32 instructions with an operand routing not cleared by tools/ubench/pk_vs_mfma.hip '
                           '(DESIGN 5b) in %s: %s -- library moved to %s.rejected' % (final, bad, final))
    if verbose:
        print('ISA scan: %d kernels, %d packed fp32 instructions, every modifier form cleared by the reproducer'
              % (r['kernels'], r['pk_total']))


if __name__ == '__main__':
    if '--flavour' in sys.argv:
        build(flavour=sys.argv[sys.argv.index('--flavour') + 1], force=True)
    else:
        build(force='--force' in sys.argv)
