#!/usr/bin/env python
"""Build-time guard for the two kernels whose operand requests are inline asm with hand-counted waits (k_w4_gemm128b,
k_w4_wgrad128b; csrc/kernels_w4.hip): the compiler treats an asm load's '=v' destination as defined at the asm statement, so
correctness relies on it never reading, copying, moving or overwriting that register while the request is in flight.  This
script compiles kernels_w4.hip to device assembly and walks both kernels instruction by instruction: every
`global_load_dwordx4` opens its destination registers, every `s_waitcnt vmcnt(N)` retires all but the N youngest vector-memory
operations (the walk is linear in program order -- both kernels are straight-line apart from their
one K loop, whose body begins and ends with nothing in flight), and ANY mention of an open register by another instruction (a use, a v_mov, an overwrite) -- or any scratch
access -- is reported.  Exit code 1 on a violation.  (Advisor finding, round 3.)

    python tools/check_asm_loads.py            # used by neural-ode-features_amd/build.py after linking
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'neural-ode-features_amd', 'csrc')
KERNELS = ('k_w4_gemm128b', 'k_w4_wgrad128b', 'k_w4_gemm128h', 'k_w4_gemm256h')
LOAD_WAW_OK = ('k_w4_gemm128h',)
# kernels that may spill OUTSIDE their K loop (k_w4_gemm256h: 256 accumulators leave the AGPRs through a few spilled words at the loop's exit):
# their scratch instructions are checked like every other instruction -- no register of an open request may appear in them -- but are not
# violations by themselves.  (The compiler's own counted waits for its scratch loads stay correct beside asm requests it does not know:
# requests return in order, so unknown OLDER ones change nothing and unknown YOUNGER ones only make its wait longer.)
SCRATCH_OK = ('k_w4_gemm256h',)
VMEM = re.compile(r'^\s*(global_|buffer_|flat_|scratch_)(load|store|atomic)')
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(asm_path):
    lines = open(asm_path).read().split('\n')
    bad = []
    for k in KERNELS:
        start = next((i for i, ln in enumerate(lines) if re.match(r'^_ZN4node\d+%s' % k, ln)), None)
        if start is None:
            bad.append('%s: kernel not found in the assembly' % k)
            continue
        inflight = []       # [destination registers or None, line number, other_lanes] of vector-memory operations, oldest first
        nloads = 0
        for i in range(start + 1, len(lines)):
            ln = lines[i].split(';')[0].strip()
            if ln.startswith('.Lfunc_end'):
                break
            if not ln or ln.endswith(':') or ln.startswith('.') or ln.startswith('s_endpgm'):
                continue
            if ln.startswith('scratch_') and k not in SCRATCH_OK:
                bad.append('%s: scratch access at line %d: %s' % (k, i + 1, ln))
            # the else side of a divergent if (`s_xor_b64 exec, exec, saved` / `s_andn2_saveexec_b64`) runs on the lanes the then side did not: what the then
            # side left in flight is on other lanes until the paths rejoin (`s_or_b64 exec, exec, saved`)
            if re.match(r's_xor_b64\s+exec,\s*exec,', ln) or ln.startswith('s_andn2_saveexec_b64'):
                for e in inflight:
                    e[2] = True
            elif re.match(r's_or_b64\s+exec,\s*exec,', ln):
                for e in inflight:
                    e[2] = False
            m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', ln)
            if m:
                keep = int(m.group(1))
                inflight = inflight[len(inflight) - keep:] if keep else []
                continue
            if ln.startswith('s_waitcnt') and 'vmcnt' not in ln:
                continue
            if VMEM.match(ln):
                ops = ln.split(None, 1)[1] if ' ' in ln else ''
                if ln.startswith('global_load_dwordx4'):
                    dest = regs_of(ops.split(',')[0])
                    src = regs_of(','.join(ops.split(',')[1:]))
                    nloads += 1
                else:
                    dest, src = None, regs_of(ops)
                # (k_w4_gemm128h also holds COMPILER-managed loads -- its shared piece's ring, refilled past the range's end: a load whose
                #  destination is that of an older load still in flight is legal there, loads return in order; an ADDRESS read from such
                #  a register is not, in any kernel)
                waw_ok = k in LOAD_WAW_OK
                hit = [r for d, _, o in inflight if d and not o for r in d if r in src or (dest and r in dest and not waw_ok)]
                if hit:
                    bad.append('%s: line %d touches v%d while its load is in flight: %s' % (k, i + 1, hit[0], ln))
                if waw_ok and dest:
                    for e in inflight:
                        if e[0] and (e[0] & dest):
                            e[0] = e[0] - dest      # superseded: the younger load owns the registers now
                inflight.append([dest, i + 1, False])
                continue
            touched = regs_of(ln)
            hit = [r for d, _, o in inflight if d and not o for r in d if r in touched]
            if hit:
                bad.append('%s: line %d touches v%d while its load is in flight: %s' % (k, i + 1, hit[0], ln))
        if nloads == 0:
            bad.append('%s: no global_load_dwordx4 found (did the kernel change?)' % k)
    return bad


def check_statement_sources():
    """Source-level rules for the OTHER hand-written vector-memory asm of the library (advisor finding, round 5): an asm statement
    that loads into registers must carry its own `s_waitcnt vmcnt(0)` (kernels_tiny_solve.hip: the results cannot be used above the
    wait, and the compiler never sees a request in flight) with EARLY-CLOBBER outputs (`=&v`: an output may not share a register with
    a pointer input of a later request of the same statement); an asm LDS-DMA piece (`global_load_lds_dwordx4`: kernels_w4.hip) has no
    register destination at all and must restore M0 in the statement that writes it.  Checked for both the product and the
    diagnostics build: the rules are on the source text."""
    bad = []
    for fn in ('kernels_tiny_solve.hip', 'kernels_tiny.hip', 'kernels_w4.hip'):
        src = open(os.path.join(CSRC, fn)).read()
        for m in re.finditer(r'asm volatile\((.*?)\);', src, re.S):
            st = m.group(1)
            line = src.count('\n', 0, m.start()) + 1
            if 'global_load_lds' in st:
                if st.count('s_mov_b32 m0') < 2 and 's_mov_b32 %0, m0' not in st:
                    bad.append('%s:%d: LDS-DMA asm does not save / restore M0 inside its statement' % (fn, line))
                continue
            loads = len(re.findall(r'global_load_dword|TS_Q\b', st))
            if loads == 0:
                continue
            if fn == 'kernels_w4.hip':
                continue      # (register loads of kernels_w4.hip carry counted waits in separate statements: walked in the assembly above)
            if 's_waitcnt vmcnt(0)' not in st:
                bad.append('%s:%d: asm load without its own s_waitcnt vmcnt(0) in the statement' % (fn, line))
            if re.search(r'"=v"', st):
                bad.append('%s:%d: asm load output without early clobber (=&v)' % (fn, line))
    return bad


def build_flags():
    """The flags the library itself is compiled with (neural-ode-features_amd/build.py: one list, no drift)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('node_amd_build_flags', os.path.join(ROOT, 'neural-ode-features_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.HIPCC, list(mod.FLAGS)


def main():
    hipcc, flags = build_flags()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'kw4.s')
        cmd = [hipcc] + flags + ['-S', '--cuda-device-only', os.path.join(CSRC, 'kernels_w4.hip'), '-o', out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        bad = check(out)
    bad += check_statement_sources()
    for b in bad:
        print('check_asm_loads:', b)
    print('check_asm_loads: %s' % ('%d violation(s)' % len(bad) if bad else 'ok (%s; statement rules: kernels_tiny_solve.hip, kernels_tiny.hip, kernels_w4.hip)' % ', '.join(KERNELS)))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
