"""Scan the gfx950 assembly hipcc wrote (-save-temps) for the result of a multi-pass vector instruction (packed fp32; also transcendentals and fp64
arithmetic) consumed, a few instructions later, by an instruction
that does not take its operands through the VALU's forwarding path: a DPP source, LDS data, a store's data.

Why: DESIGN.md section 5.  `v_pk_add_f32` (two passes over the wave) followed three instructions later by `v_mov_b32_dpp` reading its
result -- the shape hipcc gave the GroupNorm partial sums once the SLP vectoriser had paired (sum, sum of squares) -- read STALE values
in lanes 48-63 whenever a wave of another kernel shared the SIMD (two engines on two streams); alone on its SIMD it never did.
hipcc keeps two wait states between a VALU write and a DPP read, whatever the writer is.

usage: scan_pk_hazard.py [--fail] file.s [max distance, default 6]   (tests/test_build_isa.py imports scan(); the Makefile runs it with --fail on every build)"""
import re, sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok): out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok): out.add(int(m.group(1)))
    return out


def scan(path, maxd=6):
    """{(kernel, kind): [(instructions between, producer, consumer), ...]}; kind: dpp | lane | wide | lds | store."""
    cur = None; window = []; hits = {}
    for ln in open(path):
        m = re.match(r'^(_Z\w+):', ln)
        if m: cur = m.group(1); window = []; continue
        l = ln.strip()
        if not l or l[0] in '.;' or l.endswith(':') or cur is None: continue
        op, _, rest = l.partition(' ')
        ops = [o.strip() for o in rest.split(',')]
        if op.startswith('s_nop'): window = [(w, d, age + int(rest.split()[0]) + 1) for (w, d, age) in window]
        else: window = [(w, d, age + 1) for (w, d, age) in window]
        window = [(w, d, age) for (w, d, age) in window if age <= maxd + 1]
        kind = None; src = set()
        if '_dpp' in op: kind = 'dpp'; src = regs(','.join(ops[1:]))
        elif op.startswith(('ds_write_b128', 'ds_write_b96', 'global_store_dwordx4', 'global_store_dwordx3', 'buffer_store_dwordx4', 'buffer_store_dwordx3')):
            kind = 'wide'; src = regs(rest)      # (a > 64-bit store's data)
        elif op.startswith(('ds_write', 'ds_bpermute', 'ds_permute', 'ds_swizzle')): kind = 'lds'; src = regs(rest)
        elif op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'global_atomic')): kind = 'store'; src = regs(rest)
        elif op.startswith(('v_readlane', 'v_readfirstlane', 'v_permlane')): kind = 'lane'; src = regs(','.join(ops[1:]))
        if kind:
            for (w, d, age) in window:
                if d & src: hits.setdefault((cur, kind), []).append((age - 1, w, l))
        # multi-pass producers: packed fp32 (two passes), transcendentals (quarter rate), fp64 arithmetic
        if (op.startswith('v_pk_') and op.endswith('_f32')) or re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op) or re.match(r'v_(fma|add|mul|max|min|div_\w+|trig_preop|ldexp|frexp_mant|fract|rndne|floor|ceil|trunc)_f64', op):
            window.append((l, regs(ops[0]), 0))
        else:
            dst = regs(ops[0]) if op.startswith('v_') else set()
            window = [(w, d - dst, age) for (w, d, age) in window]     # overwritten since
    return hits


if __name__ == "__main__":
    fail = "--fail" in sys.argv                                   # the Makefile's form: exit 1 when a DPP / lane-read source follows a multi-pass producer
    argv = [a for a in sys.argv[1:] if a != "--fail"]
    hits = scan(argv[0], int(argv[1]) if len(argv) > 1 else 6)
    tot = {}
    for (fn, kind), hs in sorted(hits.items()):
        tot[kind] = tot.get(kind, 0) + len(hs)
        print(f"{kind:5s} {len(hs):4d} hits, closest {min(h[0] for h in hs)} instruction(s) between: {fn[:110]}")
        if kind in ('dpp', 'lane'):
            for h in hs[:2]: print(f"        {h[1]}  ->  {h[2]}   ({h[0]} between)")
    print("totals:", tot)
    if fail and (tot.get("dpp", 0) or tot.get("lane", 0)):
        sys.exit(1)
