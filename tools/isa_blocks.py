"""Development tool: per-basic-block instruction mix of one kernel in a hipcc -S listing.
usage: python tools/isa_blocks.py file.s kernel-name-substring [first_line last_line]"""
import collections
import re
import sys


def classify(x):
    if x.startswith('v_mfma'): return 'mfma'
    if x.startswith('buffer_load'): return 'bufload'
    if x.startswith('buffer_store') or x.startswith('global_store'): return 'store'
    if x.startswith('global_load') or x.startswith('global_atomic'): return 'gload'
    if x.startswith('ds_read') or x.startswith('ds_load'): return 'ds_read'
    if x.startswith('ds_write') or x.startswith('ds_store'): return 'ds_write'
    if x.startswith('v_readlane') or x.startswith('v_writelane'): return 'lane_spill'
    if x.startswith('v_'): return 'valu'
    if x.startswith('s_waitcnt'): return 'waitcnt'
    if x.startswith('s_nop'): return 's_nop'
    if x.startswith('s_load') or x.startswith('s_buffer_load'): return 'smem'
    if x.startswith('s_cbranch') or x.startswith('s_branch'): return 'branch'
    if x.startswith('s_'): return 'salu'
    return x


def main():
    lines = open(sys.argv[1]).read().split('\n')
    name = sys.argv[2]
    start = [i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0]][0]
    end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i] or lines[i].strip() == 's_endpgm')
    body = lines[start:end + 1]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
    cur, cnt, first = 'entry', collections.Counter(), 0
    def flush(i):
        if sum(cnt.values()):
            print(f"{first:5d}-{i:5d} {cur:12s} n={sum(cnt.values()):4d} " + ' '.join(f"{k}={v}" for k, v in sorted(cnt.items())))
    for i, l in enumerate(body):
        if i < lo or i >= hi: continue
        s = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            flush(i); cur, cnt, first = m.group(1), collections.Counter(), i
            continue
        if not s or s.startswith((';', '.')): continue
        op = s.split()[0]
        cnt[classify(op)] += 1
        if op.startswith('s_cbranch') or op.startswith('s_branch'):
            print(f"      {i:5d}   {s}")
    flush(hi)


if __name__ == '__main__':
    main()
