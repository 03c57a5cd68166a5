"""Instruction census of a kernel's biggest loop (the decoder step loop) from hipcc's assembly, by class and by source line.

  cd tacotron_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=hidden \
      -mllvm -amdgpu-sched-strategy=iterative-maxocc -gline-tables-only -S --cuda-device-only decoder3.hip -o /tmp/d3.s
  python tools/isa_loop_census.py /tmp/d3.s 'fwd_kernelILi4ELi2ELb1' [regex of opcodes to list by source line]

The loop is found as the longest backward branch inside the kernel's text; `.loc` directives attribute instructions to source
lines (line 0 = compiler-generated joins).  This is how round 5 found that ~70 scalar exec-mask instructions per poll iteration
sat in gather_end ("Round 5: the decoder")."""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else r'^s_(and|or|xor|andn2|orn2)\w*_b64|^s_\w*saveexec|^s_c?branch'
    text = open(path).read().split('\n')
    starts = [i for i, l in enumerate(text) if re.match(r'^_Z\w+:', l)]
    begin = next(i for i in starts if kern in text[i])
    end = next((i for i in starts if i > begin), len(text))
    lines = text[begin:end]
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            if best is None or i - a > best[1] - best[0]:
                best = (a, i)
    a, b = best
    cls, by_line, ops = collections.Counter(), collections.Counter(), collections.Counter()
    cur, n = 0, 0
    for l in lines[a:b + 1]:
        m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            cur = int(m.group(2)) if m.group(1) == '0' else -int(m.group(1))   # negative: a header (file number)
            continue
        t = l.strip()
        if not t or t.startswith(('.', ';')) or t.endswith(':'):
            continue
        op = t.split()[0]
        if not re.match(r'^[sv]_|^ds_|^buffer_|^global_|^flat_|^scratch_', op):
            continue
        n += 1
        if 'saveexec' in op: c = 'saveexec'
        elif op.startswith('s_cbranch') or op == 's_branch': c = 'branch'
        elif op.startswith('s_waitcnt'): c = 'waitcnt'
        elif op.startswith('s_'): c = 'salu'
        elif op.startswith(('v_fma', 'v_pk_fma', 'v_fmac')): c = 'fma'
        elif op.startswith('v_mov'): c = 'v_mov'
        elif op.startswith('v_cmp'): c = 'v_cmp'
        elif op.startswith('v_'): c = 'valu'
        elif op.startswith('ds_'): c = 'lds'
        else: c = 'vmem'
        cls[c] += 1
        if re.search(pat, op):
            by_line[cur] += 1
            ops[op] += 1
    print('%s: loop of %d instructions (static, one wave): %s' % (kern, n, dict(cls.most_common())))
    print('matching %r: %d; by source line (0 = compiler-generated): %s' % (pat, sum(by_line.values()), by_line.most_common(25)))
    print('opcodes:', ops.most_common(12))


if __name__ == '__main__':
    main()
