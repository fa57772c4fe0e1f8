"""Linear instruction stream of one kernel of an annotated gfx950 assembly (-gline-tables-only -S), each instruction tagged with the
source file:line it was emitted for - the reading aid behind the per-stage wait-point work of round 5.

    python tools/isa_walk.py <k.s> <kernel-name-substring> [out.txt]

Also prints a per-source-line-range summary: instructions by class and the number of s_waitcnt per file."""
import re, sys, collections

def walk(path, sub):
    files, cur, infn = {}, (0, 0), False
    out = []
    for line in open(path):
        line = line.rstrip('\n')
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
            continue
        m = re.match(r'^(_Z\w+):', line)
        if m:
            infn = sub in m.group(1)
            continue
        if not infn:
            continue
        if line.startswith('.Lfunc_end'):
            infn = False
            continue
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r'^(\.LBB\w+):', line)
        if m:
            out.append(('label', m.group(1), '', cur))
            continue
        m = re.match(r'\s+([a-z]\w+)\s*(.*)', line)
        if not m or line.lstrip().startswith('.'):
            continue
        out.append(('ins', m.group(1), m.group(2).split(';')[0].strip(), cur))
    return files, out

if __name__ == '__main__':
    files, ins = walk(sys.argv[1], sys.argv[2])
    o = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
    for kind, op, args, cur in ins:
        tag = f'{files.get(cur[0], "?")[:18]}:{cur[1]}'
        if kind == 'label':
            o.write(f'{tag:26s} {op}:\n')
        else:
            o.write(f'{tag:26s}   {op} {args}\n')
