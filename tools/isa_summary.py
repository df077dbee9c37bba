"""Summarise the control flow / waitcnt structure of one kernel in a hipcc -S listing (tuning aid)."""
import re
import sys


def main(path, sym, limit=4000):
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith(sym + ':')][0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    out = []

    def push(tag):
        if out and out[-1][0] == tag:
            out[-1][1] += 1
        else:
            out.append([tag, 1])
    for l in lines[start:end + 1]:
        t = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            out.append(['\n' + t.split()[0], 1])
        elif t.startswith('global_load') or t.startswith('buffer_load'):
            push('GL')
        elif t.startswith('v_mfma'):
            push('MFMA')
        elif t.startswith('ds_read') or t.startswith('ds_load'):
            push('DSR')
        elif t.startswith('ds_write') or t.startswith('ds_store'):
            push('DSW')
        elif t.startswith('s_barrier'):
            push('BAR')
        elif t.startswith('s_waitcnt'):
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                out.append(['W%s' % m.group(1), 1])
        elif t.startswith('s_cbranch') or t.startswith('s_branch'):
            out.append([t.split()[0][2:] + '->' + t.split()[-1], 1])
    s = ' '.join('%s%s' % (a, 'x%d' % b if b > 1 else '') for a, b in out)
    print('instructions', end - start)
    print(s[:limit])
    for l in lines[end:end + 120]:
        if '.vgpr_count' in l or 'NumVgprs' in l or 'ScratchSize' in l or 'Occupancy' in l or 'NumAgprs' in l:
            print(l.strip())


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4000)
