"""Basic blocks of one kernel in a -save-temps .s file that contain a given instruction: instruction counts per block
(which loop is hot, whether it spills).   python tools/isa_blocks.py <file.s> <kernel-substring> <instr> [<instr> ...]"""
import sys
s = open(sys.argv[1]).read()
a = s.index(sys.argv[2] + ":") if (sys.argv[2] + ":") in s else s.index(sys.argv[2])
b = s.index('.Lfunc_end', a)
blocks, cur, name = [], [], 'entry'
for l in s[a:b].split('\n'):
    t = l.strip()
    if t.startswith('.LBB') and ':' in t:
        blocks.append((name, cur)); cur = []; name = t.split(':')[0] + ' ' + t.split(';')[-1].strip()
    else:
        cur.append(l)
blocks.append((name, cur))
for name, blk in blocks:
    ins = [l.strip() for l in blk if l.strip() and not l.strip().startswith((';', '.'))]
    hits = {k: sum(1 for l in ins if l.startswith(k)) for k in sys.argv[3:]}
    if any(hits.values()):
        valu = sum(1 for l in ins if l.startswith('v_'))
        salu = sum(1 for l in ins if l.startswith('s_') and not l.startswith(('s_waitcnt', 's_nop')))
        print(name, len(ins), 'instr (VALU %d SALU %d nop %d wait %d ds %d global %d scratch %d lane-spill %d)' % (
            valu, salu, sum(1 for l in ins if l.startswith('s_nop')), sum(1 for l in ins if l.startswith('s_waitcnt')),
            sum(1 for l in ins if l.startswith('ds_')), sum(1 for l in ins if l.startswith(('global_', 'buffer_'))),
            sum(1 for l in ins if l.startswith('scratch_')), sum(1 for l in ins if l.startswith(('v_readlane', 'v_writelane')))), hits)
