"""tools/isa_blocks.py <file.s> <first line> <last line> <min VALU>: basic blocks of a stretch of hipcc -save-temps assembly with their VALU / v_mov /
v_alignbit / v_bitop3 / SALU counts and branch targets -- how the static figures of profiles/r04_contam_static.md were read."""
import re,sys
fn=sys.argv[1]; lo=int(sys.argv[2]); hi=int(sys.argv[3])
L=open(fn).read().split('\n')[lo-1:hi]
blocks=[]; cur=None
for i,l in enumerate(L):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        cur={'name':m.group(1),'line':lo+i,'valu':0,'salu':0,'mov':0,'br':[], 'n':0,'align':0,'bitop':0}
        blocks.append(cur); continue
    if cur is None: continue
    t=l.strip().split(' ')[0] if l.strip() else ''
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur['n']+=1
    if t.startswith('v_'):
        cur['valu']+=1
        if t.startswith('v_mov') or t.startswith('v_accvgpr'): cur['mov']+=1
        if t.startswith('v_alignbit'): cur['align']+=1
        if t.startswith('v_bitop3'): cur['bitop']+=1
    elif t.startswith('s_'):
        cur['salu']+=1
        if t.startswith('s_cbranch') or t=='s_branch':
            cur['br'].append(l.strip().split()[-1])
for b in blocks:
    if b['valu']>=int(sys.argv[4]) :
        print(b['line'],b['name'],'valu',b['valu'],'mov',b['mov'],'align',b['align'],'bitop3',b['bitop'],'salu',b['salu'],b['br'])
