import re,sys
def scan(path, win=3):
    lines=[l.split(';')[0].strip() for l in open(path, errors='replace')]
    ins=[(i,l) for i,l in enumerate(lines) if l and not l.endswith(':') and not l.startswith('.')]
    hits=[]
    for k,(i,l) in enumerate(ins):
        m=re.match(r'(v_pk_\w+_f32)\s+v\[(\d+):(\d+)\]',l)
        if not m: continue
        lo,hi=int(m.group(2)),int(m.group(3))
        for (j,n) in ins[k+1:k+1+win]:
            mm=re.match(r'(v_\w+)\s+v(\d+),(.*)',n)
            if mm and not mm.group(1).startswith('v_pk') and int(mm.group(2)) in (lo,hi):
                srcs=re.findall(r'\bv(\d+)\b',mm.group(3))+[str(x) for a,b in re.findall(r'v\[(\d+):(\d+)\]',mm.group(3)) for x in range(int(a),int(b)+1)]
                if str(mm.group(2)) not in srcs:
                    hits.append((i,l,n)); break
    return hits
for p in sys.argv[1:]:
    h=scan(p); print(p, len(h))
    for i,l,n in h[:4]: print('   ',l,' ...  ',n)
