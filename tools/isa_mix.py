"""Static instruction mix of one kernel in a hipcc -S listing.  usage: isa_mix.py file.s mangled-name-fragment [...]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
for tag in sys.argv[2:]:
    m = re.search(r'^(_Z\d+' + re.escape(tag) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', txt, re.S | re.M)
    if not m:
        print(tag, "not found")
        continue
    ins = [l.split()[0] for l in m.group(2).split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter()
    for i in ins:
        key = ('v_pk' if i.startswith('v_pk') else 'v_cvt' if i.startswith('v_cvt') else 'v_cndmask' if i.startswith('v_cndmask')
               else 'v_fma' if i.startswith(('v_fma', 'v_fmac')) else 'v_minmax' if i.startswith(('v_max', 'v_min', 'v_med'))
               else 'v_cmp' if i.startswith('v_cmp') else 'valu_other' if i.startswith('v_') else 'salu' if i.startswith('s_')
               else 'ds' if i.startswith('ds_') else 'vmem' if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
        c[key] += 1
    print(tag, len(ins), dict(c))
