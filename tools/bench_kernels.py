import json,sys
for f in sys.argv[1:]:
    l=[x for x in open(f) if x.startswith('{')]
    d=json.loads(l[-1])
    print(f, d['value'], d['ms_per_step'], 'bf16:', d.get('amp_bf16',{}).get('ms_per_step'))
    for leg,k in (('fp32',d.get('kernels',{})),('bf16',d.get('amp_bf16',{}).get('kernels',{}))):
        for n,v in k.items():
            if n.startswith('layer'): print('   ',leg,n, round(v['avg_ms'],3))
