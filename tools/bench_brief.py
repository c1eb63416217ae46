import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); r=d['roofline']
    ek=r.get('encoder_kernels',{})
    print('batch launch:', {k:v.get('ms') for k,v in ek.items()})
    print(d['value'], d['steps'], 'frac',r['frac'],'ach',r['achieved'], 'single',r.get('single_frame_launch'), 'faults',d['config'].get('lane_faults'))
