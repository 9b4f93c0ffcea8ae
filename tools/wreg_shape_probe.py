import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for s in d['roofline']['shapes']:
    if s['kernel'].startswith('gemm_wreg'):
        print(sys.argv[1].split('/')[-1], s['M'], s['N'], s['K'], 'launches', s['launches'], 'avg_us %.1f'%(1e3*s['avg_launch_ms']), 'frac %.3f'%s['frac'])
print(' ms_per_step %.1f'%d['ms_per_step'])
