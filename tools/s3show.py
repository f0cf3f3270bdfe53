import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else '/root/repo/gpurun_out/stream3_check.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['shape'], d['dims'][0], d['kernel'].split('[')[1][:-1] if 'planes' in d['kernel'] else 'TILE', d['us'], d['gstencils'])
    else:
        print(l.strip()[:300])
