for w in 5 25 50 100 200 400; do python3 bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('warmup', d['warmup'], 'value', d['value'], 'launch_us', d['roofline']['launch_us'])"; done
