for i in 1 2 3; do
for o in "spans3=0" "spans3=1"; do
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --shape box3d1r --option $o 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('box fp64', '$o', d['value'], d['roofline']['launch_us'])"
done; done
for o in "spans3=0" "spans3=1"; do
timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --shape box3d1r --size 512 512 512 --option $o 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('box fp64 512', '$o', d['value'], d['roofline']['launch_us'])"
done
