for sh in star3d1r; do
for n in 128 192 256 320 384; do
for o in "--option stream3=0" "--option stream3=1 --option steps_per_launch=2" "--option stream3=1 --option steps_per_launch=2 --option stream3_waves=4" ; do
python3 bench.py --shape $sh --size $n $n $n --steps 48 --warmup 6 --no-cpu-baseline $o 2>/dev/null | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$sh $n', '[$o]', d['value'], d['roofline']['kernel'][:52])"
done; done; done
