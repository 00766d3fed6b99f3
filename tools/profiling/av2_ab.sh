B="--dataset av2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --steps 30 --warmup 6"
get() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  a=$(cd ab_prev && python bench.py $B 2>/dev/null | get)
  b=$(python bench.py $B 2>/dev/null | get)
  echo "rep $rep av2 fwd [r5] $a [tree] $b"
done
a=$(cd ab_prev && python bench.py --dataset av2 --train --steps 8 --warmup 3 --no-roofline 2>/dev/null | get)
b=$(python bench.py --dataset av2 --train --steps 8 --warmup 3 --no-roofline 2>/dev/null | get)
echo "av2 train [r5] $a [tree] $b"
