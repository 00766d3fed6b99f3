mkdir -p gpurun_out/j8
for rep in 1 2 3 4; do
  for e in "FSF_READBACK_MAILBOX=0" "FSF_READBACK_MAILBOX=1"; do
    v=$(env $e python bench.py --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep 10-sweep [$e]  $v"
    v=$(env $e python bench.py --sweeps 1 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep 1-sweep  [$e]  $v"
  done
done > gpurun_out/j8/ab.txt 2>&1
cat gpurun_out/j8/ab.txt
