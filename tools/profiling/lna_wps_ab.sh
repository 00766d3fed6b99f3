out=gpurun_out/r5/lna_wps_ab.txt; mkdir -p gpurun_out/r5; : > $out
for rep in 1 2; do
for v in "" "-DLNA_SEG_WPS=2" "-DLNA_WPS=2 -DLNA_SEG_WPS=2"; do
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  echo "## rep $rep [${v:-default 3 / 3}]" >> $out
  python tools/profiling/sir_bench.py 2>/dev/null | tr "|" "\n" | grep -i "K22" >> $out
done
done
touch fullysparsefusion_amd/csrc/linear_norm_act.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
