mkdir -p gpurun_out/r04
timeout -k 10 800 python -m pytest tests/test_sw2d_gpu.py -m gpu -x -q > gpurun_out/r04/gputests_srcid.log 2>&1; tail -3 gpurun_out/r04/gputests_srcid.log
for n in "8 500x250" "6 1000x250" "5 800x400"; do
  timeout -k 10 200 python3 profiles/time_rk2.py $n 2>&1 | tail -1 | tee -a gpurun_out/r04/rk2_srcid.json
  BDG_SW2D_SOURCES_PRODUCT=1 timeout -k 10 200 python3 profiles/time_rk2.py $n 2>&1 | tail -1 | tee -a gpurun_out/r04/rk2_srcprod.json
done
